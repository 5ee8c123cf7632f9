// Host-callable launchers of the gfx950 kernels.  Each launcher only enqueues on
// the given stream; none allocates or synchronises.
#pragma once
#include <hip/hip_runtime.h>

#include <functional>

#include <cstddef>
#include <cstdint>

namespace jea {

// Passed to kernels by value: it then lives in the kernarg segment, where
// wave-uniform reads at compile-time offsets become scalar loads.
struct QuantTables {
    float lum[64];   // natural order, float(Q) as the reference converts it
    float chr[64];   // (quantiser.cpp:61,66,76,81)
    // Transposed copies ([u][v], i.e. index 8u+v holds entry 8v+u) in the order the
    // forward kernel walks them, and their reciprocals RN(1/Q) for its fast path.
    float lum_t[64];
    float chr_t[64];
    float rlum_t[64];
    float rchr_t[64];
};

// Huffman code tables: entry = (length << 16) | code word, indexed by symbol.
struct HuffTables {
    uint32_t dc[2][12];   // [0 = luminance, 1 = chrominance][SSSS]
    uint32_t ac[2][256];  // [class][RRRRSSSS]; 0x00 = EOB, 0xF0 = ZRL
};

// Huffman decode tables, one per (class, DC/AC): index 0 lum-DC, 1 lum-AC,
// 2 chr-DC, 3 chr-AC.  Entries are (length << 8) | symbol, 0 = no code word.
// fast[] resolves codes of up to 9 bits from the next 9 stream bits.  Canonical
// codes grow numerically with their length, so all longer codes sit at the top
// of the 16-bit prefix space: slow[] is indexed by (next 16 bits) - first_long
// and covers that whole range (640 prefixes for the Annex K AC tables).
constexpr int kDecodeSlowEntries = 1024;
struct DecodeTable {
    uint16_t fast[512];
    uint16_t slow[kDecodeSlowEntries];
    uint32_t first_long;  // smallest 16-bit prefix that starts a code longer than 9 bits; 65536 when none
};
struct DecodeTables {
    DecodeTable tab[4];
};

// The same four tables for the parallel decoder, whose inner loop wants everything about a symbol in
// one word so that it needs no branches:
//   byte 0  bits the entry takes from the stream: code length + amplitude bits (1 when no code matches),
//           plus the length of the end-of-block code for a combined entry (below)
//   byte 1  bits 0-6: coefficient positions skipped before the value: the run of an AC symbol, 0 for DC
//           symbols and for the end-of-block symbol;  bit 7 (kStepEndsBlock): the block ends with this entry
//   byte 2  bits 0-4: code length;  bits 5-7: 0, or for a combined entry the length of the end-of-block code
//   byte 3  bits 0-4: 32 - amplitude bits (the shift that brings the amplitude down from the top of a word);
//           kStepHasAmp, kStepBad, kStepDcAdd flags
// fast[] is indexed by the next kFastBits stream bits.  Where longer codes start with those bits the entry
// has byte 0 = 0 and points to a sub-table of slow[] indexed by the stream bits that follow:
//   byte 1  32 - m for a sub-table of 2^m entries (the right shift that leaves the m bits behind the index bits),
//   bytes 2-3  byte offset of the sub-table in the table.
// The tables of the Annex K codes (built on the host) use sub-tables of 2^(16 - kFastBits) entries throughout; the
// tables built on the device from a file's own DHT segment size each sub-table by the longest code under its
// prefix, which makes any code with the format's alphabets fit the same slow[] sizes (at most one oversized
// sub-table per code length: 162 + 2 + 4 + ... + 32 entries).
// Combined entries: sparse blocks are a DC symbol, a few AC symbols and the end-of-block code, so most look-ups
// of a smooth picture would be end-of-block codes.  Where a symbol with its amplitude bits AND the
// end-of-block code of the same table class behind it fit into the kFastBits index bits, the entry stands for
// both: a DC symbol (also one without amplitude bits) or an AC symbol with amplitude, then the end of the
// block.  If such an AC symbol fills position 63 the block is complete without an end-of-block code (the bits
// that look like one belong to the next block): the walker then gives the code's bits back (kStepEobLen).
constexpr uint32_t kStepEndsBlock = 0x80u << 8;
constexpr uint32_t kStepEobLenShift = 21;    // bits 21-23
constexpr uint32_t kStepHasAmp = 1u << 29;   // amplitude bits follow the code
constexpr uint32_t kStepBad = 1u << 30;      // no code word, or a symbol the format does not allow
constexpr uint32_t kStepDcAdd = 1u << 31;    // DC symbol with amplitude bits (the sign bit: one compare)
template <int kFastBitsT, int kSlowEntriesT>
struct StepTableT {
    static constexpr int kFastBits = kFastBitsT, kSlowEntries = kSlowEntriesT;
    uint32_t fast[1 << kFastBitsT];
    uint32_t slow[kSlowEntriesT];
};
// Table sizes are LDS, and LDS is what limits the workgroups per compute unit of the passes that walk the stream.
// Eleven index bits for the AC tables (23 KB of tables): three workgroups; ten (15 KB): four, and the synchronisation
// rounds of a batch run 13-26 % quicker, although fewer symbol + end-of-block pairs fit an entry; eleven bits for the
// DC tables too (34 KB): two, and a 4K picture of noise decoded in 0.78 instead of 0.70 ms.
using StepTableDc = StepTableT<9, 128>;    // DC codes are at most 11 bits long
using StepTableAc = StepTableT<10, 320>;   // the Annex K AC codes beyond 10 bits fit five 64-entry sub-tables
struct StepClass {
    StepTableDc dc;
    StepTableAc ac;
};
struct StepTables {
    StepClass cls[2];   // luminance, chrominance
};

// Which of the reference's colour mappers the components come from (colour_mapping.cpp):
// RGBToYCbCrMapper -- the transform of :44-92, component 0 is the only luminance component (:94-97);
// RGBToRGBMapper   -- components are R, G, B as they are (:16-37) and all three count as luminance
//                     (:39-42), i.e. use the luminance quantiser and Huffman tables.
struct ComponentMode {
    uint32_t lum_mask = 1;     // bit c set: component c uses the luminance tables
    uint32_t passthrough = 0;  // non-zero: no colour transform
    __host__ __device__ bool is_lum(uint32_t c) const { return (lum_mask >> c) & 1u; }
};

struct HeaderBytes {
    uint8_t b[608];  // 607 used
};

// Entropy coding works on tiles of kTileBlocks consecutive block-channels of
// one image (84 MCUs); tiles never straddle images.
constexpr int kTileBlocks = 252;
constexpr int kTileOverlap = 4;   // block-channels of the next tile that a tile's workgroup handles once more
// Bit string of one block-channel: at most 22 + 63 * 26 = 1660 bits = 52 words.
// A tile's slots are a [kSlotWords][kSlotStride] word array, block j in column j.
constexpr int kSlotWords = 52;
constexpr int kSlotStride = 256;

inline uint32_t tiles_per_image(size_t mcus_per_image) {
    return (uint32_t)((mcus_per_image * 3 + kTileBlocks - 1) / kTileBlocks);
}

// Position of an MCU-row stripe inside the entropy stream of its image, for
// images whose stripes are encoded by different contexts / GPUs.  All zero for a
// whole image.  The stripe's bit strings start `bit_base` bits into the image's
// stream; it owns the stream bytes from first_byte = ceil(bit_base / 8) on, and
// the byte it shares with the following stripe, which it completes with
// `next_head` (the first 32 bits of that stripe's stream, MSB first).
struct StripeInfo {
    uint64_t bit_base = 0;
    uint64_t first_byte = 0;   // ceil(bit_base / 8)
    uint64_t origin_byte = 0;  // first_byte rounded down to 4: stream byte held at unstuffed[0]
    uint32_t next_head = 0;
    uint32_t has_next = 0;
    uint32_t active = 0;       // non-zero: stripe mode (segment size written by the stuff pass)
    int16_t initial_dc[3] = {0, 0, 0};  // quantised DC of the last MCU of the previous stripe
};

// Device-side bookkeeping of one encode pass; all arrays are caller-provided
// scratch.  n_tiles = tiles per image, n = images.
// Per-image Huffman tables (the optional mode the reference names as future work, entropy_encoder.cpp:614):
// scratch of launch_entropy_encode's two extra kernels.  All null: the Annex K tables for every image.
constexpr uint32_t kHuffSpecBytes = 416;    // four tables as the header carries them: id byte, 16 counts, values
constexpr uint32_t kHuffSpecOffset = 177;   // where they sit in the 607-byte header
struct OptimalHuffman {
    uint32_t* hist = nullptr;      // [n][4][256] symbol counts in header order: lum DC, lum AC, chr DC, chr AC
    HuffTables* tables = nullptr;  // [n] code words of image i
    uint8_t* spec = nullptr;       // [n][kHuffSpecBytes]
};

struct EntropyScratch {
    uint16_t* len16;        // [n][mcus*3]      bits per block-channel
    const uint8_t* chunk_masks;  // [n][mcus*3] non-zero 16-byte chunks of each block, or null = load everything
    bool planar;            // coefficients are in the chunk-planar layout (see launch_fdct_quant); needs chunk_masks
    ComponentMode mode;
    uint32_t* slots;        // [n][n_tiles][kSlotWords][kSlotStride]  per-block bit strings
    uint32_t* tile_bits;    // [n][n_tiles]
    uint64_t* tile_base;    // [n][n_tiles]     exclusive scan of tile_bits within the image
    uint32_t* tile_ff;      // [n][n_tiles]     0xFF bytes among the tile's owned bytes
    uint64_t* tile_ffbase;  // [n][n_tiles]     exclusive scan of tile_ff within the image
    uint64_t* img_bits;     // [n]
    uint64_t* img_ubase;    // [n]  byte offset of the image's unstuffed stream in `unstuffed`
    uint64_t* img_ff;       // [n]
    uint8_t* unstuffed;     // packed, not yet byte-stuffed streams
    size_t unstuffed_capacity;
    uint32_t* tile_ff8;     // [n_images][n_tiles] x 16 bytes: 0xFF bytes a tile owns for each of the eight bit
                            // alignments it may get, as 16-bit counts
    hipEvent_t* marks = nullptr;   // timing marks (null = none): launch_entropy_encode records marks[0] after the
                                   // bit-string kernel; the caller records what comes before and after
    OptimalHuffman optimal;
    // Passes of one batch on two streams (jea_encode_batch_async): what the stream waits for before the kernels that
    // need the file offset the pass before this one ended at (null = nothing to wait for)
    hipEvent_t wait_before_offsets = nullptr;
    // ... and what it records once this pass's own offsets and sizes are final (before its files are written)
    hipEvent_t record_after_offsets = nullptr;
};

// d_status_to_clear (may be null): status word of the encode pass this launch opens; the kernel zeroes it.
// d_chunk_masks (may be null): one byte per block-channel, bit g set iff zig-zag coefficients 8g..8g+7 are not
// all zero -- lets the entropy pass skip loading the (usually many) all-zero 16-byte chunks.
// planar (needs d_chunk_masks): instead of [image][mcu][channel][64], 16-byte chunk c (zig-zag positions
// 8c..8c+7) of block (mcu, channel) of image i goes to 16-byte element ((i * 8 + c) * 3 + channel) * mcus + mcu,
// and only chunk 0 and the chunks flagged in the mask are written at all -- the entropy pass then touches a
// fraction of the memory.  Internal to the fused encode path.
void launch_fdct_quant(hipStream_t s, const uint8_t* d_rgb, size_t n_images, int width, int height,
                       const QuantTables& q, int16_t* d_coefs, int32_t* d_status_to_clear, uint8_t* d_chunk_masks,
                       bool planar = false, ComponentMode mode = ComponentMode{});

void launch_bit_lengths(hipStream_t s, const int16_t* d_coefs, size_t n_images, size_t mcus_per_image,
                        const HuffTables* d_huff, uint16_t* d_len16, uint32_t* d_tile_bits, int32_t* d_status,
                        ComponentMode mode = ComponentMode{});

// bit-length pass -> scans -> pack -> stuff.  header_len is 0 or 607; with_eoi appends FFD9.
// d_prev_offset / d_prev_size (null, or both given): a batch encoded in several passes -- offset and size (device
// memory, written by the pass before) of the file after which this pass's files continue at 64-byte steps.
void launch_entropy_encode(hipStream_t s, const int16_t* d_coefs, size_t n_images, size_t mcus_per_image,
                           const HuffTables* d_huff, const EntropyScratch& sc, const HeaderBytes& header,
                           int header_len, bool with_eoi, uint8_t* d_out, size_t out_capacity,
                           uint64_t* d_offsets, uint64_t* d_sizes, int32_t* d_status,
                           const uint64_t* d_prev_offset = nullptr, const uint64_t* d_prev_size = nullptr);

// Stripe encode in two steps around the exchange of the stripes' bit totals:
// bits: block bit strings (DC chain continued from st.initial_dc), per-tile scan,
//       total -> sc.img_bits[0], first 32 stream bits -> d_head;
// finish: pack at st.bit_base, 0xFF scan, stuffed segment (+ header / EOI) -> d_out, size -> d_sizes[0].
// The facts that travel between the stripes' owners may stay in device memory (so that a rank enqueues all steps and
// the exchanges between them without waiting for the GPU):
//   launch_stripe_last_dc: the last MCU's quantised DC values -> d_last_dc[0..2] (d_last_dc[3] = 0);
//   d_initial_dc (may be null): the DC start in device memory instead of st.initial_dc;
//   d_facts (may be null): {bit count, head bits} of the stripe as two 64-bit words;
//   launch_stripe_place: from all stripes' facts ([n_stripes][2]) the StripeInfo of stripe `index` -> *d_st;
//   d_st (may be null): the StripeInfo in device memory instead of st.
void launch_stripe_last_dc(hipStream_t s, const int16_t* d_coefs, size_t mcus, int16_t* d_last_dc);
void launch_stripe_bits(hipStream_t s, const int16_t* d_coefs, size_t mcus, const HuffTables* d_huff,
                        const EntropyScratch& sc, const StripeInfo& st, uint32_t* d_head, int32_t* d_status,
                        const int16_t* d_initial_dc = nullptr, uint64_t* d_facts = nullptr);
void launch_stripe_place(hipStream_t s, const uint64_t* d_all_facts, uint32_t index, uint32_t n_stripes, StripeInfo* d_st);
void launch_stripe_finish(hipStream_t s, size_t mcus, const EntropyScratch& sc, const StripeInfo& st,
                          const HeaderBytes& header, int header_len, bool with_eoi, uint8_t* d_out,
                          size_t out_capacity, uint64_t* d_offsets, uint64_t* d_sizes, int32_t* d_status,
                          const StripeInfo* d_st = nullptr);

// Parallel Huffman decode (decode.hip).  The scan of image i is the byte range
// [d_offsets[i], d_offsets[i] + d_sizes[i]) of d_scan and ends with the EOI marker.
// The plan is ragged: every image gets the workgroups its own scan needs -- groups_of_scan(bytes) of them, each owning
// a fixed range of raw scan bytes -- and the job's workgroups are numbered through, image after image.  Whatever
// is kept per workgroup or per sub-sequence is indexed by that running number, so one file much longer than the
// others costs its own workgroups and nothing else.
struct DecodePlan {
    uint32_t n_images = 0;
    uint32_t n_chunks = 0;      // 4096-byte pieces of the longest raw scan
    uint32_t sub_words = 0;     // 32-bit words of unstuffed stream per sub-sequence (one thread's share)
    uint32_t n_groups = 0;      // workgroups covering the longest scan
    uint32_t total_groups = 0;  // workgroups of the whole job: the sum over the images
    size_t subs_per_image = 0;  // sub-sequence records of the longest scan
    size_t total_subs = 0;      // sub-sequence records of the whole job
    uint32_t total_blocks = 0;  // block-channels per image
    bool records_in_fours = false;   // the write pass stores block records as aligned 16-byte fours
    // More than kPlanScanImages images: the kernels find a workgroup's image in a table made on the host
    // (fill_group_map); up to that many they add up the images' workgroup counts themselves.
    bool mapped = false;
    bool dense = false;         // more than 64 bits of stream per block-channel: long sub-sequences, four rounds
};
constexpr uint32_t kPlanScanImages = 16;
// Raw scan bytes that one workgroup owns, for sub-sequences of sub_words words, and workgroups of a scan.
uint32_t decode_group_bytes(uint32_t sub_words);
__host__ __device__ inline uint32_t groups_of_scan(uint64_t scan_bytes, uint32_t group_bytes) {
    const uint64_t g = (scan_bytes + group_bytes - 1) / group_bytes;
    return g ? (uint32_t)g : 1u;
}
// scan_sizes: raw scan bytes per image (null: every image max_scan_bytes long).
DecodePlan plan_entropy_decode(size_t n_images, const uint64_t* scan_sizes, size_t max_scan_bytes, size_t mcus_per_image,
                               size_t total_scan_bytes = 0);
// For a mapped plan: img_first[i] = number of image i's first workgroup; map[2 g], map[2 g + 1] = image and
// workgroup-within-the-image of workgroup g.  Images in the order given (the kernels do not care).
void fill_group_map(const DecodePlan& plan, const uint64_t* scan_sizes, uint32_t* img_first, uint32_t* map);

// Caller-provided scratch; n = images, subs = DecodePlan::total_subs, groups = DecodePlan::total_groups.
struct DecodeScratch {
    const uint32_t* grp_map = nullptr;     // mapped plans: device copies of what fill_group_map made
    const uint32_t* img_first = nullptr;
    const uint64_t* ubase;   // [n] byte offset (multiple of 16) of image i's unstuffed stream in `ustream`;
                             //     streams need as many bytes as their raw scans, rounded up to 16
    uint8_t* ustream;
    uint32_t* chunk_drops;   // [n][n_chunks] stuffed zeros per raw chunk
    uint64_t* chunk_base;    // [n][n_chunks] exclusive scan
    uint64_t* img_drops;     // [n]
    uint32_t* sub_end;       // [subs] decoder state at the end of each sub-sequence
    uint32_t* sub_used;      // [subs] state the sub-sequence was last decoded from
    int4* sub_info;          // [subs] blocks completed, DC-difference sums of Y, Cb, Cr
    int4* grp_sum;           // [groups]
    int4* grp_base;          // [groups] exclusive scan within each image
    uint32_t* changed;       // [4] round flags
    uint32_t* grp_pending;   // [groups] workgroup ran out of repair steps in the last round
    uint32_t* grp_end;       // [groups] raw decoder: end state of the group's last sub-sequence
    bool tables_per_image = false;   // the StepTables pointer given to the launch functions is an array, one per image
    uint32_t* sub_guess;     // [subs] start states offered from the stream context
    uint32_t* first_bad;     // [n] first sub-sequence that is not proven yet
    unsigned long long* ctx_keys;   // [ctx_slots] context hash table (power-of-two size): keys ...
    uint32_t* ctx_vals;             // ... and the start states remembered under them
    uint32_t ctx_slots;
    ComponentMode mode;
    uint32_t* out_records;   // null: coefficients go out as [block][64]; else they go out as block records
                             // [n][mcus*3] plus chunk planes, see launch_idct_dequant
    hipEvent_t* marks = nullptr;   // timing marks (null = none): launch_entropy_decode records marks[0] after
                                   // unstuffing, [1] after the two synchronisation rounds, [2] after scan and
                                   // write pass, [3] after follow_up
};

// Small jobs the decoder's first kernel does on the side (both optional).  A launch at the head of the chain
// costs more than its run time, because there the GPU waits for the host.
constexpr uint32_t kSeedWords = 48;   // = 3 * kPlanScanImages: jobs that small need no table either
struct DecodePrologue {
    // The scan ranges as kernel arguments instead of a copy (which on a stream costs the latency of the copy
    // engine): 3 * n_images <= kSeedWords words -- offsets | sizes | unstuffed bases -- that the kernel uses and
    // leaves at d_offsets, where d_sizes = d_offsets + n and sc.ubase = d_offsets + 2 n must point.  0: the
    // arrays are in device memory already.
    const uint64_t* h_words = nullptr;
    uint32_t n_words = 0;
    // The header_bytes bytes in front of every scan, copied to headers_out back to back ([n][header_bytes];
    // may be pinned host memory: the kernel then delivers the headers itself).  null: none.
    uint8_t* headers_out = nullptr;
    uint32_t header_bytes = 0;
};

// What the decoder tells the host, in host memory that the GPU can write (hipHostMalloc).
struct DecodeReport {
    uint32_t unsettled;   // the last of the standard synchronisation rounds still changed something
    int32_t status;       // *d_status after the write pass
};
// What has to be copied into the report once the write pass has run.  The decoder hands this to the follow-up
// work: a kernel launched behind the write pass can do the copy on the side (one thread, two words), which
// saves the launch of a kernel that does nothing else.
struct DecodeReportRelay {
    const uint32_t* unsettled = nullptr;
    const int32_t* status = nullptr;
    DecodeReport* report = nullptr;   // null: nothing to relay
};
// Work that consumes the coefficients, enqueued behind the write pass.  Returns true when one of its kernels
// relays the report (see launch_idct_dequant); otherwise the decoder launches a small kernel for that.
using DecodeFollowUp = std::function<bool(const DecodeReportRelay&)>;

// Enqueues the whole decode -- two synchronisation rounds, (scan and) write pass -- then `follow_up` (work that
// consumes the coefficients; may be empty), then a copy of the outcome to *report, and returns without
// waiting.  Every block of d_coefs is written (chunk-planar: chunk 0 and the non-zero chunks); *d_status is
// cleared first and receives the errors.
//   raw = true   the kernels read the raw scans and drop the stuffed bytes on their way into LDS: no unstuffing
//                pass, no unstuffed copy in memory.  After the stream has been waited on, report->unsettled == 0
//                is the usual case and the coefficients are final; otherwise the stream needs repair rounds,
//                which only the other variant has: call again with raw = false.
//   raw = false  an unstuffing pass first (launch_unstuff), then the same on the unstuffed copy.  If
//                report->unsettled != 0 afterwards, repair_entropy_decode has to run.
hipError_t launch_entropy_decode(hipStream_t s, const uint8_t* d_scan, const uint64_t* d_offsets,
                                 const uint64_t* d_sizes, const DecodePlan& plan, const DecodeScratch& sc,
                                 const StepTables* d_tables, int16_t* d_coefs, int32_t* d_status,
                                 const DecodeFollowUp& follow_up, DecodeReport* report,
                                 const DecodePrologue& prologue, bool raw);

// One StepTables per image from the DHT segment of its file (d_scan_offsets[i] - header_bytes is where file i's
// header starts in d_data): for files written with per-image Huffman tables.  A segment that is not a usable
// set of tables gives tables in which no code decodes.
void launch_build_step_tables(hipStream_t s, const uint8_t* d_data, const uint64_t* d_scan_offsets, size_t n_images,
                              uint32_t header_bytes, StepTables* d_tables);

// The decoder's first step on its own: the unstuffed copy of every stream (sc.ustream at sc.ubase[i]) and the
// number of dropped bytes per stream (sc.img_drops).  Needs sc.chunk_drops, chunk_base, img_drops, ubase, ustream, changed.
hipError_t launch_unstuff(hipStream_t s, const uint8_t* d_scan, const uint64_t* d_offsets, const uint64_t* d_sizes,
                          const DecodePlan& plan, const DecodeScratch& sc, int32_t* d_status, const DecodePrologue& prologue);

// For streams that the standard rounds did not settle: more rounds (waiting on the stream after each),
// then the write pass, `follow_up` and the report again, and a final wait.
hipError_t repair_entropy_decode(hipStream_t s, const uint64_t* d_sizes, const DecodePlan& plan, const DecodeScratch& sc,
                                 const StepTables* d_tables, int16_t* d_coefs, int32_t* d_status,
                                 const DecodeFollowUp& follow_up, DecodeReport* report);

// d_records null: d_coefs is [image][mcu][channel][64].  Else the decoder's own layout: one 32-bit record
// per block-channel, [image][mcu*3 + channel] = DC value (low 16 bits) | chunk mask << 16, and the chunk
// planes of launch_fdct_quant, in which only the chunks whose mask bit is set exist.  Mask bit c >= 1: chunk c
// (zig-zag positions 8c..8c+7) has a non-zero coefficient; bit 0: one of positions 1..7 is non-zero (the
// stored chunk 0 repeats the DC value).  A block without AC coefficients is its record and nothing else:
// one 4-byte store for the decoder instead of a chunk and a mask byte.
// d_group_list (decoder's layout only; may be null): idct_group_list_bytes() bytes of scratch.  With it, and when the
// caller expects most MCUs to be single colours (DC values alone: smooth pictures at ordinary qualities), groups of 64
// MCUs that are mostly such MCUs are done by a kernel made for them and only the others by the general one.
void launch_idct_dequant(hipStream_t s, const int16_t* d_coefs, size_t n_images, int width, int height,
                         const QuantTables& q, uint8_t* d_rgb, const uint32_t* d_records = nullptr,
                         ComponentMode mode = ComponentMode{}, const DecodeReportRelay& relay = DecodeReportRelay{},
                         uint32_t* d_group_list = nullptr, bool mostly_single_colours = false);
size_t idct_group_list_bytes();

// Per-block (component granularity) launchers, blocks.hip.
void launch_dct_forward_blocks(hipStream_t s, const uint8_t* d_in, size_t n, float* d_out);
void launch_dct_inverse_blocks(hipStream_t s, const float* d_in, size_t n, uint8_t* d_out);
void launch_quantise_blocks(hipStream_t s, const float* d_in, size_t n, const QuantTables& q, bool use_lum,
                            int16_t* d_out);
void launch_dequantise_blocks(hipStream_t s, const int16_t* d_in, size_t n, const QuantTables& q, bool use_lum,
                              float* d_out);
void launch_huffman_encode_blocks(hipStream_t s, const int16_t* d_zz, const int16_t* d_prev_dc, const uint8_t* d_is_lum,
                                  size_t n, const HuffTables* d_huff, uint16_t* d_lengths, uint32_t* d_bits,
                                  int32_t* d_status);
void launch_huffman_decode_block(hipStream_t s, const uint8_t* d_stream, uint64_t len, uint64_t start_bit, bool is_lum,
                                 int last_dc, const DecodeTables* d_tables, int16_t* d_zz, uint64_t* d_end_bit,
                                 int32_t* d_last_dc, int32_t* d_status);

// Run-length form in and out (what the reference hands to / gets from its final-encoding step).
// d_bits must hold (22 + 26 * n + 31) / 32 words; d_runs / d_values of the decoder 64 entries each.
void launch_huffman_encode_rle(hipStream_t s, int dc_difference, const uint16_t* d_runs, const int16_t* d_values, uint32_t n,
                               bool is_lum, const HuffTables* d_huff, uint32_t* d_bits, uint32_t* d_length, int32_t* d_status);
void launch_huffman_decode_rle(hipStream_t s, const uint8_t* d_stream, uint64_t len, uint64_t start_bit, bool is_lum,
                               const DecodeTables* d_tables, int32_t* d_dc_difference, uint16_t* d_runs, int16_t* d_values,
                               uint32_t* d_n_entries, uint64_t* d_end_bit, int32_t* d_status);

void launch_rgb_to_ycbcr(hipStream_t s, const uint8_t* d_rgb, size_t n_pixels, uint8_t* d_ycc);
void launch_ycbcr_to_rgb(hipStream_t s, const uint8_t* d_ycc, size_t n_pixels, uint8_t* d_rgb);

}  // namespace jea
