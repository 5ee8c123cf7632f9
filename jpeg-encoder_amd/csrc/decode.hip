// Parallel Huffman decode of reference-format scans for gfx950.
//
// The reference reads one image's scan bit by bit from a single serial
// BitStream (jpeg/src/encoder.cpp:46-60, jpeg/src/entropy_encoder.cpp:703-712,
// 767-866); the stream has no restart markers (encoder.cpp:121-137), so nothing
// in the file says where a block starts.  Here the same coefficients are
// produced by many threads per image:
//
//   unstuff_count_kernel / scan_tiles / unstuff_compact_kernel
//       remove the stuffed bytes exactly as BitStream::removeStuffedBytes does
//       (bit_stream.cpp:136-147: a 0x00 that follows a 0xFF located before the
//       last two bytes of the stream is dropped), into a word-aligned copy
//   decode_sync_kernel (repeated until nothing changes)
//       the unstuffed stream is cut into sub-sequences of kSubBits bits, one per
//       thread.  A decoder state is (bit position, colour channel, zig-zag
//       progress inside the block).  Every thread first decodes its sub-sequence
//       from a guessed state; Huffman streams re-synchronise after a few
//       symbols, so its END state is usually right even when the guess was not.
//       Then each thread whose predecessor's end state differs from the state it
//       started from decodes again from that state, until no thread had to: at
//       that fixed point every sub-sequence has been decoded from the state the
//       serial decoder would have been in.  Each decode also counts the blocks
//       it completes and sums the DC differences it meets per channel.
//   scan_groups_kernel
//       exclusive scan of those counts and sums over the workgroups of an image
//   decode_write_kernel
//       one more decode from the now known states, this time with the block
//       index and the running DC of each channel known at the start of every
//       sub-sequence; every block is stored, in whole 16-byte chunks, by the
//       thread in whose sub-sequence it starts
//   context_* kernels (only for streams that do not re-synchronise)
//       in a stream that repeats with a short period (flat or regularly patterned
//       pictures) a decoder in a wrong state can stay out of phase for ever, and
//       the repair would crawl through the stream one sub-sequence at a time.
//       There, the start states already proven at the head of the stream are
//       remembered under a hash of the 512 stream bits before them and offered as
//       guesses to the sub-sequences further on that are preceded by the same
//       bits; the same fixed-point check decides, so a wrong guess costs time only
//
// Run-length removal follows entropy_encoder.cpp:122-144 including its quirk:
// the last run-length entry of a block is never restored, so a block whose
// entries fill all 63 AC positions loses the final one.
#include <hip/hip_runtime.h>

#include "huffman_device.h"
#include "kernels.h"
#include "../../include/jpegenc_amd.h"

namespace jea {

namespace {

constexpr int kThreads = 256;
constexpr uint32_t kChunkBytes = 4096;            // raw bytes per workgroup of the unstuffing passes
constexpr uint32_t kFoldChunks = 4096;            // streams of up to this many chunks skip the scan launch
constexpr uint32_t kNoState = 0xffffffffu;
// Sub-sequence sizes in 32-bit words (template parameter W below).  Short ones give small streams
// enough threads to fill the GPU; long ones keep the per-sub-sequence bookkeeping of big streams small.
// A symbol is at most 27 bits, well inside the smallest size.
constexpr uint32_t kSubWordsSmall = 2, kSubWordsMedium = 4, kSubWordsLarge = 16;
// A workgroup owns kOwned consecutive sub-sequences.  In the first round its first kWarm threads
// decode the last sub-sequences of the previous workgroup once more, only to hand the first owned
// sub-sequence a start state that has most likely synchronised already.
// (8: with 16 the sync rounds and the write pass of detailed and noisy batches took 3-8 % longer -- a sixteenth of the
// lanes own nothing --, with 4 a 4K picture of noise needs a quarter more repair steps)
#ifndef JEA_WARM_SLOTS
#define JEA_WARM_SLOTS 8
#endif
constexpr uint32_t kWarm = JEA_WARM_SLOTS;
constexpr uint32_t kOwned = kThreads - kWarm;

// Which image a workgroup belongs to and which of that image's groups it is.  A job's workgroups are numbered
// through, image after image, each image having groups_of_scan(its raw scan bytes) of them; `first` is the number
// of the image's group 0.  Everything kept per group or per sub-sequence is indexed by the running number.
struct GroupMap {
    const uint2* map;            // jobs of more than kPlanScanImages images: (image, group) of every workgroup, from the host
    const uint32_t* img_first;   // ... and every image's first workgroup
    uint32_t n_images;
};
struct GroupId {
    uint32_t img, group, first;
};
// size_of(i): raw scan bytes of image i (from device memory, or from the kernel's arguments)
template <class SizeOf>
__device__ __forceinline__ GroupId locate_group(const GroupMap& gm, uint32_t wg, uint32_t group_raw, SizeOf size_of) {
    if (gm.map != nullptr) {
        const uint2 e = gm.map[wg];
        return GroupId{e.x, e.y, wg - e.y};
    }
    uint32_t first = 0, img = 0;
    for (; img + 1 < gm.n_images; ++img) {   // a few images at most: wave-uniform scalar work
        const uint32_t g = groups_of_scan(size_of(img), group_raw);
        if (wg < first + g) break;
        first += g;
    }
    return GroupId{img, wg - first, first};
}
template <class SizeOf>
__device__ __forceinline__ uint32_t first_group_of(const GroupMap& gm, uint32_t img, uint32_t group_raw, SizeOf size_of) {
    if (gm.img_first != nullptr) return gm.img_first[img];
    uint32_t first = 0;
    for (uint32_t i = 0; i < img; ++i) first += groups_of_scan(size_of(i), group_raw);
    return first;
}

// state word: bits 0-7 bit offset past the sub-sequence start, 8-9 channel, 16-21 zig-zag progress
__device__ __forceinline__ uint32_t make_state(uint32_t ovf, uint32_t c, uint32_t z) { return ovf | (c << 8) | (z << 16); }

struct Sums {
    int nb, dc0, dc1, dc2;
};
__device__ __forceinline__ Sums operator+(Sums a, Sums b) { return {a.nb + b.nb, a.dc0 + b.dc0, a.dc1 + b.dc1, a.dc2 + b.dc2}; }

__device__ __forceinline__ Sums shfl_up(Sums v, int d) {
    return {__shfl_up(v.nb, d), __shfl_up(v.dc0, d), __shfl_up(v.dc1, d), __shfl_up(v.dc2, d)};
}

// Exclusive scan over the 256 threads of a workgroup; `total` gets the sum.
__device__ __forceinline__ Sums block_exclusive_scan(Sums v, Sums* lds4, Sums& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    Sums incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const Sums t = shfl_up(incl, d);
        if (lane >= d) incl = incl + t;
    }
    if (lane == 63) lds4[wave] = incl;
    __syncthreads();
    Sums base{0, 0, 0, 0};
    for (int w = 0; w < wave; ++w) base = base + lds4[w];
    total = lds4[0] + lds4[1] + lds4[2] + lds4[3];
    __syncthreads();
    return {base.nb + incl.nb - v.nb, base.dc0 + incl.dc0 - v.dc0, base.dc1 + incl.dc1 - v.dc1,
            base.dc2 + incl.dc2 - v.dc2};
}

// ---------------------------------------------------------------------------
// Unstuffing
// ---------------------------------------------------------------------------

// Bit k of the result is set when raw byte j0 + k is a stuffed zero, as bit_stream.cpp:136-147 sees it.  The
// reference walks the scan serially: at a 0xFF it looks at the next byte, erases it when it is 0x00, and steps
// over it when it is not -- so the byte after a 0xFF is never itself taken as the 0xFF of a pair.  In a run of
// consecutive 0xFF bytes every second one is therefore passed over, and a 0x00 after the run goes exactly
// when the run's length is odd.  Streams this encoder writes only ever have runs of one (every 0xFF is followed
// by its 0x00); the rule matters for damaged streams, which must decode, or fail, as the reference's would.
// The run that ends just before this piece is counted backwards, 4096 bytes at most: an unstuffed stream
// with that many one-bits in a row holds no valid code word, so such a stream fails either way.
// Length, modulo 2, of the run of 0xFF bytes that ends at byte j0 - 1.
__device__ __forceinline__ uint32_t ff_run_parity_before(const uint8_t* __restrict__ file, uint64_t len, uint64_t j0) {
    uint32_t odd = 0;
    if (j0 >= 1 && j0 - 1 < len && file[j0 - 1] == 0xFFu) {
        uint32_t run = 1;
        while (run < 4096u && run < j0 && file[j0 - 1 - run] == 0xFFu) ++run;
        odd = run & 1u;
    }
    return odd;
}

// drop_mask16 for a piece that lies inside the file and is already in registers (byte k of the piece is byte
// k & 3 of word k >> 2, as loaded); `odd` = ff_run_parity_before(j0).
__device__ __forceinline__ uint32_t drop_mask16_loaded(uint4 piece, uint32_t odd, uint64_t j0, uint64_t scan_end,
                                                       uint8_t* bytes) {
    const uint32_t w[4] = {piece.x, piece.y, piece.z, piece.w};
    uint32_t mask = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const uint32_t b = (w[k >> 2] >> (8 * (k & 3))) & 0xffu;
        bytes[k] = (uint8_t)b;
        if (b == 0u && odd && j0 + k - 1 < scan_end) mask |= 1u << k;
        odd = (b == 0xFFu) ? (odd ^ 1u) : 0u;
    }
    return mask;
}

__device__ __forceinline__ uint32_t drop_mask16(const uint8_t* __restrict__ file, uint64_t len, uint64_t scan_end,
                                                uint64_t j0, uint8_t* bytes) {
    uint32_t odd = ff_run_parity_before(file, len, j0);   // of the run of 0xFF bytes that ends at the previous byte
    uint32_t mask = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const uint64_t j = j0 + k;
        const uint32_t b = j < len ? file[j] : 0x01u;
        bytes[k] = (uint8_t)b;
        // :137-138: only a 0xFF before the last two bytes makes the next zero vanish
        if (j < len && b == 0u && odd && j - 1 < scan_end) mask |= 1u << k;
        odd = (b == 0xFFu) ? (odd ^ 1u) : 0u;
    }
    return mask;
}

struct SegmentWords {
    uint64_t v[kSeedWords];
};

// The decoder's first kernel.  Besides counting it does the small jobs that would otherwise be launches of
// their own at the head of the chain, where the GPU waits for the host: it clears what the later kernels
// report into; with n_words != 0 the scan ranges arrive as kernel arguments (offsets | sizes | unstuffed
// bases, n images each) and are left in d_words for the kernels that follow; with headers_out given it
// copies the header_bytes bytes in front of every scan there (pinned host memory: the host validates all
// headers after the one wait of the call, and no copy sits in the stream between the decoder's kernels).
__global__ __launch_bounds__(kThreads) void unstuff_count_kernel(const uint8_t* __restrict__ scan,
                                                                 const uint64_t* __restrict__ offsets,
                                                                 const uint64_t* __restrict__ sizes,
                                                                 uint32_t n_chunks,
                                                                 uint32_t* __restrict__ chunk_drops,
                                                                 uint32_t* __restrict__ round_flags,
                                                                 int32_t* __restrict__ status,
                                                                 SegmentWords words, uint32_t n_words,
                                                                 uint64_t* __restrict__ d_words,
                                                                 uint8_t* __restrict__ headers_out,
                                                                 uint32_t header_bytes) {
    __shared__ uint32_t wsum[4];
    const uint32_t chunk = blockIdx.x, img = blockIdx.y;
    if (chunk == 0 && img == 0 && threadIdx.x < 4) round_flags[threadIdx.x] = 0;
    if (chunk == 0 && img == 0 && threadIdx.x == 4) *status = 0;
    if (chunk == 0 && img == 0 && threadIdx.x < n_words) d_words[threadIdx.x] = words.v[threadIdx.x];
    const uint64_t offset = n_words ? words.v[img] : offsets[img];
    const uint64_t len = n_words ? words.v[gridDim.y + img] : sizes[img];
    if (headers_out != nullptr && chunk == 0) {
        const uint8_t* src = scan + offset - header_bytes;
        for (uint32_t i = threadIdx.x; i < header_bytes; i += kThreads) headers_out[(size_t)img * header_bytes + i] = src[i];
    }
    const uint64_t j0 = (uint64_t)chunk * kChunkBytes + 16u * threadIdx.x;
    uint32_t n = 0;
    if (j0 < len) {
        uint8_t bytes[16];
        n = __popc(drop_mask16(scan + offset, len, len >= 2 ? len - 2 : 0, j0, bytes));
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) n += __shfl_xor(n, d);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) chunk_drops[(size_t)img * n_chunks + chunk] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// kFold (streams of few chunks): no scan launch in between; every workgroup adds up the drop counts of
// the chunks before it, and the one with the stream's last chunk also leaves the image's total.
template <bool kFold>
__global__ __launch_bounds__(kThreads) void unstuff_compact_kernel(const uint8_t* __restrict__ scan,
                                                                   const uint64_t* __restrict__ offsets,
                                                                   const uint64_t* __restrict__ sizes,
                                                                   uint32_t n_chunks,
                                                                   const uint32_t* __restrict__ chunk_drops,
                                                                   const uint64_t* __restrict__ chunk_base,
                                                                   uint64_t* __restrict__ img_drops,
                                                                   const uint64_t* __restrict__ ubase,
                                                                   uint8_t* __restrict__ ustream) {
    __shared__ __align__(16) uint8_t kept[kChunkBytes + 4];
    __shared__ uint32_t wsum[4];
    __shared__ uint64_t fold4[4];
    const uint32_t chunk = blockIdx.x, img = blockIdx.y;
    const uint64_t len = sizes[img];
    const uint64_t c0 = (uint64_t)chunk * kChunkBytes;
    if (c0 >= len) {
        if (kFold && chunk == 0 && threadIdx.x == 0) img_drops[img] = 0;   // an empty stream
        return;
    }
    uint64_t drops_before = 0;
    if (kFold) {
        for (uint32_t i = threadIdx.x; i < chunk; i += kThreads) drops_before += chunk_drops[(size_t)img * n_chunks + i];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) drops_before += __shfl_xor(drops_before, d);
        if ((threadIdx.x & 63) == 0) fold4[threadIdx.x >> 6] = drops_before;
        __syncthreads();
        drops_before = fold4[0] + fold4[1] + fold4[2] + fold4[3];
        if (c0 + kChunkBytes >= len && threadIdx.x == 0)
            img_drops[img] = drops_before + chunk_drops[(size_t)img * n_chunks + chunk];
    } else {
        drops_before = chunk_base[(size_t)img * n_chunks + chunk];
    }
    const uint64_t j0 = c0 + 16u * threadIdx.x;
    uint8_t bytes[16];
    uint32_t mask = 0, valid = 0;
    if (j0 < len) {
        mask = drop_mask16(scan + offsets[img], len, len >= 2 ? len - 2 : 0, j0, bytes);
        valid = (uint32_t)((len - j0 < 16) ? len - j0 : 16);
    }
    const uint32_t drops = __popc(mask);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = drops;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t before = incl - drops;
    for (int w = 0; w < wave; ++w) before += wsum[w];
    const uint32_t total_drops = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    uint32_t o = 16u * threadIdx.x - before;
    for (uint32_t k = 0; k < valid; ++k)
        if (!((mask >> k) & 1u)) kept[o++] = bytes[k];
    __syncthreads();
    const uint32_t in_chunk = (uint32_t)((len - c0 < kChunkBytes) ? len - c0 : kChunkBytes);
    const uint32_t n_kept = in_chunk - total_drops;
    uint8_t* dst = ustream + ubase[img] + c0 - drops_before;
    // bytes up to the first 4-byte boundary of the destination, then whole words put together from two
    // aligned LDS words each, then the last bytes
    uint32_t head = (uint32_t)((4u - (uint32_t)(reinterpret_cast<uintptr_t>(dst) & 3u)) & 3u);
    head = head < n_kept ? head : n_kept;
    if (threadIdx.x < head) dst[threadIdx.x] = kept[threadIdx.x];
    const uint32_t n_words = (n_kept - head) >> 2;
    const uint32_t* k32 = reinterpret_cast<const uint32_t*>(kept);
    uint32_t* d32 = reinterpret_cast<uint32_t*>(dst + head);
    for (uint32_t q = threadIdx.x; q < n_words; q += kThreads) {
        const uint32_t from = head + 4u * q;
        d32[q] = __builtin_amdgcn_alignbyte(k32[(from >> 2) + 1], k32[from >> 2], from & 3u);
    }
    const uint32_t rest = head + 4u * n_words + threadIdx.x;
    if (rest < n_kept) dst[rest] = kept[rest];
}

// ---------------------------------------------------------------------------
// Sub-sequence decoding
// ---------------------------------------------------------------------------

// The words of one workgroup's 256 sub-sequences plus a tail, as MSB-first values: a dummy word, then the stream
// words.  No padding between the sub-sequences: the threads start on the same word of their own sub-sequences (which
// lie W words apart: a 16-way bank conflict, once) but are at unrelated words after a symbol or two, padded or not,
// and the two instructions a padded index costs at every step of a walk were 1-2 % of the passes' time.
// kTail: words staged past the group; the window below looks three words past the one that holds the
// last bit of a sub-sequence.
template <uint32_t W, uint32_t kTail = 3>
struct GroupWords {
    static constexpr uint32_t kWords = kThreads * W + kTail;   // staged stream words
    uint32_t w[kWords + 2];
    static __device__ __forceinline__ uint32_t slot(uint32_t index) { return index; }   // index 0 = the dummy
};

// first_sub: sub-sequence held by the first slot; negative slots (before the stream) read as zero.
template <uint32_t W, uint32_t kTail>
__device__ __forceinline__ void stage_group(GroupWords<W, kTail>& gw, const uint8_t* __restrict__ stream, uint64_t ulen,
                                            int64_t first_sub) {
    using G = GroupWords<W, kTail>;
    const int64_t first_word = first_sub * (int64_t)W;
    if (threadIdx.x == 0) gw.w[0] = 0;
    // four loads under way at a time
    for (uint32_t i0 = 0; i0 < G::kWords; i0 += 4 * kThreads) {
        uint32_t v[4];
#pragma unroll
        for (uint32_t r = 0; r < 4; ++r) {
            const uint32_t i = i0 + r * kThreads + threadIdx.x;
            const int64_t word = first_word + i;
            const uint64_t byte = (uint64_t)word * 4;
            v[r] = 0;
            if (i >= G::kWords || word < 0) {
            } else if (byte + 4 <= ulen) {
                v[r] = __builtin_bswap32(*reinterpret_cast<const uint32_t*>(stream + byte));
            } else {
                for (int k = 0; k < 4; ++k)
                    if (byte + k < ulen) v[r] |= (uint32_t)stream[byte + k] << (24 - 8 * k);
            }
        }
#pragma unroll
        for (uint32_t r = 0; r < 4; ++r) {
            const uint32_t i = i0 + r * kThreads + threadIdx.x;
            if (i < G::kWords) gw.w[G::slot(i + 1)] = v[r];
        }
    }
}

// The same staging straight from the raw scan, for the decoder that works without an unstuffed copy in
// memory.  There a workgroup owns a fixed range of RAW bytes, [group_raw0, group_raw0 + kOwned * 4W); the
// stuffed zeros in it are dropped on the way into LDS, so the group holds a varying number of stream bytes and
// of sub-sequences (the last one may be short).  Bit 0 of the staged window is `warm_bytes` stream bytes before
// the group's first one (warm-up sub-sequences of the first synchronisation round; 0 otherwise): twice as many
// raw bytes are examined in front of the group, since at most every second raw byte is a stuffed zero.
// Behind the group the window takes whatever 2 * 4 * kTail raw bytes yield, at least the kTail words a walk
// may look at.
struct RawGroup {
    uint32_t group_bytes;    // stream bytes of the group's own raw range
    uint32_t window_bytes;   // stream bytes from the group's first one to the end of what was examined
    bool end_in_window;      // the scan ends inside the examined range: window_bytes then ends the stream
};
template <uint32_t W, uint32_t kTail>
__device__ __forceinline__ RawGroup stage_group_raw(GroupWords<W, kTail>& gw, const uint8_t* __restrict__ scan, uint64_t len,
                                                    uint64_t group_raw0, uint32_t warm_bytes, uint32_t* __restrict__ scratch) {
    using G = GroupWords<W, kTail>;
    constexpr uint32_t kGroupRaw = kOwned * 4 * W;
    constexpr uint32_t kTotalWords = sizeof(gw.w) / 4;
    const uint32_t tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    for (uint32_t i = tid; i < kTotalWords; i += kThreads) gw.w[i] = 0;
    const uint64_t raw0 = group_raw0 - 2 * (uint64_t)warm_bytes;            // callers give warm-up only where this is >= 0
    const uint32_t warm_pieces = 2 * warm_bytes / 16, group_pieces = kGroupRaw / 16;
    const uint32_t pieces = warm_pieces + group_pieces + (2 * 4 * kTail + 15) / 16;
    const uint64_t scan_end = len >= 2 ? len - 2 : 0;
    uint32_t carry = 0, kept_warm = 0, kept_to_group_end = 0;
    int32_t shift = 0;   // stream-byte index in the examined range -> byte index in the window
    lds_barrier();       // the zeros are in place
    // A round's piece (and the byte before it) is asked for one round ahead; the barriers of a round order LDS only,
    // so that load stays under way across them.
    uint4 ahead = make_uint4(0, 0, 0, 0);
    uint32_t ahead_before = 0;
    auto ask = [&](uint32_t piece) {
        const uint64_t j0 = raw0 + 16ull * piece;
        if (piece < pieces && j0 + 16 <= len) {
            __builtin_memcpy(&ahead, scan + j0, 16);
            ahead_before = j0 > 0 ? scan[j0 - 1] : 0u;
        }
    };
    ask(tid);
    for (uint32_t p0 = 0; p0 < pieces; p0 += kThreads) {
        const uint32_t piece = p0 + tid;
        const uint64_t j0 = raw0 + 16ull * piece;
        uint8_t bytes[16] = {};
        uint32_t mask = 0, valid = 0;
        // Nearly every piece lies inside the scan and holds no 0xFF byte: nothing to drop, and its sixteen
        // bytes go out as whole words.  (The test may take a piece without 0xFF for one with: that is only slower.)
        const uint4 raw16 = ahead;
        const bool after_ff = ahead_before == 0xFFu;
        if (p0 + kThreads < pieces) ask(piece + kThreads);
        bool plain = false;
        if (piece < pieces && j0 + 16 <= len) {
            const uint32_t n0 = ~raw16.x, n1 = ~raw16.y, n2 = ~raw16.z, n3 = ~raw16.w;   // a zero byte here was 0xFF
            const uint32_t any_ff = (((n0 - 0x01010101u) & raw16.x) | ((n1 - 0x01010101u) & raw16.y) |
                                     ((n2 - 0x01010101u) & raw16.z) | ((n3 - 0x01010101u) & raw16.w)) & 0x80808080u;
            plain = any_ff == 0 && !after_ff;
            valid = 16;
            if (!plain) mask = drop_mask16_loaded(raw16, after_ff ? ff_run_parity_before(scan, len, j0) : 0u, j0, scan_end, bytes);
        } else if (piece < pieces && j0 < len) {
            mask = drop_mask16(scan, len, scan_end, j0, bytes);
            valid = (uint32_t)(len - j0);
        }
        const uint32_t kept = valid - __popc(mask);
        uint32_t incl = kept;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = __shfl_up(incl, d);
            if (lane >= d) incl += t;
        }
        // one barrier a round: the waves' sums go to one of two sets of four words in turn (a set is written again two
        // rounds on, behind the barrier of the round between)
        uint32_t* const sums = scratch + ((p0 / kThreads) & 1u) * 4u;
        if (lane == 63) sums[wave] = incl;
        lds_barrier();
        uint32_t before = carry + incl - kept;
        for (int w = 0; w < wave; ++w) before += sums[w];
        const uint32_t round_total = sums[0] + sums[1] + sums[2] + sums[3];
        // stream bytes before the group's first raw byte, and up to the end of its raw range: both boundaries
        // fall between pieces; the warm-up pieces all lie in the first round, which needs the first figure at once
        if (piece == warm_pieces) scratch[8] = before;
        if (piece == warm_pieces + group_pieces) scratch[9] = before;
        if (p0 == 0) {
            lds_barrier();
            kept_warm = scratch[8];
            shift = (int32_t)warm_bytes - (int32_t)kept_warm;
        }
        int32_t q = (int32_t)before + shift;
        if (plain && q >= 0 && (uint32_t)q + 16u <= G::kWords * 4u) {
            // words of the window are MSB first: the piece's bytes as four such words, shifted to where they start
            const uint32_t d0 = __builtin_bswap32(raw16.x), d1 = __builtin_bswap32(raw16.y), d2 = __builtin_bswap32(raw16.z),
                           d3 = __builtin_bswap32(raw16.w);
            const uint32_t w0 = ((uint32_t)q >> 2) + 1, sh = 8u * ((uint32_t)q & 3u);
            atomicOr(&gw.w[G::slot(w0)], __builtin_amdgcn_alignbit(0u, d0, sh));
            atomicOr(&gw.w[G::slot(w0 + 1)], __builtin_amdgcn_alignbit(d0, d1, sh));
            atomicOr(&gw.w[G::slot(w0 + 2)], __builtin_amdgcn_alignbit(d1, d2, sh));
            atomicOr(&gw.w[G::slot(w0 + 3)], __builtin_amdgcn_alignbit(d2, d3, sh));
            if (sh) atomicOr(&gw.w[G::slot(w0 + 4)], __builtin_amdgcn_alignbit(d3, 0u, sh));
        } else {
            if (plain) {
#pragma unroll
                for (int k = 0; k < 16; ++k) bytes[k] = (uint8_t)((k < 4 ? raw16.x : (k < 8 ? raw16.y : (k < 12 ? raw16.z : raw16.w))) >> (8 * (k & 3)));
            }
            // (unrolled: with k a variable, bytes[] is an array in scratch memory)
#pragma unroll
            for (uint32_t k = 0; k < 16; ++k) {
                if (k >= valid || ((mask >> k) & 1u)) continue;
                if (q >= 0 && (uint32_t)q < G::kWords * 4u)
                    atomicOr(&gw.w[G::slot(((uint32_t)q >> 2) + 1)], (uint32_t)bytes[k] << (8u * (3u - ((uint32_t)q & 3u))));
                ++q;
            }
        }
        carry += round_total;
    }
    lds_barrier();   // the second figure, and the window itself
    kept_to_group_end = scratch[9];
    RawGroup rg;
    const uint64_t examined_end = raw0 + 16ull * pieces;
    rg.end_in_window = len <= examined_end;
    rg.window_bytes = carry - kept_warm;
    rg.group_bytes = kept_to_group_end - kept_warm;   // pieces past the end of the scan keep nothing, so this holds there too
    return rg;
}

// The 64 stream bits around the read position, in registers, and the word after them on its way from
// LDS: the next symbol's bits never wait for a load.  With j = ceil(p / 32), `a` and `b` are stream
// words j-1 and j (word -1 being the dummy), which hold bits p .. p+31 for every p with that j.
template <class Group>
struct BitWindow {
    const Group& gw;
    uint32_t j, a, b, n;
    __device__ __forceinline__ BitWindow(const Group& g, uint32_t p) : gw(g) {
        j = (p + 31u) >> 5;
        a = gw.w[Group::slot(j)];
        b = gw.w[Group::slot(j + 1)];
        n = gw.w[Group::slot(j + 2)];
    }
    __device__ __forceinline__ uint32_t peek(uint32_t p) const { return __builtin_amdgcn_alignbit(a, b, 0u - p); }
    // p has moved on by less than 32 bits
    __device__ __forceinline__ void follow(uint32_t p) {
        const uint32_t j2 = (p + 31u) >> 5;
        const bool crossed = j2 != j;
        a = crossed ? b : a;
        b = crossed ? n : b;
        j = j2;
        n = gw.w[Group::slot(j2 + 2)];
    }
};

// All loads are issued before the first store: written as one loop, every piece waits for the one before it
// (six round trips to the L2 for the 23 KB).
__device__ __forceinline__ void load_tables(StepTables& t, const StepTables* __restrict__ g) {
    static_assert(sizeof(StepTables) % 16 == 0, "copied in 16-byte pieces");
    constexpr uint32_t kPieces = sizeof(StepTables) / 16, kPerThread = (kPieces + kThreads - 1) / kThreads;
    const uint4* src = reinterpret_cast<const uint4*>(g);
    uint4* dst = reinterpret_cast<uint4*>(&t);
    // (every element is loaded, from piece 0 where the thread has none: an array that is only partly assigned stays
    // in scratch memory, and the copy then goes load, wait, scratch store four times over and back through scratch loads)
    uint4 v[kPerThread];
#pragma unroll
    for (uint32_t r = 0; r < kPerThread; ++r) {
        const uint32_t i = r * kThreads + threadIdx.x;
        v[r] = src[i < kPieces ? i : 0u];
    }
    // (the values are made opaque here, all loads issued: left alone the compiler sinks every load into the branch of
    // its store -- load, wait, store, four trips to memory one after the other)
#pragma unroll
    for (uint32_t r = 0; r < kPerThread; ++r) asm volatile("" : "+v"(v[r].x), "+v"(v[r].y), "+v"(v[r].z), "+v"(v[r].w));
#pragma unroll
    for (uint32_t r = 0; r < kPerThread; ++r) {
        const uint32_t i = r * kThreads + threadIdx.x;
        if (i < kPieces) dst[i] = v[r];
    }
}

constexpr uint32_t kClassBytes = sizeof(StepClass);    // luminance tables first, then chrominance
constexpr uint32_t kAcBytes = sizeof(StepTableDc);     // DC table first, then AC
constexpr uint32_t kDcShift = 32 - StepTableDc::kFastBits, kAcShift = 32 - StepTableAc::kFastBits;

// `tab`: the table to use (in LDS), v: the next 32 stream bits, shift: 32 - the table's fast bits.
__device__ __forceinline__ uint32_t step_lookup(const uint8_t* tab, uint32_t v, uint32_t shift) {
    uint32_t e = *reinterpret_cast<const uint32_t*>(tab + ((v >> shift) << 2));
    // a longer code: the bits behind the index bits select the entry of a sub-table
    if ((e & 0xffu) == 0) e = *reinterpret_cast<const uint32_t*>(tab + (e >> 16) + (((v << (32u - shift)) >> ((e >> 8) & 0xffu)) << 2));
    return e;
}

// entropy_encoder.cpp:786-806, 845-864: the amplitude whose bits follow a code of `len` bits at the
// top of v; a leading 1 means the bits are the value, a leading 0 that they are the value's magnitude
// inverted.  Meaningless when the symbol has no amplitude bits.
__device__ __forceinline__ int step_amplitude(uint32_t v, uint32_t e) {
    const uint32_t x = v << ((e >> 16) & 31u);
    const uint32_t sg = (uint32_t)((int32_t)x >> 31);          // all ones: value is positive
    const uint32_t y = ~(x ^ sg) >> ((e >> 24) & 31u);          // positive: the bits; negative: their complement
    return (int)(~(y ^ sg) + sg + 1u);
}

__device__ __forceinline__ uint32_t class_offset(uint32_t lum_mask, uint32_t c) {
    return ((lum_mask >> c) & 1u) ? 0u : kClassBytes;
}
__device__ __forceinline__ uint32_t next_channel(uint32_t c) { return c == 2u ? 0u : c + 1u; }

// Decode from `state` (of sub-sequence `sub` of the staged group) to the first
// symbol boundary at or past the end of the sub-sequence, counting the blocks
// completed and summing the DC differences met.  Malformed symbols do not stop
// the walk (they count as one bit and one coefficient), so every walk ends.
//
// The loop body has no branches.  What depends on the channel -- table class, DC sum -- is kept in
// three registers each, rotated when a block ends, so that the current channel's is always the first.
// `limit`: the sub-sequence ends there at the latest (the end of a group that holds a short last sub-sequence).
// kSums = false: the end state and the block count only (the DC sums come back zero): a quarter fewer instructions a
// step, for the repair steps of streams that take many of them (see decode_sync_kernel).
template <uint32_t W, bool kSums = true>
__device__ __forceinline__ uint32_t walk(const GroupWords<W>& gw, const StepTables& t, uint32_t lum_mask,
                                         uint32_t sub, uint32_t state, Sums& sums, uint32_t limit = 0xffffffffu) {
    constexpr uint32_t kSubBits = 32 * W;
    uint32_t p = sub * kSubBits + (state & 0xffu);
    const uint32_t c0 = (state >> 8) & 3u;
    uint32_t z = (state >> 16) & 63u;
    const uint32_t end = min((sub + 1) * kSubBits, limit);
    const uint8_t* tb = reinterpret_cast<const uint8_t*>(&t);
    const uint32_t c1 = next_channel(c0), c2 = next_channel(c1);
    uint32_t o0 = class_offset(lum_mask, c0), o1 = class_offset(lum_mask, c1), o2 = class_offset(lum_mask, c2);
    uint32_t k = z ? kAcBytes : 0u;     // DC table at the start of a block, AC table inside
    uint32_t shift = z ? kAcShift : kDcShift;
    int nb = 0, a0 = 0, a1 = 0, a2 = 0;
    BitWindow<GroupWords<W>> win(gw, p);
#pragma unroll 2   // (two steps per turn: the values carried round the loop need no copies between them)
    while (p < end) {
        const uint32_t v = win.peek(p);
        const uint32_t e = step_lookup(tb + o0 + k, v, shift);
        p += e & 0xffu;
        // AC positions consumed including this entry, plus 128 when the entry ends the block
        const uint32_t processed = z + ((e >> 8) & 0xffu);
        if (__builtin_expect(processed == 128u + 63u, 0)) p -= (e >> kStepEobLenShift) & 7u;   // see kernels.h: combined entries
        win.follow(p);
#ifdef JEA_PROBE_WALK_PAD   // experiments: that many idle vector instructions a step -- is the walk bound by issue or by latency?
#pragma unroll
        for (int pad = 0; pad < JEA_PROBE_WALK_PAD; ++pad) asm volatile("v_mov_b32 %0, %0" : "+v"(nb));
#endif
        const bool done = processed >= 63u;                  // end of block, or the block is full
        z = done ? 0u : processed + 1u;
        k = done ? 0u : kAcBytes;
        shift = done ? kDcShift : kAcShift;
        if (kSums && (int32_t)e < 0) a0 += step_amplitude(v, e);   // kStepDcAdd (the sum inside the branch: no zero to add outside it)
        nb += done ? 1 : 0;
        const uint32_t o = o0;
        const int a = a0;
        o0 = done ? o1 : o0; o1 = done ? o2 : o1; o2 = done ? o : o2;
        if (kSums) { a0 = done ? a1 : a0; a1 = done ? a2 : a1; a2 = done ? a : a2; }
    }
    // (a0, a1, a2) belong to the channel the walk ended in and the two after it
    const uint32_t turns = (uint32_t)nb % 3u;
    const uint32_t c_end = (c0 + turns) % 3u;
    const int d0 = c_end == 0 ? a0 : (c_end == 1 ? a2 : a1);
    const int d1 = c_end == 1 ? a0 : (c_end == 2 ? a2 : a1);
    const int d2 = c_end == 2 ? a0 : (c_end == 0 ? a2 : a1);
    sums = Sums{nb, d0, d1, d2};
    return make_state(p - end, c_end, z);
}

// ---------------------------------------------------------------------------
// Write pass.  A block belongs to the thread in whose sub-sequence it STARTS; that
// thread decodes it to its end, also past the end of the sub-sequence (at most
// 1660 bits, which the staged window covers), and the thread after it skips the
// part it finds itself in.  So every block is produced by exactly one thread,
// which assembles it 16 bytes (8 zig-zag positions) at a time in a small LDS
// buffer of its own and stores whole chunks: no zero-filled output is needed and
// no two threads ever write to the same block.
//
// Output layouts: kPlanar = false: [block][64] int16, all eight chunks of every
// block are stored.  kPlanar = true (inside the library): one 32-bit record per
// block = DC value | chunk mask << 16, and chunk c of block (mcu, channel) as
// 16-byte element (c * 3 + channel) * mcus + mcu of the image's coefficient
// area -- only chunks with a non-zero AC coefficient are stored (mask bit c),
// so a block that is its DC value alone costs a single 4-byte store.  Stores
// are what this pass is made of: with a mask byte and a 16-byte chunk per block
// 1024 sparse 1080p frames took 2.5 ms, without the mask bytes 1.6 ms.
// ---------------------------------------------------------------------------

constexpr uint32_t kWriteTail = 56;   // words staged past the group: 1660 bits of overrun + 96 bits of look-ahead
constexpr uint32_t kAccWords = 6;     // per-thread chunk buffer: 16 bytes + padding against bank conflicts

struct WriteTarget {
    int16_t* out;            // coefficients of this image
    uint32_t* records;       // planar: block records of this image
    uint32_t n_mcu;
    uint32_t block;          // index of the block the start state is in
    uint32_t total_blocks;
    int last0, last1, last2; // running DC per channel at the start state
    bool error;              // a malformed symbol was met in a block of this thread
    bool finished;           // this thread completed the image's last block ...
    uint32_t final_bit;      // ... at this bit position (relative to the group)
};

template <bool kPlanar, bool kFours>
struct BlockWriter {
    uint32_t* acc;       // this thread's chunk buffer in LDS, all zero between chunks
    uint4* dst;          // chunk 0 of the current block
    size_t chunk_stride; // in 16-byte units
    uint32_t* record_dst;
    uint32_t chunk;      // index of the chunk being assembled
    uint32_t mask;
    uint32_t dc;
    // A thread's blocks follow one another, so their records do too: they leave four at a time, as one aligned
    // 16-byte store (a 4-byte store per block is a write transaction per block at the L2, and those, not the bytes,
    // were half of this pass's time on smooth pictures).  q0..q2: the `pending` records before the current one,
    // oldest first when there are three; last_record: where the newest of them belongs.
    // (in fours only where the launch has workgroups to spare: the few instructions it takes lengthen a thread's
    // walk, which is all that counts when a single picture's workgroups have the GPU to themselves)
    uint32_t q0 = 0, q1 = 0, q2 = 0, pending = 0;
    uint32_t* last_record = nullptr;

    __device__ __forceinline__ void flush_records() {
        if (pending >= 1) last_record[0] = q2;
        if (pending >= 2) last_record[-1] = q1;
        if (pending >= 3) last_record[-2] = q0;
        pending = 0;
    }
    __device__ __forceinline__ void put_record(uint32_t value) {
        if (!kFours) {
            *record_dst = value;
            return;
        }
        const bool closes_four = ((uint32_t)(reinterpret_cast<uintptr_t>(record_dst) >> 2) & 3u) == 3u;
        if (closes_four && pending == 3u) {
            *reinterpret_cast<uint4*>(record_dst - 3) = make_uint4(q0, q1, q2, value);
            pending = 0;
        } else if (closes_four) {   // the thread's first records: the four began in another thread's blocks
            *record_dst = value;
            flush_records();
        } else {
            q0 = q1; q1 = q2; q2 = value;
            ++pending;
            last_record = record_dst;
        }
    }

    __device__ __forceinline__ void begin(uint4* chunk0, size_t stride, uint32_t* record) {
        dst = chunk0;
        chunk_stride = stride;
        record_dst = record;
        chunk = 0;
        mask = 0;
        dc = 0;
    }
    __device__ __forceinline__ void flush() {
        const uint2 lo = *reinterpret_cast<const uint2*>(acc), hi = *reinterpret_cast<const uint2*>(acc + 2);
        const bool nonzero = (lo.x | lo.y | hi.x | hi.y) != 0;
        // planar: a chunk exists when it holds a non-zero AC coefficient; the DC value travels in the record
        const uint32_t ac = ((chunk == 0 ? lo.x & 0xffff0000u : lo.x) | lo.y | hi.x | hi.y);
        if (!kPlanar || ac != 0) dst[chunk * chunk_stride] = make_uint4(lo.x, lo.y, hi.x, hi.y);
        if (nonzero) {
            *reinterpret_cast<uint2*>(acc) = make_uint2(0, 0);
            *reinterpret_cast<uint2*>(acc + 2) = make_uint2(0, 0);
        }
        mask |= (ac != 0 ? 1u : 0u) << chunk;
    }
    __device__ __forceinline__ void put_dc(int value) {
        dc = (uint32_t)value & 0xffffu;
        reinterpret_cast<int16_t*>(acc)[0] = (int16_t)value;
    }
    // positions arrive in increasing order
    __device__ __forceinline__ void put(uint32_t position, int value) {
        const uint32_t c = position >> 3;
        if (c != chunk) {
            flush();
            if (!kPlanar)
                for (uint32_t i = chunk + 1; i < c; ++i) dst[i] = make_uint4(0, 0, 0, 0);
            chunk = c;
        }
        reinterpret_cast<int16_t*>(acc)[position & 7u] = (int16_t)value;
    }
    __device__ __forceinline__ void end() {
        flush();
        if (kPlanar) {
            put_record(dc | (mask << 16));
        } else {
            for (uint32_t i = chunk + 1; i < 8; ++i) dst[i] = make_uint4(0, 0, 0, 0);
        }
    }
};

// Every symbol moves the position inside the block forward, so a block is at most 64 symbols of at
// most 26 bits (a damaged one included) and the walk cannot leave the staged window.
template <uint32_t W, bool kPlanar, bool kFours>
__device__ __forceinline__ void write_blocks(const GroupWords<W, kWriteTail>& gw, const StepTables& t, uint32_t* acc,
                                             uint32_t lum_mask, uint32_t sub, uint32_t state, WriteTarget& wt,
                                             uint32_t limit = 0xffffffffu) {
    constexpr uint32_t kSubBits = 32 * W;
    uint32_t p = sub * kSubBits + (state & 0xffu);
    const uint32_t c0 = (state >> 8) & 3u;
    uint32_t z = (state >> 16) & 63u;
    const uint32_t end = min((sub + 1) * kSubBits, limit);
    const uint8_t* tb = reinterpret_cast<const uint8_t*>(&t);
    const uint32_t c1 = next_channel(c0), c2 = next_channel(c1);
    // table class and running DC of the current channel and the two after it (see walk)
    uint32_t o0 = class_offset(lum_mask, c0), o1 = class_offset(lum_mask, c1), o2 = class_offset(lum_mask, c2);
    // (as opaque register values: left to itself the compiler turns these selections among three neighbouring fields
    // into indexed loads, and the three fields into an array in scratch memory)
    int by_channel0 = wt.last0, by_channel1 = wt.last1, by_channel2 = wt.last2;
    asm volatile("" : "+v"(by_channel0), "+v"(by_channel1), "+v"(by_channel2));
    int l0 = c0 == 0 ? by_channel0 : (c0 == 1 ? by_channel1 : by_channel2);
    int l1 = c1 == 0 ? by_channel0 : (c1 == 1 ? by_channel1 : by_channel2);
    int l2 = c2 == 0 ? by_channel0 : (c2 == 1 ? by_channel1 : by_channel2);
    uint32_t k = z ? kAcBytes : 0u;
    uint32_t shift = z ? kAcShift : kDcShift;
    uint32_t block = wt.block;
    // Where chunk 0 of the current block and of the two after it go; rotated like the rest, so that no
    // block has to derive its place from its index.
    uint4* const out4 = reinterpret_cast<uint4*>(wt.out);
    const uint32_t mcu0 = block / 3u;
    const size_t stride = kPlanar ? (size_t)3 * wt.n_mcu : 1;
    uint4 *d0, *d1, *d2;
    if (kPlanar) {
        d0 = out4 + (size_t)c0 * wt.n_mcu + mcu0;
        d1 = out4 + (size_t)c1 * wt.n_mcu + mcu0 + (c1 < c0 ? 1 : 0);
        d2 = out4 + (size_t)c2 * wt.n_mcu + mcu0 + (c2 < c0 ? 1 : 0);
    } else {
        d0 = out4 + (size_t)block * 8;
        d1 = d0 + 8;
        d2 = d0 + 16;
    }
    uint32_t* record = kPlanar ? wt.records + block : nullptr;
    bool finished = false, owning = false;
    uint32_t final_bit = 0;
    uint32_t trouble = 0;   // bit 6: an entry reaches past position 63; kStepBad: no code word -- in an own block
    BlockWriter<kPlanar, kFours> bw;
    bw.acc = acc;
    BitWindow<GroupWords<W, kWriteTail>> win(gw, p);
    for (;;) {
        if (k == 0 && !owning) {
            // at a block boundary: the block that starts here is this thread's, if it starts in its sub-sequence
            if (p >= end) break;
            owning = true;
            bw.begin(d0, stride, record);
        }
        const uint32_t v = win.peek(p);
        const uint32_t e = step_lookup(tb + o0 + k, v, shift);
        const bool is_dc = k == 0;
        p += e & 0xffu;
        // AC positions consumed including this entry, plus 128 when the entry ends the block (the end-of-block
        // symbol, or a symbol with that code behind it).  Bit 6 -- 64..127 and 192..255 -- can only come from
        // an entry that reaches past position 63: the serial decoder's error
        const uint32_t processed = z + ((e >> 8) & 0xffu);
        if (__builtin_expect(processed == 128u + 63u, 0)) p -= (e >> kStepEobLenShift) & 7u;   // see kernels.h: combined entries
        win.follow(p);
#ifdef JEA_PROBE_WRITE_PAD   // experiments: that many idle vector instructions a step -- what is an instruction worth here?
#pragma unroll
        for (int pad = 0; pad < JEA_PROBE_WRITE_PAD; ++pad) asm volatile("v_mov_b32 %0, %0" : "+v"(trouble));
#endif
        const bool done = processed >= 63u;
        const uint32_t position = processed & 0x7fu;
        z = done ? 0u : processed + 1u;
        k = done ? 0u : kAcBytes;
        shift = done ? kDcShift : kAcShift;
        if (owning) {
            const int amp = (e & kStepHasAmp) ? step_amplitude(v, e) : 0;
            trouble |= processed | (e & kStepBad);
            if (is_dc) {
                l0 = (int)(int16_t)(l0 + amp);   // entropy_encoder.cpp:125-126: the DC chain runs in int16
                bw.put_dc(l0);
            } else if (position < 63u && (e & kStepHasAmp)) {
                // an entry that fills position 63 is the block's last one and is dropped (entropy_encoder.cpp:129)
                bw.put(position, amp);
            }
        }
        if (done) {
            const uint32_t o = o0;
            const int l = l0;
            uint4* const d = d0 + (kPlanar ? 1 : 24);   // the same channel, one MCU on
            o0 = o1; o1 = o2; o2 = o;
            l0 = l1; l1 = l2; l2 = l;
            d0 = d1; d1 = d2; d2 = d;
            if (kPlanar) ++record;
            if (owning) bw.end();
            owning = false;
            if (++block == wt.total_blocks) {
                finished = true;
                final_bit = p;
                break;
            }
        }
    }
    if (kPlanar && kFours) bw.flush_records();
    wt.error = (trouble & (64u | kStepBad)) != 0;
    wt.finished = finished;
    wt.final_bit = final_bit;
}

constexpr int kSumsMissing = 0x40000000;   // in a record's block count: its DC sums are still to be computed (decode_sync_kernel)
#ifndef JEA_MAX_LOCAL_ITERATIONS
#define JEA_MAX_LOCAL_ITERATIONS 24
#endif
#ifndef JEA_DENSE_EXTRA_ROUNDS
#define JEA_DENSE_EXTRA_ROUNDS 2
#endif
constexpr uint32_t kMaxLocalIterations = JEA_MAX_LOCAL_ITERATIONS;   // repair steps per launch; what is left is flagged pending

// What the kernels of the decoder that reads the raw scan need instead of the unstuffed copy.
struct RawSource {
    const uint8_t* scan = nullptr;       // all files; image i's scan starts at offsets[i] and is sizes[i] raw bytes long
    const uint64_t* offsets = nullptr;
    uint32_t* grp_end = nullptr;         // [groups] end state of each group's last sub-sequence
};
// Jobs the first kernel of the chain does on the side (see unstuff_count_kernel, which does them when there is
// an unstuffing pass): clearing *status and *check_flag, scan ranges from kernel arguments, header gather.
struct RawPrologue {
    SegmentWords words;
    uint32_t n_words = 0;
    uint64_t* d_words = nullptr;
    uint8_t* headers_out = nullptr;
    uint32_t header_bytes = 0;
    int32_t* status = nullptr;
    uint32_t* check_flag = nullptr;
};

// kRaw: the stream is read from the raw scan (stage_group_raw); a group is then a range of raw bytes, holds
// n_active <= kOwned sub-sequences, and hands its end state to the next group through raw.grp_end.
template <uint32_t W, bool kRaw>
__global__ __launch_bounds__(kThreads) void decode_sync_kernel(const uint8_t* __restrict__ ustream,
                                                               const uint64_t* __restrict__ ubase,
                                                               const uint64_t* __restrict__ sizes,
                                                               const uint64_t* __restrict__ img_drops,
                                                               GroupMap gm, int first_round,
                                                               const StepTables* __restrict__ tables,
                                                               uint32_t* __restrict__ sub_end,
                                                               uint32_t* __restrict__ sub_used,
                                                               int4* __restrict__ sub_info,
                                                               int4* __restrict__ grp_sum,
                                                               uint32_t* __restrict__ grp_pending,
                                                               const uint32_t* __restrict__ sub_guess,
                                                               const uint32_t* __restrict__ first_bad,
                                                               uint32_t lum_mask,
                                                               uint32_t* __restrict__ changed, RawSource raw,
                                                               RawPrologue pro, uint32_t tables_per_image) {
    constexpr uint32_t kSubBits = 32 * W;
    constexpr uint32_t kGroupRaw = kOwned * 4 * W;
    __shared__ GroupWords<W> gw;
    __shared__ StepTables t;
    __shared__ uint32_t ends[kThreads + 1];   // ends[i] = state at the start of the sub-sequence in slot i
    __shared__ Sums red[4];
    __shared__ uint32_t scratch[12];
    __shared__ uint32_t wave_lanes[4];        // repair steps: lanes with work, by wave
    __shared__ uint32_t work[kThreads];       // ... their slots, listed
    __shared__ uint32_t res_end[kThreads];    // ... and what the walk of a listed slot gave
    __shared__ Sums res_sums[kThreads];

    const uint32_t tid = threadIdx.x;
    const bool from_args = kRaw && first_round && pro.n_words != 0;   // the scan ranges are kernel arguments
    const GroupId gid = from_args ? locate_group(gm, blockIdx.x, kGroupRaw, [&](uint32_t i) { return pro.words.v[gm.n_images + i]; })
                                  : locate_group(gm, blockIdx.x, kGroupRaw, [&](uint32_t i) { return sizes[i]; });
    const uint32_t group = gid.group, img = gid.img;
    const size_t group_index = blockIdx.x;
    uint64_t raw_len = 0, raw_offset = 0, ulen = 0, total_bits = 0;
    bool empty;
    if (kRaw) {
        if (first_round) {
            if (blockIdx.x == 0 && tid == 0) {
                *pro.status = 0;
                *pro.check_flag = 0;
            }
            if (blockIdx.x == 0 && tid < pro.n_words) pro.d_words[tid] = pro.words.v[tid];
        }
        raw_offset = from_args ? pro.words.v[img] : raw.offsets[img];
        raw_len = from_args ? pro.words.v[gm.n_images + img] : sizes[img];
        if (first_round && pro.headers_out != nullptr && group == 0) {
            const uint8_t* src = raw.scan + raw_offset - pro.header_bytes;
            for (uint32_t i = tid; i < pro.header_bytes; i += kThreads) pro.headers_out[(size_t)img * pro.header_bytes + i] = src[i];
        }
        empty = (uint64_t)group * kGroupRaw >= raw_len;
    } else {
        ulen = sizes[img] - img_drops[img];
        total_bits = 8 * ulen;
        empty = (uint64_t)group * kOwned * kSubBits >= total_bits;
    }
    if (empty) {   // nothing of the stream in this group
        if (first_round && tid == 0) {
            grp_sum[group_index] = make_int4(0, 0, 0, 0);
            grp_pending[group_index] = 0;
        }
        return;
    }
    // Classic: slot tid holds sub-sequence first_sub + tid; slots below kWarm belong to the previous group.
    // Raw: the same, except that the first group has no warm-up slots in front: its owned sub-sequences start at slot 0.
    const int64_t first_sub = (int64_t)group * kOwned - kWarm;
    const int64_t sub = first_sub + tid;
    const bool owned = tid >= kWarm;
    const size_t sub_index = group_index * kOwned + (tid - kWarm);   // owned slots only
    const uint32_t warm_slots = (!kRaw || group > 0) ? kWarm : 0u;
    const uint32_t slot = tid - kWarm + warm_slots;                  // of an owned thread; warm-up threads: tid
    bool active;
    if (kRaw) active = false;   // known after staging (first round) or from the records (later rounds)
    else active = sub >= 0 && (uint64_t)sub * kSubBits < total_bits && (owned || (first_round && group > 0));

    uint32_t used = kNoState, my_end = 0, guess = kNoState;
    Sums sums{0, 0, 0, 0};
    bool sums_missing = false, carried = false;
    if (!first_round) {
        // Unless the previous launch ran out of repair steps, it left the group consistent in itself,
        // and only the first owned sub-sequence can have a new start state: the end state of the
        // group before.  Guesses from the stream context (see context_guess_kernel) are tried on top.
        uint32_t entry = 0;
        bool need = false;
        if (kRaw) {
            if (owned) {
                used = sub_used[sub_index];
                active = used != kNoState;    // the first round marked the slots the group does not fill
            }
        } else if (owned && active) {
            used = sub_used[sub_index];
            if (sub_guess != nullptr && (uint64_t)sub >= first_bad[img]) {
                guess = sub_guess[sub_index];
                if (guess == used) guess = kNoState;
            }
            need = guess != kNoState;
        }
        if (tid == kWarm) {
            if (group > 0)
                entry = __atomic_load_n(kRaw ? &raw.grp_end[group_index - 1] : &sub_end[sub_index - 1], __ATOMIC_RELAXED);
            need = need || (active && entry != used) || grp_pending[group_index] != 0;
        }
        if (!__syncthreads_or(need ? 1 : 0)) return;
        carried = grp_pending[group_index] != 0;
        if (owned) {
            my_end = sub_end[sub_index];
            const int4 si = sub_info[sub_index];
            sums_missing = (si.x & kSumsMissing) != 0;   // left by a launch that ran out of steps
            sums = Sums{si.x & ~kSumsMissing, si.y, si.z, si.w};
        }
        if (kRaw) {
            if (tid == kWarm) ends[warm_slots] = entry;
            if (owned) ends[slot + 1] = my_end;
        } else {
            if (tid == kWarm) ends[kWarm] = entry;
            if (owned) ends[tid + 1] = my_end;
        }
    } else {
        // Every slot starts from the guess "state 0 at its first bit"; for the first slot of the
        // image that is the true state.  The warm-up slots of the first group lie before the
        // stream and stay idle.
        ends[tid + 1] = 0;
        if (tid == 0) ends[0] = 0;
    }

    load_tables(t, tables + (tables_per_image ? img : 0u));
    uint32_t limit = 0xffffffffu, n_active = kOwned;
    if (kRaw) {
        const RawGroup rg = stage_group_raw(gw, raw.scan + raw_offset, raw_len, (uint64_t)group * kGroupRaw,
                                            warm_slots * 4 * W, scratch);
        n_active = (rg.group_bytes * 8 + kSubBits - 1) / kSubBits;
        limit = warm_slots * kSubBits + rg.group_bytes * 8;
        if (first_round) active = owned ? (tid - kWarm) < n_active : group > 0;
    } else {
        stage_group(gw, ustream + ubase[img], ulen, first_sub);
    }
    __syncthreads();
#if defined(JEA_PROBE_SYNC) && JEA_PROBE_SYNC == 1   // experiments (tools/probe_sync_steps.sh): the kernel up to here
    return;
#endif
    const uint32_t my_slot = (kRaw && owned) ? slot : tid;

    bool dirty = false;
    if (__syncthreads_or(guess != kNoState ? 1 : 0)) {
        // offered start states first; the loop below keeps them only where they fit their predecessors
        if (guess != kNoState) {
            my_end = walk<W>(gw, t, lum_mask, my_slot, guess, sums, limit);
            sums_missing = false;
            used = guess;
            dirty = true;
        }
        __syncthreads();
        if (guess != kNoState) ends[my_slot + 1] = my_end;
        __syncthreads();
    }
    // A repair step costs a wave the same whether one of its lanes walks or all of them, and from the third step on
    // few do (those whose predecessor's end state changed once more).  So the lanes with work are counted, and where
    // they would leave a wave's worth of lanes idle, their slots are listed and walked by the first lanes of the
    // workgroup -- as few waves as it takes; the others wait at the barrier and leave their issue slots to the other
    // workgroups of the compute unit.  Which wave comes first turns with the group and the step, so that the
    // workgroups of a compute unit do not all pick the same SIMD.
#ifndef JEA_COMPACT_BELOW
#define JEA_COMPACT_BELOW 192
#endif
    constexpr uint32_t kCompactBelow = JEA_COMPACT_BELOW;
    // A stream whose listed lanes are still at it after a few steps is one whose wrong start states travel far (noise:
    // through dozens of sub-sequences, one a step).  From then on the listed walks leave the DC sums out -- a quarter
    // of a step's instructions -- and the lanes whose last walk was of that kind walk once more, in full, when the
    // group has settled.  A launch that runs out of steps leaves that walk to the next one: it marks the records
    // (kSumsMissing), and the next launch starts with the light walks at once.  Only where the launch leaves
    // workgroup slots free, though (four a compute unit): there a step costs its latency whatever its width, and the
    // full walk at the end costs one step; on a full GPU a listed step costs a quarter of that walk's issue slots,
    // and a stream would have to take dozens of steps (those that run out of them do) for the exchange to pay.
#ifndef JEA_LIGHT_FROM
#define JEA_LIGHT_FROM 4
#endif
#ifndef JEA_OPENING_LIGHT
#define JEA_OPENING_LIGHT 1
#endif
    constexpr uint32_t kLightFrom = JEA_LIGHT_FROM;
    const bool roomy = gridDim.x <= 1024u;
    // The opening step of the first round walks every sub-sequence from a guess that is all but never right, so all
    // of it is walked again in the second step anyway: the opening walks leave the sums out too, and what they walked
    // is walked again whether or not its start state changed (`unsummed`).
    bool unsummed = false;
    uint32_t pending = 0;
    for (uint32_t iteration = 0;; ++iteration) {
        const uint32_t start = ends[my_slot];
        const bool redo = active && (start != used || unsummed);
        const bool opening = JEA_OPENING_LIGHT && first_round && iteration == 0;
        const uint64_t votes = __ballot(redo ? 1 : 0);
        if ((tid & 63u) == 0) wave_lanes[tid >> 6] = (uint32_t)__popcll(votes);
        __syncthreads();   // every start state has been read, every count is there
        const uint32_t n0 = wave_lanes[0], n1 = wave_lanes[1], n2 = wave_lanes[2], n3 = wave_lanes[3];
        const uint32_t lanes = n0 + n1 + n2 + n3;
        if (lanes == 0) break;
        if (iteration == kMaxLocalIterations) {   // the rest of the repair is left to the next launch
            pending = 1;
            break;
        }
        if (lanes >= kCompactBelow) {
            if (redo) {
                if (opening) my_end = walk<W, false>(gw, t, lum_mask, my_slot, start, sums, limit);
                else my_end = walk<W>(gw, t, lum_mask, my_slot, start, sums, limit);
                unsummed = opening;
                sums_missing = false;
                used = start;
                dirty = true;
                ends[my_slot + 1] = my_end;
            }
        } else {
            const bool light = opening || carried || (roomy && iteration >= kLightFrom);
            const uint32_t wave = tid >> 6;
            const uint32_t before = (wave > 0 ? n0 : 0u) + (wave > 1 ? n1 : 0u) + (wave > 2 ? n2 : 0u);
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(votes >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)votes, 0u));
            if (redo) work[before + rank] = my_slot;
            __syncthreads();
            const uint32_t turn = (tid + 64u * ((group + iteration) & 3u)) & (kThreads - 1u);
            if (turn < lanes) {
                const uint32_t its_slot = work[turn];
                Sums its_sums;
                if (light) res_end[its_slot] = walk<W, false>(gw, t, lum_mask, its_slot, ends[its_slot], its_sums, limit);
                else res_end[its_slot] = walk<W>(gw, t, lum_mask, its_slot, ends[its_slot], its_sums, limit);
                res_sums[its_slot] = its_sums;
            }
            __syncthreads();
            if (redo) {
                my_end = res_end[my_slot];
                sums = res_sums[my_slot];
                sums_missing = light && !opening;
                unsummed = opening;
                used = start;
                dirty = true;
                ends[my_slot + 1] = my_end;
            }
        }
        __syncthreads();
#if defined(JEA_PROBE_SYNC) && JEA_PROBE_SYNC >= 2   // experiments: JEA_PROBE_SYNC - 1 walk steps
        if (iteration + 2 == JEA_PROBE_SYNC) break;
#endif
    }

    if (sums_missing && owned && !pending) {
        (void)walk<W>(gw, t, lum_mask, my_slot, used, sums, limit);
        sums_missing = false;
        dirty = true;
    }
    dirty = dirty && owned;
    const int any_dirty = __syncthreads_or(dirty ? 1 : 0);
    if (dirty) {
        __atomic_store_n(&sub_end[sub_index], my_end, __ATOMIC_RELAXED);
        sub_used[sub_index] = used;
        sub_info[sub_index] = make_int4(sums.nb | (sums_missing ? kSumsMissing : 0), sums.dc0, sums.dc1, sums.dc2);
        if (kRaw && (tid - kWarm) + 1 == n_active) __atomic_store_n(&raw.grp_end[group_index], my_end, __ATOMIC_RELAXED);
    } else if (first_round && owned) {   // sub-sequences past the end of the stream
        sub_end[sub_index] = 0;
        sub_used[sub_index] = kNoState;
        sub_info[sub_index] = make_int4(0, 0, 0, 0);
    }
    if (tid == 0) grp_pending[group_index] = pending;
    if (any_dirty || first_round || pending) {
        Sums v = owned ? sums : Sums{0, 0, 0, 0};
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1)
            v = v + Sums{__shfl_xor(v.nb, d), __shfl_xor(v.dc0, d), __shfl_xor(v.dc1, d), __shfl_xor(v.dc2, d)};
        if ((tid & 63) == 0) red[tid >> 6] = v;
        __syncthreads();
        if (tid == 0) {
            const Sums sm = red[0] + red[1] + red[2] + red[3];
            grp_sum[group_index] = make_int4(sm.nb, sm.dc0, sm.dc1, sm.dc2);
            if (any_dirty || pending) atomicOr(changed, 1u);
        }
    }
}

// ---------------------------------------------------------------------------
// Guesses from the stream context, for streams that do not re-synchronise.
// ---------------------------------------------------------------------------

constexpr uint32_t kContextWords = 16;   // stream words before a sub-sequence that identify its context

// first_bad[img] (preset to 0xffffffff) = first sub-sequence that is not proven: the state it was
// decoded from is not the end state of the one before it (state 0 for the first).  Everything before
// it has been decoded from the serial decoder's states.
template <uint32_t W>
__global__ __launch_bounds__(kThreads) void first_unproven_kernel(const uint64_t* __restrict__ sizes,
                                                                  const uint64_t* __restrict__ img_drops,
                                                                  GroupMap gm,
                                                                  const uint32_t* __restrict__ sub_end,
                                                                  const uint32_t* __restrict__ sub_used,
                                                                  uint32_t* __restrict__ first_bad) {
    const uint32_t tid = threadIdx.x;
    const GroupId gid = locate_group(gm, blockIdx.x, kOwned * 4 * W, [&](uint32_t i) { return sizes[i]; });
    const uint32_t group = gid.group, img = gid.img;
    if (tid >= kOwned) return;
    const uint64_t total_bits = 8 * (sizes[img] - img_drops[img]);
    const uint64_t sub = (uint64_t)group * kOwned + tid;
    if (sub * 32 * W >= total_bits) return;
    const size_t sub_index = (size_t)blockIdx.x * kOwned + tid;
    const uint32_t before = sub == 0 ? 0u : sub_end[sub_index - 1];
    if (sub_used[sub_index] != before) atomicMin(&first_bad[img], (uint32_t)sub);
}

__device__ __forceinline__ unsigned long long context_hash(const uint8_t* __restrict__ stream, uint64_t first_word,
                                                           uint32_t img) {
    unsigned long long h = 0x9e3779b97f4a7c15ull ^ img;
    const uint32_t* w = reinterpret_cast<const uint32_t*>(stream) + first_word;
#pragma unroll
    for (uint32_t i = 0; i < kContextWords; ++i) {
        h ^= w[i];
        h *= 0xff51afd7ed558ccdull;
        h ^= h >> 29;
    }
    return h | 1ull;   // 0 marks an empty table entry
}

// kLearn: the proven sub-sequences (those before first_bad) enter their start state into the table
// under the hash of the stream words before them.  Otherwise: the others look theirs up.
template <uint32_t W, bool kLearn>
__global__ __launch_bounds__(kThreads) void context_kernel(const uint8_t* __restrict__ ustream,
                                                           const uint64_t* __restrict__ ubase,
                                                           const uint64_t* __restrict__ sizes,
                                                           const uint64_t* __restrict__ img_drops,
                                                           GroupMap gm, const uint32_t* __restrict__ first_bad,
                                                           const uint32_t* __restrict__ sub_used,
                                                           unsigned long long* __restrict__ keys,
                                                           uint32_t* __restrict__ vals, uint32_t table_mask,
                                                           uint32_t* __restrict__ sub_guess) {
    const uint32_t tid = threadIdx.x;
    const GroupId gid = locate_group(gm, blockIdx.x, kOwned * 4 * W, [&](uint32_t i) { return sizes[i]; });
    const uint32_t group = gid.group, img = gid.img;
    if (tid >= kOwned) return;
    const uint64_t sub = (uint64_t)group * kOwned + tid;
    const bool proven = sub < first_bad[img];
    if (kLearn != proven) return;
    const uint64_t ulen = sizes[img] - img_drops[img];
    const size_t sub_index = (size_t)blockIdx.x * kOwned + tid;
    const uint64_t first_word = sub * W;
    const bool usable = first_word >= kContextWords && (first_word + 1) * 4 <= ulen;
    if (!kLearn) sub_guess[sub_index] = kNoState;
    if (!usable) return;
    const unsigned long long key = context_hash(ustream + ubase[img], first_word - kContextWords, img);
    uint32_t slot = (uint32_t)(key >> 20) & table_mask;
    for (int probe = 0; probe < 8; ++probe, slot = (slot + 1) & table_mask) {
        if (kLearn) {
            const unsigned long long old = atomicCAS(&keys[slot], 0ull, key);
            if (old == 0ull || old == key) {
                vals[slot] = sub_used[sub_index];
                return;
            }
        } else {
            const unsigned long long k = keys[slot];
            if (k == key) {
                sub_guess[sub_index] = vals[slot];
                return;
            }
            if (k == 0ull) return;
        }
    }
}

// One workgroup per image: exclusive scan of the group sums; an image whose
// stream holds fewer blocks than the picture needs is a stream error (the serial
// decoder would run off the end of the file).
//
// With `unsettled` given it also does what one more synchronisation round would find out: whether
// every workgroup's first sub-sequence was decoded from the end state of the sub-sequence before it
// and no workgroup ran out of repair steps.  If so the fixed point is proven; if not the flag is set.
__global__ __launch_bounds__(kThreads) void scan_groups_kernel(const int4* __restrict__ grp_sum, GroupMap gm,
                                                               uint32_t total_blocks, int4* __restrict__ grp_base,
                                                               int32_t* __restrict__ status,
                                                               const uint64_t* __restrict__ sizes,
                                                               const uint64_t* __restrict__ img_drops,
                                                               uint32_t sub_bits, const uint32_t* __restrict__ sub_end,
                                                               const uint32_t* __restrict__ sub_used,
                                                               const uint32_t* __restrict__ grp_pending,
                                                               uint32_t* __restrict__ unsettled,
                                                               const uint32_t* __restrict__ raw_grp_end) {
    __shared__ Sums lds4[4];
    __shared__ Sums carry_s;
    const uint32_t img = blockIdx.x;
    // this image's groups: n_groups of them from number `first` on
    const uint32_t group_raw = kOwned * sub_bits / 8;
    const size_t first = first_group_of(gm, img, group_raw, [&](uint32_t i) { return sizes[i]; });
    const uint32_t n_groups = groups_of_scan(sizes[img], group_raw);
    if (threadIdx.x == 0) carry_s = Sums{0, 0, 0, 0};
    __syncthreads();
    // raw_grp_end given: the decoder that reads the raw scan -- a group is kOwned * sub_bits / 8 raw bytes, and the
    // state it hands on is in raw_grp_end
    const bool raw = raw_grp_end != nullptr;
    const uint64_t total_bits = unsettled != nullptr ? 8 * (raw ? sizes[img] : sizes[img] - img_drops[img]) : 0;
    bool open_end = false;
    for (uint32_t g0 = 0; g0 < n_groups; g0 += kThreads) {
        const uint32_t g = g0 + threadIdx.x;
        Sums v{0, 0, 0, 0};
        if (g < n_groups) {
            const int4 s = grp_sum[first + g];
            v = Sums{s.x, s.y, s.z, s.w};
            if (unsettled != nullptr && (uint64_t)g * kOwned * sub_bits < total_bits) {
                const size_t first_sub = (first + g) * kOwned;
                uint32_t entry = 0;   // the stream starts in state 0
                if (g > 0) entry = raw ? raw_grp_end[first + g - 1] : sub_end[first_sub - 1];
                open_end = open_end || entry != sub_used[first_sub] || grp_pending[first + g] != 0;
            }
        }
        Sums total;
        const Sums excl = block_exclusive_scan(v, lds4, total);
        const Sums base = carry_s + excl;
        if (g < n_groups) grp_base[first + g] = make_int4(base.nb, base.dc0, base.dc1, base.dc2);
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry_s + total;
        __syncthreads();
    }
    if (open_end) atomicOr(unsettled, 1u);
    if (threadIdx.x == 0 && (uint32_t)carry_s.nb < total_blocks) atomicCAS(status, 0, JEA_ERR_BAD_STREAM);
}

// kFold (streams of up to kFoldGroups workgroups): there is no scan launch before this kernel.  Every workgroup
// adds up the sums of the workgroups before it, checks what scan_groups_kernel checks for its own part -- that
// its first sub-sequence was decoded from the end state of the sub-sequence before it and that it did not run out
// of repair steps (else *unsettled is set) -- and the workgroup with the end of the stream sees whether the
// stream held enough blocks.
constexpr uint32_t kFoldGroups = 1024;

template <uint32_t W, bool kPlanar, bool kFold, bool kRaw, bool kFours>
__global__ __launch_bounds__(kThreads) void decode_write_kernel(const uint8_t* __restrict__ ustream,
                                                                const uint64_t* __restrict__ ubase,
                                                                const uint64_t* __restrict__ sizes,
                                                                const uint64_t* __restrict__ img_drops,
                                                                GroupMap gm, uint32_t total_blocks,
                                                                const StepTables* __restrict__ tables,
                                                                const uint32_t* __restrict__ sub_used,
                                                                const int4* __restrict__ sub_info,
                                                                const int4* __restrict__ grp_base,
                                                                int16_t* __restrict__ coefs,
                                                                uint32_t* __restrict__ records, uint32_t lum_mask,
                                                                int32_t* __restrict__ status,
                                                                const int4* __restrict__ grp_sum,
                                                                const uint32_t* __restrict__ sub_end,
                                                                const uint32_t* __restrict__ grp_pending,
                                                                uint32_t* __restrict__ unsettled, RawSource raw,
                                                                uint32_t tables_per_image) {
    constexpr uint32_t kSubBits = 32 * W;
    constexpr uint32_t kGroupRaw = kOwned * 4 * W;
    __shared__ GroupWords<W, kWriteTail> gw;
    __shared__ StepTables t;
    __shared__ Sums lds4[4];
    __shared__ __align__(8) uint32_t acc[kThreads * kAccWords];
    __shared__ uint32_t scratch[12];

    const uint32_t tid = threadIdx.x;
    const GroupId gid = locate_group(gm, blockIdx.x, kGroupRaw, [&](uint32_t i) { return sizes[i]; });
    const uint32_t group = gid.group, img = gid.img;
    const size_t group_index = blockIdx.x, first_group = gid.first;
    // classic: lengths of the unstuffed stream; raw: raw_len is the scan's length in the file
    const uint64_t raw_len = sizes[img];
    const uint64_t ulen = kRaw ? 0 : raw_len - img_drops[img];
    const uint64_t total_bits = 8 * ulen;
    const bool empty = kRaw ? (uint64_t)group * kGroupRaw >= raw_len : (uint64_t)group * kOwned * kSubBits >= total_bits;
    if (empty) {
        if (kFold && group == 0 && tid == 0 && total_blocks != 0) atomicCAS(status, 0, JEA_ERR_BAD_STREAM);   // an empty stream
        return;
    }
    const bool holds_end = kRaw ? (uint64_t)(group + 1) * kGroupRaw >= raw_len : (uint64_t)(group + 1) * kOwned * kSubBits >= total_bits;
    int4 gb;
    if (kFold) {
        Sums before{0, 0, 0, 0};
        for (uint32_t g = tid; g < group; g += kThreads) {
            const int4 v = grp_sum[first_group + g];
            before = before + Sums{v.x, v.y, v.z, v.w};
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1)
            before = before + Sums{__shfl_xor(before.nb, d), __shfl_xor(before.dc0, d), __shfl_xor(before.dc1, d),
                                   __shfl_xor(before.dc2, d)};
        if ((tid & 63) == 0) lds4[tid >> 6] = before;
        __syncthreads();
        before = lds4[0] + lds4[1] + lds4[2] + lds4[3];
        __syncthreads();
        gb = make_int4(before.nb, before.dc0, before.dc1, before.dc2);
        if (tid == 0) {
            const size_t first = group_index * kOwned;
            if (unsettled != nullptr) {
                uint32_t entry = 0;   // the stream starts in state 0
                if (group > 0) entry = kRaw ? raw.grp_end[group_index - 1] : sub_end[first - 1];
                if (entry != sub_used[first] || grp_pending[group_index] != 0) atomicOr(unsettled, 1u);
            }
            // the workgroup that holds the end of the stream: fewer blocks than the picture needs is a stream
            // error (the serial decoder would run off the end of the file)
            if (holds_end) {
                const int4 own = grp_sum[group_index];
                if ((uint32_t)(before.nb + own.x) < total_blocks) atomicCAS(status, 0, JEA_ERR_BAD_STREAM);
            }
        }
    } else {
        gb = grp_base[group_index];
    }
    if ((uint32_t)gb.x >= total_blocks) return;   // the picture is complete before this group
    // slot tid holds the group's tid-th sub-sequence; the last kWarm threads have none
    const size_t sub_index = group_index * kOwned + tid;

    // asked for first, so that they are under way while tables and stream are staged
    const int4 si = tid < kOwned ? sub_info[sub_index] : make_int4(0, 0, 0, 0);
    const uint32_t start_state = tid < kOwned ? sub_used[sub_index] : 0u;
    load_tables(t, tables + (tables_per_image ? img : 0u));
    const uint8_t* stream = kRaw ? nullptr : ustream + ubase[img];
    bool active;
    uint32_t limit = 0xffffffffu;
    RawGroup rg{0, 0, false};
    if (kRaw) {
        rg = stage_group_raw(gw, raw.scan + raw.offsets[img], raw_len, (uint64_t)group * kGroupRaw, 0u, scratch);
        limit = rg.group_bytes * 8;
        active = tid < kOwned && (uint64_t)tid * kSubBits < limit;
    } else {
        stage_group(gw, stream, ulen, (int64_t)group * kOwned);
        active = tid < kOwned && ((uint64_t)group * kOwned + tid) * kSubBits < total_bits;
    }
    for (uint32_t i = 0; i < kAccWords; ++i) acc[tid * kAccWords + i] = 0;

    Sums total;
    const Sums before = block_exclusive_scan(Sums{si.x & ~kSumsMissing, si.y, si.z, si.w}, lds4, total);   // also orders the staging
    WriteTarget wt;
    wt.out = coefs + (size_t)img * total_blocks * 64;
    wt.records = kPlanar ? records + (size_t)img * total_blocks : nullptr;
    wt.n_mcu = total_blocks / 3u;
    wt.block = (uint32_t)(gb.x + before.nb);
    wt.total_blocks = total_blocks;
    wt.last0 = (int)(int16_t)(gb.y + before.dc0);
    wt.last1 = (int)(int16_t)(gb.z + before.dc1);
    wt.last2 = (int)(int16_t)(gb.w + before.dc2);
    wt.error = false;
    wt.finished = false;
    wt.final_bit = 0;
    if (!active || wt.block >= total_blocks) return;

    write_blocks<W, kPlanar, kFours>(gw, t, acc + tid * kAccWords, lum_mask, tid, start_state, wt, limit);
    if (wt.error) {
        // a malformed symbol outranks a missing end marker, as in the serial order of checks
        const int32_t old = atomicCAS(status, 0, JEA_ERR_BAD_STREAM);
        if (old == JEA_ERR_NO_EOI) atomicCAS(status, JEA_ERR_NO_EOI, JEA_ERR_BAD_STREAM);
    }
    if (wt.finished) {
        // positions from here on are relative to the group's first stream bit
        bool overrun, marker;
        const uint32_t q = (wt.final_bit + 7) >> 3;   // encoder.cpp:62-65: skip to the byte boundary, the next two bytes must be the EOI marker
        if (kRaw) {
            overrun = rg.end_in_window && wt.final_bit > 8 * rg.window_bytes;
            const uint8_t* wb = reinterpret_cast<const uint8_t*>(gw.w);
            using G = GroupWords<W, kWriteTail>;
            const uint32_t b0 = wb[G::slot((q >> 2) + 1) * 4u + (3u - (q & 3u))];
            const uint32_t b1 = wb[G::slot(((q + 1) >> 2) + 1) * 4u + (3u - ((q + 1) & 3u))];
            marker = !(rg.end_in_window && q + 2 > rg.window_bytes) && b0 == 0xFFu && b1 == 0xD9u;
        } else {
            const uint64_t end_bit = (uint64_t)group * kOwned * kSubBits + wt.final_bit;
            const uint64_t qq = (end_bit + 7) >> 3;
            overrun = end_bit > total_bits;
            marker = !overrun && qq + 2 <= ulen && stream[qq] == 0xFFu && stream[qq + 1] == 0xD9u;
        }
        if (overrun) {   // bits past the end of the file were consumed
            const int32_t old = atomicCAS(status, 0, JEA_ERR_BAD_STREAM);
            if (old == JEA_ERR_NO_EOI) atomicCAS(status, JEA_ERR_NO_EOI, JEA_ERR_BAD_STREAM);
        } else if (!marker) {
            atomicCAS(status, 0, JEA_ERR_NO_EOI);
        }
    }
}

// Same exclusive scan as the encoder's per-image tile scan, for the chunk drop counts.
__global__ __launch_bounds__(1024) void scan_chunks_kernel(const uint32_t* __restrict__ in, uint32_t n,
                                                           uint64_t* __restrict__ out,
                                                           uint64_t* __restrict__ img_total) {
    __shared__ uint64_t wave_sum[16];
    __shared__ uint64_t carry_s;
    const uint32_t img = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t t0 = 0; t0 < n; t0 += 1024) {
        const uint32_t t = t0 + threadIdx.x;
        const uint64_t v = (t < n) ? (uint64_t)in[(size_t)img * n + t] : 0;
        uint64_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t u = __shfl_up(incl, d);
            if (lane >= d) incl += u;
        }
        if (lane == 63) wave_sum[wave] = incl;
        __syncthreads();
        uint64_t base = carry_s;
        for (int w = 0; w < wave; ++w) base += wave_sum[w];
        if (t < n) out[(size_t)img * n + t] = base + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = base + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) img_total[img] = carry_s;
}

}  // namespace

// ---------------------------------------------------------------------------
// Step tables from a file's own DHT segment (files written with per-image Huffman tables): the device-side twin of
// the host's build_step_table, one workgroup per image.  The segment is expected where this library's and the
// reference's headers have it, with the alphabets' full sizes (12 and 162 values); anything else, or counts that
// are no prefix code, or sub-tables that do not fit, yields tables in which nothing decodes, and the stream fails.
// ---------------------------------------------------------------------------
namespace {
struct CanonicalCode {
    int mincode[17], maxcode[17], valptr[17];   // per code length 1..16; maxcode = -1: no code of that length
};

// The code word that starts the 16-bit prefix p16, among lengths lmin..lmax: (length << 8) | symbol, or 0.
__device__ __forceinline__ uint32_t canonical_decode(const CanonicalCode& cc, const uint8_t* vals, uint32_t p16, int lmin, int lmax) {
    for (int l = lmin; l <= lmax; ++l) {
        const int code = (int)(p16 >> (16 - l));
        if (cc.maxcode[l] >= 0 && code <= cc.maxcode[l] && code >= cc.mincode[l])
            return ((uint32_t)l << 8) | vals[cc.valptr[l] + code - cc.mincode[l]];
    }
    return 0;
}

// Entry of symbol `sym` with a code of l bits (build_step_table's rules).
__device__ __forceinline__ uint32_t step_entry(uint32_t sym, uint32_t l, bool is_dc) {
    const uint32_t ss = is_dc ? sym : (sym & 15u), run = is_dc ? 0u : (sym >> 4);
    const bool valid = is_dc ? sym <= 11u : (ss <= 10u && (ss != 0u || sym == 0x00u || sym == 0xF0u));
    if (!valid) return kStepBad | 1u;
    uint32_t entry = (l + ss) | (run << 8) | (l << 16) | (((32u - ss) & 31u) << 24);
    if (!is_dc && sym == 0x00u) entry |= kStepEndsBlock;
    if (ss) entry |= kStepHasAmp | (is_dc ? kStepDcAdd : 0u);
    return entry;
}

template <class Table>
__device__ __forceinline__ void fill_step_table(Table& t, const CanonicalCode& cc, const uint8_t* bits, const uint8_t* vals,
                                                bool is_dc, uint32_t eob_code, uint32_t eob_len, bool usable,
                                                uint32_t* scan4) {
    constexpr uint32_t F = Table::kFastBits;
    const uint32_t tid = threadIdx.x, bad = kStepBad | 1u;
    const int lane = tid & 63, wave = tid >> 6;
    for (uint32_t i = tid; i < (uint32_t)Table::kSlowEntries; i += kThreads) t.slow[i] = bad;
    // the first index that leads to codes longer than the index: canonical codes grow with their length
    uint32_t first_bucket = 1u << F;
    if (usable)
        for (int l = (int)F + 1; l <= 16; ++l)
            if (bits[l - 1] != 0) {
                first_bucket = (uint32_t)cc.mincode[l] >> (l - F);
                break;
            }
    uint32_t carry = 0;   // sub-table entries handed out so far
    bool overflow = false;
    for (uint32_t j0 = 0; j0 < (1u << F); j0 += kThreads) {
        const uint32_t j = j0 + tid;
        uint32_t entry = bad, m = 0;
        if (usable && j < first_bucket) {
            const uint32_t d = canonical_decode(cc, vals, j << (16 - F), 1, (int)F);
            if (d != 0) {
                entry = step_entry(d & 0xffu, d >> 8, is_dc);
                // the symbol, its amplitude bits and the end-of-block code all inside the index bits (kernels.h)
                const uint32_t used = entry & 0xffu;
                if (!(entry & (kStepBad | kStepEndsBlock)) && (is_dc || (entry & kStepHasAmp)) && eob_len >= 1 &&
                    eob_len <= 7 && used + eob_len <= F && ((j >> (F - used - eob_len)) & ((1u << eob_len) - 1u)) == eob_code)
                    entry = (entry + eob_len) | kStepEndsBlock | (eob_len << kStepEobLenShift);
            }
        } else if (usable) {
            // longest code under this prefix = size of its sub-table
            for (int l = (int)F + 1; l <= 16; ++l)
                if (cc.maxcode[l] >= 0 && j >= ((uint32_t)cc.mincode[l] >> (l - F)) && j <= ((uint32_t)cc.maxcode[l] >> (l - F)))
                    m = (uint32_t)l - F;
        }
        const uint32_t need = m ? 1u << m : 0u;
        uint32_t incl = need;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t u = __shfl_up(incl, d);
            if (lane >= d) incl += u;
        }
        if (lane == 63) scan4[wave] = incl;
        __syncthreads();
        uint32_t at = carry + incl - need;
        for (int w = 0; w < wave; ++w) at += scan4[w];
        carry += scan4[0] + scan4[1] + scan4[2] + scan4[3];
        __syncthreads();
        if (m) {
            if (at + need <= (uint32_t)Table::kSlowEntries) {
                entry = ((32u - m) << 8) | (((uint32_t)offsetof(Table, slow) + at * 4u) << 16);
                for (uint32_t k = 0; k < need; ++k) {
                    const uint32_t d = canonical_decode(cc, vals, (j << (16 - F)) | (k << (16 - F - m)), (int)F + 1, (int)(F + m));
                    t.slow[at + k] = d ? step_entry(d & 0xffu, d >> 8, is_dc) : bad;
                }
            } else {
                overflow = true;
            }
        }
        t.fast[j] = entry;
    }
    // sub-tables that do not fit: nothing of this table may decode
    if (__syncthreads_or(overflow ? 1 : 0))
        for (uint32_t j = tid; j < (1u << F); j += kThreads) t.fast[j] = bad;
}

__global__ __launch_bounds__(kThreads) void build_step_tables_kernel(const uint8_t* __restrict__ data,
                                                                     const uint64_t* __restrict__ scan_offsets,
                                                                     uint32_t header_bytes, uint32_t lum_mask_unused,
                                                                     StepTables* __restrict__ out) {
    __shared__ uint8_t spec[kHuffSpecBytes];
    __shared__ CanonicalCode cc[4];
    __shared__ uint32_t eob_code[2], eob_len[2], ok[4], scan4[4];
    const uint32_t img = blockIdx.x, tid = threadIdx.x;
    const uint8_t* hdr = data + scan_offsets[img] - header_bytes;
    for (uint32_t i = tid; i < kHuffSpecBytes; i += kThreads) spec[i] = hdr[kHuffSpecOffset + i];
    __syncthreads();
    if (tid < 4) {
        const uint32_t at = tid == 0 ? 0u : (tid == 1 ? 29u : (tid == 2 ? 208u : 237u));
        const bool is_ac = (tid & 1u) != 0;
        const uint8_t* bits = spec + at + 1;
        const uint8_t* vals = spec + at + 17;
        const int n = is_ac ? 162 : 12;
        bool good = spec[at] == ((is_ac ? 0x10u : 0x00u) | (tid >> 1));
        int code = 0, k = 0;
        for (int l = 1; l <= 16; ++l) {
            cc[tid].mincode[l] = code;
            cc[tid].valptr[l] = k;
            k += bits[l - 1];
            code += bits[l - 1];
            cc[tid].maxcode[l] = bits[l - 1] ? code - 1 : -1;
            if (code > (1 << l)) good = false;   // more codes than the length has
            code <<= 1;
        }
        if (k != n) good = false;
        ok[tid] = good ? 1u : 0u;
        if (is_ac) {
            uint32_t ec = 0, el = 0;
            if (good)
                for (int l = 1; l <= 16; ++l)
                    for (int i = 0; i < bits[l - 1]; ++i)
                        if (vals[cc[tid].valptr[l] + i] == 0x00u) {
                            ec = (uint32_t)(cc[tid].mincode[l] + i);
                            el = (uint32_t)l;
                        }
            eob_code[tid >> 1] = ec;
            eob_len[tid >> 1] = el;
        }
    }
    __syncthreads();
    const bool usable = ok[0] && ok[1] && ok[2] && ok[3];
    StepTables& t = out[img];
    for (uint32_t c = 0; c < 2; ++c) {
        const uint32_t dc_at = c == 0 ? 0u : 208u, ac_at = c == 0 ? 29u : 237u;
        fill_step_table(t.cls[c].dc, cc[2 * c], spec + dc_at + 1, spec + dc_at + 17, true, eob_code[c], eob_len[c], usable, scan4);
        fill_step_table(t.cls[c].ac, cc[2 * c + 1], spec + ac_at + 1, spec + ac_at + 17, false, eob_code[c], eob_len[c], usable, scan4);
    }
}
}  // namespace

void launch_build_step_tables(hipStream_t s, const uint8_t* d_data, const uint64_t* d_scan_offsets, size_t n_images,
                              uint32_t header_bytes, StepTables* d_tables) {
    build_step_tables_kernel<<<(unsigned)n_images, kThreads, 0, s>>>(d_data, d_scan_offsets, header_bytes, 0u, d_tables);
}

uint32_t decode_group_bytes(uint32_t sub_words) { return kOwned * 4u * sub_words; }

DecodePlan plan_entropy_decode(size_t n_images, const uint64_t* scan_sizes, size_t max_scan_bytes, size_t mcus_per_image,
                               size_t total_scan_bytes) {
    DecodePlan p;
    p.n_images = (uint32_t)n_images;
    p.n_chunks = (uint32_t)((max_scan_bytes + kChunkBytes - 1) / kChunkBytes);
    if (p.n_chunks == 0) p.n_chunks = 1;
    // the longest sub-sequences that still give the GPU a few workgroups per compute unit -- unless the streams are
    // dense.  A decoder that starts in a wrong state falls into step only where a block ends, so the distance it
    // needs grows with the blocks' length, and what the two rounds' 32 repair steps cover grows with the
    // sub-sequences' length: one 1920x1080 picture of noise at quality 50 (145 bits per block) does not settle
    // with 64-bit sub-sequences and went through the fallback decoder (0.81 ms), with 512-bit ones it settles.
    // The time of such a stream is the serial walk over that distance whatever the sub-sequences' length, so the
    // extra workgroups of the short ones would buy nothing either.
    if (total_scan_bytes == 0) total_scan_bytes = max_scan_bytes * n_images;
    const bool dense = (uint64_t)total_scan_bytes * 8 > 64ull * mcus_per_image * 3 * n_images;
    const uint32_t sizes[3] = {kSubWordsLarge, kSubWordsMedium, kSubWordsSmall};
    for (uint32_t w : sizes) {
        const uint32_t group_bytes = decode_group_bytes(w);
        p.sub_words = w;
        p.n_groups = groups_of_scan(max_scan_bytes, group_bytes);
        uint64_t total = 0;
        if (scan_sizes != nullptr)
            for (size_t i = 0; i < n_images; ++i) total += groups_of_scan(scan_sizes[i], group_bytes);
        else
            total = (uint64_t)p.n_groups * n_images;
        p.total_groups = (uint32_t)total;
#ifdef JEA_PLAN_MIN_GROUPS   // experiments: tools/build_variant.sh <name> -DJEA_PLAN_MIN_GROUPS=<n>
        if (total >= JEA_PLAN_MIN_GROUPS || dense) break;
#else
        if (total >= 512 || dense) break;
#endif
    }
    p.subs_per_image = (size_t)p.n_groups * kOwned;
    p.total_subs = (size_t)p.total_groups * kOwned;
    p.mapped = n_images > kPlanScanImages;
    p.dense = dense;
    p.total_blocks = (uint32_t)(mcus_per_image * 3);
    // Block records four at a time (BlockWriter::put_record): where the launch has workgroups with work to spare, and
    // not for pictures full of detail, where a thread has few blocks.  Write pass, records singly / in fours:
    // 256 smooth 1080p q=50 frames (1024 workgroups, 5 bits per block) 220 / 149 us, q=90 (4352; 21 bits) 879 / 813,
    // eight photo-like 4K pictures (1440; 55 bits) 277 / 281.5, one smooth 4K picture (180) level.
    const uint64_t busy_groups = total_scan_bytes / ((uint64_t)kOwned * 4 * p.sub_words);
    p.records_in_fours = busy_groups >= 256 && (uint64_t)total_scan_bytes * 8 < 40ull * p.total_blocks * n_images;
    return p;
}

void fill_group_map(const DecodePlan& plan, const uint64_t* scan_sizes, uint32_t* img_first, uint32_t* map) {
    const uint32_t group_bytes = decode_group_bytes(plan.sub_words);
    uint32_t at = 0;
    for (uint32_t i = 0; i < plan.n_images; ++i) {
        const uint32_t n = scan_sizes != nullptr ? groups_of_scan(scan_sizes[i], group_bytes) : plan.n_groups;
        img_first[i] = at;
        for (uint32_t g = 0; g < n; ++g, ++at) {
            map[2 * (size_t)at] = i;
            map[2 * (size_t)at + 1] = g;
        }
    }
}

namespace {
__global__ void report_kernel(const uint32_t* __restrict__ unsettled, const int32_t* __restrict__ status,
                              DecodeReport* __restrict__ report) {
    report->unsettled = *unsettled;
    report->status = *status;
}

template <uint32_t W>
struct DecodeRun {
    hipStream_t s;
    const uint64_t* d_sizes;
    const DecodePlan& p;
    const DecodeScratch& sc;
    const StepTables* d_tables;
    int16_t* d_coefs;
    int32_t* d_status;
    // reading the raw scan (no unstuffed copy): the files, the scans' offsets in them, and the first kernel's side jobs
    bool raw = false;
    const uint8_t* d_scan = nullptr;
    const uint64_t* d_offsets = nullptr;
    const DecodePrologue* prologue = nullptr;
    bool tables_per_image = false;   // d_tables is an array, one set per image (launch_build_step_tables)

    dim3 grid() const { return dim3(p.total_groups); }
    GroupMap group_map() const { return GroupMap{reinterpret_cast<const uint2*>(sc.grp_map), sc.img_first, p.n_images}; }
    RawSource source() const { return raw ? RawSource{d_scan, d_offsets, sc.grp_end} : RawSource{}; }
    void sync_round(int first, uint32_t* flag, bool with_guesses) const {
        if (raw) {
            RawPrologue rp{};
            if (first) {
                rp.status = d_status;
                rp.check_flag = sc.changed + 2;
                if (prologue != nullptr) {
                    for (uint32_t i = 0; i < prologue->n_words; ++i) rp.words.v[i] = prologue->h_words[i];
                    rp.n_words = prologue->n_words;
                    rp.d_words = const_cast<uint64_t*>(d_offsets);
                    rp.headers_out = prologue->headers_out;
                    rp.header_bytes = prologue->header_bytes;
                }
            }
            decode_sync_kernel<W, true><<<grid(), kThreads, 0, s>>>(nullptr, nullptr, d_sizes, nullptr, group_map(), first, d_tables,
                                                                   sc.sub_end, sc.sub_used, sc.sub_info, sc.grp_sum,
                                                                   sc.grp_pending, nullptr, nullptr, sc.mode.lum_mask, flag,
                                                                   source(), rp, tables_per_image ? 1u : 0u);
        } else {
            decode_sync_kernel<W, false><<<grid(), kThreads, 0, s>>>(sc.ustream, sc.ubase, d_sizes, sc.img_drops, group_map(), first,
                                                                    d_tables, sc.sub_end, sc.sub_used, sc.sub_info, sc.grp_sum,
                                                                    sc.grp_pending, with_guesses ? sc.sub_guess : nullptr,
                                                                    sc.first_bad, sc.mode.lum_mask, flag, RawSource{}, RawPrologue{},
                                                                    tables_per_image ? 1u : 0u);
        }
    }
    template <bool kPlanar, bool kFold, bool kRaw, bool kFours>
    void write_launch(uint32_t* check) const {
        decode_write_kernel<W, kPlanar, kFold, kRaw, kFours><<<grid(), kThreads, 0, s>>>(
            sc.ustream, sc.ubase, d_sizes, sc.img_drops, group_map(), p.total_blocks, d_tables, sc.sub_used, sc.sub_info,
            sc.grp_base, d_coefs, kPlanar ? sc.out_records : nullptr, sc.mode.lum_mask, d_status, sc.grp_sum, sc.sub_end,
            sc.grp_pending, check, source(), tables_per_image ? 1u : 0u);
    }
    template <bool kPlanar, bool kFold, bool kRaw>
    void write(uint32_t* check) const {
        if (kPlanar && p.records_in_fours) write_launch<kPlanar, kFold, kRaw, kPlanar>(check);
        else write_launch<kPlanar, kFold, kRaw, false>(check);
    }
    // The unstuffed-copy decoder is the fall-back for streams that need repair rounds: it always scans in a launch
    // of its own.
    void scan_and_write(uint32_t* check = nullptr) const {
        const bool fold = raw && p.n_groups <= kFoldGroups;
        if (!fold)
            scan_groups_kernel<<<p.n_images, kThreads, 0, s>>>(sc.grp_sum, group_map(), p.total_blocks, sc.grp_base,
                                                               d_status, d_sizes, sc.img_drops, 32 * W, sc.sub_end,
                                                               sc.sub_used, sc.grp_pending, check, raw ? sc.grp_end : nullptr);
        const bool planar = sc.out_records != nullptr;
        if (raw) {
            if (planar && fold) write<true, true, true>(check);
            else if (planar) write<true, false, true>(check);
            else if (fold) write<false, true, true>(check);
            else write<false, false, true>(check);
        } else {
            if (planar) write<true, false, false>(check);
            else write<false, false, false>(check);
        }
    }

    // Two rounds up front: the first decodes everything, the second repairs the starts of the
    // workgroups.  The scan then checks what a third round would: that nothing is left to repair --
    // which is the proof of the fixed point.  The write pass and whatever consumes the coefficients
    // follow at once; the outcome of the check is reported together with the status, so the usual
    // decode never waits for the GPU in the middle.
    hipError_t first_pass(const DecodeFollowUp& follow_up, DecodeReport* report) const {
        sync_round(1, sc.changed + 0, false);
        sync_round(0, sc.changed + 1, false);
        // Dense streams synchronise late (see plan_entropy_decode): two more rounds of repair steps before the check.
        // A round in which nothing is left to repair costs its launch, some 10 us.
        if (p.dense) {
            for (int extra = 0; extra < JEA_DENSE_EXTRA_ROUNDS; ++extra) sync_round(0, sc.changed + 1, false);
        }
        if (sc.marks != nullptr) (void)hipEventRecord(sc.marks[1], s);
        scan_and_write(sc.changed + 2);
        if (sc.marks != nullptr) (void)hipEventRecord(sc.marks[2], s);
        const bool relayed = follow_up && follow_up(DecodeReportRelay{sc.changed + 2, d_status, report});
        if (sc.marks != nullptr) (void)hipEventRecord(sc.marks[3], s);
        if (!relayed) report_kernel<<<1, 1, 0, s>>>(sc.changed + 2, d_status, report);
        return hipGetLastError();
    }

    // The stream does not re-synchronise somewhere.  More rounds follow, each of which extends the
    // proven head of the stream by at least kMaxLocalIterations sub-sequences; in the first few, the
    // proven start states are offered as guesses wherever the same 512 stream bits come before
    // (periodic streams of flat or patterned pictures then settle in a handful of rounds).  What the
    // write pass produced from the unfinished states (bounded, but meaningless) is produced again.
    hipError_t repair(const DecodeFollowUp& follow_up, DecodeReport* report) const {
        hipError_t e;
        const uint64_t max_rounds = (uint64_t)p.subs_per_image / kMaxLocalIterations + p.n_groups + 8;
        if ((e = hipMemsetAsync(sc.ctx_keys, 0, (size_t)sc.ctx_slots * sizeof(unsigned long long), s)) != hipSuccess) return e;
        for (uint64_t extra = 0; report->unsettled != 0; ++extra) {
            if (extra > max_rounds) return hipErrorUnknown;
            const bool guesses = extra < 64;
            if (guesses) {
                if ((e = hipMemsetAsync(sc.first_bad, 0xff, p.n_images * sizeof(uint32_t), s)) != hipSuccess) return e;
                first_unproven_kernel<W><<<grid(), kThreads, 0, s>>>(d_sizes, sc.img_drops, group_map(), sc.sub_end,
                                                                    sc.sub_used, sc.first_bad);
                context_kernel<W, true><<<grid(), kThreads, 0, s>>>(sc.ustream, sc.ubase, d_sizes, sc.img_drops, group_map(),
                                                                   sc.first_bad, sc.sub_used, sc.ctx_keys, sc.ctx_vals,
                                                                   sc.ctx_slots - 1, sc.sub_guess);
                context_kernel<W, false><<<grid(), kThreads, 0, s>>>(sc.ustream, sc.ubase, d_sizes, sc.img_drops, group_map(),
                                                                    sc.first_bad, sc.sub_used, sc.ctx_keys, sc.ctx_vals,
                                                                    sc.ctx_slots - 1, sc.sub_guess);
            }
            if ((e = hipMemsetAsync(sc.changed + 3, 0, sizeof(uint32_t), s)) != hipSuccess) return e;
            sync_round(0, sc.changed + 3, guesses);
            report_kernel<<<1, 1, 0, s>>>(sc.changed + 3, d_status, report);
            if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;
        }
        if ((e = hipMemsetAsync(d_status, 0, sizeof(int32_t), s)) != hipSuccess) return e;
        scan_and_write();
        const bool relayed = follow_up && follow_up(DecodeReportRelay{sc.changed + 3, d_status, report});
        if (!relayed) report_kernel<<<1, 1, 0, s>>>(sc.changed + 3, d_status, report);
        if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;
        return hipGetLastError();
    }
};
}  // namespace

// bit_stream.cpp:136-147 for every stream of the job: count, (scan,) compact.  Clears *d_status and the round flags.
hipError_t launch_unstuff(hipStream_t s, const uint8_t* d_scan, const uint64_t* d_offsets, const uint64_t* d_sizes,
                          const DecodePlan& p, const DecodeScratch& sc, int32_t* d_status, const DecodePrologue& pro) {
    const dim3 cgrid(p.n_chunks, p.n_images);
    SegmentWords words{};
    if (pro.n_words > kSeedWords || (pro.n_words != 0 && pro.n_words != 3 * p.n_images)) return hipErrorInvalidValue;
    for (uint32_t i = 0; i < pro.n_words; ++i) words.v[i] = pro.h_words[i];
    unstuff_count_kernel<<<cgrid, kThreads, 0, s>>>(d_scan, d_offsets, d_sizes, p.n_chunks, sc.chunk_drops, sc.changed,
                                                    d_status, words, pro.n_words, const_cast<uint64_t*>(d_offsets),
                                                    pro.headers_out, pro.header_bytes);
    if (p.n_chunks <= kFoldChunks) {
        unstuff_compact_kernel<true><<<cgrid, kThreads, 0, s>>>(d_scan, d_offsets, d_sizes, p.n_chunks, sc.chunk_drops,
                                                                nullptr, sc.img_drops, sc.ubase, sc.ustream);
    } else {
        scan_chunks_kernel<<<p.n_images, 1024, 0, s>>>(sc.chunk_drops, p.n_chunks, sc.chunk_base, sc.img_drops);
        unstuff_compact_kernel<false><<<cgrid, kThreads, 0, s>>>(d_scan, d_offsets, d_sizes, p.n_chunks, sc.chunk_drops,
                                                                 sc.chunk_base, sc.img_drops, sc.ubase, sc.ustream);
    }
    return hipGetLastError();
}

hipError_t launch_entropy_decode(hipStream_t s, const uint8_t* d_scan, const uint64_t* d_offsets,
                                 const uint64_t* d_sizes, const DecodePlan& p, const DecodeScratch& sc,
                                 const StepTables* d_tables, int16_t* d_coefs, int32_t* d_status,
                                 const DecodeFollowUp& follow_up, DecodeReport* report, const DecodePrologue& pro, bool raw) {
    hipError_t e;
    if (pro.n_words > kSeedWords || (pro.n_words != 0 && pro.n_words != 3 * p.n_images)) return hipErrorInvalidValue;
    if (!raw && (e = launch_unstuff(s, d_scan, d_offsets, d_sizes, p, sc, d_status, pro)) != hipSuccess) return e;
    if (sc.marks != nullptr) (void)hipEventRecord(sc.marks[0], s);
    switch (p.sub_words) {
        case kSubWordsSmall:
            return DecodeRun<kSubWordsSmall>{s, d_sizes, p, sc, d_tables, d_coefs, d_status, raw, d_scan, d_offsets, &pro, sc.tables_per_image}.first_pass(follow_up, report);
        case kSubWordsMedium:
            return DecodeRun<kSubWordsMedium>{s, d_sizes, p, sc, d_tables, d_coefs, d_status, raw, d_scan, d_offsets, &pro, sc.tables_per_image}.first_pass(follow_up, report);
        default:
            return DecodeRun<kSubWordsLarge>{s, d_sizes, p, sc, d_tables, d_coefs, d_status, raw, d_scan, d_offsets, &pro, sc.tables_per_image}.first_pass(follow_up, report);
    }
}

hipError_t repair_entropy_decode(hipStream_t s, const uint64_t* d_sizes, const DecodePlan& p, const DecodeScratch& sc,
                                 const StepTables* d_tables, int16_t* d_coefs, int32_t* d_status,
                                 const DecodeFollowUp& follow_up, DecodeReport* report) {
    switch (p.sub_words) {
        case kSubWordsSmall:
            return DecodeRun<kSubWordsSmall>{s, d_sizes, p, sc, d_tables, d_coefs, d_status, false, nullptr, nullptr, nullptr, sc.tables_per_image}.repair(follow_up, report);
        case kSubWordsMedium:
            return DecodeRun<kSubWordsMedium>{s, d_sizes, p, sc, d_tables, d_coefs, d_status, false, nullptr, nullptr, nullptr, sc.tables_per_image}.repair(follow_up, report);
        default:
            return DecodeRun<kSubWordsLarge>{s, d_sizes, p, sc, d_tables, d_coefs, d_status, false, nullptr, nullptr, nullptr, sc.tables_per_image}.repair(follow_up, report);
    }
}

}  // namespace jea
