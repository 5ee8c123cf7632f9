/*
 * jpegenc_amd.h -- C-ABI of the MI355X (gfx950) baseline-JPEG encode/decode library.
 *
 * This is the boundary the C++ host classes (namespace jpeg: Encoder,
 * BaselineEncoder and their components) call through, and what any other host
 * language would bind.  Plain pointers and sizes only; no exceptions cross it.
 * Every function returns JEA_OK (0) or a JEA_ERR_* code; jea_last_error() gives
 * the text, which for the decode-side checks is the reference's own message.
 *
 * The functions replace these pieces of the reference (paths relative to the
 * reference root):
 *   whole encode   jpeg/src/encoder.cpp:14-42   (Encoder::encode)
 *   whole decode   jpeg/src/encoder.cpp:44-71   (Encoder::decode)
 *   stage entry points: see each declaration.
 *
 * Image layout everywhere: RGB8, row-major, top-down, tightly packed
 * (jpeg/inc/bitmap_image.hpp:20-25: m_imageData[y * m_width + x] = {r,g,b}).
 * Coefficient layout: int16, [image][mcu][channel][64] in zig-zag order, MCUs in
 * raster order, channels Y, Cb, Cr -- the order the reference entropy-codes them
 * in (encoder.cpp:21-27).
 * Quantisation tables: 64 x uint16, natural (row-major) order, as held by the
 * reference's Quantiser (jpeg/inc/quantiser.hpp:24-25).
 *
 * Pointers named d_* are device (HIP) pointers on the context's GPU; pointers
 * named h_* are host pointers.  A context owns one HIP stream and its scratch
 * buffers and must be used from one thread at a time; distinct contexts are
 * independent (one per GPU, or several per GPU).
 */
#ifndef JPEGENC_AMD_H_
#define JPEGENC_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JEA_OK 0
#define JEA_ERR_INVALID_ARG 1   /* null pointer, zero size, width/height out of 1..65535 */
#define JEA_ERR_HIP 2           /* a HIP runtime call failed; text has the HIP error */
#define JEA_ERR_CAPACITY 3      /* output (or scratch sized from it) too small */
#define JEA_ERR_RANGE 4         /* coefficient outside the Huffman tables' categories */
#define JEA_ERR_BAD_HEADER 5    /* decode: a header field check failed (encoder.cpp:140-283) */
#define JEA_ERR_BAD_STREAM 6    /* decode: invalid Huffman code / run (entropy_encoder.cpp:776,826,842) */
#define JEA_ERR_NO_EOI 7        /* decode: "Failed to find EOI marker" (encoder.cpp:63-65) */
#define JEA_ERR_NO_DEVICE 8     /* no usable gfx950 device / library built without device code */

#define JEA_HEADER_BYTES 607    /* SOI..SOS written by encoder.cpp:74-138 */

typedef struct jea_ctx jea_ctx;

/* ---- context ---------------------------------------------------------- */

/* Creates a context on HIP device `device_id` with its own stream. */
int jea_create(int device_id, jea_ctx** out_ctx);
void jea_destroy(jea_ctx* ctx);
/* Message of the last failing call on this context (or of jea_create when ctx is NULL). */
const char* jea_last_error(const jea_ctx* ctx);
/* Use a caller-owned hipStream_t for all following work (NULL = the context's own stream). */
int jea_set_stream(jea_ctx* ctx, void* hip_stream);
/* The hipStream_t the context launches on (so callers can record events on it). */
void* jea_get_stream(jea_ctx* ctx);
/* Blocks until all work enqueued on the context's stream has finished. */
int jea_synchronize(jea_ctx* ctx);

/* Upper bound (bytes, default 32 GiB) of the scratch one whole-encode call may keep per context: the
 * coefficients and worst-case bit strings of an image take 15.9 bytes per pixel, so a batch of 1024
 * 1920x1080 frames would take 34 GB.  A batch that needs more than the budget is encoded in passes of as
 * many images as fit, on the same stream; files, offsets and sizes are the same as from one pass.  When
 * even a buffer within the budget does not fit the GPU's free memory the call returns JEA_ERR_CAPACITY.
 * No counterpart in the reference (which holds one block at a time). */
int jea_set_scratch_budget(jea_ctx* ctx, size_t bytes);

/* Timing of the library's own kernels, for benchmarks (no counterpart in the reference, whose demo
 * times whole calls with std::chrono, examples/encode-decode.cpp:103-121).  While profiling is on, the
 * whole-encode and whole-decode entry points record HIP events on the context's stream at their phase
 * boundaries; jea_last_phase_ms waits for the last of them and returns the phases' durations in
 * milliseconds:
 *   encode (jea_encode_batch[_async], jea_encode_batch_static_async, jea_encode_host): 3 phases --
 *     forward transform kernel | bit-string kernel | scans + pack-and-stuff kernel;
 *   decode (jea_decode_batch, jea_decode_host, jea_entropy_decode): 4 phases --
 *     unstuffing | the two synchronisation rounds | scan + write pass | inverse transform kernel.
 * A decode whose stream the two standard rounds do not settle (it is decoded again by the variant with repair
 * rounds) reports no phases: jea_last_phase_ms then fails as if no profiled call had been made.
 * Events between kernels cost a little; leave profiling off when measuring throughput. */
int jea_set_profiling(jea_ctx* ctx, int on);
int jea_last_phase_ms(jea_ctx* ctx, float* ms, size_t capacity, size_t* n_phases);

/* ---- tables (host-side helpers, no GPU work) -------------------------- */

/* Colour mapper of everything the context does from now on (default JEA_MAPPER_YCBCR):
 * JEA_MAPPER_YCBCR  RGBToYCbCrMapper, colour_mapping.cpp:44-97 -- what BaselineEncoder uses;
 * JEA_MAPPER_RGB    RGBToRGBMapper, colour_mapping.cpp:16-42: the three components are R, G, B
 *                   as they are and all of them use the luminance quantiser and Huffman tables.
 * The file header does not depend on it (encoder.cpp:74-138 writes the same component table for
 * any mapper), so only a decoder set to the same mapper reads such a file back. */
#define JEA_MAPPER_YCBCR 0
#define JEA_MAPPER_RGB 1
int jea_set_colour_mapper(jea_ctx* ctx, int mapper);

/* Per-image Huffman tables (default off).  The reference writes the Annex K tables into every file and names
 * tailored ones as future work ("implement frequency analysis to generate tailored Huffman tables - requires
 * two passes in encoder", entropy_encoder.cpp:614; the header writer's note at :637-638 says the same).  With
 * on != 0 jea_encode_batch[_async] and jea_encode_host make that second pass: symbols are counted per image,
 * each of the four tables gets the code lengths of a Huffman tree over (count + 1) for every symbol of its
 * alphabet -- so each table keeps all its symbols and the header keeps its 607 bytes and its layout -- limited to
 * 16 bits as in T.81 K.3, and the DHT segment carries these tables.  jea_decode_batch and jea_decode_host then
 * read the tables from each file's header instead of assuming the Annex K ones (the reference's decoder skips
 * the segment, encoder.cpp:240).  Files are smaller (a smooth 512x512 picture at q=50: 6393 instead of 10570
 * bytes) and decode to the same pixels.  Not available through the stripe, static-scratch, scan-only and
 * per-block entry points, which keep the Annex K tables (the first two refuse the mode). */
int jea_set_huffman_optimised(jea_ctx* ctx, int on);

/* Quality -> tables, the arithmetic of Quantiser::Quantiser (quantiser.cpp:3-55). */
void jea_quant_tables(int quality, uint16_t lum_q[64], uint16_t chr_q[64]);
/* The fixed 607-byte header of encoder.cpp:74-138 (+ quantiser.cpp:87-102,
 * entropy_encoder.cpp:636-694) for a width x height image. */
void jea_write_header(uint8_t out[JEA_HEADER_BYTES], uint16_t width, uint16_t height,
                      const uint16_t lum_q[64], const uint16_t chr_q[64]);
/* Upper bound of one JPEG file: header + worst-case stuffed scan + pad + EOI. */
size_t jea_max_jpeg_bytes(uint16_t width, uint16_t height);
/* Header validation of Encoder::decodeHeader (encoder.cpp:140-283).  Returns
 * JEA_OK or JEA_ERR_BAD_HEADER; `msg` (may be NULL) receives the reference's
 * message for the first failing field; hdr_width/hdr_height the SOF0 fields. */
int jea_check_header(const uint8_t* h_jpeg, size_t len, uint16_t* hdr_width, uint16_t* hdr_height,
                     const char** msg);

/* Reads what a file of this format says about itself: the SOF0 dimensions and the
 * two quantisation tables of its DQT segment (natural order; the segment holds the
 * 8-bit truncation of each entry in zig-zag order, quantiser.cpp:87-102).  The
 * reference decoder ignores both (it takes the size from the JPEGImage fields and
 * the tables from the encoder object, encoder.cpp:46,187-188); with this call a
 * file can be decoded without knowing the quality it was written with:
 *   jea_read_header(file, len, &w, &h, lum, chr);  jea_decode_host(ctx, file, len, w, h, lum, chr, rgb);
 * which equals the reference's decode whenever no table entry exceeded 255 at
 * encode time.  Fails (JEA_ERR_BAD_HEADER) on what jea_check_header rejects, on
 * a table entry of 0, and on a DHT segment that declares other Huffman tables than
 * the reference's own (entropy_encoder.cpp:614-694), which are the only ones the
 * decoder has.  No counterpart in the reference. */
int jea_read_header(const uint8_t* h_jpeg, size_t len, uint16_t* width, uint16_t* height, uint16_t lum_q[64],
                    uint16_t chr_q[64]);

/* The same for files that may carry per-image Huffman tables (jea_set_huffman_optimised): those are accepted
 * when their DHT segment has this library's layout -- the four table ids in the reference's order
 * (entropy_encoder.cpp:672-693), 12 and 162 values, DC codes of at most 11 bits -- and *own_huffman_tables
 * says whether the tables differ from the reference's.  If they do, decode with a context on which
 * jea_set_huffman_optimised(ctx, 1) has been called.  No counterpart in the reference. */
int jea_read_header_ex(const uint8_t* h_jpeg, size_t len, uint16_t* width, uint16_t* height, uint16_t lum_q[64],
                       uint16_t chr_q[64], int* own_huffman_tables);

/* ---- whole encode: Encoder::encode, encoder.cpp:14-42 ------------------ */

/* Encodes `n_images` equally sized device-resident images in one pass and
 * returns when the result is complete.  Image i is at d_rgb + i*3*width*height.
 * JPEG file i (header, scan, EOI) is written to d_out + h_offsets[i] and is
 * h_sizes[i] bytes long; files are packed back to back at 64-byte aligned
 * offsets.  `out_capacity` bounds d_out; n_images * jea_max_jpeg_bytes() always
 * suffices, less is fine when the images compress (JEA_ERR_CAPACITY otherwise). */
int jea_encode_batch(jea_ctx* ctx, const uint8_t* d_rgb, size_t n_images, uint16_t width, uint16_t height,
                     const uint16_t lum_q[64], const uint16_t chr_q[64], uint8_t* d_out, size_t out_capacity,
                     uint64_t* h_offsets, uint64_t* h_sizes);

/* Same work, asynchronous: enqueues on the context's stream and returns.
 * d_offsets / d_sizes (n_images x uint64 each) and d_status (one int32: 0 or a
 * JEA_ERR_* code) are device buffers filled by the kernels; read them after
 * jea_synchronize() or from later work on the same stream.  On the context's own stream, batches of 128 images
 * and more run as pieces of 64 that alternate between that stream and a second stream of the context (the
 * transform of one piece beside the entropy coding of the one before); the second stream starts after what is
 * already on the first and the first ends the call, so the ordering a caller sees is that of a single stream.
 * On a stream given with jea_set_stream everything stays on that stream. */
int jea_encode_batch_async(jea_ctx* ctx, const uint8_t* d_rgb, size_t n_images, uint16_t width, uint16_t height,
                           const uint16_t lum_q[64], const uint16_t chr_q[64], uint8_t* d_out,
                           size_t out_capacity, uint64_t* d_offsets, uint64_t* d_sizes, int32_t* d_status);

/* jea_encode_batch_async for callers that cannot allocate device memory while they
 * call: code under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc.  It has no counterpart in the reference.  Same arguments, same
 * kernels, same output, but the scratch (coefficients, bit strings, per-tile
 * sums) lives in a buffer of JEA_STATIC_SCRATCH_BYTES that is part of the loaded
 * library -- one copy per GPU, held until the process ends -- so the call makes no
 * hipMalloc / hipFree and never synchronises.  Limits:
 *   - jea_static_scratch_needed(n_images, width, height) must not exceed
 *     JEA_STATIC_SCRATCH_BYTES (one 1920x1080 image needs 33.3 MB; the need grows
 *     with n_images * MCUs); larger jobs get JEA_ERR_CAPACITY and a message that
 *     states both numbers, and nothing is enqueued;
 *   - all contexts of a GPU share that GPU's buffer.  The library keeps its users
 *     apart: each call's kernels wait (an event of the GPU, no host wait, no
 *     allocation) for the kernels of the call that used the buffer before, from
 *     whichever context, stream or host thread that came -- so such calls never
 *     overlap on a GPU, they queue.  The exception is a stream under HIP graph
 *     capture, where an event cannot order work outside the graph: launches of
 *     graphs that contain this call, and other calls of it on the same GPU, must
 *     not overlap, or they race on the buffer.  Nothing in the buffer has to
 *     survive or to be cleared between calls;
 *   - the buffer costs JEA_STATIC_SCRATCH_BYTES of memory on every GPU that a
 *     process with this library loaded uses, whether or not the call is made. */
#define JEA_STATIC_SCRATCH_BYTES (48u << 20)
/* The same number from the loaded library (for bindings that cannot read the macro). */
size_t jea_static_scratch_bytes(void);
size_t jea_static_scratch_needed(size_t n_images, uint16_t width, uint16_t height);
int jea_encode_batch_static_async(jea_ctx* ctx, const uint8_t* d_rgb, size_t n_images, uint16_t width,
                                  uint16_t height, const uint16_t lum_q[64], const uint16_t chr_q[64],
                                  uint8_t* d_out, size_t out_capacity, uint64_t* d_offsets, uint64_t* d_sizes,
                                  int32_t* d_status);

/* One image held in host memory: copies it to the GPU, encodes, copies the file
 * back.  This is what BaselineEncoder::encode(BitmapImageRGB const&, JPEGImage&)
 * calls.  h_out must hold out_capacity bytes; *out_len receives the file size. */
int jea_encode_host(jea_ctx* ctx, const uint8_t* h_rgb, uint16_t width, uint16_t height,
                    const uint16_t lum_q[64], const uint16_t chr_q[64], uint8_t* h_out, size_t out_capacity,
                    size_t* out_len);

/* Batches held in host memory, as the reference's caller holds them (examples/encode-decode.cpp:104-118): n_images
 * bitmaps back to back in, files out at 64-byte aligned offsets (h_offsets / h_sizes as jea_encode_batch returns them) --
 * and the reverse.  The batch goes through the GPU in pieces of about 32 MB of pixels on two staging slots: the next
 * piece's way over the link, the kernels of the current piece and the way back of the piece before overlap.  Caller
 * memory from jea_host_malloc (pinned) is read and written by the GPU directly; other memory is copied through the
 * context's pinned staging buffers by a few host threads.  Both calls return when the results are complete. */
int jea_encode_host_batch(jea_ctx* ctx, const uint8_t* h_rgb, size_t n_images, uint16_t width, uint16_t height,
                          const uint16_t lum_q[64], const uint16_t chr_q[64], uint8_t* h_out, size_t out_capacity,
                          uint64_t* h_offsets, uint64_t* h_sizes);
int jea_decode_host_batch(jea_ctx* ctx, const uint8_t* h_jpeg, const uint64_t* h_offsets, const uint64_t* h_sizes,
                          size_t n_images, uint16_t width, uint16_t height, const uint16_t lum_q[64],
                          const uint16_t chr_q[64], uint8_t* h_rgb);
/* Pinned host memory for the calls above (hipHostMalloc / hipHostFree). */
int jea_host_malloc(jea_ctx* ctx, size_t bytes, void** h_ptr);
int jea_host_free(jea_ctx* ctx, void* h_ptr);

/* ---- one image as MCU-row stripes on several contexts / GPUs -----------
 * The reference codes an image as one serial chain (encoder.cpp:20-28): the DC
 * predictors and the bit position carry from MCU to MCU.  A stripe (a run of
 * whole MCU rows; every stripe but the last must be a multiple of 8 pixel rows
 * high) can still be encoded on its own context when three small facts travel
 * between the stripes' owners, in this order:
 *   1. jea_stripe_transform: transform the stripe; yields the quantised DC values
 *      of its last MCU.  -> every stripe needs the values of the stripe above.
 *   2. jea_stripe_bits: Huffman-code the stripe with that DC start; yields its bit
 *      count and the first 32 bits of its stream.  -> every stripe needs the sum
 *      of the counts above it and the head bits of the stripe below.
 *   3. jea_stripe_finish: pack at the global bit offset, byte-stuff; writes the
 *      stripe's segment of the file (the first stripe's starts with the header,
 *      the last one's ends with padding and EOI).
 * The file is the segments concatenated in stripe order.  The caller moves the
 * facts (all-gather between ranks, or a loop in one process). */
int jea_stripe_transform(jea_ctx* ctx, const uint8_t* d_rgb_stripe, uint16_t width, uint16_t stripe_height,
                         const uint16_t lum_q[64], const uint16_t chr_q[64], int16_t last_dc[3]);
int jea_stripe_bits(jea_ctx* ctx, const int16_t prev_dc[3], uint64_t* total_bits, uint32_t* head_bits);
/* is_first: write the 607-byte header for an image_width x image_height image first.
 * has_next: a stripe follows; next_head_bits are the head bits it reported. */
int jea_stripe_finish(jea_ctx* ctx, uint64_t bits_before, int is_first, int has_next, uint32_t next_head_bits,
                      uint16_t image_width, uint16_t image_height, const uint16_t lum_q[64],
                      const uint16_t chr_q[64], uint8_t* d_segment, size_t capacity, size_t* segment_bytes);

/* The same three steps for owners that exchange the facts from device memory to device memory (RCCL): every
 * call only enqueues on the context's stream, nothing is copied to the host and nothing waits, so a rank enqueues
 * transform, all-gather, bits, all-gather, finish and waits once at the end.  What replaces the host values:
 *   d_last_dc        4 x int16: the last MCU's quantised DC values and a zero;
 *   d_prev_dc        3 x int16 of the stripe above (e.g. inside the gathered array), NULL for the first stripe;
 *   d_facts          2 x uint64 of this stripe: its bit count and its first 32 stream bits;
 *   d_all_facts      n_stripes x 2 x uint64, every stripe's facts in stripe order: the bit offset of stripe
 *                    stripe_index (the sum of the counts above it, as the serial encoder's stream position would
 *                    be, bit_stream.cpp:44-67) and the head bits of the stripe below are taken from it on the GPU;
 *   d_segment_bytes  1 x uint64: the segment's size;   d_status  1 x int32: 0 or a JEA_ERR_* code.
 * Read d_segment_bytes and d_status after jea_synchronize() or from later work on the stream. */
int jea_stripe_transform_async(jea_ctx* ctx, const uint8_t* d_rgb_stripe, uint16_t width, uint16_t stripe_height,
                               const uint16_t lum_q[64], const uint16_t chr_q[64], int16_t* d_last_dc);
int jea_stripe_bits_async(jea_ctx* ctx, const int16_t* d_prev_dc, uint64_t* d_facts);
int jea_stripe_finish_async(jea_ctx* ctx, const uint64_t* d_all_facts, uint32_t stripe_index, uint32_t n_stripes,
                            uint16_t image_width, uint16_t image_height, const uint16_t lum_q[64],
                            const uint16_t chr_q[64], uint8_t* d_segment, size_t capacity, uint64_t* d_segment_bytes,
                            int32_t* d_status);

/* ---- whole decode: Encoder::decode, encoder.cpp:44-71 ------------------ */

/* Decodes n_images JPEG files produced by this encoder family (fixed header
 * layout, Annex-K Huffman tables).  File i is d_jpeg + h_offsets[i], h_sizes[i]
 * bytes.  As in the reference, the output dimensions are the caller's
 * (JPEGImage::m_width/m_height, encoder.cpp:46), not the SOF0 fields, and the
 * tables are the caller's, not the file's DQT (encoder.cpp:187-188).  Image i is
 * written to d_rgb + i*3*width*height.  When an error is returned the contents
 * of d_rgb are unspecified (jea_decode_host and the C++ Encoder::decode leave
 * the caller's output untouched in that case, as encoder.cpp:63-70 does). */
int jea_decode_batch(jea_ctx* ctx, const uint8_t* d_jpeg, const uint64_t* h_offsets, const uint64_t* h_sizes,
                     size_t n_images, uint16_t width, uint16_t height, const uint16_t lum_q[64],
                     const uint16_t chr_q[64], uint8_t* d_rgb);

/* One file in host memory -> one image in host memory
 * (BaselineEncoder::decode(JPEGImage, BitmapImageRGB&)). */
int jea_decode_host(jea_ctx* ctx, const uint8_t* h_jpeg, size_t len, uint16_t width, uint16_t height,
                    const uint16_t lum_q[64], const uint16_t chr_q[64], uint8_t* h_rgb);

/* ---- stage entry points (device pointers; synchronous) ---------------- */

/* Fused block fetch with edge replication + RGB->YCbCr + level shift + separable
 * 8x8 forward DCT + quantise + zig-zag.  Replaces block_grid.cpp:8-58,
 * colour_mapping.cpp:44-67, discrete_cosine_transform.cpp:3-6,13-19,77-124,
 * quantiser.cpp:57-70, entropy_encoder.cpp:15-77.
 * d_coefs: n_images * mcus * 3 * 64 int16. */
int jea_fdct_quant(jea_ctx* ctx, const uint8_t* d_rgb, size_t n_images, uint16_t width, uint16_t height,
                   const uint16_t lum_q[64], const uint16_t chr_q[64], int16_t* d_coefs);

/* jea_fdct_quant without the final synchronisation: enqueues on the context's
 * stream and returns (for callers that time or pipeline the kernel themselves). */
int jea_fdct_quant_async(jea_ctx* ctx, const uint8_t* d_rgb, size_t n_images, uint16_t width, uint16_t height,
                         const uint16_t lum_q[64], const uint16_t chr_q[64], int16_t* d_coefs);

/* The same transform as the fused encode path runs it (same reference lines): the
 * coefficients go out in the library's chunk-planar layout -- the 16-byte chunk c
 * (zig-zag positions 8c..8c+7) of block (mcu, channel) of image i is 16-byte element
 * ((i * 8 + c) * 3 + channel) * mcus + mcu of d_coefs_planar -- and only chunk 0 and
 * the chunks that hold a non-zero coefficient are written; d_chunk_masks
 * ([image][mcu][channel], one byte each) has bit c set for exactly those non-zero
 * chunks.  Unwritten chunks keep whatever the buffer held.  Asynchronous, like
 * jea_fdct_quant_async.  d_coefs_planar: n_images * mcus * 3 * 64 int16. */
int jea_fdct_quant_planar_async(jea_ctx* ctx, const uint8_t* d_rgb, size_t n_images, uint16_t width,
                                uint16_t height, const uint16_t lum_q[64], const uint16_t chr_q[64],
                                int16_t* d_coefs_planar, uint8_t* d_chunk_masks);

/* Bit length of every block-channel ([image][mcu][channel] order) with the DC
 * prediction chain of encoder.cpp:20 restarted per image.  Replaces
 * entropy_encoder.cpp:94-120 and the length side of :715-765.
 * d_lengths: n_images * mcus_per_image * 3 uint16. */
int jea_block_bit_lengths(jea_ctx* ctx, const int16_t* d_coefs, size_t n_images, size_t mcus_per_image,
                          uint16_t* d_lengths);

/* Entropy-coded scan segments from coefficient blocks: Huffman coding
 * (entropy_encoder.cpp:94-120, 696-765), MSB-first packing
 * (bit_stream.cpp:44-81), byte stuffing (:127-134) and zero-bit padding
 * (:101-105).  No header, no EOI.  Segment i is written to
 * d_out + h_offsets[i] (64-byte aligned), h_sizes[i] bytes. */
int jea_entropy_encode(jea_ctx* ctx, const int16_t* d_coefs, size_t n_images, size_t mcus_per_image,
                       uint8_t* d_out, size_t out_capacity, uint64_t* h_offsets, uint64_t* h_sizes);

/* Inverse of jea_entropy_encode on stuffed scan bytes that are followed by the
 * EOI marker: unstuffing (bit_stream.cpp:136-147), Huffman decoding
 * (entropy_encoder.cpp:703-712, 767-866), run-length removal including the
 * dropped last entry (:122-144), DC chain.  Segment i starts at
 * d_scan + h_offsets[i] and, with its 2-byte EOI, is h_sizes[i] bytes. */
int jea_entropy_decode(jea_ctx* ctx, const uint8_t* d_scan, const uint64_t* h_offsets, const uint64_t* h_sizes,
                       size_t n_images, size_t mcus_per_image, int16_t* d_coefs);

/* BitStream::removeStuffedBytes (bit_stream.cpp:136-147) on its own: d_scan[0 .. len) is scan data followed by
 * the two bytes of the EOI marker; the 0x00 after a 0xFF that lies before those two bytes is dropped --
 * with the reference's way of walking: the byte after a 0xFF is never itself the 0xFF of a pair, so of a run
 * of 0xFF bytes every second one counts.  d_out (len bytes of room) receives the *out_len remaining bytes. */
int jea_remove_stuffing(jea_ctx* ctx, const uint8_t* d_scan, size_t len, uint8_t* d_out, size_t* out_len);

/* De-zigzag + dequantise + separable 8x8 inverse DCT + level unshift +
 * YCbCr->RGB + cropped store.  Replaces entropy_encoder.cpp:79-92,
 * quantiser.cpp:72-85, discrete_cosine_transform.cpp:8-11,21-27,90-156,
 * colour_mapping.cpp:69-92, block_grid.cpp:87-99. */
int jea_idct_dequant(jea_ctx* ctx, const int16_t* d_coefs, size_t n_images, uint16_t width, uint16_t height,
                     const uint16_t lum_q[64], const uint16_t chr_q[64], uint8_t* d_rgb);

/* Per-pixel colour maps on n interleaved triples (exhaustive-sweep test surface
 * for colour_mapping.cpp:44-67 and :69-92). */
int jea_rgb_to_ycbcr(jea_ctx* ctx, const uint8_t* d_rgb, size_t n_pixels, uint8_t* d_ycc);
int jea_ycbcr_to_rgb(jea_ctx* ctx, const uint8_t* d_ycc, size_t n_pixels, uint8_t* d_rgb);

/* ---- component-level entry points (host pointers, synchronous) ---------
 * The granularity of the reference's injected components: one 8x8 block-channel
 * (or pixel block) per element.  They run the same device code as the image
 * kernels; they are meant for the per-block methods of the host classes and for
 * tests, not for throughput. */

/* ColourMapper::map / unmap on n pixels (colour_mapping.cpp:44-67, :69-92). */
int jea_rgb_to_ycbcr_host(jea_ctx* ctx, const uint8_t* h_rgb, size_t n_pixels, uint8_t* h_ycc);
int jea_ycbcr_to_rgb_host(jea_ctx* ctx, const uint8_t* h_ycc, size_t n_pixels, uint8_t* h_rgb);

/* DiscreteCosineTransformer::transform / inverseTransform with the separated
 * transform (discrete_cosine_transform.cpp:3-27, 77-156): n blocks of 64 u8
 * samples (index x + 8y) <-> 64 floats (index u + 8v). */
int jea_dct_forward_blocks(jea_ctx* ctx, const uint8_t* h_samples, size_t n_blocks, float* h_dct);
int jea_dct_inverse_blocks(jea_ctx* ctx, const float* h_dct, size_t n_blocks, uint8_t* h_samples);

/* Quantiser::quantise / dequantise (quantiser.cpp:57-85) with one table, natural order. */
int jea_quantise_blocks(jea_ctx* ctx, const float* h_dct, size_t n_blocks, const uint16_t table[64],
                        int16_t* h_quantised);
int jea_dequantise_blocks(jea_ctx* ctx, const int16_t* h_quantised, size_t n_blocks, const uint16_t table[64],
                          float* h_dct);

/* HuffmanEncoder encode of independent block-channels (entropy_encoder.cpp:94-120,
 * 696-765): zig-zag ordered coefficients, the previous DC value and the table
 * class (non-zero = luminance) per block.  h_lengths[i] receives the bit count,
 * h_bits[i*52 .. i*52+51] the bits, MSB first in each 32-bit word, zero padded. */
int jea_huffman_encode_blocks(jea_ctx* ctx, const int16_t* h_zigzag, const int16_t* h_prev_dc,
                              const uint8_t* h_is_luminance, size_t n_blocks, uint16_t* h_lengths, uint32_t* h_bits);

/* HuffmanEncoder decode of one block-channel (entropy_encoder.cpp:122-144, 703-712,
 * 767-866) from an unstuffed stream.  jea_stream_upload copies the stream to the
 * GPU once; jea_huffman_decode_block then decodes the block that starts at
 * `start_bit`, returns its zig-zag coefficients (with the reference's dropped
 * last run-length entry), the bit position after it and the updated DC value. */
int jea_stream_upload(jea_ctx* ctx, const uint8_t* h_stream, size_t len);
int jea_huffman_decode_block(jea_ctx* ctx, uint64_t start_bit, int is_luminance, int16_t* last_dc,
                             int16_t h_zigzag[64], uint64_t* end_bit);

/* HuffmanEncoder::applyFinalEncoding / removeFinalEncoding (entropy_encoder.cpp:696-712, 715-866): the
 * final-encoding step on the run-length form of one block-channel, the level at which the reference lets a
 * subclass of EntropyEncoder replace the coder (entropy_encoder.hpp:50-51).
 * Encode: the DC difference and then every (run, value) entry as given, the run taken modulo 256 as the
 * reference narrows it (entropy_encoder.cpp:733); (0, 0) is the end-of-block code, (15, 0) the zero-run code,
 * another entry with value 0 fails with the reference's "Invalid runtime encoding encountered."
 * (JEA_ERR_RANGE), as do entries the tables have no code for (the reference indexes past its tables there).  h_bits needs
 * (22 + 26 * n_entries + 31) / 32 words; bits are MSB first in each word, zero padded.
 * Decode: from the stream given to jea_stream_upload, starting at start_bit; entries until one is the end of
 * block or the positions they cover add up to exactly 63 (:707-710).  A damaged stream whose count steps over
 * 63 keeps the reference reading; here the call fails (JEA_ERR_BAD_STREAM) once 64 entries are out. */
int jea_huffman_encode_rle(jea_ctx* ctx, int16_t dc_difference, const uint16_t* h_runs, const int16_t* h_values,
                           size_t n_entries, int is_luminance, uint32_t* length_bits, uint32_t* h_bits,
                           size_t bits_capacity_words);
int jea_huffman_decode_rle(jea_ctx* ctx, uint64_t start_bit, int is_luminance, int16_t* dc_difference,
                           uint16_t h_runs[64], int16_t h_values[64], size_t* n_entries, uint64_t* end_bit);

/* ---- device memory helpers for callers without a HIP binding ---------- */
int jea_malloc(jea_ctx* ctx, size_t bytes, void** d_ptr);
int jea_free(jea_ctx* ctx, void* d_ptr);
int jea_memcpy_h2d(jea_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);
int jea_memcpy_d2h(jea_ctx* ctx, void* h_dst, const void* d_src, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* JPEGENC_AMD_H_ */
