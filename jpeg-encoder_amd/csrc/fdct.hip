// Fused forward kernel: block fetch with edge replication + RGB->YCbCr + level
// shift + separable 8x8 forward DCT + quantise + zig-zag, one pass over the
// image, plus the matching inverse kernel and the per-pixel colour kernels.
//
// Arithmetic contract (bit-exact with the reference, paths relative to its root):
//   colour      jpeg/src/colour_mapping.cpp:44-67 / :69-92   double, left to right
//   level shift jpeg/src/discrete_cosine_transform.cpp:13-27
//   FDCT        :77-88, :103-124   float, every product and sum rounded, no FMA,
//               8-term sums in index order, scale applied after the sum
//   IDCT        :90-101, :126-156  (scale*cos) pre-multiplied, clamp, truncation
//   quantise    jpeg/src/quantiser.cpp:57-70   float divide, +0.5 and floor in double
//   dequantise  :72-85
//   zig-zag     jpeg/src/entropy_encoder.cpp:15-92
//   edges       jpeg/src/block_grid.cpp:8-28 (replicate), :87-99 (crop)
// This file must be compiled with -ffp-contract=off.
//
// Two forward kernels share the arithmetic (dct_device.h).
// fdct_quant_rows_kernel -- pictures whose sides are multiples of 8 (both output layouts): lane = one pixel row /
// one column of a block, the block's values change hands inside the wave through LDS; see the comment above it.
// fdct_quant_kernel -- every other case (ragged sides, rows of fewer than 8 MCUs): a workgroup of 192 threads
// owns 64 consecutive MCUs.
// Wave w (64 lanes) is channel w (Y, Cb, Cr) and lane l is MCU l, so channel
// dependent constants are wave-uniform (scalar registers) and one thread carries
// one whole 8x8 block-channel in registers through both DCT passes -- no
// cross-lane traffic in the transform.  Pixels come in through LDS with
// coalesced 8-byte loads of the 8 row strips; coefficient blocks go out through
// LDS so that the 24 KiB the workgroup produces is stored as one contiguous run
// of 16-byte stores.
#include <hip/hip_runtime.h>

#include "dct_device.h"
#include "kernels.h"
#include "tables.h"

namespace jea {

namespace {

constexpr int kMcusPerGroup = 64;
constexpr int kFdctThreads = 192;
// idct_sparse_kernel's occupancy target.  256 smooth 1920x1080 q=50 frames: 8 waves per SIMD (64 registers, spills)
// 598 us, 7: 475, 6: 391, 5 and 4: 372.
#ifndef JEA_SPARSE_MIN_WAVES   // experiments: tools/build_variant.sh <name> -DJEA_SPARSE_MIN_WAVES=<n>
#define JEA_SPARSE_MIN_WAVES 5
#endif
constexpr uint32_t kSparseListCapacity = 32768;   // groups idct_sparse_kernel can hand to idct_dequant_kernel

#ifndef JEA_K1_MIN_WAVES
#define JEA_K1_MIN_WAVES 5
#endif
template <bool kFast, bool kPlanar>
__global__ __launch_bounds__(kFdctThreads, JEA_K1_MIN_WAVES) void fdct_quant_kernel(const uint8_t* __restrict__ rgb, int width,
                                                                  int height, int blocks_w, uint32_t n_mcu,
                                                                  size_t image_bytes, QuantTables q,
                                                                  int16_t* __restrict__ coefs,
                                                                  int32_t* __restrict__ status_to_clear,
                                                                  uint8_t* __restrict__ chunk_masks,
                                                                  ComponentMode mode) {
    // First kernel of an encode pass: clears the pass's status word so that no
    // separate memset launch is needed (later kernels only ever raise it).
    if (status_to_clear != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *status_to_clear = 0;
    // One buffer, used twice: first the pixel rows of the 64 MCUs
    // ([row 0..7][mcu 0..63][8 px * 3 B], 12 KiB), then, after every wave has
    // finished its row pass, the finished blocks ([mcu][channel][128 B] with the
    // 16-byte chunks rotated by the block index, 24 KiB).
    // (chunk-planar output leaves from registers: only the 12 KiB of pixel rows are needed then,
    // which leaves LDS for the entropy kernels of other images running beside this one)
    __shared__ __align__(16) uint8_t lds[kPlanar ? 8 * kMcusPerGroup * 24 : kFdctThreads * 128];
    uint8_t* const lds_in = lds;
    uint8_t* const lds_out = lds;
    // {RN(1/Q), Q} of both quantisation tables in the order the column pass walks them ([u][v]): read from
    // here they reach the quantiser in vector registers (a scalar-register operand would halve its issue rate)
    __shared__ float2 lds_qt[128];

    const int tid = threadIdx.x;
    if (tid < 128) {
        // lum_t | chr_t and rlum_t | rchr_t lie side by side in QuantTables
        const float* qb = reinterpret_cast<const float*>(&q);
        lds_qt[tid] = make_float2(qb[256 + tid], qb[128 + tid]);
    }
    const uint32_t g0 = blockIdx.x * kMcusPerGroup;
    const uint8_t* img = rgb + (size_t)blockIdx.y * image_bytes;
    const uint32_t mcus_here = (n_mcu - g0 < (uint32_t)kMcusPerGroup) ? (n_mcu - g0) : (uint32_t)kMcusPerGroup;

    if (kFast) {
        // width % 8 == 0: every MCU row strip is 24 in-bounds bytes at an 8-byte
        // aligned address; thread t moves piece t%3 of MCU t/3 for each of the 8 rows.
        const int m = tid / 3, part = tid - 3 * m;
        if ((uint32_t)m < mcus_here) {
            const uint32_t mcu = g0 + m;
            const uint32_t by = mcu / (uint32_t)blocks_w, bx = mcu - by * (uint32_t)blocks_w;
            uint2 v[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                int y = (int)by * 8 + r;
                y = y < height ? y : height - 1;  // bottom edge: repeat the last row
                v[r] = *reinterpret_cast<const uint2*>(img + ((size_t)y * width + bx * 8) * 3 + part * 8);
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) *reinterpret_cast<uint2*>(lds_in + r * 1536 + tid * 8) = v[r];
        }
    } else {
        // Any width or alignment: the same split of the work; MCUs that lie wholly inside the picture still
        // load their row strips eight bytes at a time (unaligned loads), only the MCUs of a ragged last
        // column gather bytes with the x clamp of block_grid.cpp:16-26 (the y clamp serves both).
        const int m = tid / 3, part = tid - 3 * m;
        if ((uint32_t)m < mcus_here) {
            const uint32_t mcu = g0 + m;
            const uint32_t by = mcu / (uint32_t)blocks_w, bx = mcu - by * (uint32_t)blocks_w;
            const bool whole = (int)bx * 8 + 8 <= width;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                int y = (int)by * 8 + r;
                y = y < height ? y : height - 1;
                const uint8_t* row = img + (size_t)y * width * 3;
                uint2 v = make_uint2(0, 0);
                if (whole) {
                    __builtin_memcpy(&v, row + (size_t)bx * 24 + part * 8, 8);
                } else {
                    uint32_t w2[2] = {0, 0};
                    for (int k = 0; k < 8; ++k) {
                        const int byte = part * 8 + k, px = byte / 3, comp = byte - 3 * px;
                        int x = (int)bx * 8 + px;
                        x = x < width ? x : width - 1;
                        w2[k >> 2] |= (uint32_t)row[(size_t)x * 3 + comp] << (8 * (k & 3));
                    }
                    v = make_uint2(w2[0], w2[1]);
                }
                *reinterpret_cast<uint2*>(lds_in + r * 1536 + tid * 8) = v;
            }
        }
    }
    __syncthreads();

    const int channel = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = tid & 63;
    const uint32_t j = (uint32_t)m * 3 + (uint32_t)channel;  // block index inside the group
    const bool active = (uint32_t)m < mcus_here;
    float row[8][8];
    if (active) {
        double ck, ca, cb, cc;
        colour_constants(channel, ck, ca, cb, cc);
        ColourInt ci = colour_int_constants(channel);
        // all four constants live in vector registers: an FMA with a scalar-register operand issues at half
        // the rate of one whose operands are all vector registers
        asm volatile("v_mov_b32 %0, %1" : "=v"(ci.k) : "s"(ci.k));
        asm volatile("v_mov_b32 %0, %1" : "=v"(ci.a) : "s"(ci.a));
        asm volatile("v_mov_b32 %0, %1" : "=v"(ci.b) : "s"(ci.b));
        asm volatile("v_mov_b32 %0, %1" : "=v"(ci.c) : "s"(ci.c));
        if (mode.passthrough) {
            // RGBToRGBMapper (colour_mapping.cpp:16-26): the component as it is.  A loop of its own,
            // so that the usual path carries no test per pixel.
#pragma unroll
            for (int y = 0; y < 8; ++y) {
                const uint2* src = reinterpret_cast<const uint2*>(lds_in + y * 1536 + m * 24);
                const uint2 w0 = src[0], w1 = src[1], w2 = src[2];
                const uint32_t d[6] = {w0.x, w0.y, w1.x, w1.y, w2.x, w2.y};
                float f[8];
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    const int b0 = 3 * p, b1 = 3 * p + 1, b2 = 3 * p + 2;
                    const uint32_t ri = (d[b0 >> 2] >> (8 * (b0 & 3))) & 0xffu;
                    const uint32_t gi = (d[b1 >> 2] >> (8 * (b1 & 3))) & 0xffu;
                    const uint32_t bi = (d[b2 >> 2] >> (8 * (b2 & 3))) & 0xffu;
                    f[p] = (float)(channel == 0 ? ri : (channel == 1 ? gi : bi)) - 128.0f;
                }
                fdct_row(f, y, row);
            }
        } else {
#pragma unroll
            for (int y = 0; y < 8; ++y) {
                const uint2* src = reinterpret_cast<const uint2*>(lds_in + y * 1536 + m * 24);
                const uint2 w0 = src[0], w1 = src[1], w2 = src[2];
                const uint32_t d[6] = {w0.x, w0.y, w1.x, w1.y, w2.x, w2.y};
                float f[8];
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    const int b0 = 3 * p, b1 = 3 * p + 1, b2 = 3 * p + 2;
                    const uint32_t ri = (d[b0 >> 2] >> (8 * (b0 & 3))) & 0xffu;
                    const uint32_t gi = (d[b1 >> 2] >> (8 * (b1 & 3))) & 0xffu;
                    const uint32_t bi = (d[b2 >> 2] >> (8 * (b2 & 3))) & 0xffu;
                    bool tie;
                    f[p] = colour_forward_shifted_fast(ci, (float)ri, (float)gi, (float)bi, tie);
                    if (__builtin_expect(__any(tie), 0)) {
                        // the literal formula; floor(v + 0.5) is an integer in [-1, 256]: exact as a float, so
                        // the clamp to [0, 255] (:49-51) can be done there; then "- 128" into int8 (:13-19)
                        const float rounded =
                            (float)floor((((ck + ca * (double)ri) + cb * (double)gi) + cc * (double)bi) + 0.5);
                        f[p] = __builtin_amdgcn_fmed3f(rounded, 0.0f, 255.0f) - 128.0f;
                    }
                }
                fdct_row(f, y, row);
            }
        }
    }
    __syncthreads();  // every wave is done with lds_in: the buffer becomes lds_out

    if (active) {
        const float2* qtab = lds_qt + (mode.is_lum(channel) ? 0 : 64);
        int qv[64];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            float o[8];
            fdct_col(row, u, o);
            float2 c[8];
#pragma unroll
            for (int v = 0; v < 8; ++v) c[v] = qtab[8 * u + v];
            quantise_col(o, c, u, qv);
        }
        // zig-zag (entropy_encoder.cpp:15-77); hand-over through LDS, or, chunk-planar, straight to
        // memory: the lanes of a wave are consecutive MCUs of one channel, i.e. consecutive 16-byte
        // elements of a plane, and chunks that are all zero (most, in most pictures) are not stored
        uint4* plane = reinterpret_cast<uint4*>(coefs) + ((size_t)blockIdx.y * 24 + channel) * n_mcu + g0 + m;
        uint32_t mask = 0;
#pragma unroll
        for (int k8 = 0; k8 < 8; ++k8) {
            uint4 chunk;
            chunk.x = pack16(qv[kZigzag[8 * k8 + 0]], qv[kZigzag[8 * k8 + 1]]);
            chunk.y = pack16(qv[kZigzag[8 * k8 + 2]], qv[kZigzag[8 * k8 + 3]]);
            chunk.z = pack16(qv[kZigzag[8 * k8 + 4]], qv[kZigzag[8 * k8 + 5]]);
            chunk.w = pack16(qv[kZigzag[8 * k8 + 6]], qv[kZigzag[8 * k8 + 7]]);
            const bool nonzero = (chunk.x | chunk.y | chunk.z | chunk.w) != 0;
            mask |= (nonzero ? 1u : 0u) << k8;
            if (kPlanar) {
                if (k8 == 0 || nonzero) plane[(size_t)k8 * 3 * n_mcu] = chunk;
            } else {
                *reinterpret_cast<uint4*>(lds_out + j * 128 + ((((uint32_t)k8 + j) & 7u) << 4)) = chunk;
            }
        }
        if (chunk_masks != nullptr) chunk_masks[((size_t)blockIdx.y * n_mcu + g0) * 3 + j] = (uint8_t)mask;
    }
    if (kPlanar) return;
    __syncthreads();

    // [mcu][channel][64] int16 of this group is one contiguous run in memory
    uint4* out = reinterpret_cast<uint4*>(coefs + ((size_t)blockIdx.y * n_mcu + g0) * 192);
    const uint32_t n_chunks = mcus_here * 3 * 8;
    for (uint32_t i = tid; i < n_chunks; i += kFdctThreads) {
        const uint32_t jj = i >> 3, kk = i & 7u;
        out[i] = *reinterpret_cast<const uint4*>(lds_out + jj * 128 + (((kk + jj) & 7u) << 4));
    }
}

// ---------------------------------------------------------------------------
// The forward kernel for pictures whose sides are multiples of 8 (rows of at least 8 MCUs): the same arithmetic, laid
// out over the lanes by block ROW instead of by block.
//
// A wave works on 8 consecutive MCUs at a time (a "unit").  Lane l = 8 m + s: m is the MCU; s is, in turn, the
// pixel row the lane converts and transforms (row pass), the horizontal frequency whose column it transforms
// and quantises (column pass), and the 16-byte chunk of the finished block that it stores.  Between the phases
// the 8 x 8 values of a block change hands inside the wave through LDS (no workgroup barrier: a wave's LDS
// instructions complete in order).  A lane carries one row or column -- 8 values, not 64 -- so
//   * the three channels of a pixel row are done by one lane one after the other and every byte is converted
//     to float once, not once per channel,
//   * the quantised values leave as 16-bit LDS stores at their zig-zag place: no packing instructions,
//   * the kernel needs few registers, and a wave's work per unit is a few hundred instructions instead of three
//     thousand: launches have no long tail, and the loads of the next unit are in flight during the arithmetic.
// A wave takes `units_per_wave` consecutive units; the launcher picks that number so that the waves of a launch
// fill the compute units evenly.
#ifndef JEA_ROW_WAVES   // experiments: tools/build_variant.sh <name> -DJEA_ROW_WAVES=<n> -DJEA_ROW_MIN_WAVES=<n>
#define JEA_ROW_WAVES 4
#endif
#ifndef JEA_ROW_MIN_WAVES
#define JEA_ROW_MIN_WAVES 4
#endif
constexpr int kRowWaves = JEA_ROW_WAVES;
#ifndef JEA_ROW_DYN_LDS   // experiments: bytes of unused LDS per workgroup, which limits how many share a compute unit
#define JEA_ROW_DYN_LDS 0
#endif
constexpr int kRowDynLds = JEA_ROW_DYN_LDS;
constexpr int kRowThreads = 64 * kRowWaves;
// LDS layouts, chosen with the banking rules of the LDS in mind (stores: bank = dword mod 32, conflicts among the 32
// lanes of a half wave, two-way conflicts hidden behind the store's register transfer; 16-byte loads: bank = dword
// mod 64, conflicts among four fixed groups of 16 lanes that each take m = 0..3 or 4..7 with s = 0..3 or 4..7):
//  * transposition buffer, in 16-byte slots: value (frequency u, row y) of MCU m lives in slot
//    kRowTSlot[m & 3] + 128 (m >> 2) + 66 (y >> 2) + 2 u, word y & 3.  The row pass stores with word offsets 8 u
//    (two-way conflicts), the column pass loads two slots 66 apart (conflict-free).
//  * zig-zag buffer: block m starts at word kRowZWord[m & 3] + 148 (m >> 2).
//  * quantisation tables: row u of a table is 16 floats {RN(1/Q), Q} x 8 at a stride of 20 floats, so that the
//    eight rows a group of lanes reads lie in different banks.
constexpr int kRowTSlot[4] = {0, 1, 41, 40};
constexpr int kRowTWords = 1024;
constexpr int kRowZWord[4] = {0, 36, 116, 80};
constexpr int kRowZWords = 320;
constexpr int kRowQStride = 20;

// The eight sums of discrete_cosine_transform.cpp:103-112 (and :114-124, which has the same shape) over one row
// or column; products and sums round one by one, in index order (see fdct_row).
__device__ __forceinline__ void fdct_1d(const float (&in)[8], float (&out)[8]) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        float acc = fwd_c(u, 0) * in[0];
#pragma unroll
        for (int x = 1; x < 8; ++x) acc += fwd_c(u, x) * in[x];
        out[u] = (u == 0) ? scale_s(0) * acc : acc;
    }
}

// byte s of the result: twice the zig-zag position of natural index 8 v + s
constexpr uint64_t zigzag_row_doubled(int v) {
    uint64_t r = 0;
    for (int s = 0; s < 8; ++s)
        for (int k = 0; k < 64; ++k)
            if (kZigzag[k] == 8 * v + s) r |= (uint64_t)(2 * k) << (8 * s);
    return r;
}

constexpr uint64_t kZigzagRowDoubled[8] = {zigzag_row_doubled(0), zigzag_row_doubled(1), zigzag_row_doubled(2), zigzag_row_doubled(3),
                                           zigzag_row_doubled(4), zigzag_row_doubled(5), zigzag_row_doubled(6), zigzag_row_doubled(7)};

// byte v of entry u: twice the zig-zag position of natural index 8 v + u (the inverse kernel's lanes hold a row v)
constexpr uint64_t zigzag_col_doubled(int u) {
    uint64_t r = 0;
    for (int v = 0; v < 8; ++v)
        for (int k = 0; k < 64; ++k)
            if (kZigzag[k] == 8 * v + u) r |= (uint64_t)(2 * k) << (8 * v);
    return r;
}
constexpr uint64_t kZigzagColDoubled[8] = {zigzag_col_doubled(0), zigzag_col_doubled(1), zigzag_col_doubled(2), zigzag_col_doubled(3),
                                           zigzag_col_doubled(4), zigzag_col_doubled(5), zigzag_col_doubled(6), zigzag_col_doubled(7)};

// what one wave's LDS traffic must not be reordered across (the lanes exchange data with each other)
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <bool kPlanar, bool kStrided = false>
__global__ __launch_bounds__(kRowThreads, JEA_ROW_MIN_WAVES) void fdct_quant_rows_kernel(
    const uint8_t* __restrict__ rgb, int width, int height, int blocks_w, uint32_t n_mcu, size_t image_bytes, QuantTables q,
    int16_t* __restrict__ coefs, int32_t* __restrict__ status_to_clear, uint8_t* __restrict__ chunk_masks, ComponentMode mode,
    uint32_t units_per_wave) {
    if (status_to_clear != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *status_to_clear = 0;
    __shared__ __align__(16) float lds_q[2 * 8 * kRowQStride];   // [table][u]{RN(1/Q), Q}[v]
    __shared__ __align__(16) float lds_t[kRowWaves][kRowTWords];
    __shared__ __align__(16) uint32_t lds_z[kRowWaves][kRowZWords];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t lane = (uint32_t)tid & 63u, m = lane >> 3, s = lane & 7u;
    const uint32_t n_units = (n_mcu + 7u) >> 3;
    // kStrided (experiment): as many waves as the GPU holds at once, wave w takes units w, w + waves, w + 2 waves, ...
    const uint32_t unit_step = kStrided ? gridDim.x * kRowWaves : 1u;
    uint32_t unit = kStrided ? (uint32_t)blockIdx.x * kRowWaves + (uint32_t)wave
                             : ((uint32_t)blockIdx.x * kRowWaves + (uint32_t)wave) * units_per_wave;
    const uint32_t unit_end = kStrided ? n_units : (unit + units_per_wave < n_units ? unit + units_per_wave : n_units);
    const bool has_work = unit < unit_end;   // per wave; false only in the last workgroup of a picture

    // MCU of this lane in the first unit; later units are 8 MCUs further each.  Width and height are multiples
    // of 8 here (the launcher's condition), so a row of MCUs is a row of 24 blocks_w bytes, no row is repeated,
    // and a lane's pixels move 192 bytes per unit, plus seven pixel rows when its MCU wraps into the next row.
    // The first unit's pixels are asked for before anything else: every other step of the set-up hides behind them.
    // (asked for even earlier, so that they arrive first: the two table entries this thread copies into LDS;
    // lum_t | chr_t and rlum_t | rchr_t lie side by side in QuantTables, entry 8 u + v each)
    float2 q_pair = make_float2(0.0f, 0.0f);
    if (tid < 128) {
        const float* qb = reinterpret_cast<const float*>(&q);
        q_pair = make_float2(qb[256 + tid], qb[128 + tid]);
    }
    uint32_t mcu = unit * 8u + m;
    const uint32_t row_bytes = (uint32_t)width * 3u;
    // the tail of a picture's last unit reads the last MCU and is never stored; so does a wave without work, so
    // that the loads are issued unconditionally and the copy of the tables need not wait for them
    const uint32_t mcu_in = mcu < n_mcu ? mcu : n_mcu - 1u;
    // mcu_in / blocks_w through a float quotient and one correction either way (exact: both are below 2^24, the
    // launcher's condition; the integer division costs twice the instructions)
    uint32_t by0 = (uint32_t)((float)mcu_in * __builtin_amdgcn_rcpf((float)blocks_w));
    int32_t bx_signed = (int32_t)mcu_in - (int32_t)(by0 * (uint32_t)blocks_w);
    by0 = bx_signed < 0 ? by0 - 1u : (bx_signed >= blocks_w ? by0 + 1u : by0);
    bx_signed = bx_signed < 0 ? bx_signed + blocks_w : (bx_signed >= blocks_w ? bx_signed - blocks_w : bx_signed);
    uint32_t bx = (uint32_t)bx_signed;
    const uint8_t* src = rgb + (size_t)blockIdx.y * image_bytes + ((size_t)by0 * 8 + s) * row_bytes + (size_t)bx * 24;
    uint2 cur[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) cur[i] = reinterpret_cast<const uint2*>(src)[i];
    if (tid < 128) {
        float* to = lds_q + (tid >> 6) * 8 * kRowQStride + ((tid >> 3) & 7) * kRowQStride + 2 * (tid & 7);
        to[0] = q_pair.x;
        to[1] = q_pair.y;
    }
    __syncthreads();
    if (!has_work) return;   // no workgroup barrier below

    // kRowTSlot[m & 3] and kRowZWord[m & 3] without a table in memory
    const uint32_t m_lo = m & 1u, m_hi = (m >> 1) & 1u;
    const uint32_t t_slot = 40u * m_hi + (m_lo ^ m_hi) + 128u * (m >> 2);
    const uint32_t z_word = 36u * (m_lo ^ m_hi) + 80u * m_hi + 148u * (m >> 2);
    static_assert(kRowTSlot[0] == 0 && kRowTSlot[1] == 1 && kRowTSlot[2] == 41 && kRowTSlot[3] == 40, "layout");
    static_assert(kRowZWord[0] == 0 && kRowZWord[1] == 36 && kRowZWord[2] == 116 && kRowZWord[3] == 80, "layout");
    float* const t_write = lds_t[wave] + 4u * (t_slot + 66u * (s >> 2)) + (s & 3u);   // + 8 u: value of frequency u, row s
    const float* const t_read = lds_t[wave] + 4u * (t_slot + 2u * s);   // rows 0-3 of frequency s; rows 4-7 66 slots on
    uint8_t* const z_block = reinterpret_cast<uint8_t*>(lds_z[wave] + z_word);
    // Where coefficient (v, u = s) goes: its zig-zag position, two bytes each.  The eight positions of row v,
    // doubled, are the bytes of one 64-bit constant; the lane takes byte s.
    uint32_t z_off[8];
    const uint32_t pick_byte_s = 0x0c0c0c00u | s;   // byte-permute selector: byte s of the eight, zero-extended
#pragma unroll
    for (int v = 0; v < 8; ++v) {
        const uint64_t row = kZigzagRowDoubled[v];
        z_off[v] = __builtin_amdgcn_perm((uint32_t)(row >> 32), (uint32_t)row, pick_byte_s);
    }

    // colour constants of the three channels in vector registers (an operand in a scalar register halves the
    // issue rate of an FMA)
    ColourInt ci[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        ci[ch] = colour_int_constants(ch);
        asm volatile("v_mov_b32 %0, %1" : "=v"(ci[ch].k) : "s"(ci[ch].k));
        asm volatile("v_mov_b32 %0, %1" : "=v"(ci[ch].a) : "s"(ci[ch].a));
        asm volatile("v_mov_b32 %0, %1" : "=v"(ci[ch].b) : "s"(ci[ch].b));
        asm volatile("v_mov_b32 %0, %1" : "=v"(ci[ch].c) : "s"(ci[ch].c));
    }
    // Output addresses as a uniform base plus a 32-bit offset per lane (the launcher checks that a picture's
    // planes stay below 4 GiB): chunk s of channel ch of this lane's MCU, and the MCU's three mask bytes.
    // (kPlanar: the chunk planes of the fused encode, chunk 0 and non-zero chunks only; else [mcu][channel][64],
    // every chunk, and the mask bytes only if the caller wants them)
    const uint32_t plane_bytes = n_mcu * 16u;
    uint8_t* const out_image = reinterpret_cast<uint8_t*>(coefs) + (size_t)blockIdx.y * 24 * plane_bytes;
    uint8_t* const mask_image = chunk_masks + (size_t)blockIdx.y * n_mcu * 3;
    const uint32_t channel_step = kPlanar ? plane_bytes : 128u;
    uint32_t out_off = kPlanar ? s * 3u * plane_bytes + mcu * 16u : mcu * 384u + 16u * s, mask_off = mcu * 3u;

    for (; unit < unit_end; unit += unit_step) {
        const bool valid = mcu < n_mcu;
        const uint32_t d[6] = {cur[0].x, cur[0].y, cur[1].x, cur[1].y, cur[2].x, cur[2].y};
        float px[24];
#pragma unroll
        for (int i = 0; i < 24; ++i) {
            px[i] = (float)((d[i >> 2] >> (8 * (i & 3))) & 0xffu);
            asm volatile("" : "+v"(px[i]));   // from here on the floats are all there is (no second, integer copy of the bytes)
        }
        // the next unit's pixels are on their way while this one is worked on
        if (kStrided) {
            if (unit + unit_step < unit_end) {
                const uint32_t next = mcu + 8u * unit_step, next_in = next < n_mcu ? next : n_mcu - 1u;
                uint32_t by = (uint32_t)((float)next_in * __builtin_amdgcn_rcpf((float)blocks_w));
                int32_t bxs = (int32_t)next_in - (int32_t)(by * (uint32_t)blocks_w);
                by = bxs < 0 ? by - 1u : (bxs >= blocks_w ? by + 1u : by);
                bxs = bxs < 0 ? bxs + blocks_w : (bxs >= blocks_w ? bxs - blocks_w : bxs);
                src = rgb + (size_t)blockIdx.y * image_bytes + ((size_t)by * 8 + s) * row_bytes + (size_t)bxs * 24;
#pragma unroll
                for (int i = 0; i < 3; ++i) cur[i] = reinterpret_cast<const uint2*>(src)[i];
            }
        } else if (unit + 1 < unit_end) {
            bx += 8u;
            const bool wraps = bx >= (uint32_t)blocks_w;
            bx -= wraps ? (uint32_t)blocks_w : 0u;
            uint32_t step = wraps ? 192u + 7u * row_bytes : 192u;
            step = mcu + 8u < n_mcu ? step : 0u;   // lanes past the last MCU stay where they are
            src += step;
#pragma unroll
            for (int i = 0; i < 3; ++i) cur[i] = reinterpret_cast<const uint2*>(src)[i];
        }

#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            float f[8];
            if (mode.passthrough) {   // RGBToRGBMapper (colour_mapping.cpp:16-26): the component as it is
#pragma unroll
                for (int p = 0; p < 8; ++p) f[p] = px[3 * p + ch] - 128.0f;
            } else {
                // One test per pixel row: the smallest fraction of the eight values against the window (four min3 and
                // a compare instead of eight half-rate compares and eight branches).  Only where some lane's row has a
                // value near a tie (46 % of a wave's luminance rows, 10 % of the others) are the values looked at one
                // by one, and the literal formula runs for those that are.  2 % on the kernel.
                float fr[8];
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    const float qv = __builtin_fmaf(ci[ch].c, px[3 * p + 2], __builtin_fmaf(ci[ch].b, px[3 * p + 1], __builtin_fmaf(ci[ch].a, px[3 * p], ci[ch].k)));
                    fr[p] = __builtin_amdgcn_fractf(qv);
                    f[p] = qv - fr[p];
                }
                // (min3 throughout: fminf would first canonicalise both operands, two more instructions)
                const float smallest = min3(min3(min3(fr[0], fr[1], fr[2]), fr[3], fr[4]), min3(fr[5], fr[6], fr[7]), fr[7]);
                if (__builtin_expect(__any(smallest < ci[ch].window), 0)) {
#pragma unroll
                    for (int p = 0; p < 8; ++p) {
                        bool tie;
                        (void)colour_forward_shifted_fast(ci[ch], px[3 * p], px[3 * p + 1], px[3 * p + 2], tie);
                        if (__any(tie)) {   // the literal formula, as in fdct_quant_kernel
                            double ck, ca, cb, cc;
                            colour_constants(ch, ck, ca, cb, cc);
                            const float rounded = (float)floor(
                                (((ck + ca * (double)px[3 * p]) + cb * (double)px[3 * p + 1]) + cc * (double)px[3 * p + 2]) + 0.5);
                            f[p] = __builtin_amdgcn_fmed3f(rounded, 0.0f, 255.0f) - 128.0f;
                        }
                    }
                }
            }
            float o[8];
            fdct_1d(f, o);   // row pass: o[u] = rowDCT[8u + s]
#pragma unroll
            for (int u = 0; u < 8; ++u) t_write[8 * u] = o[u];
            wave_lds_fence();
            float in[8];
            {
                const float4 a = *reinterpret_cast<const float4*>(t_read), b = *reinterpret_cast<const float4*>(t_read + 4 * 66);
                in[0] = a.x; in[1] = a.y; in[2] = a.z; in[3] = a.w;
                in[4] = b.x; in[5] = b.y; in[6] = b.z; in[7] = b.w;
            }
            fdct_1d(in, o);   // column pass: o[v] = out[8v + s]
            const float4* qrow = reinterpret_cast<const float4*>(lds_q + (mode.is_lum(ch) ? 0 : 8 * kRowQStride) + kRowQStride * s);
#pragma unroll
            for (int v2 = 0; v2 < 4; ++v2) {
                const float4 c = qrow[v2];
                *reinterpret_cast<int16_t*>(z_block + z_off[2 * v2]) = (int16_t)quantise_rn(o[2 * v2], c.x, c.y);
                *reinterpret_cast<int16_t*>(z_block + z_off[2 * v2 + 1]) = (int16_t)quantise_rn(o[2 * v2 + 1], c.z, c.w);
            }
            wave_lds_fence();
            const uint4 chunk = *reinterpret_cast<const uint4*>(z_block + 16 * s);
            wave_lds_fence();   // the next channel's values overwrite both buffers
            const bool nonzero = (chunk.x | chunk.y | chunk.z | chunk.w) != 0;
            const uint64_t nz = __ballot(nonzero);
            if (valid) {
                if (!kPlanar || s == 0 || nonzero) *reinterpret_cast<uint4*>(out_image + (size_t)ch * channel_step + out_off) = chunk;
                if (s == 0 && (kPlanar || chunk_masks != nullptr))
                    (mask_image + ch)[mask_off] = (uint8_t)((m < 4 ? (uint32_t)nz : (uint32_t)(nz >> 32)) >> (8 * (m & 3u)));
            }
        }
        mcu += 8u * unit_step;
        out_off += (kPlanar ? 8u * 16u : 8u * 384u) * unit_step;
        mask_off += 8u * 3u * unit_step;
    }
}

// zig-zag position of natural index n
constexpr int zigzag_position(int n) {
    for (int i = 0; i < 64; ++i)
        if (kZigzag[i] == n) return i;
    return 0;
}

// Inverse transform of one block in two steps around a workgroup barrier, so that the sample
// buffer may reuse the LDS space of the coefficients.  Coefficients outside the triangle
// u + v <= D are known to be zero (D = 7: no such knowledge).  The sums keep the reference's order
// and simply stop where only zero terms would follow: adding c * 0 can change nothing but the
// sign of a zero, which the final truncation to an integer does not see.
//
// Row step: de-zigzag (entropy_encoder.cpp:79-92), dequantise (quantiser.cpp:72-85) and row pass
// (discrete_cosine_transform.cpp:126-135): tmp[x][v] = sum_u (s(u)C[u][x]) * src[u + 8v].
// `chunk[c]` is the LDS address of the block's 16-byte chunk c (zig-zag positions 8c..8c+7).
template <int D>
__device__ __forceinline__ void idct_rows(const uint8_t* const (&chunk)[8], const float* __restrict__ qt,
                                          float (&tmp)[8][8]) {
    constexpr int kRows = D < 7 ? D + 1 : 8;
#pragma unroll
    for (int v = 0; v < kRows; ++v) {
        const int terms = (D < 7) ? (D - v + 1) : 8;
        float deq[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (u < terms) {
                const int k = zigzag_position(8 * v + u);
                const int c = (int)*reinterpret_cast<const int16_t*>(chunk[k >> 3] + 2 * (k & 7));
                deq[u] = (float)c * qt[8 * v + u];
            }
        }
#pragma unroll
        for (int x = 0; x < 8; ++x) {
            float acc = inv_c(0, x) * deq[0];
#pragma unroll
            for (int u = 1; u < terms; ++u) acc += inv_c(u, x) * deq[u];
            tmp[x][v] = acc;
        }
    }
}

// Column step :136-156: out[8y+x] = clamp/truncate(sum_v (s(v)C[v][y]) * tmp[x][v]); +128 (:21-27).
template <int D>
__device__ __forceinline__ void idct_cols(const float (&tmp)[8][8], uint8_t* __restrict__ px) {
    constexpr int kRows = D < 7 ? D + 1 : 8;
#pragma unroll
    for (int y = 0; y < 8; ++y) {
        uint32_t packed[2] = {0, 0};
#pragma unroll
        for (int x = 0; x < 8; ++x) {
            float acc = inv_c(0, y) * tmp[x][0];
#pragma unroll
            for (int v = 1; v < kRows; ++v) acc += inv_c(v, y) * tmp[x][v];
            // :144-154: below -128 -> -128, above 127 -> 127, else truncation toward zero; then + 128 (:21-27).
            // Truncating first and letting the byte conversion saturate gives the same byte: a sum below
            // -128 truncates to at most -128, i.e. at most 0 after the shift, one above 127 to at least 255
            packed[x >> 2] = insert_byte(__builtin_truncf(acc) + 128.0f, x & 3, packed[x >> 2]);
        }
        *reinterpret_cast<uint2*>(px + y * 8) = make_uint2(packed[0], packed[1]);
    }
}

// ---------------------------------------------------------------------------
// Decode side: de-zigzag + dequantise + IDCT + level unshift + YCbCr -> RGB.
// Same work split as the forward kernel (wave = channel, lane = MCU); the three
// channels of the MCUs meet in LDS, then each thread colour-converts whole pixels.
// ---------------------------------------------------------------------------

// Eight bytes of decoded pixels to memory.  Non-temporal: the kernels do not read them again, and kept out of the
// caches they leave alone what the write pass put there for these kernels to read.
__device__ __forceinline__ void store_pixels(uint8_t* to, uint32_t lo, uint32_t hi) {
#ifdef JEA_TEMPORAL_PIXEL_STORES   // experiments
    *reinterpret_cast<uint2*>(to) = make_uint2(lo, hi);
#else
    __builtin_nontemporal_store(lo, reinterpret_cast<uint32_t*>(to));
    __builtin_nontemporal_store(hi, reinterpret_cast<uint32_t*>(to) + 1);
#endif
}

template <bool kPlanar>
// Five waves per SIMD (96 registers, 104 bytes of spill) against four (122 registers): 1.5 % quicker on batches of
// detailed pictures, level elsewhere.
#ifndef JEA_IDCT_MIN_WAVES   // experiments: tools/build_variant.sh <name> -DJEA_IDCT_MIN_WAVES=<n>
#define JEA_IDCT_MIN_WAVES 5
#endif
__global__ __launch_bounds__(kFdctThreads, JEA_IDCT_MIN_WAVES) void idct_dequant_kernel(const int16_t* __restrict__ coefs,
                                                                    const uint32_t* __restrict__ records, int width,
                                                                    int height, int blocks_w, uint32_t n_mcu,
                                                                    size_t image_bytes, QuantTables q,
                                                                    uint8_t* __restrict__ rgb, ComponentMode mode,
                                                                    DecodeReportRelay relay,
                                                                    const uint32_t* __restrict__ group_list,
                                                                    uint32_t groups_per_image) {
    // Which group of 64 MCUs of which image: the grid says (x = group, y = image), or, behind idct_sparse_kernel,
    // the list of the groups that kernel left alone does (group_list[0] = entries, then image * groups + group each).
    uint32_t group_x = blockIdx.x, image = blockIdx.y;
    if (group_list != nullptr) {
        const uint32_t listed = group_list[0] < kSparseListCapacity ? group_list[0] : kSparseListCapacity;
        if (blockIdx.x >= listed) return;
        const uint32_t id = group_list[1 + blockIdx.x];
        image = id / groups_per_image;
        group_x = id - image * groups_per_image;
    }
    // the decoder's outcome, known since the kernels before this one ended, goes to the host on the side
    if (relay.report != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        relay.report->unsettled = *relay.unsettled;
        relay.report->status = *relay.status;
    }
    // coefficients as rotated 16-byte chunks (as in the forward kernel); after the row step the
    // first half is reused for the samples, [channel][mcu][64] u8
    __shared__ __align__(16) uint8_t lds_coef[kFdctThreads * 128];
    uint8_t* const lds_px = lds_coef;
    // per channel, the MCUs whose block is a DC value alone (all 64 samples equal); and the MCUs that have a
    // block with more than that, listed
    __shared__ uint64_t dc_alone_of[3];
    __shared__ uint8_t varied[kMcusPerGroup];

    const int tid = threadIdx.x;
    const uint32_t g0 = group_x * kMcusPerGroup;
    uint8_t* img = rgb + (size_t)image * image_bytes;
    const uint32_t mcus_here = (n_mcu - g0 < (uint32_t)kMcusPerGroup) ? (n_mcu - g0) : (uint32_t)kMcusPerGroup;

    if (!kPlanar) {
        const uint4* in = reinterpret_cast<const uint4*>(coefs + ((size_t)image * n_mcu + g0) * 192);
        const uint32_t n_chunks = mcus_here * 3 * 8;
        for (uint32_t i = tid; i < n_chunks; i += kFdctThreads) {
            const uint32_t jj = i >> 3, kk = i & 7u;
            *reinterpret_cast<uint4*>(lds_coef + jj * 128 + (((kk + jj) & 7u) << 4)) = in[i];
        }
        __syncthreads();
    }

    const int channel = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = tid & 63;
    const uint32_t j = (uint32_t)m * 3 + (uint32_t)channel;
    const bool valid = (uint32_t)m < mcus_here;
    const float* qt = mode.is_lum(channel) ? q.lum : q.chr;
    float tmp[8][8];
    float dc_only = 0.0f;
    int diag = 7;
    bool dc_alone = false;
    if (valid) {
        const uint8_t* chunk[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) chunk[c] = lds_coef + j * 128 + ((((uint32_t)c + j) & 7u) << 4);
        // How far do the non-zero coefficients of this wave's 64 blocks reach along the zig-zag?
        // The transform is then run on the smallest triangle u + v <= D that holds them all.
        bool rest_nonzero, far_nonzero;   // anything beyond chunk 0 / beyond chunk 1
        if (kPlanar) {
            // The decoder's layout: each thread fetches its block's record and the chunks that exist (the
            // lanes of a wave are consecutive 16-byte elements of a plane) into its LDS slot; chunks
            // beyond the first are not even looked at when the whole wave has none.
            const uint32_t mcu = g0 + (uint32_t)m;
            const uint32_t rec = records[((size_t)image * n_mcu + mcu) * 3 + channel];
            const uint32_t mask = (rec >> 16) & 0xffu;
            const uint4* plane = reinterpret_cast<const uint4*>(coefs) + ((size_t)image * 24 + channel) * n_mcu + mcu;
            rest_nonzero = (mask >> 1) != 0;
            far_nonzero = (mask >> 2) != 0;
            dc_alone = mask == 0;
            uint4 first = make_uint4(rec & 0xffffu, 0, 0, 0);   // DC alone
            if (mask & 1u) first = plane[0];
            *reinterpret_cast<uint4*>(const_cast<uint8_t*>(chunk[0])) = first;
            if (__ballot(rest_nonzero) != 0) {
#pragma unroll
                for (int c = 1; c < 8; ++c) {
                    uint4 v = make_uint4(0, 0, 0, 0);
                    if ((mask >> c) & 1u) v = plane[(size_t)c * 3 * n_mcu];
                    *reinterpret_cast<uint4*>(const_cast<uint8_t*>(chunk[c])) = v;
                }
            } else {
                // the widest of the small variants (u + v <= 3) reaches zig-zag positions 8 and 9
                *reinterpret_cast<uint4*>(const_cast<uint8_t*>(chunk[1])) = make_uint4(0, 0, 0, 0);
            }
        } else {
            uint32_t rest = 0, far = 0;
#pragma unroll
            for (int c = 1; c < 8; ++c) {
                const uint4 c4 = *reinterpret_cast<const uint4*>(chunk[c]);
                rest |= c4.x | c4.y | c4.z | c4.w;
                if (c >= 2) far |= c4.x | c4.y | c4.z | c4.w;
            }
            rest_nonzero = rest != 0;
            far_nonzero = far != 0;
            const uint4 c0 = *reinterpret_cast<const uint4*>(chunk[0]);
            dc_alone = (rest | (c0.x >> 16) | c0.y | c0.z | c0.w) == 0;
        }
        if (__ballot(rest_nonzero) == 0) {
            // words of the first chunk: zig-zag positions 0-1 | 2-3 | 4-5 | 6-7, which lie on the
            // diagonals u + v = 0,1 | 1,2 | 2,2 | 3,3
            const uint4 c0 = *reinterpret_cast<const uint4*>(chunk[0]);
            diag = __ballot(c0.w != 0) != 0 ? 3 : ((__ballot((c0.y | c0.z) != 0) != 0) ? 2 : 1);
            if (diag == 1 && __ballot((c0.x >> 16) != 0) == 0) {
                // DC alone in all 64 blocks: every sample of a block is the same value,
                // s(0)C[0][y] * (s(0)C[0][x] * deq[0]) with C[0][.] = 1, clamped and truncated as in idct_cols
                diag = 0;
                dc_only = inv_c(0, 0) * (inv_c(0, 0) * ((float)(int)(int16_t)(c0.x & 0xffffu) * qt[0]));
            }
        } else if (__ballot(far_nonzero) == 0) {
            diag = 5;   // zig-zag positions 0..15 (chunks 0 and 1) lie on the diagonals u + v <= 5
        }
        switch (diag) {
            case 0: break;
            case 1: idct_rows<1>(chunk, qt, tmp); break;
            case 2: idct_rows<2>(chunk, qt, tmp); break;
            case 3: idct_rows<3>(chunk, qt, tmp); break;
            case 5: idct_rows<5>(chunk, qt, tmp); break;
            default: idct_rows<7>(chunk, qt, tmp); break;
        }
    }
    {
        const uint64_t alone = __ballot(dc_alone);
        if (m == 0) dc_alone_of[channel] = alone;
    }
    __syncthreads();   // every coefficient has been read: the samples may overwrite them
    // An MCU whose three blocks are DC values alone is one colour: it is converted once.  The others go on a list.
    const uint64_t flat = dc_alone_of[0] & dc_alone_of[1] & dc_alone_of[2];
    const uint32_t n_varied = mcus_here - (uint32_t)__popcll(flat);
    if (channel == 0 && valid && !((flat >> m) & 1u)) {
        const uint64_t others = ~flat;
        varied[__builtin_amdgcn_mbcnt_hi((uint32_t)(others >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)others, 0u))] = (uint8_t)m;
    }
    if (valid) {
        uint8_t* dst = lds_px + (channel * kMcusPerGroup + m) * 64;
        switch (diag) {
            case 0: {
                const uint32_t o = (uint32_t)((int)__builtin_amdgcn_fmed3f(dc_only, -128.0f, 127.0f) + 128);
                const uint32_t four = o * 0x01010101u;
#pragma unroll
                for (int i = 0; i < 4; ++i) reinterpret_cast<uint4*>(dst)[i] = make_uint4(four, four, four, four);
                break;
            }
            case 1: idct_cols<1>(tmp, dst); break;
            case 2: idct_cols<2>(tmp, dst); break;
            case 3: idct_cols<3>(tmp, dst); break;
            case 5: idct_cols<5>(tmp, dst); break;
            default: idct_cols<7>(tmp, dst); break;
        }
    }
    __syncthreads();

    // Colour conversion and cropped store.  Fast path (width a multiple of 8, so every block row
    // segment is whole and 8-byte aligned): work item = (row r, mcu mm) = 8 pixels = 24 contiguous
    // bytes; consecutive threads take consecutive MCUs, i.e. consecutive 24-byte pieces of an image row.
    if ((width & 7) == 0) {
        // One colour: thread t takes MCU t / 3 and the 8-byte piece t % 3 of its 24-byte rows, so that the lanes of
        // a wave write consecutive bytes.
        const uint32_t fm = (uint32_t)tid / 3u, piece = (uint32_t)tid - 3u * fm;
        if ((flat >> fm) & 1u) {
            const uint32_t mcu = g0 + fm;
            const uint32_t by = mcu / (uint32_t)blocks_w, bx = mcu - by * (uint32_t)blocks_w;
            const int c0 = lds_px[(0 * kMcusPerGroup + fm) * 64], c1 = lds_px[(1 * kMcusPerGroup + fm) * 64],
                      c2 = lds_px[(2 * kMcusPerGroup + fm) * 64];
            float px3[3];
            if (mode.passthrough) {
                px3[0] = (float)c0; px3[1] = (float)c1; px3[2] = (float)c2;
            } else {
                bool tie;
                colour_inverse_fast_unclamped((float)c0, (float)c1, (float)c2, px3[0], px3[1], px3[2], tie);
                if (tie) {   // the literal formula
                    uint8_t rr, gg, bb;
                    colour_inverse(c0, c1, c2, rr, gg, bb);
                    px3[0] = (float)rr; px3[1] = (float)gg; px3[2] = (float)bb;
                }
            }
            // eight pixels are the 24 bytes RGBR GBRG BRGB RGBR GBRG BRGB; piece p starts at byte 8 p, with colour 2 p mod 3
            uint32_t w2[2] = {0, 0};
#pragma unroll
            for (int byte = 0; byte < 8; ++byte) {
                const uint32_t c = (2u * piece + (uint32_t)byte) % 3u;
                w2[byte >> 2] = insert_byte(c == 0 ? px3[0] : (c == 1 ? px3[1] : px3[2]), byte & 3, w2[byte >> 2]);
            }
            uint8_t* to = img + ((size_t)by * 8 * width + (size_t)bx * 8) * 3 + 8 * piece;
            const int rows = height - (int)by * 8 < 8 ? height - (int)by * 8 : 8;
            for (int r = 0; r < rows; ++r) store_pixels(to + (size_t)r * width * 3, w2[0], w2[1]);
        }
        for (uint32_t i = tid; i < 8u * n_varied; i += kFdctThreads) {
            const uint32_t r = i / n_varied, mm = varied[i - r * n_varied];
            const uint32_t mcu = g0 + mm;
            const uint32_t by = mcu / (uint32_t)blocks_w, bx = mcu - by * (uint32_t)blocks_w;
            const int y = (int)by * 8 + (int)r;
            if (y >= height) continue;
            const uint2 y8 = *reinterpret_cast<const uint2*>(lds_px + (0 * kMcusPerGroup + mm) * 64 + r * 8);
            const uint2 b8 = *reinterpret_cast<const uint2*>(lds_px + (1 * kMcusPerGroup + mm) * 64 + r * 8);
            const uint2 r8 = *reinterpret_cast<const uint2*>(lds_px + (2 * kMcusPerGroup + mm) * 64 + r * 8);
            const uint32_t yy[2] = {y8.x, y8.y}, cbb[2] = {b8.x, b8.y}, crr[2] = {r8.x, r8.y};
            uint32_t out[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const int sh = 8 * (p & 3);
                const int c0 = (int)((yy[p >> 2] >> sh) & 0xffu), c1 = (int)((cbb[p >> 2] >> sh) & 0xffu),
                          c2 = (int)((crr[p >> 2] >> sh) & 0xffu);
                float px3[3];
                if (mode.passthrough) {   // colour_mapping.cpp:28-37
                    px3[0] = (float)c0; px3[1] = (float)c1; px3[2] = (float)c2;
                } else {
                    bool tie;
                    colour_inverse_fast_unclamped((float)c0, (float)c1, (float)c2, px3[0], px3[1], px3[2], tie);   // insert_byte clamps
                    if (__builtin_expect(__any(tie), 0)) {   // the literal formula
                        uint8_t rr, gg, bb;
                        colour_inverse(c0, c1, c2, rr, gg, bb);
                        px3[0] = (float)rr; px3[1] = (float)gg; px3[2] = (float)bb;
                    }
                }
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int byte = 3 * p + c;
                    out[byte >> 2] = insert_byte(px3[c], byte & 3, out[byte >> 2]);
                }
            }
            uint8_t* dst = img + ((size_t)y * width + (size_t)bx * 8) * 3;
            store_pixels(dst, out[0], out[1]);
            store_pixels(dst + 8, out[2], out[3]);
            store_pixels(dst + 16, out[4], out[5]);
        }
        return;
    }
    // General path: work item = (row r, mcu m, pixel p), ordered so that consecutive threads
    // write consecutive pixels of a row strip; pixels outside the picture are dropped
    // (block_grid.cpp:87-99).
    for (uint32_t i = tid; i < 8u * mcus_here * 8u; i += kFdctThreads) {
        const uint32_t r = i / (mcus_here * 8u), rem = i - r * (mcus_here * 8u);
        const uint32_t mm = rem >> 3, p = rem & 7u;
        const uint32_t mcu = g0 + mm;
        const uint32_t by = mcu / (uint32_t)blocks_w, bx = mcu - by * (uint32_t)blocks_w;
        const int y = (int)by * 8 + (int)r, x = (int)bx * 8 + (int)p;
        if (y < height && x < width) {
            const int Y = lds_px[(0 * kMcusPerGroup + mm) * 64 + r * 8 + p];
            const int Cb = lds_px[(1 * kMcusPerGroup + mm) * 64 + r * 8 + p];
            const int Cr = lds_px[(2 * kMcusPerGroup + mm) * 64 + r * 8 + p];
            uint8_t rr, gg, bb;
            if (mode.passthrough) {
                rr = (uint8_t)Y; gg = (uint8_t)Cb; bb = (uint8_t)Cr;
            } else {
                float fr, fg, fb;
                bool tie;
                colour_inverse_fast((float)Y, (float)Cb, (float)Cr, fr, fg, fb, tie);
                rr = (uint8_t)fr; gg = (uint8_t)fg; bb = (uint8_t)fb;
                if (tie) colour_inverse(Y, Cb, Cr, rr, gg, bb);
            }
            uint8_t* dst = img + ((size_t)y * width + x) * 3;
            dst[0] = rr; dst[1] = gg; dst[2] = bb;
        }
    }
}

// ---------------------------------------------------------------------------
// The decoder's inverse transform for groups of MCUs that are mostly single colours (smooth pictures at
// ordinary qualities: all but two or three of a group's 64 MCUs have DC values alone).  In idct_dequant_kernel
// every lane carries a block, so two blocks with detail cost the wave the whole transform, and its 24 KiB of
// coefficients in LDS keep the compute units half empty.  Here a WAVE takes a group of 64 MCUs, lane = MCU:
//   1. the lane reads its MCU's three records; a block that is its DC value alone is one sample value;
//   2. MCUs whose three blocks are such values are one colour: converted once and stored as eight rows, from
//      the records alone (colour_mapping.cpp:69-92, block_grid.cpp:87-99);
//   3. the other MCUs are listed, and their blocks are transformed eight at a time, channel after channel, with
//      the lanes laid out by block ROW (lane = 8 b + s, as in fdct_quant_rows_kernel): chunks -> LDS -> de-zigzag
//      and dequantise a row of coefficients (entropy_encoder.cpp:79-92, quantiser.cpp:72-85) -> row pass ->
//      transposition through LDS -> column pass, clamp, truncation (discrete_cosine_transform.cpp:126-156) ->
//      samples to LDS; then colour conversion and store of those MCUs' rows.
// What binds this kernel is the time a group spends waiting for its records and chunks while the whole chip is
// storing (the stores alone would run at 5.5 TB/s, tools/store_patterns.hip): with a wave per group, and nothing
// in a group's way that waits for another wave, a compute unit has three times the groups under way that a
// workgroup of three waves per group had.
// A group with more than kSparseMaxVaried such MCUs is put on a list and left to idct_dequant_kernel, which is
// launched behind this kernel with one workgroup per place on the list (a few in the pictures this kernel is
// for; the launcher sends batches of detailed pictures to idct_dequant_kernel alone).  When the list is full the
// group is done here after all, in as many passes as it takes.  Needs width % 8 == 0.
#ifndef JEA_SPARSE_KEPT_PASSES   // experiments: tools/build_variant.sh <name> -DJEA_SPARSE_KEPT_PASSES=<n>
#define JEA_SPARSE_KEPT_PASSES 2
#endif
constexpr int kSparseKeptPasses = JEA_SPARSE_KEPT_PASSES;   // passes (of eight listed MCUs) whose pixels have a place in LDS
constexpr int kSparseMaxVaried = 8 * kSparseKeptPasses;
constexpr int kSparseTStride = 72;    // floats per block in the transposition buffer (stores of one x go to 32 different banks)
constexpr int kSparseWaves = 4;       // groups per workgroup
static_assert(kRowZWords <= 8 * kSparseTStride, "the chunks share the transposition buffer's space");

// Row-lane inverse transform of eight blocks per wave on the triangle u + v <= D (D = 7: everything).
// z_block: the block's chunks (zig-zag order) in LDS; qrow: row v of the dequantisation table;
// out: the eight samples of column x = s, rows y = 0..7, as bytes, in two words.
// The transposition buffer may lie over the chunks: they are read before it is written.
template <int D>
__device__ __forceinline__ void idct_rowlane(const uint8_t* z_block, const uint32_t (&z_off)[8], const float* qrow,
                                             float* t_write, const float* t_read, uint32_t (&out)[2]) {
    constexpr int kTerms = D < 7 ? D + 1 : 8;
    float deq[8];
    {
        const float4 qa = *reinterpret_cast<const float4*>(qrow), qb = *reinterpret_cast<const float4*>(qrow + 4);
        const float qv[8] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w};
#pragma unroll
        for (int u = 0; u < kTerms; ++u)
            deq[u] = (float)(int)*reinterpret_cast<const int16_t*>(z_block + z_off[u]) * qv[u];
    }
    wave_lds_fence();
    // row pass :126-135: tmp[x][v] = sum_u (s(u)C[u][x]) * src[u + 8v]; rows beyond the triangle hold zeros
#pragma unroll
    for (int x = 0; x < 8; ++x) {
        float acc = inv_c(0, x) * deq[0];
#pragma unroll
        for (int u = 1; u < kTerms; ++u) acc += inv_c(u, x) * deq[u];
        t_write[8 * x] = acc;
    }
    wave_lds_fence();
    float in[8];
    {
        const float4 a = *reinterpret_cast<const float4*>(t_read), b = *reinterpret_cast<const float4*>(t_read + 4);
        in[0] = a.x; in[1] = a.y; in[2] = a.z; in[3] = a.w;
        in[4] = b.x; in[5] = b.y; in[6] = b.z; in[7] = b.w;
    }
    wave_lds_fence();
    // column pass :136-156, then + 128 (:21-27); the byte conversion saturates as the clamp of :144-154 does
    out[0] = out[1] = 0;
#pragma unroll
    for (int y = 0; y < 8; ++y) {
        float acc = inv_c(0, y) * in[0];
#pragma unroll
        for (int v = 1; v < kTerms; ++v) acc += inv_c(v, y) * in[v];
        out[y >> 2] = insert_byte(__builtin_truncf(acc) + 128.0f, y & 3, out[y >> 2]);
    }
}

__global__ __launch_bounds__(64 * kSparseWaves, JEA_SPARSE_MIN_WAVES) void idct_sparse_kernel(
    const int16_t* __restrict__ coefs, const uint32_t* __restrict__ records, int width, int height, int blocks_w,
    uint32_t n_mcu, size_t image_bytes, QuantTables q, uint8_t* __restrict__ rgb, ComponentMode mode,
    DecodeReportRelay relay, uint32_t* __restrict__ group_list, uint32_t groups_per_image) {
    // the decoder's outcome, known since the kernels before this one ended, goes to the host on the side
    if (relay.report != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        relay.report->unsettled = *relay.unsettled;
        relay.report->status = *relay.status;
    }
    __shared__ __align__(16) float lds_qt[128];      // dequantisation tables, natural order: luminance | chrominance
    __shared__ __align__(16) float lds_tz[kSparseWaves][8 * kSparseTStride];   // a wave's chunks, then its transposition buffer
    // samples of the eight MCUs of a pass, [channel][b][y][x], which their pixels then replace, [b][y][24 bytes]; kSparseKeptPasses passes' worth
    __shared__ __align__(16) uint8_t lds_px[kSparseWaves][kSparseKeptPasses][3 * 8 * 64];
    __shared__ uint8_t lds_varied[kSparseWaves][kMcusPerGroup];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t lane = (uint32_t)tid & 63u;
    const uint32_t group = blockIdx.x * (uint32_t)kSparseWaves + (uint32_t)wave;
    const bool has_group = group < groups_per_image;   // per wave
    const uint32_t g0 = group * kMcusPerGroup;
    const uint32_t mcus_here = !has_group ? 0u : ((n_mcu - g0 < (uint32_t)kMcusPerGroup) ? (n_mcu - g0) : (uint32_t)kMcusPerGroup);
    const bool valid = lane < mcus_here;
    const uint32_t mcu_own = g0 + lane;
    const uint32_t image = blockIdx.y;
    uint8_t* const img = rgb + (size_t)image * image_bytes;

    uint32_t rec[3] = {0u, 0u, 0u};
    if (valid) {
        const uint32_t* r = records + ((size_t)image * n_mcu + mcu_own) * 3;
        rec[0] = r[0]; rec[1] = r[1]; rec[2] = r[2];
    }
    if (tid < 128) lds_qt[tid] = reinterpret_cast<const float*>(&q)[tid];
    __syncthreads();   // the only workgroup barrier
    if (!has_group) return;

    const bool lane_flat = valid && (((rec[0] | rec[1] | rec[2]) >> 16) & 0xffu) == 0;
    const uint64_t flat = __ballot(lane_flat), others = __ballot(valid && !lane_flat);
    const uint32_t n_varied = (uint32_t)__popcll(others);
    if (n_varied > (uint32_t)kSparseMaxVaried) {   // for idct_dequant_kernel, if the list has room
        uint32_t place = 0;
        if (lane == 0) {
            place = atomicAdd(group_list, 1u);
            if (place < kSparseListCapacity) group_list[1 + place] = image * groups_per_image + group;
        }
        place = (uint32_t)__builtin_amdgcn_readfirstlane((int)place);
        if (place < kSparseListCapacity) return;
    }
    // Chunk s of channel c of MCU m of this picture is chunk_image[(3 s + c) n_mcu + m].  What the transform of a
    // listed MCU will want first is asked for now: the first chunk of each of its blocks that has one stored (a
    // block without carries its DC value in the record).
    const uint4* const chunk_image = reinterpret_cast<const uint4*>(coefs) + (size_t)image * 24 * n_mcu;
    uint4 fc[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        fc[c] = make_uint4(rec[c] & 0xffffu, 0u, 0u, 0u);
        if ((rec[c] >> 16) & 1u) fc[c] = chunk_image[(size_t)c * n_mcu + mcu_own];
    }

    // One colour.  DC alone: every sample is s(0)C[0][y] * (s(0)C[0][x] * deq[0]) with C[0][.] = 1, clamped and
    // truncated (:144-154); the lane converts its MCU's colour (used if the MCU is one) and notes where the MCU lies.
    uint32_t own_colour = 0, own_place;
    {
        float dcs[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float q0 = mode.is_lum(c) ? q.lum[0] : q.chr[0];
            const float dc_only = inv_c(0, 0) * (inv_c(0, 0) * ((float)(int)(int16_t)(rec[c] & 0xffffu) * q0));
            dcs[c] = (float)((int)__builtin_amdgcn_fmed3f(dc_only, -128.0f, 127.0f) + 128);
        }
        float px3[3];
        if (mode.passthrough) {
            px3[0] = dcs[0]; px3[1] = dcs[1]; px3[2] = dcs[2];
        } else {
            bool tie;
            colour_inverse_fast_unclamped(dcs[0], dcs[1], dcs[2], px3[0], px3[1], px3[2], tie);
            if (tie) {   // the literal formula
                uint8_t rr, gg, bb;
                colour_inverse((int)dcs[0], (int)dcs[1], (int)dcs[2], rr, gg, bb);
                px3[0] = (float)rr; px3[1] = (float)gg; px3[2] = (float)bb;
            }
        }
        own_colour = insert_byte(px3[0], 0, insert_byte(px3[1], 1, insert_byte(px3[2], 2, 0u)));
        const uint32_t mcu_in = mcu_own < n_mcu ? mcu_own : n_mcu - 1u;
        const uint32_t by = mcu_in / (uint32_t)blocks_w, bx = mcu_in - by * (uint32_t)blocks_w;
        own_place = by << 16 | bx;   // both below 8192
    }

    // place of a listed MCU on the list (for the others: how many listed MCUs come before them)
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(others >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)others, 0u));
    if (n_varied != 0) {
        const bool listed = (others >> lane) & 1u;
        if (listed) lds_varied[wave][rank] = (uint8_t)lane;
        // the listed MCUs, eight per pass, channel after channel: lane = 8 b + s
        const uint32_t b = lane >> 3, s = lane & 7u;
        float* const tz = lds_tz[wave];
        uint8_t* const z_block = reinterpret_cast<uint8_t*>(tz + 36u * ((b & 1u) ^ ((b >> 1) & 1u)) + 80u * ((b >> 1) & 1u) + 148u * (b >> 2));
        float* const t_write = tz + b * kSparseTStride + s;        // + 8 x: value (x, row v = s)
        const float* const t_read = tz + b * kSparseTStride + 8 * s;   // + v: value (x = s, row v)
        uint32_t z_off[8];   // byte offset of coefficient (v = s, u) in the block's zig-zag order
        {
            const uint32_t pick_byte_s = 0x0c0c0c00u | s;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint64_t col = kZigzagColDoubled[u];
                z_off[u] = __builtin_amdgcn_perm((uint32_t)(col >> 32), (uint32_t)col, pick_byte_s);
            }
        }
        wave_lds_fence();
        // The pixels of the first kSparseMaxVaried listed MCUs (all of them, unless the group was too detailed for this kernel and
        // the list was full) stay in LDS and leave with the single colours' rows; passes beyond those store their
        // own, and run first, while the LDS space is free.
        for (int pass = (int)((n_varied - 1u) >> 3); pass >= 0; --pass) {
            const uint32_t first = 8u * (uint32_t)pass;
            uint8_t* const px = lds_px[wave][pass < kSparseKeptPasses ? pass : 0];
            const uint32_t n_here = n_varied - first < 8u ? n_varied - first : 8u;
            const bool live = b < n_here;
            const uint32_t mm = lds_varied[wave][live ? first + b : first];
            // a listed lane whose MCU is in this pass puts its first chunks where block (rank - first) has its chunk 0
            const uint32_t slot = rank - first;
            const bool in_pass = listed && slot < 8u;
            uint8_t* const z_mine = reinterpret_cast<uint8_t*>(tz + 36u * ((slot & 1u) ^ ((slot >> 1) & 1u)) + 80u * ((slot >> 1) & 1u) + 148u * ((slot >> 2) & 1u));
#pragma unroll 1
            for (int c = 0; c < 3; ++c) {
                const uint32_t rec_c = c == 0 ? rec[0] : (c == 1 ? rec[1] : rec[2]);
                uint4 fc_c;
                fc_c.x = c == 0 ? fc[0].x : (c == 1 ? fc[1].x : fc[2].x);
                fc_c.y = c == 0 ? fc[0].y : (c == 1 ? fc[1].y : fc[2].y);
                fc_c.z = c == 0 ? fc[0].z : (c == 1 ? fc[1].z : fc[2].z);
                fc_c.w = c == 0 ? fc[0].w : (c == 1 ? fc[1].w : fc[2].w);
                const uint32_t rec_mm = (uint32_t)__shfl((int)rec_c, (int)mm);
                const uint32_t mask = live ? (rec_mm >> 16) & 0xffu : 0u;
                const bool have = (mask >> s) & 1u;
                if (in_pass) *reinterpret_cast<uint4*>(z_mine) = fc_c;
                if (s == 0) {
                    if (!live) *reinterpret_cast<uint4*>(z_block) = make_uint4(0, 0, 0, 0);
                } else {
                    uint4 chunk = make_uint4(0, 0, 0, 0);
                    if (have) chunk = chunk_image[((size_t)s * 3 + (size_t)c) * n_mcu + g0 + mm];
                    *reinterpret_cast<uint4*>(z_block + 16 * s) = chunk;
                }
                // how far along the zig-zag do the coefficients of the wave's eight blocks reach?  (chunk 0: positions 0-7
                // = diagonals u + v <= 3, words 0-1 | 2-3 | 4-5 | 6-7 on diagonals 0,1 | 1,2 | 2,2 | 3,3; chunk 1: <= 5)
                int diag;
                if (__ballot(have && s >= 2) != 0) diag = 7;
                else if (__ballot(have && s == 1) != 0) diag = 5;
                else if (__ballot(in_pass && fc_c.w != 0) != 0) diag = 3;
                else if (__ballot(in_pass && (fc_c.y | fc_c.z) != 0) != 0) diag = 2;
                else diag = 1;
                wave_lds_fence();
                const float* const qrow = lds_qt + (mode.is_lum(c) ? 0 : 64) + 8 * s;
                uint32_t col8[2];
                switch (diag) {
                    case 1: idct_rowlane<1>(z_block, z_off, qrow, t_write, t_read, col8); break;
                    case 2: idct_rowlane<2>(z_block, z_off, qrow, t_write, t_read, col8); break;
                    case 3: idct_rowlane<3>(z_block, z_off, qrow, t_write, t_read, col8); break;
                    case 5: idct_rowlane<5>(z_block, z_off, qrow, t_write, t_read, col8); break;
                    default: idct_rowlane<7>(z_block, z_off, qrow, t_write, t_read, col8); break;
                }
                // column x = s of block b: bytes y = 0..7 go to [channel][b][y][x]
                uint8_t* to = px + ((uint32_t)c * 8u + b) * 64u + s;
#pragma unroll
                for (int y = 0; y < 8; ++y) to[8 * y] = (uint8_t)(col8[y >> 2] >> (8 * (y & 3)));
            }
            wave_lds_fence();
            // colour conversion: lane = (row r, MCU bi of the pass) = 8 pixels = 24 bytes
            {
                const uint32_t r = lane >> 3, bi = lane & 7u;
                const bool mine = bi < n_here;
                uint32_t out[6] = {0, 0, 0, 0, 0, 0};
                if (mine) {
                    const uint2 y8 = *reinterpret_cast<const uint2*>(px + (0 * 8 + bi) * 64 + r * 8);
                    const uint2 b8 = *reinterpret_cast<const uint2*>(px + (1 * 8 + bi) * 64 + r * 8);
                    const uint2 r8 = *reinterpret_cast<const uint2*>(px + (2 * 8 + bi) * 64 + r * 8);
                    const uint32_t yy[2] = {y8.x, y8.y}, cbb[2] = {b8.x, b8.y}, crr[2] = {r8.x, r8.y};
#pragma unroll
                    for (int p = 0; p < 8; ++p) {
                        const int sh = 8 * (p & 3);
                        const int c0 = (int)((yy[p >> 2] >> sh) & 0xffu), c1 = (int)((cbb[p >> 2] >> sh) & 0xffu),
                                  c2 = (int)((crr[p >> 2] >> sh) & 0xffu);
                        float px3[3];
                        if (mode.passthrough) {   // colour_mapping.cpp:28-37
                            px3[0] = (float)c0; px3[1] = (float)c1; px3[2] = (float)c2;
                        } else {
                            bool tie;
                            colour_inverse_fast_unclamped((float)c0, (float)c1, (float)c2, px3[0], px3[1], px3[2], tie);   // insert_byte clamps
                            if (tie) {   // the literal formula
                                uint8_t rr, gg, bb;
                                colour_inverse(c0, c1, c2, rr, gg, bb);
                                px3[0] = (float)rr; px3[1] = (float)gg; px3[2] = (float)bb;
                            }
                        }
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            const int byte = 3 * p + c;
                            out[byte >> 2] = insert_byte(px3[c], byte & 3, out[byte >> 2]);
                        }
                    }
                }
                wave_lds_fence();   // every lane has its samples: the pixels may take their place
                if (mine) {
                    if (pass < kSparseKeptPasses) {
                        uint2* dst = reinterpret_cast<uint2*>(px + (bi * 8u + r) * 24u);
                        dst[0] = make_uint2(out[0], out[1]);
                        dst[1] = make_uint2(out[2], out[3]);
                        dst[2] = make_uint2(out[4], out[5]);
                    } else {
                        const uint32_t mcu2 = g0 + lds_varied[wave][first + bi];
                        const uint32_t by = mcu2 / (uint32_t)blocks_w, bx = mcu2 - by * (uint32_t)blocks_w;
                        const int y = (int)by * 8 + (int)r;
                        if (y < height) {
                            uint8_t* dst = img + ((size_t)y * width + (size_t)bx * 8) * 3;
                            store_pixels(dst, out[0], out[1]);
                            store_pixels(dst + 8, out[2], out[3]);
                            store_pixels(dst + 16, out[4], out[5]);
                        }
                    }
                }
            }
            wave_lds_fence();   // the next pass may use the same space
        }
    }

    // The group's rows, last: the counter a wave waits on for a load also counts its stores, so stores issued ahead
    // of the listed MCUs' loads would have to drain first.  A row of the group is 64 x 24 bytes; lane l writes the
    // 8-byte pieces l, l + 64, l + 128 of it (piece p: MCU p / 3, bytes 8 (p % 3) ...), so that a store instruction
    // of the wave covers 512 consecutive bytes, single colours and listed MCUs alike (rows with holes, filled
    // by other stores later, cost a tenth of the store rate: tools/store_patterns.hip).  Eight pixels of one colour
    // are the 24 bytes RGBR GBRG BRGB RGBR GBRG BRGB: piece 0 = RGBR GBRG, piece 1 = BRGB RGBR, piece 2 = GBRG BRGB.
#pragma unroll
    for (uint32_t k = 0; k < 3; ++k) {
        const uint32_t p = lane + 64u * k, fm = p / 3u, piece = p - 3u * fm;
        const uint32_t colour = (uint32_t)__shfl((int)own_colour, (int)fm), place = (uint32_t)__shfl((int)own_place, (int)fm);
        const uint32_t rank_fm = (uint32_t)__shfl((int)rank, (int)fm);
        const bool is_flat = (flat >> fm) & 1u, is_kept = ((others >> fm) & 1u) && rank_fm < (uint32_t)kSparseMaxVaried;
        if (is_flat || is_kept) {
            const uint32_t by = place >> 16, bx = place & 0xffffu;
            constexpr uint32_t kRgbr = 0x00020100u, kGbrg = 0x01000201u, kBrgb = 0x02010002u;
            const uint32_t w0 = __builtin_amdgcn_perm(colour, colour, piece == 0 ? kRgbr : (piece == 1 ? kBrgb : kGbrg));
            const uint32_t w1 = __builtin_amdgcn_perm(colour, colour, piece == 0 ? kGbrg : (piece == 1 ? kRgbr : kBrgb));
            uint2 row[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) row[r] = make_uint2(w0, w1);
            if (is_kept) {
                const uint8_t* from = lds_px[wave][rank_fm >> 3] + (rank_fm & 7u) * 192u + 8u * piece;
#pragma unroll
                for (int r = 0; r < 8; ++r) row[r] = *reinterpret_cast<const uint2*>(from + 24 * r);
            }
            uint8_t* to = img + ((size_t)by * 8 * width + (size_t)bx * 8) * 3 + 8 * piece;
            const int rows = height - (int)by * 8 < 8 ? height - (int)by * 8 : 8;
#pragma unroll
            for (int r = 0; r < 8; ++r)
                if (r < rows) {
                    store_pixels(to + (size_t)r * width * 3, row[r].x, row[r].y);   // 1024 frames: 1.29 instead of 1.36 ms
                }
        }
    }
}

__global__ void rgb_to_ycbcr_kernel(const uint8_t* __restrict__ rgb, size_t n, uint8_t* __restrict__ ycc) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t ri = rgb[3 * i], gi = rgb[3 * i + 1], bi = rgb[3 * i + 2];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            // the fused kernel's two paths: exact integer arithmetic, literal double formula on rounding ties
            bool tie;
            float v = colour_forward_shifted_fast(colour_int_constants(ch), (float)ri, (float)gi, (float)bi, tie) + 128.0f;
            if (tie) {
                double ck, ca, cb, cc;
                colour_constants(ch, ck, ca, cb, cc);
                v = (float)colour_forward(ck, ca, cb, cc, (double)ri, (double)gi, (double)bi);
            }
            ycc[3 * i + ch] = (uint8_t)v;
        }
    }
}

__global__ void ycbcr_to_rgb_kernel(const uint8_t* __restrict__ ycc, size_t n, uint8_t* __restrict__ rgb) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        // the inverse kernel's two paths: exact arithmetic, literal double formula near rounding ties
        const int Y = ycc[3 * i], Cb = ycc[3 * i + 1], Cr = ycc[3 * i + 2];
        float fr, fg, fb;
        bool tie;
        colour_inverse_fast((float)Y, (float)Cb, (float)Cr, fr, fg, fb, tie);
        uint8_t r = (uint8_t)fr, g = (uint8_t)fg, b = (uint8_t)fb;
        if (tie) colour_inverse(Y, Cb, Cr, r, g, b);
        rgb[3 * i] = r; rgb[3 * i + 1] = g; rgb[3 * i + 2] = b;
    }
}

}  // namespace

void launch_fdct_quant(hipStream_t s, const uint8_t* d_rgb, size_t n_images, int width, int height,
                       const QuantTables& q, int16_t* d_coefs, int32_t* d_status_to_clear, uint8_t* d_chunk_masks,
                       bool planar, ComponentMode mode) {
    const int bw = (width + 7) / 8, bh = (height + 7) / 8;
    const uint32_t n_mcu = (uint32_t)bw * (uint32_t)bh;
    const size_t image_bytes = (size_t)3 * width * height;
    const dim3 grid((n_mcu + kMcusPerGroup - 1) / kMcusPerGroup, (unsigned)n_images);
    const bool fast = (width % 8 == 0) && ((reinterpret_cast<uintptr_t>(d_rgb) & 7u) == 0);
    planar = planar && d_chunk_masks != nullptr;
    if (fast && bw >= 8 && height % 8 == 0 && (uint64_t)n_mcu * 384 < (1ull << 32)) {
        // Units (8 MCUs) per wave.  Short waves suit a launch best -- workgroups then start and end at different
        // times, so loads, arithmetic and stores of neighbours overlap and the launch has no long tail -- but a
        // wave's set-up is worth a tenth of a unit's work.  Measured on 3840x2160 pictures (one picture / eight
        // per launch, microseconds): 1 unit 26.3 / 178, 2 units 26.9 / 164, 4 units 27.8 / 161, 8 units 33.5 / 164.
        // So: one unit per wave until the launch has about three rounds of resident waves (1024 SIMDs x 5), then
        // longer waves, up to 6 units.
        const uint64_t units = (n_mcu + 7) / 8;
        uint64_t k64 = (units * n_images + 8192) / 16384;
#ifdef JEA_K1_UNITS   // experiments: tools/build_variant.sh <name> -DJEA_K1_UNITS=<n>
        const uint32_t best_k = JEA_K1_UNITS;
        (void)k64;
#else
        const uint32_t best_k = (uint32_t)(k64 < 1 ? 1 : (k64 > 6 ? 6 : k64));
#endif
        const dim3 grid_rows((unsigned)((units + (uint64_t)kRowWaves * best_k - 1) / ((uint64_t)kRowWaves * best_k)),
                             (unsigned)n_images);
#ifdef JEA_K1_STRIDED_WAVES   // experiments: a launch of that many waves, each striding over the units
        if (planar && n_images == 1) {
            fdct_quant_rows_kernel<true, true><<<dim3(JEA_K1_STRIDED_WAVES / kRowWaves, 1), kRowThreads, kRowDynLds, s>>>(
                d_rgb, width, height, bw, n_mcu, image_bytes, q, d_coefs, d_status_to_clear, d_chunk_masks, mode, 1);
            return;
        }
#endif
        if (planar)
            fdct_quant_rows_kernel<true><<<grid_rows, kRowThreads, kRowDynLds, s>>>(d_rgb, width, height, bw, n_mcu, image_bytes, q, d_coefs,
                                                                           d_status_to_clear, d_chunk_masks, mode, best_k);
        else
            fdct_quant_rows_kernel<false><<<grid_rows, kRowThreads, kRowDynLds, s>>>(d_rgb, width, height, bw, n_mcu, image_bytes, q, d_coefs,
                                                                            d_status_to_clear, d_chunk_masks, mode, best_k);
        return;
    }
#define JEA_LAUNCH_FDCT(F, P)                                                                                      \
    fdct_quant_kernel<F, P><<<grid, kFdctThreads, 0, s>>>(d_rgb, width, height, bw, n_mcu, image_bytes, q, d_coefs, \
                                                          d_status_to_clear, d_chunk_masks, mode)
    if (fast && planar) JEA_LAUNCH_FDCT(true, true);
    else if (fast) JEA_LAUNCH_FDCT(true, false);
    else if (planar) JEA_LAUNCH_FDCT(false, true);
    else JEA_LAUNCH_FDCT(false, false);
#undef JEA_LAUNCH_FDCT
}

void launch_idct_dequant(hipStream_t s, const int16_t* d_coefs, size_t n_images, int width, int height,
                         const QuantTables& q, uint8_t* d_rgb, const uint32_t* d_records, ComponentMode mode,
                         const DecodeReportRelay& relay, uint32_t* d_group_list, bool mostly_single_colours) {
    const int bw = (width + 7) / 8, bh = (height + 7) / 8;
    const uint32_t n_mcu = (uint32_t)bw * (uint32_t)bh;
    const size_t image_bytes = (size_t)3 * width * height;
    const dim3 grid((n_mcu + kMcusPerGroup - 1) / kMcusPerGroup, (unsigned)n_images);
    const uint64_t total_groups = (uint64_t)grid.x * n_images;
    // The two-kernel route is for batches of smooth pictures: it pays off once the launch fills the GPU several
    // times over (a single picture is quicker through the general kernel alone: one launch less in a chain
    // that the host waits for), and only where most MCUs are single colours.
#ifdef JEA_NO_SPARSE   // experiments: the general kernel alone
    mostly_single_colours = false;
#endif
    if (d_records != nullptr && d_group_list != nullptr && mostly_single_colours && width % 8 == 0 &&
#ifdef JEA_SPARSE_FROM_GROUPS   // experiments: tools/build_variant.sh <name> -DJEA_SPARSE_FROM_GROUPS=<n>
        total_groups >= JEA_SPARSE_FROM_GROUPS && total_groups < (1ull << 32)) {
#else
        total_groups >= 4096 && total_groups < (1ull << 32)) {
#endif
        (void)hipMemsetAsync(d_group_list, 0, sizeof(uint32_t), s);
        const dim3 grid_sparse((grid.x + kSparseWaves - 1) / kSparseWaves, (unsigned)n_images);
        idct_sparse_kernel<<<grid_sparse, 64 * kSparseWaves, 0, s>>>(d_coefs, d_records, width, height, bw, n_mcu, image_bytes, q,
                                                                     d_rgb, mode, relay, d_group_list, grid.x);
        const dim3 grid_listed((unsigned)(total_groups < kSparseListCapacity ? total_groups : kSparseListCapacity), 1);
        idct_dequant_kernel<true><<<grid_listed, kFdctThreads, 0, s>>>(d_coefs, d_records, width, height, bw, n_mcu,
                                                                       image_bytes, q, d_rgb, mode, DecodeReportRelay{},
                                                                       d_group_list, grid.x);
    } else if (d_records != nullptr) {
        idct_dequant_kernel<true><<<grid, kFdctThreads, 0, s>>>(d_coefs, d_records, width, height, bw, n_mcu,
                                                                image_bytes, q, d_rgb, mode, relay, nullptr, grid.x);
    } else {
        idct_dequant_kernel<false><<<grid, kFdctThreads, 0, s>>>(d_coefs, nullptr, width, height, bw, n_mcu,
                                                                 image_bytes, q, d_rgb, mode, relay, nullptr, grid.x);
    }
}

size_t idct_group_list_bytes() { return sizeof(uint32_t) * (1 + (size_t)kSparseListCapacity); }

void launch_rgb_to_ycbcr(hipStream_t s, const uint8_t* d_rgb, size_t n_pixels, uint8_t* d_ycc) {
    const unsigned grid = (unsigned)((n_pixels + 255) / 256 < 4096 ? (n_pixels + 255) / 256 : 4096);
    rgb_to_ycbcr_kernel<<<grid ? grid : 1, 256, 0, s>>>(d_rgb, n_pixels, d_ycc);
}
void launch_ycbcr_to_rgb(hipStream_t s, const uint8_t* d_ycc, size_t n_pixels, uint8_t* d_rgb) {
    const unsigned grid = (unsigned)((n_pixels + 255) / 256 < 4096 ? (n_pixels + 255) / 256 : 4096);
    ycbcr_to_rgb_kernel<<<grid ? grid : 1, 256, 0, s>>>(d_ycc, n_pixels, d_rgb);
}

}  // namespace jea
