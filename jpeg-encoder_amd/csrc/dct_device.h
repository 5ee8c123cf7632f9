// Device functions of the colour / DCT / quantiser arithmetic, shared by the fused
// image kernels (fdct.hip) and the per-block kernels (blocks.hip) so that both
// run the very same instruction sequences.  Arithmetic contract: see fdct.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "tables.h"

namespace jea {

__device__ __forceinline__ float min3(float a, float b, float c) {
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// colour_mapping.cpp:48-51: floor(in + 0.5), clamp to [0, 255]
__device__ __forceinline__ double round_to_range(double v) {
    v = floor(v + 0.5);
    v = (0.0 < v) ? v : 0.0;
    v = (255.0 < v) ? 255.0 : v;
    return v;
}

// One output channel of RGBToYCbCrMapper::applyMapping, colour_mapping.cpp:52-60.
// Each formula there has the shape ((k + a*r) + b*g) + c*b evaluated left to
// right in double; subtracting a product equals adding the product of the
// negated constant, so one expression serves the three channels:
//   Y  : k = 0,   a =  0.299,    b =  0.587,    c =  0.114
//   Cb : k = 128, a = -0.168736, b = -0.331264, c =  0.5
//   Cr : k = 128, a =  0.5,      b = -0.418688, c = -0.081312
__device__ __forceinline__ double colour_forward(double k, double a, double b, double c, double r, double g,
                                                 double bl) {
    return round_to_range(((k + a * r) + b * g) + c * bl);
}

__device__ __forceinline__ void colour_constants(int channel, double& k, double& a, double& b, double& c) {
    k = channel == 0 ? 0.0 : 128.0;
    a = channel == 0 ? 0.299 : (channel == 1 ? -0.168736 : 0.5);
    b = channel == 0 ? 0.587 : (channel == 1 ? -0.331264 : -0.418688);
    c = channel == 0 ? 0.114 : (channel == 1 ? 0.5 : -0.081312);
}

// The same formulas without double arithmetic.  All constants are multiples of 1/M with M = 1000 for Y
// and M = 31250 for Cb, Cr, so the exact value is t / M with t an integer, and the reference's result is
// floor(t / M + 0.5) wherever its double arithmetic, whose error is some 1e-13, cannot land on the other
// side of an integer: everywhere but at the ties, where t / M + 0.5 is itself an integer and only the
// literal double formula can tell which way the reference rounds.
//
// Here q = value + 0.5 - 128 + d is formed by three f32 FMAs with constants rounded to f32; the
// magnitudes stay below 256, so every rounding is at most 2^-17 and the constants contribute at most
// 255 * 2^-24 in all: |q - true| < e = 4e-5.  With d = 1e-4 and the test fract(q) < 2e-4:
//   every tie (true fraction exactly d) has fract(q) in [d - e, d + e] and is flagged;
//   an unflagged q has a true fraction of at least 2e-4 - e > d, so the exact value + 0.5 - 128 lies in the
//   same unit interval as q, and floor(q) is the reference's result (a q within e below an integer may
//   belong to the next interval only if the true fraction is below e, which is flagged territory).
// Flagged pixels (2e-4 of them, plus the ties: 1e-3 of the Y values) are redone with colour_forward.
// An unflagged value + 0.5 lies strictly inside (0, 256), so no clamp is needed.  All 2^24 inputs of
// every channel are compared with the oracle by the tests.
struct ColourInt {
    float k, a, b, c, window;
};
__device__ __forceinline__ ColourInt colour_int_constants(int channel) {
    constexpr float d = 1e-4f;
    ColourInt ci;
    if (channel == 0) ci = ColourInt{0.5f - 128.0f + d, 0.299f, 0.587f, 0.114f, 2e-4f};
    else if (channel == 1) ci = ColourInt{0.5f + d, -0.168736f, -0.331264f, 0.5f, 2e-4f};
    else ci = ColourInt{0.5f + d, 0.5f, -0.418688f, -0.081312f, 2e-4f};
    return ci;
}
// Level-shifted sample (value rounded, minus 128) as a float; `tie` is set when it must be redone.
__device__ __forceinline__ float colour_forward_shifted_fast(const ColourInt& ci, float r, float g, float b, bool& tie) {
    const float q = __builtin_fmaf(ci.c, b, __builtin_fmaf(ci.b, g, __builtin_fmaf(ci.a, r, ci.k)));
    const float fr = __builtin_amdgcn_fractf(q);      // q - floor(q), exact
    tie = fr < ci.window;
    return q - fr;
}

// low halves of two ints side by side: one byte-permute instruction
__device__ __forceinline__ uint32_t pack16(int lo, int hi) {
    return __builtin_amdgcn_perm((uint32_t)hi, (uint32_t)lo, 0x05040100u);
}

// Forward-transform constants with the scale folded in where that is exact:
// s(u) = 0.5 for u > 0 is a power of two, so  0.5 * sum_x C[u][x] f[x]  equals
// sum_x (0.5 C[u][x]) f[x]  bit for bit (scaling by two commutes with every
// rounding; nothing here comes near the subnormal range).  Row u = 0 has
// C = 1 exactly, so its sum needs no products and keeps the final multiply by
// s(0) = 0.5/sqrt(2), which is not a power of two.
constexpr float fwd_c(int u, int x) { return u == 0 ? 1.0f : 0.5f * cos_c(u, x); }

// Row pass of discrete_cosine_transform.cpp:103-112 for image row y of the block:
//   row[u][y] = s(u) * (((C[u][0] f[0] + C[u][1] f[1]) + ...) + C[u][7] f[7])
__device__ __forceinline__ void fdct_row(const float (&f)[8], int y, float (&row)[8][8]) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        float acc = fwd_c(u, 0) * f[0];
#pragma unroll
        for (int x = 1; x < 8; ++x) acc += fwd_c(u, x) * f[x];
        row[u][y] = (u == 0) ? scale_s(0) * acc : acc;
    }
}

// Column pass :114-124 for horizontal frequency u: o[v] = out[8v+u].
__device__ __forceinline__ void fdct_col(const float (&row)[8][8], int u, float (&o)[8]) {
#pragma unroll
    for (int v = 0; v < 8; ++v) {
        float acc = fwd_c(v, 0) * row[u][0];
#pragma unroll
        for (int y = 1; y < 8; ++y) acc += fwd_c(v, y) * row[u][y];
        o[v] = (v == 0) ? scale_s(0) * acc : acc;
    }
}

// quantiser.cpp:57-70, literally: IEEE float divide, then 0.5 + and floor in double.
__device__ __forceinline__ int quantise_exact(float o, float q) {
    const float quot = o / q;
    return (int)floor(0.5 + (double)quot);
}

// The same value without a division.  With rq = RN(1/q):
//   q0 = RN(o * rq)            within an ulp of o / q
//   r  = RN(o - q0 * q)        one FMA; exact, because q0 is that close
//   q1 = RN(q0 + r * rq)       one FMA; the correctly rounded quotient RN(o / q)
// (Markstein's correction step; it needs rq to be the correctly rounded reciprocal, which it is for
// every integer q = 1..65535.  Checked against the division on the host over all those q with
// millions of dividends including the neighbours of every q * (n + 1/2); the two differ only where the
// quotient is subnormal, and those round to 0 either way.)  Then floor(0.5 + double(q1)): the double
// sum is exact, and v_cvt_rpi_i32_f32 returns floor(x + 0.5) of the exact sum for every float x
// (compared with the double formula over all 2^32 bit patterns, tools/probe_insts.hip).
// Every operand is a vector register: on gfx950 an f32 multiply or FMA with a scalar-register
// operand issues at half the rate, so the callers bring rq and q in through LDS or registers.
__device__ __forceinline__ int quantise_rn(float o, float rq, float q) {
    const float q0 = o * rq;
    const float r = __builtin_fmaf(-q0, q, o);
    const float q1 = __builtin_fmaf(r, rq, q0);
    int n;
    asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(n) : "v"(q1));
    return n;
}

// Quantises the eight coefficients out[8v+u], v = 0..7, of one column u; c[v] = {RN(1/Q), Q} of entry 8v+u.
__device__ __forceinline__ void quantise_col(const float (&o)[8], const float2 (&c)[8], int u, int* __restrict__ qv) {
#pragma unroll
    for (int v = 0; v < 8; ++v) qv[8 * v + u] = quantise_rn(o[v], c[v].x, c[v].y);
}

// colour_mapping.cpp:77-85; (C - 128) is formed in integer arithmetic first.
__device__ __forceinline__ void colour_inverse(int Y, int Cb, int Cr, uint8_t& r, uint8_t& g, uint8_t& b) {
    const double y = (double)Y, cb = (double)(Cb - 128), cr = (double)(Cr - 128);
    r = (uint8_t)round_to_range(y + 1.402 * cr);
    g = (uint8_t)round_to_range((y - 0.344136 * cb) - 0.714136 * cr);
    b = (uint8_t)round_to_range(y + 1.772 * cb);
}

// The inverse formulas by the method of colour_forward_shifted_fast.  With cb = Cb - 128, cr = Cr - 128
// the exact values are multiples of 1/1000 (R, B) and 1/125000 (G); q = value + 0.5 + d comes from one or
// two f32 FMAs at magnitudes below 512 (roundings of at most 2^-16 each, constants 1.8 * 128 * 2^-24):
// |q - true| < e = 5e-5 < d = 1e-4, and the test is fract(q) < 2e-4 as there.  Values still need the
// clamp to [0, 255] (:49-51); they are integers, so the conversion to bytes is exact.
// This variant leaves the clamp to the caller: r, g, b are whole numbers between about -180 and 435.
__device__ __forceinline__ void colour_inverse_fast_unclamped(float Y, float Cb, float Cr, float& r, float& g, float& b,
                                                              bool& tie) {
    const float cb = Cb - 128.0f, cr = Cr - 128.0f;
    const float y = Y + (0.5f + 1e-4f);
    const float qr = __builtin_fmaf(1.402f, cr, y);
    const float qb = __builtin_fmaf(1.772f, cb, y);
    const float qg = __builtin_fmaf(-0.714136f, cr, __builtin_fmaf(-0.344136f, cb, y));
    const float fr = __builtin_amdgcn_fractf(qr), fb = __builtin_amdgcn_fractf(qb), fg = __builtin_amdgcn_fractf(qg);
    tie = min3(fr, fb, fg) < 2e-4f;
    r = qr - fr;
    b = qb - fb;
    g = qg - fg;
}
__device__ __forceinline__ void colour_inverse_fast(float Y, float Cb, float Cr, float& r, float& g, float& b, bool& tie) {
    colour_inverse_fast_unclamped(Y, Cb, Cr, r, g, b, tie);
    r = __builtin_amdgcn_fmed3f(r, 0.0f, 255.0f);
    b = __builtin_amdgcn_fmed3f(b, 0.0f, 255.0f);
    g = __builtin_amdgcn_fmed3f(g, 0.0f, 255.0f);
}

// Byte `index` (0..3) of `word` replaced by the value, a whole number held as a float.  v_cvt_pk_u8_f32
// saturates: whole numbers below 0 give 0 and above 255 give 255 (checked for -2^20..2^20 on the GPU,
// tools/probe_insts.hip), so a clamp in front of it is redundant.
__device__ __forceinline__ uint32_t insert_byte(float value, uint32_t index, uint32_t word) {
    return __builtin_amdgcn_cvt_pk_u8_f32(value, index, word);
}

constexpr float inv_c(int u, int x) { return scale_s(u) * cos_c(u, x); }  // fl32(s(u) * C[u][x]), :131,:142


// Inverse transform of one dequantised block (natural order): row pass
// discrete_cosine_transform.cpp:126-135, column pass with clamp and truncation
// :136-156, level unshift :21-27.  px[8y+x] receives the u8 sample.
__device__ __forceinline__ void idct_block(const float (&deq)[64], uint8_t (&px)[64]) {
    float tmp[8][8];
#pragma unroll
    for (int v = 0; v < 8; ++v) {
#pragma unroll
        for (int x = 0; x < 8; ++x) {
            float acc = inv_c(0, x) * deq[8 * v];
#pragma unroll
            for (int u = 1; u < 8; ++u) acc += inv_c(u, x) * deq[u + 8 * v];
            tmp[x][v] = acc;
        }
    }
#pragma unroll
    for (int y = 0; y < 8; ++y) {
#pragma unroll
        for (int x = 0; x < 8; ++x) {
            float acc = inv_c(0, y) * tmp[x][0];
#pragma unroll
            for (int v = 1; v < 8; ++v) acc += inv_c(v, y) * tmp[x][v];
            int o;
            if (acc < -128.0f) o = -128;
            else if (acc > 127.0f) o = 127;
            else o = (int)acc;
            px[8 * y + x] = (uint8_t)(o + 128);
        }
    }
}

}  // namespace jea
