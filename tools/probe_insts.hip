// Probes of single gfx950 instructions whose exact behaviour the kernels would like to rely on.
//   1. v_cvt_rpi_i32_f32 against floor(double(x) + 0.5) over every float of magnitude below 2^30
//   2. v_fma_mix_f32 reading an f16 subnormal operand (a byte value n held as the 16-bit integer n, i.e. n * 2^-24)
//      against the f32 fma with float(n), for every n and a set of coefficients
//   3. v_cvt_pk_u8_f32 on whole numbers from -2^20 to 2^20: does it saturate to 0..255 (so that a clamp before it is
//      redundant)?  Also which way it rounds fractions, for the record
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off probe_insts.hip -o probe_insts
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>

__global__ void rpi_kernel(unsigned long long* mismatches, unsigned* examples) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
        const unsigned bits = (unsigned)i;
        const float x = __uint_as_float(bits);
        if (!(fabsf(x) < 1073741824.0f)) continue;
        int n;
        asm volatile("v_cvt_rpi_i32_f32 %0, %1" : "=v"(n) : "v"(x));
        const int want = (int)floor((double)x + 0.5);
        if (n != want) {
            const unsigned long long k = atomicAdd(mismatches, 1ull);
            if (k < 16) examples[k] = bits;
        }
    }
}

__global__ void mix_kernel(const float* coef, int n_coef, unsigned long long* mismatches, unsigned* examples) {
    // thread = (coefficient, constant term index, byte value)
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = t & 255, ci = (t >> 8) % n_coef, ki = (t >> 8) / n_coef;
    if (ki >= 64) return;
    const float a = coef[ci], k = -130.0f + 4.1f * ki + 1e-4f;
    const float want = __builtin_fmaf(a, (float)n, k);
    const float a24 = a * 16777216.0f;   // exact
    const unsigned lo = (unsigned)n, hi = (unsigned)n << 16;
    float got_lo, got_hi;
    asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,0]" : "=v"(got_lo) : "v"(a24), "v"(lo), "v"(k));
    asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "=v"(got_hi) : "v"(a24), "v"(hi), "v"(k));
    if (__float_as_uint(got_lo) != __float_as_uint(want) || __float_as_uint(got_hi) != __float_as_uint(want)) {
        const unsigned long long kk = atomicAdd(mismatches, 1ull);
        if (kk < 16) examples[kk] = (unsigned)t;
    }
}

__global__ void pk_u8_kernel(unsigned long long* mismatches, unsigned* round_kind) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x) - (1 << 20);
    if (i > (1 << 20)) return;
    const float x = (float)i;
    const unsigned want = i < 0 ? 0u : (i > 255 ? 255u : (unsigned)i);
    for (unsigned byte = 0; byte < 4; ++byte) {
        const unsigned got = __builtin_amdgcn_cvt_pk_u8_f32(x, byte, 0xa5a5a5a5u);
        const unsigned expect = (0xa5a5a5a5u & ~(0xffu << (8 * byte))) | (want << (8 * byte));
        if (got != expect) atomicAdd(mismatches, 1ull);
    }
    if (i >= 0 && i < 255) {   // fractions: bit 0 set if x.25 -> x, bit 1 if x.5 -> x (for even and odd x), bit 2 if x.75 -> x
        const unsigned a = __builtin_amdgcn_cvt_pk_u8_f32(x + 0.25f, 0, 0), b = __builtin_amdgcn_cvt_pk_u8_f32(x + 0.5f, 0, 0),
                       c = __builtin_amdgcn_cvt_pk_u8_f32(x + 0.75f, 0, 0);
        atomicOr(&round_kind[i & 1], (a == (unsigned)i ? 1u : 0u) | (b == (unsigned)i ? 2u : 0u) | (c == (unsigned)i ? 4u : 0u));
        atomicOr(&round_kind[2 + (i & 1)], (a == (unsigned)i + 1 ? 1u : 0u) | (b == (unsigned)i + 1 ? 2u : 0u) | (c == (unsigned)i + 1 ? 4u : 0u));
    }
}

int main() {
    unsigned long long* d_mis; unsigned* d_ex;
    (void)hipMalloc(&d_mis, 8); (void)hipMalloc(&d_ex, 64);
    unsigned long long mis = 0; unsigned ex[16];
    (void)hipMemset(d_mis, 0, 8);
    rpi_kernel<<<4096, 256>>>(d_mis, d_ex);
    (void)hipMemcpy(&mis, d_mis, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(ex, d_ex, 64, hipMemcpyDeviceToHost);
    printf("v_cvt_rpi_i32_f32 vs floor(double(x)+0.5): %llu mismatches\n", mis);
    for (unsigned i = 0; i < mis && i < 16; ++i) { float f; memcpy(&f, &ex[i], 4); printf("  x = 0x%08x (%.9g)\n", ex[i], f); }

    const float coef[] = {0.299f, 0.587f, 0.114f, -0.168736f, -0.331264f, 0.5f, -0.418688f, -0.081312f, 1.0f, -1.0f};
    const int n_coef = sizeof(coef) / 4;
    float* d_coef; (void)hipMalloc(&d_coef, sizeof(coef)); (void)hipMemcpy(d_coef, coef, sizeof(coef), hipMemcpyHostToDevice);
    (void)hipMemset(d_mis, 0, 8);
    mix_kernel<<<(256 * n_coef * 64 + 255) / 256, 256>>>(d_coef, n_coef, d_mis, d_ex);
    (void)hipMemcpy(&mis, d_mis, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(ex, d_ex, 64, hipMemcpyDeviceToHost);
    printf("v_fma_mix_f32 with an f16 subnormal operand vs v_fma_f32 on float(n): %llu mismatches of %d\n", mis, 256 * n_coef * 64);
    for (unsigned i = 0; i < mis && i < 16; ++i) printf("  case %u\n", ex[i]);

    (void)hipMemset(d_mis, 0, 8); (void)hipMemset(d_ex, 0, 64);
    pk_u8_kernel<<<((2 << 20) + 256) / 256, 256>>>(d_mis, d_ex);
    (void)hipMemcpy(&mis, d_mis, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(ex, d_ex, 64, hipMemcpyDeviceToHost);
    printf("v_cvt_pk_u8_f32 on whole numbers -2^20..2^20 vs clamp to 0..255, all four byte positions: %llu mismatches\n", mis);
    printf("  fractions .25/.5/.75 (bits 0/1/2) that gave x: even x %u, odd x %u; that gave x+1: even x %u, odd x %u\n", ex[0], ex[1], ex[2], ex[3]);
    return 0;
}
