// Micro-benchmark: issue cost (cycles per wave64 instruction per SIMD) of the VALU
// instructions the fused JPEG kernels lean on, measured with enough waves to fill
// every SIMD.  Build: hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>
#include <algorithm>

#define REP16(x) x x x x x x x x x x x x x x x x
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
    unsigned u0 = threadIdx.x * 2654435761u + 1, u1 = u0 * 3, u2 = u0 * 5, u3 = u0 * 7;
    typedef float __attribute__((ext_vector_type(2))) f2;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a1, a2}, p3 = {a3, a0};
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { REP16(asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %2, %2, %1\n v_add_f32 %3, %3, %1\n v_add_f32 %4, %4, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed) :);) }
        if (OP == 1) { REP16(asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(d0) :);) }
        if (OP == 2) { REP16(asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(d0) :);) }
        if (OP == 3) { REP16(asm volatile("v_cvt_f64_u32 %0, %4\n v_cvt_f64_u32 %1, %5\n v_cvt_f64_u32 %2, %6\n v_cvt_f64_u32 %3, %7" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(u0), "v"(u1), "v"(u2), "v"(u3) :);) }
        if (OP == 4) { REP16(asm volatile("v_floor_f64 %0, %0\n v_floor_f64 %1, %1\n v_floor_f64 %2, %2\n v_floor_f64 %3, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) ::);) }
        if (OP == 5) { REP16(asm volatile("v_cvt_i32_f64 %0, %4\n v_cvt_i32_f64 %1, %5\n v_cvt_i32_f64 %2, %6\n v_cvt_i32_f64 %3, %7" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(d0), "v"(d1), "v"(d2), "v"(d3) :);) }
        if (OP == 6) { REP16(asm volatile("v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f32_f64 %2, %6\n v_cvt_f32_f64 %3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(d0), "v"(d1), "v"(d2), "v"(d3) :);) }
        if (OP == 7) { REP16(asm volatile("v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) :);) }
        if (OP == 8) { REP16(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(p0) :);) }
        if (OP == 9) { REP16(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(p0) :);) }
        if (OP == 10) { REP16(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) ::);) }
        if (OP == 11) { REP16(asm volatile("v_mad_i32_i24 %0, %0, %4, %1\n v_mad_i32_i24 %1, %1, %4, %2\n v_mad_i32_i24 %2, %2, %4, %3\n v_mad_i32_i24 %3, %3, %4, %0" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(u0) :);) }
        if (OP == 12) { REP16(asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(u0) :);) }
        if (OP == 13) { REP16(asm volatile("v_bfe_u32 %0, %0, 8, 8\n v_bfe_u32 %1, %1, 8, 8\n v_bfe_u32 %2, %2, 8, 8\n v_bfe_u32 %3, %3, 8, 8" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) ::);) }
        if (OP == 14) { REP16(asm volatile("v_cvt_f32_ubyte1 %0, %4\n v_cvt_f32_ubyte1 %1, %5\n v_cvt_f32_ubyte1 %2, %6\n v_cvt_f32_ubyte1 %3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(u0), "v"(u1), "v"(u2), "v"(u3) :);) }
        if (OP == 15) { REP16(asm volatile("v_fma_f32 %0, %0, %4, %1\n v_fma_f32 %1, %1, %4, %2\n v_fma_f32 %2, %2, %4, %3\n v_fma_f32 %3, %3, %4, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed) :);) }
        if (OP == 16) { REP16(asm volatile("v_fma_f64 %0, %0, %4, %1\n v_fma_f64 %1, %1, %4, %2\n v_fma_f64 %2, %2, %4, %3\n v_fma_f64 %3, %3, %4, %0" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(d0) :);) }
        if (OP == 17) { REP16(asm volatile("v_floor_f32 %0, %0\n v_floor_f32 %1, %1\n v_floor_f32 %2, %2\n v_floor_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) ::);) }
        if (OP == 18) { REP16(asm volatile("v_cvt_i32_f32 %0, %4\n v_cvt_i32_f32 %1, %5\n v_cvt_i32_f32 %2, %6\n v_cvt_i32_f32 %3, %7" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) :);) }
        if (OP == 19) { REP16(asm volatile("v_max_f64 %0, %0, %4\n v_max_f64 %1, %1, %4\n v_max_f64 %2, %2, %4\n v_max_f64 %3, %3, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(d0) :);) }
        if (OP == 20) { REP16(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) :: "vcc");) }
        if (OP == 21) { REP16(asm volatile("v_lshlrev_b64 %0, 3, %0\n v_lshlrev_b64 %1, 3, %1\n v_lshlrev_b64 %2, 3, %2\n v_lshlrev_b64 %3, 3, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) ::);) }
        if (OP == 22) { REP16(asm volatile("v_cvt_f64_i32 %0, %4\n v_cvt_f64_i32 %1, %5\n v_cvt_f64_i32 %2, %6\n v_cvt_f64_i32 %3, %7" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(u0), "v"(u1), "v"(u2), "v"(u3) :);) }
        if (OP == 23) { REP16(asm volatile("v_fract_f32 %0, %0\n v_fract_f32 %1, %1\n v_fract_f32 %2, %2\n v_fract_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) ::);) }
        if (OP == 24) { REP16(asm volatile("v_rndne_f32 %0, %0\n v_rndne_f32 %1, %1\n v_rndne_f32 %2, %2\n v_rndne_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) ::);) }
        if (OP == 25) { REP16(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %1, %2\n v_cmp_lt_f32 vcc, %2, %3\n v_cmp_lt_f32 vcc, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) :: "vcc");) }
        if (OP == 26) { REP16(asm volatile("v_cmp_lt_f64 vcc, %0, %1\n v_cmp_lt_f64 vcc, %1, %2\n v_cmp_lt_f64 vcc, %2, %3\n v_cmp_lt_f64 vcc, %3, %0" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) :: "vcc");) }
        if (OP == 27) { REP16(asm volatile("v_mul_f32 %0, %0, %1\n v_mul_f32 %2, %2, %1\n v_mul_f32 %3, %3, %1\n v_mul_f32 %4, %4, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed) :);) }

        if (OP == 30) { REP16(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cmp_lt_f32 vcc, %2, %3\n v_cndmask_b32 %2, %2, %3, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) :: "vcc");) }
        if (OP == 31) { REP16(asm volatile("v_cndmask_b32 %0, %4, %5, s[20:21]\n v_cndmask_b32 %1, %5, %6, s[20:21]\n v_cndmask_b32 %2, %6, %7, s[20:21]\n v_cndmask_b32 %3, %7, %4, s[20:21]" : "=v"(u0), "=v"(u1), "=v"(u2), "=v"(u3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "s20", "s21");) }
        if (OP == 32) { REP16(asm volatile("v_cvt_rpi_i32_f32 %0, %4\n v_cvt_rpi_i32_f32 %1, %5\n v_cvt_rpi_i32_f32 %2, %6\n v_cvt_rpi_i32_f32 %3, %7" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) :);) }
        if (OP == 33) { REP16(asm volatile("v_cvt_flr_i32_f32 %0, %4\n v_cvt_flr_i32_f32 %1, %5\n v_cvt_flr_i32_f32 %2, %6\n v_cvt_flr_i32_f32 %3, %7" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) :);) }
        if (OP == 34) { REP16(asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0\n v_div_scale_f32 %2, vcc, %2, %3, %2\n v_div_scale_f32 %1, vcc, %1, %0, %1\n v_div_scale_f32 %3, vcc, %3, %2, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) :: "vcc");) }
        if (OP == 35) { REP16(asm volatile("v_div_fixup_f32 %0, %0, %1, %2\n v_div_fixup_f32 %1, %1, %2, %3\n v_div_fixup_f32 %2, %2, %3, %0\n v_div_fixup_f32 %3, %3, %0, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) ::);) }
        if (OP == 36) { REP16(asm volatile("v_div_fmas_f32 %0, %0, %1, %2\n v_div_fmas_f32 %1, %1, %2, %3\n v_div_fmas_f32 %2, %2, %3, %0\n v_div_fmas_f32 %3, %3, %0, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) :: "vcc");) }
        if (OP == 37) { REP16(asm volatile("v_min_f32 %0, %0, %1\n v_min_f32 %1, %1, %2\n v_min_f32 %2, %2, %3\n v_min_f32 %3, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) ::);) }
        if (OP == 38) { REP16(asm volatile("v_med3_f32 %0, %0, %1, %2\n v_med3_f32 %1, %1, %2, %3\n v_med3_f32 %2, %2, %3, %0\n v_med3_f32 %3, %3, %0, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) ::);) }
        if (OP == 39) { REP16(asm volatile("v_perm_b32 %0, %0, %1, %2\n v_perm_b32 %1, %1, %2, %3\n v_perm_b32 %2, %2, %3, %0\n v_perm_b32 %3, %3, %0, %1" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) ::);) }
        if (OP == 40) { REP16(asm volatile("v_lshl_or_b32 %0, %0, 16, %1\n v_lshl_or_b32 %1, %1, 16, %2\n v_lshl_or_b32 %2, %2, 16, %3\n v_lshl_or_b32 %3, %3, 16, %0" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) ::);) }
        if (OP == 41) { REP16(asm volatile("v_and_b32 %0, %0, %1\n v_and_b32 %1, %1, %2\n v_and_b32 %2, %2, %3\n v_and_b32 %3, %3, %0" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) ::);) }
        if (OP == 42) { REP16(asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) ::);) }
        if (OP == 43) { REP16(asm volatile("v_cvt_f32_i32 %0, %4\n v_cvt_f32_i32 %1, %5\n v_cvt_f32_i32 %2, %6\n v_cvt_f32_i32 %3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(u0), "v"(u1), "v"(u2), "v"(u3) :);) }
        if (OP == 44) { REP16(asm volatile("v_lshrrev_b32 %0, 3, %0\n v_lshrrev_b32 %1, 3, %1\n v_lshrrev_b32 %2, 3, %2\n v_lshrrev_b32 %3, 3, %3" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) ::);) }
        if (OP == 45) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %1\n v_pk_fma_f32 %1, %1, %4, %2\n v_pk_fma_f32 %2, %2, %4, %3\n v_pk_fma_f32 %3, %3, %4, %0" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(p0) :);) }
        if (OP == 46) { REP16(asm volatile("v_sub_f32 %0, |%0|, %1\n v_sub_f32 %1, |%1|, %2\n v_sub_f32 %2, |%2|, %3\n v_sub_f32 %3, |%3|, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) ::);) }
        if (OP == 47) { REP16(asm volatile("v_cmp_lt_f32 s[20:21], %0, %1\n v_cmp_lt_f32 s[22:23], %1, %2\n v_cmp_lt_f32 s[24:25], %2, %3\n v_cmp_lt_f32 s[26:27], %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) :: "s20","s21","s22","s23","s24","s25","s26","s27");) }
        if (OP == 48) { REP16(asm volatile("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(u0) :);) }
        if (OP == 49) { REP16(asm volatile("v_ffbh_u32 %0, %0\n v_ffbh_u32 %1, %1\n v_ffbh_u32 %2, %2\n v_ffbh_u32 %3, %3" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) ::);) }
        if (OP == 50) { REP16(asm volatile("v_fma_mix_f32 %0, %4, %5, %0 op_sel_hi:[0,1,0]\n v_fma_mix_f32 %1, %4, %6, %1 op_sel:[0,1,0] op_sel_hi:[0,1,0]\n v_fma_mix_f32 %2, %4, %7, %2 op_sel_hi:[0,1,0]\n v_fma_mix_f32 %3, %4, %5, %3 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed), "v"(u0), "v"(u1), "v"(u2) :);) }
        if (OP == 51) { REP16(asm volatile("v_min_f32 %0, %0, %4\n v_max_f32 %1, %1, %4\n v_min_f32 %2, %2, %4\n v_max_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed) :);) }
        if (OP == 52) { REP16(asm volatile("v_mul_f32 %0, 0x3f123456, %0\n v_mul_f32 %1, 0x3f223456, %1\n v_mul_f32 %2, 0x3f323456, %2\n v_mul_f32 %3, 0x3f423456, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) ::);) }
        if (OP == 53) { REP16(asm volatile("v_mul_f32 %0, s20, %0\n v_mul_f32 %1, s21, %1\n v_mul_f32 %2, s22, %2\n v_mul_f32 %3, s23, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) :: "s20", "s21", "s22", "s23");) }
        if (OP == 54) { REP16(asm volatile("v_and_or_b32 %0, %0, %1, %2\n v_and_or_b32 %1, %1, %2, %3\n v_and_or_b32 %2, %2, %3, %0\n v_and_or_b32 %3, %3, %0, %1" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) ::);) }
        if (OP == 55) { REP16(asm volatile("v_bfi_b32 %0, %0, %1, %2\n v_bfi_b32 %1, %1, %2, %3\n v_bfi_b32 %2, %2, %3, %0\n v_bfi_b32 %3, %3, %0, %1" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) ::);) }
        if (OP == 56) { REP16(asm volatile("v_alignbit_b32 %0, %0, %1, 8\n v_alignbit_b32 %1, %1, %2, 8\n v_alignbit_b32 %2, %2, %3, 8\n v_alignbit_b32 %3, %3, %0, 8" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) ::);) }
        if (OP == 57) { REP16(asm volatile("v_cvt_pkrtz_f16_f32 %0, %4, %5\n v_cvt_pkrtz_f16_f32 %1, %5, %6\n v_cvt_pkrtz_f16_f32 %2, %6, %7\n v_cvt_pkrtz_f16_f32 %3, %7, %4" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) :);) }
        if (OP == 58) { REP16(asm volatile("v_add3_u32 %0, %0, %1, %2\n v_add3_u32 %1, %1, %2, %3\n v_add3_u32 %2, %2, %3, %0\n v_add3_u32 %3, %3, %0, %1" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) ::);) }
        if (OP == 59) { REP16(asm volatile("v_or_b32 %0, %0, %1\n v_or_b32 %1, %1, %2\n v_or_b32 %2, %2, %3\n v_or_b32 %3, %3, %0" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) ::);) }
        if (OP == 60) { REP16(asm volatile("v_fmac_f32 %0, %4, %1\n v_fmac_f32 %1, %4, %2\n v_fmac_f32 %2, %4, %3\n v_fmac_f32 %3, %4, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed) :);) }
        if (OP == 61) { REP16(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) ::);) }
        if (OP == 62) { REP16(asm volatile("v_cvt_pk_u8_f32 %0, %4, 1, %0\n v_cvt_pk_u8_f32 %1, %5, 2, %1\n v_cvt_pk_u8_f32 %2, %6, 3, %2\n v_cvt_pk_u8_f32 %3, %7, 0, %3" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) :);) }
        if (OP == 63) { REP16(asm volatile("v_sub_f32 %0, %0, %4\n v_subrev_f32 %1, %1, %4\n v_sub_f32 %2, %2, %4\n v_subrev_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed) :);) }
        if (OP == 65) { REP16(asm volatile("v_cmp_class_f32 vcc, %0, %4\n v_cmp_class_f32 vcc, %1, %4\n v_cmp_class_f32 vcc, %2, %4\n v_cmp_class_f32 vcc, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(u0) : "vcc");) }
        if (OP == 66) { REP16(asm volatile("v_max_i32 %0, %0, %1\n v_min_u32 %1, %1, %2\n v_max_i32 %2, %2, %3\n v_min_u32 %3, %3, %0" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) ::);) }
        if (OP == 67) { REP16(asm volatile("v_lshl_add_u32 %0, %0, 3, %1\n v_lshl_add_u32 %1, %1, 3, %2\n v_lshl_add_u32 %2, %2, 3, %3\n v_lshl_add_u32 %3, %3, 3, %0" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) ::);) }
        if (OP == 68) { REP16(asm volatile("v_pk_mul_f32 %0, %0, s[20:21]\n v_pk_mul_f32 %1, %1, s[22:23]\n v_pk_mul_f32 %2, %2, s[20:21]\n v_pk_mul_f32 %3, %3, s[22:23]" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) :: "s20","s21","s22","s23");) }
        if (OP == 69) { REP16(asm volatile("v_mul_legacy_f32 %0, %0, %4\n v_mul_legacy_f32 %1, %1, %4\n v_mul_legacy_f32 %2, %2, %4\n v_mul_legacy_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed) :);) }
        if (OP == 70) { REP16(asm volatile("v_mul_f32_sdwa %0, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n v_mul_f32_sdwa %1, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_mul_f32_sdwa %2, %4, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n v_mul_f32_sdwa %3, %4, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed), "v"(u0), "v"(u1), "v"(u2) :);) }
        if (OP == 71) { REP16(asm volatile("v_add_f32_sdwa %0, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n v_add_f32_sdwa %1, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_add_f32_sdwa %2, %4, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n v_add_f32_sdwa %3, %4, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed), "v"(u0), "v"(u1), "v"(u2) :);) }
        if (OP == 72) { REP16(asm volatile("v_add_f32 %0, s20, %0\n v_add_f32 %1, s21, %1\n v_add_f32 %2, s22, %2\n v_add_f32 %3, s23, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) :: "s20", "s21", "s22", "s23");) }
        if (OP == 73) { REP16(asm volatile("v_fma_f32 %0, s20, %0, %1\n v_fma_f32 %1, s21, %1, %2\n v_fma_f32 %2, s22, %2, %3\n v_fma_f32 %3, s23, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) :: "s20", "s21", "s22", "s23");) }
        if (OP == 74) { REP16(asm volatile("v_fmac_f32 %0, s20, %1\n v_fmac_f32 %1, s21, %2\n v_fmac_f32 %2, s22, %3\n v_fmac_f32 %3, s23, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) :: "s20", "s21", "s22", "s23");) }
        if (OP == 75) { REP16(asm volatile("v_mov_b32 %0, s20\n v_mov_b32 %1, s21\n v_mov_b32 %2, s22\n v_mov_b32 %3, s23" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) :: "s20", "s21", "s22", "s23");) }
        if (OP == 76) { REP16(asm volatile("v_mul_f32 %0, 0.5, %0\n v_mul_f32 %1, 2.0, %1\n v_mul_f32 %2, 0.5, %2\n v_mul_f32 %3, 2.0, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) ::);) }
        if (OP == 77) { REP16(asm volatile("v_fma_f32 %0, %4, %0, %1\n v_fma_f32 %1, %4, %1, %2\n v_fma_f32 %2, %4, %2, %3\n v_fma_f32 %3, %4, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed) :);) }
        if (OP == 78) { REP16(asm volatile("v_fma_f32 %0, -%4, %0, %1\n v_fma_f32 %1, -%4, %1, %2\n v_fma_f32 %2, %4, |%2|, %3\n v_fma_f32 %3, %4, |%3|, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed) :);) }
        if (OP == 79) { REP16(asm volatile("v_lshlrev_b32 %0, 3, %0\n v_lshlrev_b32 %1, 5, %1\n v_ashrrev_i32 %2, 3, %2\n v_xor_b32 %3, %3, %0" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) ::);) }
        if (OP == 80) { REP16(asm volatile("v_sub_u32 %0, %0, %1\n v_subrev_u32 %1, %1, %2\n v_sub_u32 %2, %2, %3\n v_subrev_u32 %3, %3, %0" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) ::);) }
        if (OP == 81) { REP16(asm volatile("v_cvt_f32_ubyte0_sdwa %0, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n v_cvt_f32_ubyte0_sdwa %1, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2\n v_cvt_f32_ubyte0_sdwa %2, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3\n v_cvt_f32_ubyte0_sdwa %3, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(u0), "v"(u1), "v"(u2), "v"(u3) :);) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + (float)(d0 + d1 + d2 + d3) + (float)(u0 + u1 + u2 + u3) + p0.x + p1.y + p2.x + p3.y;
}

template <int OP>
void run(const char* name, float* out, int waves_per_simd) {
    const int iters = 2000;
    // 256 CUs x 4 SIMDs; 256-thread blocks = one wave per SIMD per block
    const int blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(out, 10, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // instructions per wave: iters * 64; per SIMD: * waves_per_simd
    double inst_per_simd = (double)iters * 64 * waves_per_simd;
    double ns_per_inst = ms * 1e6 / inst_per_simd;
    printf("%-18s waves/SIMD=%d  %.3f ns per wave-instruction per SIMD  (%.2f cycles @2.4GHz)\n", name, waves_per_simd, ns_per_inst, ns_per_inst * 2.4);
}

// Shader clock under a pure f32 VALU load: every workgroup stamps the shader-cycle counter (s_memtime)
// and the 100 MHz wall counter (s_memrealtime) around its loop; clock = d(cycles) / d(ticks) * 100 MHz.
__global__ __launch_bounds__(256) void clock_probe(float* out, unsigned long long* stamps, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        REP16(asm volatile("v_mul_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_add_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed) :);)
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}

void run_clock(float* out) {
    const int blocks = 256 * 5;   // five waves per SIMD, as the fused forward kernel runs
    unsigned long long* stamps; hipMalloc(&stamps, blocks * 16);
    std::vector<unsigned long long> h(2 * blocks);
    for (int rep = 0; rep < 12; ++rep) {          // about 2 s in all; the clock settles over the first launches
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        clock_probe<<<blocks, 256>>>(out, stamps, 400000, 1.000001f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), stamps, blocks * 16, hipMemcpyDeviceToHost);
        std::vector<double> mhz;
        for (int i = 0; i < blocks; ++i) mhz.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 100.0);
        std::sort(mhz.begin(), mhz.end());
        const double inst = 400000.0 * 64 * 5;
        printf("clock probe launch %2d: %.1f ms, median in-kernel clock %.0f MHz (min %.0f, max %.0f), %.3f ns = %.2f cycles per wave-instruction per SIMD\n",
               rep, ms, mhz[blocks / 2], mhz.front(), mhz.back(), ms * 1e6 / inst, ms * 1e6 / inst * mhz[blocks / 2] * 1e-3);
    }
}

int main(int argc, char** argv) {
    float* out; hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    if (argc > 1 && std::string(argv[1]) == "clock") { run_clock(out); return 0; }
    if (argc > 1 && std::string(argv[1]) == "more") {   // further instructions, five waves per SIMD
        const int w = 5;
        run<27>("v_mul_f32", out, w); run<50>("v_fma_mix_f32 (f16 src)", out, w); run<51>("v_min/max_f32", out, w); run<52>("v_mul_f32 literal", out, w);
        run<53>("v_mul_f32 sgpr", out, w); run<60>("v_fmac_f32", out, w); run<63>("v_sub/subrev_f32", out, w); run<46>("v_sub_f32 |abs|", out, w);
        run<54>("v_and_or_b32", out, w); run<55>("v_bfi_b32", out, w); run<56>("v_alignbit_b32", out, w); run<57>("v_cvt_pkrtz_f16_f32", out, w);
        run<58>("v_add3_u32", out, w); run<59>("v_or_b32", out, w); run<41>("v_and_b32", out, w); run<44>("v_lshrrev_b32", out, w); run<61>("v_mov_b32", out, w);
        run<62>("v_cvt_pk_u8_f32", out, w); run<65>("v_cmp_class_f32", out, w); run<66>("v_max_i32/min_u32", out, w); run<67>("v_lshl_add_u32", out, w);
        run<68>("v_pk_mul_f32 sgpr pair", out, w); run<8>("v_pk_mul_f32 vgpr", out, w); run<25>("v_cmp_lt_f32", out, w); run<20>("v_cndmask_b32", out, w); run<38>("v_med3_f32", out, w);
        run<13>("v_bfe_u32", out, w); run<40>("v_lshl_or_b32", out, w); run<42>("v_add_u32", out, w); run<17>("v_floor_f32", out, w); run<43>("v_cvt_f32_i32", out, w); run<18>("v_cvt_i32_f32", out, w);
        run<69>("v_mul_legacy_f32", out, w);
        return 0;
    }
    if (argc > 1 && std::string(argv[1]) == "operands") {   // operand kinds and sub-dword addressing, five waves per SIMD
        const int w = 5;
        run<27>("v_mul_f32 vgpr", out, w); run<53>("v_mul_f32 sgpr", out, w); run<52>("v_mul_f32 literal", out, w); run<76>("v_mul_f32 inline const", out, w);
        run<72>("v_add_f32 sgpr", out, w); run<77>("v_fma_f32 vgpr", out, w); run<73>("v_fma_f32 sgpr", out, w); run<78>("v_fma_f32 neg/abs", out, w); run<74>("v_fmac_f32 sgpr", out, w); run<75>("v_mov_b32 sgpr", out, w);
        run<70>("v_mul_f32_sdwa byte", out, w); run<71>("v_add_f32_sdwa byte", out, w); run<81>("v_cvt_f32_ubyte0_sdwa", out, w); run<79>("v_lshl/ashr/xor", out, w); run<80>("v_sub_u32", out, w);
        return 0;
    }
    if (argc > 1 && std::string(argv[1]) == "sweep") {   // issue rate against waves per SIMD
        for (int w : {1, 2, 3, 4, 5, 6, 8}) { run<0>("v_add_f32", out, w); run<27>("v_mul_f32", out, w); run<9>("v_pk_add_f32", out, w); run<14>("v_cvt_f32_ubyte1", out, w); run<23>("v_fract_f32", out, w); run<32>("v_cvt_rpi_i32_f32", out, w); run<39>("v_perm_b32", out, w); }
        return 0;
    }
    for (int w : {4}) {
        run<0>("v_add_f32", out, w); run<27>("v_mul_f32", out, w); run<15>("v_fma_f32", out, w);
        run<8>("v_pk_mul_f32", out, w); run<9>("v_pk_add_f32", out, w);
        run<1>("v_mul_f64", out, w); run<2>("v_add_f64", out, w); run<16>("v_fma_f64", out, w);
        run<3>("v_cvt_f64_u32", out, w); run<22>("v_cvt_f64_i32", out, w); run<4>("v_floor_f64", out, w); run<5>("v_cvt_i32_f64", out, w);
        run<6>("v_cvt_f32_f64", out, w); run<7>("v_cvt_f64_f32", out, w); run<19>("v_max_f64", out, w); run<26>("v_cmp_lt_f64", out, w);
        run<10>("v_rcp_f32", out, w); run<11>("v_mad_i32_i24", out, w); run<12>("v_mul_lo_u32", out, w);
        run<13>("v_bfe_u32", out, w); run<14>("v_cvt_f32_ubyte1", out, w); run<17>("v_floor_f32", out, w);
        run<18>("v_cvt_i32_f32", out, w); run<20>("v_cndmask_b32", out, w); run<21>("v_lshlrev_b64", out, w);
        run<23>("v_fract_f32", out, w); run<24>("v_rndne_f32", out, w); run<25>("v_cmp_lt_f32", out, w);

        run<30>("cmp+cndmask (pair)", out, w); run<31>("v_cndmask sgpr", out, w); run<32>("v_cvt_rpi_i32_f32", out, w); run<33>("v_cvt_flr_i32_f32", out, w);
        run<34>("v_div_scale_f32", out, w); run<35>("v_div_fixup_f32", out, w); run<36>("v_div_fmas_f32", out, w); run<37>("v_min_f32", out, w);
        run<38>("v_med3_f32", out, w); run<39>("v_perm_b32", out, w); run<40>("v_lshl_or_b32", out, w); run<41>("v_and_b32", out, w); run<42>("v_add_u32", out, w);
        run<43>("v_cvt_f32_i32", out, w); run<44>("v_lshrrev_b32", out, w); run<45>("v_pk_fma_f32", out, w); run<46>("v_sub_f32 abs", out, w); run<47>("v_cmp sgpr dst", out, w);
        run<48>("v_mul_hi_u32", out, w); run<49>("v_ffbh_u32", out, w);
        printf("\n");
    }
    return 0;
}
