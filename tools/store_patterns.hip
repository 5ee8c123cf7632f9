// What store rate the inverse kernels' output pattern can reach: 1024 frames of 1920x1080 RGB8 written as groups of
// 64 MCUs (8 rows of 1536 bytes, 5760 bytes apart), with nothing loaded.  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
constexpr int W = 1920, H = 1080, N = 1024;
constexpr size_t kImage = (size_t)W * H * 3;
constexpr int kGroups = (W / 8) * (H / 8) / 64;   // 506.25 -> the pictures' 32400 MCUs in groups of 64
constexpr int kMcus = (W / 8) * (H / 8);

// pattern A: 192 threads, thread t = MCU t/3, 8-byte piece t%3, eight row stores
__global__ __launch_bounds__(192) void rows8(uint8_t* out, uint32_t v) {
    const uint32_t g0 = blockIdx.x * 64, fm = threadIdx.x / 3, piece = threadIdx.x - 3 * fm, mcu = g0 + fm;
    if (mcu >= kMcus) return;
    const uint32_t by = mcu / (W / 8), bx = mcu - by * (W / 8);
    uint8_t* to = out + blockIdx.y * kImage + ((size_t)by * 8 * W + bx * 8) * 3 + 8 * piece;
    for (int r = 0; r < 8; ++r) *reinterpret_cast<uint2*>(to + (size_t)r * W * 3) = make_uint2(v + mcu, v);
}
// pattern B: 16-byte pieces: a group's row of 1536 bytes = 96 pieces; 192 threads do two rows per step
__global__ __launch_bounds__(192) void rows16(uint8_t* out, uint32_t v) {
    const uint32_t g0 = blockIdx.x * 64;
    const uint32_t half = threadIdx.x / 96, j = threadIdx.x - 96 * half;
    // the group may wrap to the next MCU row: byte 16 j of the group's row lies in MCU g0 + (16 j) / 24
    const uint32_t mcu = g0 + (16 * j) / 24, within = (16 * j) % 24;
    if (mcu >= kMcus) return;
    const uint32_t by = mcu / (W / 8), bx = mcu - by * (W / 8);
    uint8_t* to = out + blockIdx.y * kImage + ((size_t)by * 8 * W + bx * 8) * 3 + within;   // (a piece never straddles a row end: 240 MCUs per row, 5760 % 16 == 0)
    for (int r = half; r < 8; r += 2) *reinterpret_cast<uint4*>(to + (size_t)r * W * 3) = make_uint4(v + mcu, v, v, v);
}
// pattern C: 256 threads, a workgroup writes one whole MCU row of the picture (8 rows x 5760 bytes), 16 bytes per lane
__global__ __launch_bounds__(256) void mcurow16(uint8_t* out, uint32_t v) {
    uint8_t* base = out + blockIdx.y * kImage + (size_t)blockIdx.x * 8 * W * 3;
    for (int r = 0; r < 8; ++r)
        for (uint32_t j = threadIdx.x; j < W * 3 / 16; j += 256) *reinterpret_cast<uint4*>(base + (size_t)r * W * 3 + 16 * j) = make_uint4(v + j, v, v, v);
}
// pattern D: as A, but a workgroup does `per` consecutive groups
__global__ __launch_bounds__(192) void rows8_loop(uint8_t* out, uint32_t v, int per) {
    for (int k = 0; k < per; ++k) {
        const uint32_t g0 = (blockIdx.x * per + k) * 64, fm = threadIdx.x / 3, piece = threadIdx.x - 3 * fm, mcu = g0 + fm;
        if (mcu >= kMcus) return;
        const uint32_t by = mcu / (W / 8), bx = mcu - by * (W / 8);
        uint8_t* to = out + blockIdx.y * kImage + ((size_t)by * 8 * W + bx * 8) * 3 + 8 * piece;
        for (int r = 0; r < 8; ++r) *reinterpret_cast<uint2*>(to + (size_t)r * W * 3) = make_uint2(v + mcu, v);
    }
}
// pattern E: plain linear fill, 16 bytes per lane
__global__ __launch_bounds__(256) void linear16(uint8_t* out, uint32_t v) {
    const size_t i = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x;
    if (i * 16 < (size_t)N * kImage) reinterpret_cast<uint4*>(out)[i] = make_uint4(v, v, v, v);
}
// pattern F: as A with 64 threads handling 64/3 MCUs ... (one wave per workgroup, a third of a group)
__global__ __launch_bounds__(64) void rows8_wave(uint8_t* out, uint32_t v) {
    const uint32_t t = blockIdx.x * 64 + threadIdx.x, mcu = t / 3, piece = t - 3 * mcu;
    if (mcu >= kMcus) return;
    const uint32_t by = mcu / (W / 8), bx = mcu - by * (W / 8);
    uint8_t* to = out + blockIdx.y * kImage + ((size_t)by * 8 * W + bx * 8) * 3 + 8 * piece;
    for (int r = 0; r < 8; ++r) *reinterpret_cast<uint2*>(to + (size_t)r * W * 3) = make_uint2(v + mcu, v);
}
// pattern G: as A, but what is stored depends on a 4-byte record per block read first (12 bytes per MCU, as the decoder's records)
__global__ __launch_bounds__(192) void rows8_after_load(uint8_t* out, const uint32_t* __restrict__ records) {
    __shared__ uint32_t lds[192];
    const uint32_t g0 = blockIdx.x * 64, lane = threadIdx.x & 63, channel = threadIdx.x >> 6;
    lds[threadIdx.x] = (g0 + lane < kMcus) ? records[((size_t)blockIdx.y * kMcus + g0 + lane) * 3 + channel] : 0u;
    __syncthreads();
    const uint32_t fm = threadIdx.x / 3, piece = threadIdx.x - 3 * fm, mcu = g0 + fm;
    if (mcu >= kMcus) return;
    const uint32_t v = lds[fm] + lds[64 + fm] + lds[128 + fm];
    const uint32_t by = mcu / (W / 8), bx = mcu - by * (W / 8);
    uint8_t* to = out + blockIdx.y * kImage + ((size_t)by * 8 * W + bx * 8) * 3 + 8 * piece;
    for (int r = 0; r < 8; ++r) *reinterpret_cast<uint2*>(to + (size_t)r * W * 3) = make_uint2(v + mcu, v);
}
// pattern H: a wave per group, lane = MCU reads its 12 bytes; rows stored as in A through shuffles
__global__ __launch_bounds__(256) void rows8_wave_after_load(uint8_t* out, const uint32_t* __restrict__ records) {
    const uint32_t group = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, g0 = group * 64;
    if (g0 >= kMcus) return;
    uint32_t v = 0;
    if (g0 + lane < kMcus) {
        const uint32_t* r = records + ((size_t)blockIdx.y * kMcus + g0 + lane) * 3;
        v = r[0] + r[1] + r[2];
    }
    for (uint32_t k = 0; k < 3; ++k) {
        const uint32_t p = lane + 64 * k, fm = p / 3, piece = p - 3 * fm, mcu = g0 + fm;
        const uint32_t c = __shfl(v, fm);
        if (mcu < kMcus) {
            const uint32_t by = mcu / (W / 8), bx = mcu - by * (W / 8);
            uint8_t* to = out + blockIdx.y * kImage + ((size_t)by * 8 * W + bx * 8) * 3 + 8 * piece;
            for (int r = 0; r < 8; ++r) *reinterpret_cast<uint2*>(to + (size_t)r * W * 3) = make_uint2(c + mcu, c);
        }
    }
}
// pattern I: as A, but three MCUs of each group are left out of the row stores and written by other lanes as 24 bytes x 8 rows
// (lane = (row, MCU)), first or afterwards -- what the MCUs with detail do to the output pattern
template <bool kHolesFirst>
__global__ __launch_bounds__(192) void rows8_holes(uint8_t* out, uint32_t v) {
    const uint32_t g0 = blockIdx.x * 64, fm = threadIdx.x / 3, piece = threadIdx.x - 3 * fm, mcu = g0 + fm;
    const uint32_t h0 = (blockIdx.x * 7 + blockIdx.y) & 63, h1 = (h0 + 21) & 63, h2 = (h0 + 42) & 63;
    auto holes = [&] {
        if (threadIdx.x < 24) {
            const uint32_t r = threadIdx.x >> 3 ? threadIdx.x & 7 : threadIdx.x & 7, which = threadIdx.x >> 3;
            const uint32_t m2 = g0 + (which == 0 ? h0 : which == 1 ? h1 : h2);
            if (m2 < kMcus) {
                const uint32_t by = m2 / (W / 8), bx = m2 - by * (W / 8);
                uint2* dst = reinterpret_cast<uint2*>(out + blockIdx.y * kImage + ((size_t)(by * 8 + r) * W + bx * 8) * 3);
                dst[0] = make_uint2(v, m2); dst[1] = make_uint2(v, r); dst[2] = make_uint2(v, v);
            }
        }
    };
    if (kHolesFirst) holes();
    if (mcu < kMcus && fm != h0 && fm != h1 && fm != h2) {
        const uint32_t by = mcu / (W / 8), bx = mcu - by * (W / 8);
        uint8_t* to = out + blockIdx.y * kImage + ((size_t)by * 8 * W + bx * 8) * 3 + 8 * piece;
        for (int r = 0; r < 8; ++r) *reinterpret_cast<uint2*>(to + (size_t)r * W * 3) = make_uint2(v + mcu, v);
    }
    if (!kHolesFirst) holes();
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <class F> int timed(const char* name, F launch) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int i = 0; i < 5; ++i) launch(); CK(hipEventRecord(b)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
    printf("%-34s %.3f ms  %.2f TB/s\n", name, ms, (double)N * kImage / ms / 1e9); fflush(stdout);
    return 0;
}
int main() {
    uint8_t* out; CK(hipMalloc(&out, (size_t)N * kImage));
    const int groups = (kMcus + 63) / 64;
    timed("A rows of 8 B, 192 threads", [&] { rows8<<<dim3(groups, N), 192>>>(out, 1); });
    timed("B rows of 16 B, 192 threads", [&] { rows16<<<dim3(groups, N), 192>>>(out, 2); });
    timed("C MCU row per workgroup, 16 B", [&] { mcurow16<<<dim3(H / 8, N), 256>>>(out, 3); });
    timed("D 2 groups per workgroup", [&] { rows8_loop<<<dim3((groups + 1) / 2, N), 192>>>(out, 4, 2); });
    timed("D 4 groups per workgroup", [&] { rows8_loop<<<dim3((groups + 3) / 4, N), 192>>>(out, 4, 4); });
    timed("D 8 groups per workgroup", [&] { rows8_loop<<<dim3((groups + 7) / 8, N), 192>>>(out, 4, 8); });
    timed("E linear 16 B", [&] { linear16<<<dim3(4096, (unsigned)(((size_t)N * kImage / 16 + 4096 * 256 - 1) / (4096 * 256))), 256>>>(out, 5); });
    timed("F one wave per workgroup, 8 B", [&] { rows8_wave<<<dim3((kMcus * 3 + 63) / 64, N), 64>>>(out, 6); });
    uint32_t* records; CK(hipMalloc(&records, (size_t)N * kMcus * 12)); CK(hipMemset(records, 1, (size_t)N * kMcus * 12));
    timed("G as A after a record load", [&] { rows8_after_load<<<dim3(groups, N), 192>>>(out, records); });
    timed("H wave per group after a load", [&] { rows8_wave_after_load<<<dim3((groups + 3) / 4, N), 256>>>(out, records); });
    timed("I holes written afterwards", [&] { rows8_holes<false><<<dim3(groups, N), 192>>>(out, 7); });
    timed("I holes written first", [&] { rows8_holes<true><<<dim3(groups, N), 192>>>(out, 8); });
    timed("A again", [&] { rows8<<<dim3(groups, N), 192>>>(out, 1); });
    CK(hipFree(records));
    CK(hipFree(out));
    return 0;
}
