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
    CK(hipFree(out));
    return 0;
}
