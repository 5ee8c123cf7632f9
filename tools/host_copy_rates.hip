// Rates that bound a host-to-host pipeline on this box: pageable -> pinned memcpy with 1..16 threads, pinned -> device and
// device -> pinned copies (one direction and both at once on two streams), pageable -> device through the runtime.
// build: hipcc --offload-arch=gfx950 -O2 -o tools/host_copy_rates tools/host_copy_rates.hip -lpthread
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void copy_threads(char* dst, const char* src, size_t bytes, int threads) {
    std::vector<std::thread> w;
    const size_t piece = ((bytes + threads - 1) / threads + 63) & ~(size_t)63;
    for (int t = 1; t < threads; ++t) {
        const size_t at = t * piece, len = at < bytes ? (bytes - at < piece ? bytes - at : piece) : 0;
        w.emplace_back([=] { if (len) std::memcpy(dst + at, src + at, len); });
    }
    std::memcpy(dst, src, piece < bytes ? piece : bytes);
    for (auto& t : w) t.join();
}

int main() {
    const size_t bytes = (size_t)3 * 3840 * 2160, n = 16;
    char* pageable = (char*)std::malloc(bytes * n);
    std::memset(pageable, 1, bytes * n);
    char *pin0, *pin1, *d0, *d1;
    CK(hipHostMalloc((void**)&pin0, bytes, hipHostMallocDefault));
    CK(hipHostMalloc((void**)&pin1, bytes, hipHostMallocDefault));
    CK(hipMalloc((void**)&d0, bytes));
    CK(hipMalloc((void**)&d1, bytes));
    hipStream_t s0, s1;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    std::memset(pin0, 2, bytes);
    std::memset(pin1, 3, bytes);
    for (int threads : {1, 2, 4, 8, 12, 16}) {
        double best = 0;
        for (int rep = 0; rep < 3; ++rep) {
            const double t0 = now();
            for (size_t i = 0; i < n; ++i) copy_threads(pin0, pageable + i * bytes, bytes, threads);
            const double gbs = bytes * n / (now() - t0) / 1e9;
            if (gbs > best) best = gbs;
        }
        std::printf("pageable -> pinned, %2d threads: %6.1f GB/s\n", threads, best);
    }
    for (int threads : {4, 8}) {
        const double t0 = now();
        for (size_t i = 0; i < n; ++i) copy_threads(pageable + i * bytes, pin0, bytes, threads);
        std::printf("pinned -> pageable, %2d threads: %6.1f GB/s\n", threads, bytes * n / (now() - t0) / 1e9);
    }
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now();
        for (size_t i = 0; i < n; ++i) CK(hipMemcpyAsync(d0, pin0, bytes, hipMemcpyHostToDevice, s0));
        CK(hipStreamSynchronize(s0));
        std::printf("pinned -> device: %6.1f GB/s\n", bytes * n / (now() - t0) / 1e9);
        t0 = now();
        for (size_t i = 0; i < n; ++i) CK(hipMemcpyAsync(pin1, d1, bytes, hipMemcpyDeviceToHost, s1));
        CK(hipStreamSynchronize(s1));
        std::printf("device -> pinned: %6.1f GB/s\n", bytes * n / (now() - t0) / 1e9);
        t0 = now();
        for (size_t i = 0; i < n; ++i) {
            CK(hipMemcpyAsync(d0, pin0, bytes, hipMemcpyHostToDevice, s0));
            CK(hipMemcpyAsync(pin1, d1, bytes, hipMemcpyDeviceToHost, s1));
        }
        CK(hipStreamSynchronize(s0));
        CK(hipStreamSynchronize(s1));
        std::printf("both ways at once: %6.1f GB/s each\n", bytes * n / (now() - t0) / 1e9);
    }
    {
        const double t0 = now();
        for (size_t i = 0; i < n; ++i) CK(hipMemcpyAsync(d0, pageable + i * bytes, bytes, hipMemcpyHostToDevice, s0));
        CK(hipStreamSynchronize(s0));
        std::printf("pageable -> device through the runtime: %6.1f GB/s\n", bytes * n / (now() - t0) / 1e9);
    }
    {   // memcpy of image i+1 while image i travels
        const double t0 = now();
        for (size_t i = 0; i < n; ++i) {
            char* pin = (i & 1) ? pin1 : pin0;
            copy_threads(pin, pageable + i * bytes, bytes, 8);
            CK(hipMemcpyAsync((i & 1) ? d1 : d0, pin, bytes, hipMemcpyHostToDevice, s0));
            if (i) { /* the other pinned buffer is free once the copy before this one is done */ }
            if (i >= 1) CK(hipStreamSynchronize(s0));
        }
        CK(hipStreamSynchronize(s0));
        std::printf("pageable -> pinned (8 threads) -> device, naive overlap: %6.1f GB/s\n", bytes * n / (now() - t0) / 1e9);
    }
    void* reg = pageable;
    hipError_t e = hipHostRegister(reg, bytes, hipHostRegisterDefault);
    std::printf("hipHostRegister of malloc memory: %s\n", hipGetErrorString(e));
    if (e == hipSuccess) {
        const double t0 = now();
        for (size_t i = 0; i < n; ++i) CK(hipMemcpyAsync(d0, pageable, bytes, hipMemcpyHostToDevice, s0));
        CK(hipStreamSynchronize(s0));
        std::printf("registered -> device: %6.1f GB/s\n", bytes * n / (now() - t0) / 1e9);
    }
    return 0;
}
