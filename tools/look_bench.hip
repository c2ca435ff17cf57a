// How long does a "look" from the host cost the device?  A chain of tiny kernels with a look between them:
//   (a) hipMemcpyAsync of 64 bytes to pinned memory + hipStreamSynchronize (what pipeline.hip's d2h_small / stream_sync do)
//   (b) a kernel that writes the 64 bytes and a sequence number to host-coherent pinned memory, the host spins on the number
// Prints microseconds per look (device idle included: wall time of N rounds of kernel + look, minus N kernels back to back).
// hipcc --offload-arch=gfx950 -O3 tools/look_bench.hip -o /tmp/look_bench && /tmp/look_bench
// MI355X, round 4: 2.7 us per kernel back to back; (a) 14.8 us per round, (b) 9.5 us - a look costs the device about 12 us
// either way and spinning on host memory would save 5 of them: not built into the pipeline.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void work(uint32_t* p) { if (threadIdx.x == 0) p[0] += 1; }
__global__ void publish(const uint32_t* src, uint32_t* dst, uint32_t words, uint32_t* seq, uint32_t value) {
    if (threadIdx.x < words) dst[threadIdx.x] = src[threadIdx.x];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(seq, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

int main() {
    uint32_t *d = nullptr, *h = nullptr;
    CHECK(hipMalloc(&d, 256));
    CHECK(hipMemset(d, 0, 256));
    CHECK(hipHostMalloc(&h, 256, hipHostMallocDefault));
    h[32] = 0;
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    const int N = 2000;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    for (int rep = 0; rep < 3; ++rep) {
        auto t0 = now();
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(work, dim3(1), dim3(64), 0, s, d);
        CHECK(hipStreamSynchronize(s));
        auto t1 = now();
        for (int i = 0; i < N; ++i) {
            hipLaunchKernelGGL(work, dim3(1), dim3(64), 0, s, d);
            CHECK(hipMemcpyAsync(h, d, 64, hipMemcpyDeviceToHost, s));
            CHECK(hipStreamSynchronize(s));
        }
        auto t2 = now();
        volatile uint32_t* seq = h + 32;
        uint32_t v = *seq;
        for (int i = 0; i < N; ++i) {
            hipLaunchKernelGGL(work, dim3(1), dim3(64), 0, s, d);
            hipLaunchKernelGGL(publish, dim3(1), dim3(64), 0, s, (const uint32_t*)d, h, 16u, (uint32_t*)(h + 32), ++v);
            while (__atomic_load_n(h + 32, __ATOMIC_ACQUIRE) != v) {}
        }
        auto t3 = now();
        CHECK(hipStreamSynchronize(s));
        printf("back to back %.2f us per kernel; copy + synchronize %.2f us per round; publish + spin %.2f us per round (value %u)\n",
               us(t0, t1) / N, us(t1, t2) / N, us(t2, t3) / N, h[0]);
    }
    return 0;
}
