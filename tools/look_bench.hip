// Microbenchmark (round 6): what does one "look" from the host cost - a few words produced by a kernel, needed by the host before it
// launches the next kernel?  (a) hipMemcpyAsync into pinned memory + hipStreamSynchronize (what d2h_small / stream_sync do);
// (b) the producing side writes the words and then a ticket into host-mapped pinned memory, the host spins on the ticket;
// (c) as (a) with an event: hipEventRecord + hipEventSynchronize.  Each measured as the time from the launch of a 20 us kernel to the
// start of the kernel that follows the look (hipEvent timing over N rounds, minus the kernels' own time).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <chrono>

__global__ void work(uint32_t* out, uint32_t spin) {
    uint32_t x = threadIdx.x;
    for (uint32_t k = 0; k < spin; ++k) x = x * 1664525u + 1013904223u;
    if (threadIdx.x == 0) out[0] = x | 1u;
}
__global__ void to_host(const uint32_t* src, volatile uint32_t* dst, uint32_t n, volatile uint32_t* ticket, uint32_t value) {
    if (threadIdx.x < n) dst[threadIdx.x] = src[threadIdx.x];
    __threadfence_system();
    if (threadIdx.x == 0) *ticket = value;
}
int main() {
    hipStream_t s; hipStreamCreate(&s);
    uint32_t* d; hipMalloc(&d, 256);
    uint32_t* h; hipHostMalloc(&h, 4096, hipHostMallocCoherent | hipHostMallocMapped);
    uint32_t* hd = nullptr; hipHostGetDevicePointer((void**)&hd, h, 0);
    volatile uint32_t* ticket = h + 512;
    const int N = 2000;
    const uint32_t spin = 4000;
    auto run = [&](const char* name, int mode) {
        hipStreamSynchronize(s);
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; ++i) {
            hipLaunchKernelGGL(work, dim3(1), dim3(64), 0, s, d, spin);
            if (mode == 0) { hipMemcpyAsync(h, d, 16, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); }
            else if (mode == 1) {
                hipLaunchKernelGGL(to_host, dim3(1), dim3(64), 0, s, (const uint32_t*)d, (volatile uint32_t*)hd, 4u, (volatile uint32_t*)(hd + 512), (uint32_t)(i + 1));
                while (*ticket != (uint32_t)(i + 1)) { __builtin_ia32_pause(); }
            } else if (mode == 2) {
                hipStreamSynchronize(s);                    // (no copy at all: the bare wait)
            } else {
                hipLaunchKernelGGL(to_host, dim3(1), dim3(64), 0, s, (const uint32_t*)d, (volatile uint32_t*)hd, 4u, (volatile uint32_t*)(hd + 512), (uint32_t)(i + 1));
                hipStreamSynchronize(s);
            }
            if (h[0] == 0xdeadbeef) printf("!");
        }
        hipStreamSynchronize(s);
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
        printf("%-64s %7.2f us per round\n", name, us);
    };
    run("warm-up", 0);
    run("work kernel + bare hipStreamSynchronize", 2);
    run("work kernel + hipMemcpyAsync to pinned + hipStreamSynchronize", 0);
    run("work kernel + to_host kernel (mapped memory) + spin on ticket", 1);
    run("work kernel + to_host kernel + hipStreamSynchronize", 3);
    return 0;
}
