// Diagnostic: the b-side of fixed-slot bucketing in isolation - a random atomic add on counts[b]
// that hands out the position, then an 8-byte store into slot b - for several slot strides and
// read counts, with the atomic alone and the store alone beside it.
//   hipcc --offload-arch=gfx950 -O3 tools/scatter_bench.hip -o tools/_bin/scatter_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__device__ __forceinline__ uint32_t mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return (uint32_t)x;
}

template <int kMode>   // 0 atomic + store, 1 atomic only, 2 store only
__global__ __launch_bounds__(256) void scatter(uint64_t n, uint32_t n_reads, uint32_t stride, uint32_t* counts,
                                               uint32_t* ev, uint32_t* sink) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t b = mix(i) % n_reads;
    uint32_t p;
    if (kMode == 2) p = (mix(i * 7 + 1) % 100u) * 2u;
    else p = atomicAdd(&counts[b], 2u);
    if (kMode == 1) { if (p == 0xFFFFFFFFu) *sink = 1; return; }
    if (p + 2u <= stride) *(uint2*)(ev + (size_t)b * stride + p) = make_uint2((uint32_t)i, p);
}

int main() {
    const uint64_t n = 50000000ull;
    uint32_t *counts, *ev, *sink;
    const size_t max_ev = (size_t)4000000 * 2048;
    hipMalloc(&counts, 4000000 * 4); hipMalloc(&ev, max_ev * 4); hipMalloc(&sink, 4);
    hipMemset(ev, 0, max_ev * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const uint32_t reads[2] = {1000000, 4000000};
    const uint32_t strides[4] = {256, 512, 1024, 2048};
    for (uint32_t nr : reads) for (uint32_t st : strides) for (int mode = 0; mode < 3; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(counts, 0, (size_t)nr * 4);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            const uint32_t grid = (uint32_t)((n + 255) / 256);
            if (mode == 0) scatter<0><<<grid, 256>>>(n, nr, st, counts, ev, sink);
            else if (mode == 1) scatter<1><<<grid, 256>>>(n, nr, st, counts, ev, sink);
            else scatter<2><<<grid, 256>>>(n, nr, st, counts, ev, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("reads %8u stride %5u (%5.1f GB) mode %s: %7.3f ms  %6.1f G/s\n", nr, st, (double)nr * st * 4 / 1e9,
               mode == 0 ? "atomic+store" : mode == 1 ? "atomic only " : "store only  ", best, n / best / 1e6);
    }
    return 0;
}
