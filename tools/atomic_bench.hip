// Diagnostic: throughput of random 4-byte atomic adds, device scope on one counter array vs
// workgroup scope on XCD-private copies (see tools/xcc_probe.hip), for several array sizes.
//   hipcc --offload-arch=gfx950 -O3 tools/atomic_bench.hip -o tools/_bin/atomic_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7; }

template <int kMode>
__global__ void bench(const unsigned* __restrict__ idx, unsigned long long n, unsigned* counters, unsigned n_counters,
                      unsigned* __restrict__ ranks) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned c = idx[i] % n_counters;
    if (kMode == 0) {
        ranks[i] = atomicAdd(&counters[c], 2u);
    } else {
        unsigned* mine = counters + (size_t)xcc_id() * n_counters;
        ranks[i] = __hip_atomic_fetch_add(&mine[c], 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

int main() {
    const unsigned long long n = 50000000ull;
    std::vector<unsigned> h(n);
    unsigned long long s = 88172645463325252ull;
    for (auto& x : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x = (unsigned)(s >> 11); }
    unsigned *idx, *counters, *ranks;
    hipMalloc(&idx, n * 4); hipMalloc(&ranks, n * 4); hipMalloc(&counters, 8ull * (1 << 20) * 4 + 64);
    hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (unsigned nc : {1u << 20, 1u << 18, 1u << 16}) {
        for (int mode = 0; mode < 2; ++mode) {
            float best = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                hipMemset(counters, 0, 8ull * nc * 4);
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(bench<0>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, idx, n, counters, nc, ranks);
                else hipLaunchKernelGGL(bench<1>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, idx, n, counters, nc, ranks);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("counters %8u  %s  %.3f ms  %.1f G atomics/s\n", nc, mode ? "workgroup scope, XCD-private" : "device scope              ",
                   best, n / best / 1e6);
        }
    }
    return 0;
}
