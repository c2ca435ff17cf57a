// Diagnostic: which XCD a workgroup runs on (HW_REG_XCC_ID) and whether workgroup-scope
// atomics on XCD-private counters add up.
//   hipcc --offload-arch=gfx950 -O3 tools/xcc_probe.hip -o tools/_bin/xcc_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xF; }

__global__ void probe(unsigned* ids, unsigned* counters, unsigned n_counters, unsigned* ranks) {
    const unsigned x = xcc_id();
    if (threadIdx.x == 0) ids[blockIdx.x] = x;
    // every thread bumps a pseudo-random counter of its XCD's private copy
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned c = (t * 2654435761u) % n_counters;
    ranks[t] = __hip_atomic_fetch_add(&counters[x * n_counters + c], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

int main() {
    const unsigned blocks = 4096, threads = 256, n_counters = 1 << 16;
    unsigned *ids, *counters, *ranks;
    hipMalloc(&ids, blocks * 4); hipMalloc(&counters, 16 * n_counters * 4); hipMalloc(&ranks, blocks * threads * 4);
    hipMemset(counters, 0, 16 * n_counters * 4);
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), 0, 0, ids, counters, n_counters, ranks);
    hipDeviceSynchronize();
    std::vector<unsigned> h(blocks), c(16 * n_counters), r(blocks * threads);
    hipMemcpy(h.data(), ids, blocks * 4, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), counters, c.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(r.data(), ranks, r.size() * 4, hipMemcpyDeviceToHost);
    unsigned hist[16] = {0};
    for (unsigned b = 0; b < blocks; ++b) hist[h[b] & 15]++;
    printf("blocks per XCC id:"); for (int i = 0; i < 16; ++i) printf(" %u", hist[i]); printf("\n");
    printf("first 24 blocks:"); for (int i = 0; i < 24; ++i) printf(" %u", h[i]); printf("\n");
    unsigned long long total = 0; for (unsigned v : c) total += v;
    // ranks must be a permutation of 0..count-1 per counter
    std::vector<unsigned> seen(16 * n_counters, 0);
    bool ok = total == (unsigned long long)blocks * threads;
    for (unsigned b = 0; b < blocks; ++b) for (unsigned t = 0; t < threads; ++t) {
        const unsigned g = b * threads + t;
        const unsigned k = h[b] * n_counters + (g * 2654435761u) % n_counters;
        if (r[g] >= c[k]) ok = false; else seen[k] += 1;
    }
    for (size_t k = 0; k < seen.size(); ++k) if (seen[k] != c[k]) ok = false;
    printf("total %llu expected %u -> %s\n", total, blocks * threads, ok ? "ok" : "MISMATCH");
    return ok ? 0 : 1;
}
