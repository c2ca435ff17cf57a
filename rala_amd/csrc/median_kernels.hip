// Median of the pile medians per connected component (reference: rvaser/rala
// src/graph.cpp:777-783, std::nth_element at size / 2 over the component's reads).
// Keys (component label << 16 | pile median) of the reads that carry an overlap are sorted
// with one device radix sort (rocPRIM); a read then finds its component's segment with two
// binary searches and takes the element at segment size / 2.
#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "kernels.h"

namespace rala_hip {

namespace {

constexpr int kBlock = 256;
constexpr unsigned kKeyBits = 48;      // 16 bits of median + 32 bits of label

__global__ __launch_bounds__(kBlock) void median_keys_kernel(const uint32_t* __restrict__ label,
                                                             const uint8_t* __restrict__ touched,
                                                             const uint32_t* __restrict__ reads,
                                                             const uint16_t* __restrict__ median, uint32_t n,
                                                             uint64_t* __restrict__ keys) {
    const uint32_t q = blockIdx.x * kBlock + threadIdx.x;
    if (q >= n) return;
    keys[q] = touched[q] ? ((uint64_t)label[q] << 16) | median[reads[q]] : ~0ull;
}

__device__ __forceinline__ uint32_t lower_bound_u64(const uint64_t* __restrict__ a, uint32_t n, uint64_t x) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (a[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(kBlock) void median_pick_kernel(const uint32_t* __restrict__ label,
                                                             const uint8_t* __restrict__ touched,
                                                             const uint64_t* __restrict__ sorted, uint32_t n,
                                                             uint16_t* __restrict__ cmed) {
    const uint32_t q = blockIdx.x * kBlock + threadIdx.x;
    if (q >= n || !touched[q]) return;
    const uint64_t lab = label[q];
    const uint32_t lo = lower_bound_u64(sorted, n, lab << 16);
    const uint32_t hi = lower_bound_u64(sorted, n, (lab + 1) << 16);
    cmed[q] = (uint16_t)(sorted[lo + (hi - lo) / 2] & 0xFFFFu);
}

}  // namespace

size_t component_median_workspace(uint32_t n) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_keys(nullptr, bytes, (const uint64_t*)nullptr, (uint64_t*)nullptr, (size_t)n, 0u, kKeyBits,
                                   (hipStream_t)0);
    return bytes + 256;
}

hipError_t launch_component_medians(const uint32_t* label, const uint8_t* touched, const uint32_t* alive_reads,
                                    const uint16_t* median, uint32_t n_alive, uint64_t* keys, uint64_t* sorted, void* tmp,
                                    size_t tmp_bytes, uint16_t* cmed, hipStream_t s, bool keys_ready) {
    if (n_alive == 0) return hipSuccess;
    const dim3 grid((n_alive + kBlock - 1) / kBlock);
    if (!keys_ready) hipLaunchKernelGGL(median_keys_kernel, grid, dim3(kBlock), 0, s, label, touched, alive_reads, median, n_alive, keys);
    const hipError_t e = rocprim::radix_sort_keys(tmp, tmp_bytes, (const uint64_t*)keys, sorted, (size_t)n_alive, 0u,
                                                  kKeyBits, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(median_pick_kernel, grid, dim3(kBlock), 0, s, label, touched, (const uint64_t*)sorted, n_alive,
                       cmed);
    return hipSuccess;
}

}  // namespace rala_hip
