// Microbenchmark (round 4): what store shape writes 20 KB rows fastest?  One wavefront per row in all variants but the last two.
//   x4 / x2 / x1: 16, 8, 4 bytes per lane and instruction (1 KiB, 512 B, 256 B per wave-instruction)
//   nt: non-temporal;  half: two wavefronts per row (a workgroup of 128), each one half;  rows2: a wavefront writes two
//   neighbouring rows alternately
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
template <class T, bool kNt>
__global__ __launch_bounds__(64) void row_fill(T* dst, uint32_t row_elems, uint32_t n_rows, T v) {
    const uint32_t r = blockIdx.x;
    T* p = dst + (size_t)r * row_elems;
    for (uint32_t g = threadIdx.x; g < row_elems; g += 64) {
        if (kNt) __builtin_nontemporal_store(v, p + g); else p[g] = v;
    }
}
__global__ __launch_bounds__(128) void row_fill_half(uint4* dst, uint32_t row_vec, uint32_t n_rows) {
    const uint32_t r = blockIdx.x, w = threadIdx.x >> 6, l = threadIdx.x & 63;
    uint4* p = dst + (size_t)r * row_vec;
    const uint32_t h = (row_vec + 1) / 2;
    const uint4 v = make_uint4(r, r, r, r);
    for (uint32_t g = w * h + l; g < min(row_vec, (w + 1) * h); g += 64) p[g] = v;
}
__global__ __launch_bounds__(128) void two_rows_per_group(uint4* dst, uint32_t row_vec, uint32_t n_rows) {
    const uint32_t r = blockIdx.x * 2 + (threadIdx.x >> 6), l = threadIdx.x & 63;
    if (r >= n_rows) return;
    uint4* p = dst + (size_t)r * row_vec;
    const uint4 v = make_uint4(r, r, r, r);
    for (uint32_t g = l; g < row_vec; g += 64) p[g] = v;
}
__global__ __launch_bounds__(256) void four_rows_per_group(uint4* dst, uint32_t row_vec, uint32_t n_rows) {
    const uint32_t r = blockIdx.x * 4 + (threadIdx.x >> 6), l = threadIdx.x & 63;
    if (r >= n_rows) return;
    uint4* p = dst + (size_t)r * row_vec;
    const uint4 v = make_uint4(r, r, r, r);
    for (uint32_t g = l; g < row_vec; g += 64) p[g] = v;
}
// with arithmetic between the stores (a wave that computes what it stores): `work` dependent multiply-adds per store
__global__ __launch_bounds__(64) void row_fill_work(uint4* dst, uint32_t row_vec, uint32_t n_rows, uint32_t work) {
    const uint32_t r = blockIdx.x;
    uint4* p = dst + (size_t)r * row_vec;
    uint32_t x = r + threadIdx.x;
    for (uint32_t g = threadIdx.x; g < row_vec; g += 64) {
        for (uint32_t k = 0; k < work; ++k) x = x * 1664525u + 1013904223u;
        p[g] = make_uint4(x, x, x, x);
    }
}

int main() {
    const uint32_t n_rows = 400000, row_vec = 1250;      // 20 KB rows, 8 GB
    const size_t n = (size_t)n_rows * row_vec;
    uint4* d; hipMalloc(&d, n * 16);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto time = [&](const char* name, auto launch) {
        launch(); hipDeviceSynchronize();
        float best = 1e9;
        for (int i = 0; i < 4; ++i) {
            hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        printf("%-44s %7.3f ms  %7.1f GB/s\n", name, best, n * 16 / best / 1e6);
    };
    time("x4 (16 B per lane)", [&] { hipLaunchKernelGGL((row_fill<u32x4, false>), dim3(n_rows), dim3(64), 0, 0, (u32x4*)d, row_vec, n_rows, u32x4{1, 2, 3, 4}); });
    time("x2 (8 B per lane)", [&] { hipLaunchKernelGGL((row_fill<u32x2, false>), dim3(n_rows), dim3(64), 0, 0, (u32x2*)d, row_vec * 2, n_rows, u32x2{1, 2}); });
    time("x1 (4 B per lane)", [&] { hipLaunchKernelGGL((row_fill<uint32_t, false>), dim3(n_rows), dim3(64), 0, 0, (uint32_t*)d, row_vec * 4, n_rows, 7u); });
    time("x4 non-temporal", [&] { hipLaunchKernelGGL((row_fill<u32x4, true>), dim3(n_rows), dim3(64), 0, 0, (u32x4*)d, row_vec, n_rows, u32x4{1, 2, 3, 4}); });
    time("x1 non-temporal", [&] { hipLaunchKernelGGL((row_fill<uint32_t, true>), dim3(n_rows), dim3(64), 0, 0, (uint32_t*)d, row_vec * 4, n_rows, 7u); });
    time("two wavefronts per row", [&] { hipLaunchKernelGGL(row_fill_half, dim3(n_rows), dim3(128), 0, 0, d, row_vec, n_rows); });
    time("two rows per workgroup (2 waves)", [&] { hipLaunchKernelGGL(two_rows_per_group, dim3((n_rows + 1) / 2), dim3(128), 0, 0, d, row_vec, n_rows); });
    time("four rows per workgroup (4 waves)", [&] { hipLaunchKernelGGL(four_rows_per_group, dim3((n_rows + 3) / 4), dim3(256), 0, 0, d, row_vec, n_rows); });
    for (uint32_t work : {8u, 32u, 64u, 128u}) {
        char name[64]; snprintf(name, sizeof name, "x4 with %u multiply-adds per store", work);
        time(name, [&] { hipLaunchKernelGGL(row_fill_work, dim3(n_rows), dim3(64), 0, 0, d, row_vec, n_rows, work); });
    }
    time("hipMemset", [&] { hipMemsetAsync(d, 1, n * 16, 0); });
    return 0;
}
