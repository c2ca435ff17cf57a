// Microbenchmark: HBM write ceiling for "one wavefront streams one 20 KB row" vs a flat fill.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

__global__ __launch_bounds__(64) void row_fill(uint4* dst, uint32_t row_vec, uint32_t n_rows) {
    for (uint32_t r = blockIdx.x; r < n_rows; r += gridDim.x) {
        uint4* p = dst + (size_t)r * row_vec;
        const uint4 v = make_uint4(r, r, r, r);
        for (uint32_t g = threadIdx.x; g < row_vec; g += 64) p[g] = v;
    }
}
// each lane owns 4 consecutive 16-byte vectors (64 contiguous bytes): four store instructions
// whose lanes are 64 bytes apart
__global__ __launch_bounds__(64) void row_fill_lane64(uint4* dst, uint32_t row_vec, uint32_t n_rows) {
    for (uint32_t r = blockIdx.x; r < n_rows; r += gridDim.x) {
        uint4* p = dst + (size_t)r * row_vec;
        const uint4 v = make_uint4(r, r, r, r);
        for (uint32_t g = threadIdx.x * 4; g < row_vec; g += 256) {
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) if (g + u < row_vec) p[g + u] = v;
        }
    }
}
// each lane owns 2 consecutive 16-byte vectors (32 contiguous bytes): two store instructions whose lanes are 32 bytes apart
__global__ __launch_bounds__(64) void row_fill_lane32(uint4* dst, uint32_t row_vec, uint32_t n_rows) {
    for (uint32_t r = blockIdx.x; r < n_rows; r += gridDim.x) {
        uint4* p = dst + (size_t)r * row_vec;
        const uint4 v = make_uint4(r, r, r, r);
        for (uint32_t g = threadIdx.x * 2; g < row_vec; g += 128) {
#pragma unroll
            for (uint32_t u = 0; u < 2; ++u) if (g + u < row_vec) p[g + u] = v;
        }
    }
}
__global__ __launch_bounds__(256) void row_fill256(uint4* dst, uint32_t row_vec, uint32_t n_rows) {
    for (uint32_t r = blockIdx.x; r < n_rows; r += gridDim.x) {
        uint4* p = dst + (size_t)r * row_vec;
        const uint4 v = make_uint4(r, r, r, r);
        for (uint32_t g = threadIdx.x; g < row_vec; g += 256) p[g] = v;
    }
}
__global__ __launch_bounds__(256) void flat_fill(uint4* dst, size_t n) {
    const uint4 v = make_uint4(1, 2, 3, 4);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = v;
}

int main() {
    const uint32_t n_rows = 100000, row_vec = 1250;      // 20 KB rows, 2 GB
    const size_t n = (size_t)n_rows * row_vec;
    uint4* d; hipMalloc(&d, n * 16);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto time = [&](const char* name, auto launch) {
        launch(); hipDeviceSynchronize();
        float best = 1e9;
        for (int i = 0; i < 5; ++i) {
            hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        printf("%-34s %7.3f ms  %7.1f GB/s\n", name, best, n * 16 / best / 1e6);
    };
    time("row_fill wave/row grid=n_rows", [&] { hipLaunchKernelGGL(row_fill, dim3(n_rows), dim3(64), 0, 0, d, row_vec, n_rows); });
    time("lane owns 64 B, grid=n_rows", [&] { hipLaunchKernelGGL(row_fill_lane64, dim3(n_rows), dim3(64), 0, 0, d, row_vec, n_rows); });
    time("lane owns 32 B, grid=n_rows", [&] { hipLaunchKernelGGL(row_fill_lane32, dim3(n_rows), dim3(64), 0, 0, d, row_vec, n_rows); });
    time("row_fill wave/row grid=8192", [&] { hipLaunchKernelGGL(row_fill, dim3(8192), dim3(64), 0, 0, d, row_vec, n_rows); });
    time("row_fill wave/row grid=2304", [&] { hipLaunchKernelGGL(row_fill, dim3(2304), dim3(64), 0, 0, d, row_vec, n_rows); });
    time("row_fill 256thr/row grid=n_rows", [&] { hipLaunchKernelGGL(row_fill256, dim3(n_rows), dim3(256), 0, 0, d, row_vec, n_rows); });
    time("row_fill 256thr/row grid=2048", [&] { hipLaunchKernelGGL(row_fill256, dim3(2048), dim3(256), 0, 0, d, row_vec, n_rows); });
    time("flat_fill grid=2048", [&] { hipLaunchKernelGGL(flat_fill, dim3(2048), dim3(256), 0, 0, d, n); });
    time("flat_fill grid=8192", [&] { hipLaunchKernelGGL(flat_fill, dim3(8192), dim3(256), 0, 0, d, n); });
    time("hipMemset", [&] { hipMemsetAsync(d, 1, n * 16, 0); });
    return 0;
}
