// Microbenchmark (round 3): what separates a 5.1 TB/s row fill from hipMemset's 6.6 TB/s?
// Sweeps a flat grid-stride fill over grid sizes / workgroup sizes / stores per thread, and the
// one-wavefront-per-row fill with streaming (nt) stores and with fewer wavefronts in flight.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int kU>
__global__ void flat_fill(uint4* dst, size_t n) {
    const uint4 v = make_uint4(1, 2, 3, 4);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (kU - 1) * stride < n; i += kU * stride) {
#pragma unroll
        for (int u = 0; u < kU; ++u) dst[i + u * stride] = v;
    }
    for (; i < n; i += stride) dst[i] = v;
}
// a workgroup owns a contiguous block of `chunk` vectors at a time
__global__ void block_fill(uint4* dst, size_t n, uint32_t chunk) {
    const uint4 v = make_uint4(1, 2, 3, 4);
    for (size_t c = (size_t)blockIdx.x * chunk; c < n; c += (size_t)gridDim.x * chunk) {
        const size_t e = c + chunk < n ? c + chunk : n;
        for (size_t i = c + threadIdx.x; i < e; i += blockDim.x) dst[i] = v;
    }
}
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int kMode>
__global__ __launch_bounds__(64) void row_fill(uint4* dst, uint32_t row_vec, uint32_t n_rows) {
    for (uint32_t r = blockIdx.x; r < n_rows; r += gridDim.x) {
        uint4* p = dst + (size_t)r * row_vec;
        const u32x4 v = {r, r, r, r};
        for (uint32_t g = threadIdx.x; g < row_vec; g += 64) {
            if (kMode == 0) ((u32x4*)p)[g] = v;
            else if (kMode == 1) __builtin_nontemporal_store(v, (u32x4*)p + g);
            else if (kMode == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p + g), "v"(v) : "memory");
            else asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p + g), "v"(v) : "memory");
        }
    }
}

int main() {
    const uint32_t n_rows = 400000, row_vec = 1250;      // 20 KB rows, 8 GB
    const size_t n = (size_t)n_rows * row_vec;
    uint4* d; hipMalloc(&d, n * 16);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto time = [&](const char* name, int x, int y, auto launch) {
        launch(); hipDeviceSynchronize();
        float best = 1e9;
        for (int i = 0; i < 4; ++i) {
            hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        printf("%-28s %6d %6d %7.3f ms  %7.1f GB/s\n", name, x, y, best, n * 16 / best / 1e6);
    };
    time("hipMemset", 0, 0, [&] { hipMemsetAsync(d, 1, n * 16, 0); });
    for (int threads : {64, 256, 1024}) {
        for (int grid : {256, 512, 1024, 2048, 4096, 16384}) {
            time("flat_fill<1> grid threads", grid, threads, [&] { hipLaunchKernelGGL(flat_fill<1>, dim3(grid), dim3(threads), 0, 0, d, n); });
        }
    }
    for (int grid : {256, 512, 1024, 2048}) {
        time("flat_fill<4> grid 256thr", grid, 256, [&] { hipLaunchKernelGGL(flat_fill<4>, dim3(grid), dim3(256), 0, 0, d, n); });
        time("flat_fill<8> grid 256thr", grid, 256, [&] { hipLaunchKernelGGL(flat_fill<8>, dim3(grid), dim3(256), 0, 0, d, n); });
    }
    for (uint32_t chunk : {256u, 1024u, 4096u, 16384u}) {
        for (int grid : {512, 2048, 8192}) {
            time("block_fill chunk*16B grid", (int)chunk, grid, [&] { hipLaunchKernelGGL(block_fill, dim3(grid), dim3(256), 0, 0, d, n, chunk); });
        }
    }
    for (int grid : {1024, 2048, 4096, 6144, 400000}) {
        time("row_fill plain grid", grid, 64, [&] { hipLaunchKernelGGL(row_fill<0>, dim3(grid), dim3(64), 0, 0, d, row_vec, n_rows); });
        time("row_fill nt grid", grid, 64, [&] { hipLaunchKernelGGL(row_fill<1>, dim3(grid), dim3(64), 0, 0, d, row_vec, n_rows); });
        time("row_fill sc0 sc1 grid", grid, 64, [&] { hipLaunchKernelGGL(row_fill<2>, dim3(grid), dim3(64), 0, 0, d, row_vec, n_rows); });
        time("row_fill sc1 grid", grid, 64, [&] { hipLaunchKernelGGL(row_fill<3>, dim3(grid), dim3(64), 0, 0, d, row_vec, n_rows); });
    }
    time("hipMemset", 0, 0, [&] { hipMemsetAsync(d, 1, n * 16, 0); });
    return 0;
}
