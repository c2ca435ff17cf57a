// Microbenchmark (round 3): one wavefront per 20 KB row - does it matter WHERE adjacent rows are written?
//   plain      row = workgroup index (adjacent rows go to different XCDs, round robin)
//   xcd        row = (wg % 8) * (n / 8) + wg / 8: every XCD writes one contiguous range of rows
//   wgK        workgroups of K wavefronts, wavefront w of workgroup g writes row K * g + w
//   wgK + xcd  both
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int kWaves, bool kXcd>
__global__ __launch_bounds__(64 * kWaves) void row_fill(uint4* dst, uint32_t row_vec, uint32_t n_rows) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t g = blockIdx.x;
    if (kXcd) {
        const uint32_t per = gridDim.x / 8;                 // (grid is a multiple of 8)
        g = (g % 8) * per + g / 8;
    }
    const uint32_t r = g * kWaves + wave;
    if (r >= n_rows) return;
    uint4* p = dst + (size_t)r * row_vec;
    const uint4 v = make_uint4(r, r, r, r);
    for (uint32_t k = lane; k < row_vec; k += 64) p[k] = v;
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
        printf("%-28s %7.3f ms  %7.1f GB/s\n", name, best, n * 16 / best / 1e6);
    };
    time("hipMemset", [&] { hipMemsetAsync(d, 1, n * 16, 0); });
#define RUN(K, X, name) time(name, [&] { hipLaunchKernelGGL((row_fill<K, X>), dim3(n_rows / K), dim3(64 * K), 0, 0, d, row_vec, n_rows); })
    RUN(1, false, "plain");
    RUN(1, true, "xcd");
    RUN(2, false, "wg2");
    RUN(2, true, "wg2 xcd");
    RUN(4, false, "wg4");
    RUN(4, true, "wg4 xcd");
    RUN(8, false, "wg8");
    RUN(8, true, "wg8 xcd");
    RUN(16, false, "wg16");
    RUN(16, true, "wg16 xcd");
    time("hipMemset", [&] { hipMemsetAsync(d, 1, n * 16, 0); });
    return 0;
}
