// How many cycles does a wave64 integer vector instruction occupy a SIMD on gfx950?  w waves per
// SIMD each run a chain of dependent v_add_u32 / v_and_b32 / v_bfe_u32; the clock comes from
// s_memtime inside the kernel, so the answer does not depend on the clock rate.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/valu_bench tools/valu_bench.hip && tools/_bin/valu_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

constexpr int kChain = 4096;      // dependent instructions per loop body repetition (unrolled by 16)

__global__ void chain_kernel(uint32_t* out, uint64_t* cycles, uint32_t seed) {
    uint32_t a = threadIdx.x + seed, b = seed | 1u;
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < kChain / 16; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            asm volatile("v_add_u32 %0, %0, %1\n v_and_b32 %0, %0, %2\n v_bfe_u32 %0, %0, 0, 31\n v_xor_b32 %0, %0, %1"
                         : "+v"(a) : "v"(b), "v"(0x7FFFFFFFu));
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
    if (threadIdx.x % 64 == 0) cycles[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0;
}

int main() {
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    uint32_t* out;
    uint64_t* cyc;
    hipMalloc(&out, 64u << 20);
    hipMalloc(&cyc, 8u << 20);
    for (int waves_per_simd : {1, 2, 3, 4, 5, 8}) {
        const int threads = 64 * 4 * waves_per_simd;          // one workgroup per CU, 4 SIMDs
        if (threads > 1024) {
            // two workgroups per CU instead
        }
        const int wg_threads = threads > 1024 ? threads / 2 : threads;
        const int wgs = cus * (threads > 1024 ? 2 : 1);
        hipLaunchKernelGGL(chain_kernel, dim3(wgs), dim3(wg_threads), 0, 0, out, cyc, 1u);
        hipDeviceSynchronize();
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(chain_kernel, dim3(wgs), dim3(wg_threads), 0, 0, out, cyc, 3u);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        uint64_t h[64];
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double avg = 0;
        for (int k = 0; k < 16; ++k) avg += (double)h[k];
        avg /= 16;
        // s_memtime counts at a fixed 100 MHz on recent parts: report wall time per instruction instead
        const double ns_per_instr_per_wave = 1e6 * ms / kChain;
        printf("waves/SIMD %d: kernel %.3f ms, %.2f ns per instruction of one wave, %.2f ns per instruction per SIMD "
               "(at 2.4 GHz: %.1f cycles), counter delta %.0f\n",
               waves_per_simd, ms, ns_per_instr_per_wave, ns_per_instr_per_wave / waves_per_simd,
               2.4 * ns_per_instr_per_wave / waves_per_simd, avg);
    }
    return 0;
}
