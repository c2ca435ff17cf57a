// Diagnostic: checks rala_amd/csrc/wave_sort.h against std::sort on random keys.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I rala_amd/csrc tools/sort_probe.hip -o /tmp/sort_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <random>
#include <vector>

#include "wave_sort.h"

template <int C>
__global__ __launch_bounds__(64) void sort_kernel(uint32_t* keys) {
    const uint32_t lane = threadIdx.x;
    uint32_t* base = keys + (size_t)blockIdx.x * C * 64;
    uint32_t v[C];
#pragma unroll
    for (int t = 0; t < C; ++t) v[t] = base[t * 64 + lane];
    rala_hip::wave_sort_dpp<C>(v, lane);
#pragma unroll
    for (int t = 0; t < C; ++t) base[t * 64 + lane] = v[t];
}

template <uint32_t M>
__global__ void map_kernel(uint32_t* out) { out[threadIdx.x] = rala_hip::lane_xor<M>(threadIdx.x, threadIdx.x); }

template <uint32_t M>
int check_map() {
    uint32_t* d; uint32_t h[64];
    hipMalloc(&d, 256);
    hipLaunchKernelGGL(map_kernel<M>, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    hipFree(d);
    int bad = 0;
    for (uint32_t i = 0; i < 64; ++i) bad += h[i] != (i ^ M);
    if (bad) { printf("lane_xor<%u> wrong:", M); for (int i = 0; i < 64; ++i) printf(" %u", h[i]); printf("\n"); }
    return bad;
}

template <int C>
int check_sort() {
    const int blocks = 1000;
    std::vector<uint32_t> h((size_t)blocks * C * 64), want;
    std::mt19937 rng(C);
    for (auto& x : h) x = rng() % 5000u;
    want = h;
    for (int b = 0; b < blocks; ++b) std::sort(want.begin() + (size_t)b * C * 64, want.begin() + (size_t)(b + 1) * C * 64);
    uint32_t* d;
    hipMalloc(&d, h.size() * 4);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(sort_kernel<C>, dim3(blocks), dim3(64), 0, 0, d);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    hipFree(d);
    const bool ok = h == want;
    printf("sort C=%d: %s\n", C, ok ? "ok" : "MISMATCH");
    return ok ? 0 : 1;
}

int main() {
    int bad = 0;
    bad += check_map<1>(); bad += check_map<2>(); bad += check_map<3>(); bad += check_map<4>(); bad += check_map<7>();
    bad += check_map<8>(); bad += check_map<15>(); bad += check_map<16>(); bad += check_map<31>(); bad += check_map<32>();
    bad += check_map<63>();
    bad += check_sort<1>(); bad += check_sort<2>(); bad += check_sort<4>(); bad += check_sort<8>();
    printf(bad ? "FAILED\n" : "all ok\n");
    return bad ? 1 : 0;
}
