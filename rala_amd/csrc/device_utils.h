// Wavefront (64 lanes) and workgroup primitives for gfx950.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rala_hip {

constexpr int kWave = 64;

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

struct OpAdd {
    template <class T> __device__ T operator()(T a, T b) const { return a + b; }
};
struct OpMax {
    template <class T> __device__ T operator()(T a, T b) const { return a > b ? a : b; }
};
struct OpMin {
    template <class T> __device__ T operator()(T a, T b) const { return a < b ? a : b; }
};

__device__ __forceinline__ uint32_t shfl_up_t(uint32_t v, int d) { return (uint32_t)__shfl_up((int)v, d, 64); }
__device__ __forceinline__ int32_t shfl_up_t(int32_t v, int d) { return __shfl_up(v, d, 64); }
__device__ __forceinline__ uint64_t shfl_up_t(uint64_t v, int d) {
    const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)v, d, 64);
    const uint32_t hi = (uint32_t)__shfl_up((int)(uint32_t)(v >> 32), d, 64);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint32_t shfl_xor_t(uint32_t v, int m) { return (uint32_t)__shfl_xor((int)v, m, 64); }
__device__ __forceinline__ int32_t shfl_xor_t(int32_t v, int m) { return __shfl_xor(v, m, 64); }
__device__ __forceinline__ uint64_t shfl_xor_t(uint64_t v, int m) {
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, m, 64);
    const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), m, 64);
    return ((uint64_t)hi << 32) | lo;
}

// inclusive scan across the 64 lanes of a wave
template <class T, class Op>
__device__ __forceinline__ T wave_scan_incl(T v, Op op) {
    const int l = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const T o = shfl_up_t(v, d);
        if (l >= d) v = op(v, o);
    }
    return v;
}

template <class T, class Op>
__device__ __forceinline__ T wave_reduce(T v, Op op) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = op(v, shfl_xor_t(v, m));
    return v;
}

// Exclusive scan of one value per thread over a workgroup of kBlock threads.
// `tmp` is LDS scratch of kBlock/64 + 1 elements; `total` receives the
// reduction of all values.  Contains two barriers.
template <int kBlock, class T, class Op>
__device__ __forceinline__ T block_scan_excl(T v, Op op, T identity, T* tmp, T& total) {
    constexpr int kW = kBlock / 64;
    const T incl = wave_scan_incl(v, op);
    const int l = lane_id(), w = wave_id();
    if (l == 63) tmp[w] = incl;
    __syncthreads();
    T prefix = identity;
    T tot = identity;
#pragma unroll
    for (int k = 0; k < kW; ++k) {
        const T x = tmp[k];
        if (k < w) prefix = op(prefix, x);
        tot = op(tot, x);
    }
    total = tot;
    T prev = shfl_up_t(incl, 1);
    if (l == 0) prev = identity;
    __syncthreads();
    return op(prefix, prev);
}

template <int kBlock, class T, class Op>
__device__ __forceinline__ T block_reduce(T v, Op op, T identity, T* tmp) {
    constexpr int kW = kBlock / 64;
    const T r = wave_reduce(v, op);
    if (lane_id() == 0) tmp[wave_id()] = r;
    __syncthreads();
    T tot = identity;
#pragma unroll
    for (int k = 0; k < kW; ++k) tot = op(tot, tmp[k]);
    __syncthreads();
    return tot;
}

}  // namespace rala_hip
