// Wavefront (64 lanes) and workgroup primitives for gfx950.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <limits>

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

// value of lane - 1 (lane 0 keeps its own): wave_shr:1, no LDS round trip
__device__ __forceinline__ uint32_t lane_above(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x138, 0xF, 0xF, false);
}
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

// ---- wave-wide scan / reduction through DPP operands ------------------------------------------
// A ds_bpermute is an LDS-crossbar round trip that the wave has to wait for (about 60 cycles per
// step, six steps per scan); a DPP operand reads the neighbour lane inside the vector ALU.  The
// sequence is the one LLVM emits for gfx9 wave64 scans: row_shr 1, 2, 4, 8 inside rows of 16
// lanes, then row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2 and 3.  Lanes whose
// source does not exist keep `identity`, so op(v, identity) leaves them unchanged.
template <int kCtrl, int kRowMask>
__device__ __forceinline__ uint32_t dpp_from(uint32_t identity, uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)v, kCtrl, kRowMask, 0xF, false);
}
template <int kCtrl, int kRowMask>
__device__ __forceinline__ int32_t dpp_from(int32_t identity, int32_t v) {
    return __builtin_amdgcn_update_dpp(identity, v, kCtrl, kRowMask, 0xF, false);
}
template <int kCtrl, int kRowMask>
__device__ __forceinline__ uint64_t dpp_from(uint64_t identity, uint64_t v) {
    const uint32_t lo = dpp_from<kCtrl, kRowMask>((uint32_t)identity, (uint32_t)v);
    const uint32_t hi = dpp_from<kCtrl, kRowMask>((uint32_t)(identity >> 32), (uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

template <class T> __device__ __forceinline__ constexpr T op_identity(OpAdd) { return (T)0; }
template <class T> __device__ __forceinline__ constexpr T op_identity(OpMax) { return std::numeric_limits<T>::lowest(); }
template <class T> __device__ __forceinline__ constexpr T op_identity(OpMin) { return std::numeric_limits<T>::max(); }

// inclusive scan across the 64 lanes of a wave
template <class T, class Op>
__device__ __forceinline__ T wave_scan_incl(T v, Op op) {
    const T id = op_identity<T>(op);
    v = op(v, dpp_from<0x111, 0xF>(id, v));        // row_shr:1
    v = op(v, dpp_from<0x112, 0xF>(id, v));        // row_shr:2
    v = op(v, dpp_from<0x114, 0xF>(id, v));        // row_shr:4
    v = op(v, dpp_from<0x118, 0xF>(id, v));        // row_shr:8
    v = op(v, dpp_from<0x142, 0xA>(id, v));        // row_bcast:15 -> rows 1, 3
    v = op(v, dpp_from<0x143, 0xC>(id, v));        // row_bcast:31 -> rows 2, 3
    return v;
}

__device__ __forceinline__ uint32_t read_lane63(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)v, 63); }
__device__ __forceinline__ int32_t read_lane63(int32_t v) { return __builtin_amdgcn_readlane(v, 63); }
__device__ __forceinline__ uint64_t read_lane63(uint64_t v) {
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), 63) << 32) |
           (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, 63);
}

// reduction over the wave, result in every lane (a scalar register)
template <class T, class Op>
__device__ __forceinline__ T wave_reduce(T v, Op op) {
    return read_lane63(wave_scan_incl(v, op));
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
