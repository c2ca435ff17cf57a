// Bitonic sort of C * 64 uint32 keys held by one wavefront as v[t] = key[t * 64 + lane],
// ascending.  "Flip" formulation: for every merge size k the first step compares element e
// with e ^ (k - 1), the half-cleaners that follow compare e with e ^ j (j = k/4 .. 1); every
// compare-exchange leaves the minimum at the lower element, so there is no direction logic.
// Partners inside a row of 16 lanes come from DPP operands (no LDS crossbar round trip):
// xor 1 / 2 / 3 = quad_perm, xor 7 = row_half_mirror, xor 15 = row_mirror, xor 8 =
// row_ror:8, xor 4 = row_shl:4 / row_shr:4 on alternate banks.  Partners in another row go
// through ds_bpermute.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rala_hip {

template <int kCtrl, int kBankMask>
__device__ __forceinline__ uint32_t dpp_move(uint32_t old, uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, kCtrl, 0xF, kBankMask, false);
}

// value of lane (lane ^ kMask), kMask < 64
template <uint32_t kMask>
__device__ __forceinline__ uint32_t lane_xor(uint32_t v, uint32_t lane) {
    if (kMask == 1) return dpp_move<0xB1, 0xF>(v, v);            // quad_perm [1,0,3,2]
    if (kMask == 2) return dpp_move<0x4E, 0xF>(v, v);            // quad_perm [2,3,0,1]
    if (kMask == 3) return dpp_move<0x1B, 0xF>(v, v);            // quad_perm [3,2,1,0]
    if (kMask == 7) return dpp_move<0x141, 0xF>(v, v);           // row_half_mirror
    if (kMask == 15) return dpp_move<0x140, 0xF>(v, v);          // row_mirror
    if (kMask == 8) return dpp_move<0x128, 0xF>(v, v);           // row_ror:8
    if (kMask == 4) {
        const uint32_t t = dpp_move<0x104, 0x5>(v, v);           // row_shl:4 -> banks 0, 2 read lane + 4
        return dpp_move<0x114, 0xA>(t, v);                       // row_shr:4 -> banks 1, 3 read lane - 4
    }
    return (uint32_t)__builtin_amdgcn_ds_bpermute((int)((lane ^ kMask) << 2), (int)v);
}

template <uint32_t kMask, uint32_t kLowerBit>
__device__ __forceinline__ uint32_t lane_cmpx(uint32_t v, uint32_t lane) {
    const uint32_t o = lane_xor<kMask>(v, lane);
    const uint32_t mn = v < o ? v : o, mx = v < o ? o : v;
    return (lane & kLowerBit) ? mx : mn;
}

template <int C, uint32_t K>
struct SortStage {
    // merge size K (elements); registers hold 64 elements each
    static __device__ __forceinline__ void run(uint32_t (&v)[C], uint32_t lane) {
        SortStage<C, K / 2>::run(v, lane);
        // flip: e ^ (K - 1)
        if (K <= 64) {
#pragma unroll
            for (int t = 0; t < C; ++t) v[t] = lane_cmpx<(K - 1) & 63, (K / 2) & 63>(v[t], lane);
        } else {
            constexpr uint32_t m = K / 64 - 1;                   // register partner: t ^ m, lanes reversed
#pragma unroll
            for (uint32_t t = 0; t < (uint32_t)C; ++t) {
                if ((t & (K / 128)) == 0) {
                    const uint32_t tp = t ^ m;
                    const uint32_t a = v[t], b = v[tp];
                    const uint32_t ra = lane_xor<63>(a, lane), rb = lane_xor<63>(b, lane);
                    v[t] = a < rb ? a : rb;
                    v[tp] = b > ra ? b : ra;
                }
            }
        }
        half_cleaners<K / 4>(v, lane);
    }
    template <uint32_t J>
    static __device__ __forceinline__ void half_cleaners(uint32_t (&v)[C], uint32_t lane) {
        if constexpr (J >= 1) {
            if constexpr (J >= 64) {
                constexpr uint32_t dt = J / 64;
#pragma unroll
                for (uint32_t t = 0; t < (uint32_t)C; ++t) {
                    if ((t & dt) == 0) {
                        const uint32_t a = v[t], b = v[t | dt];
                        v[t] = a < b ? a : b;
                        v[t | dt] = a < b ? b : a;
                    }
                }
            } else {
#pragma unroll
                for (int t = 0; t < C; ++t) v[t] = lane_cmpx<J, J>(v[t], lane);
            }
            half_cleaners<J / 2>(v, lane);
        }
    }
};

template <int C>
struct SortStage<C, 1> {
    static __device__ __forceinline__ void run(uint32_t (&)[C], uint32_t) {}
};

template <int C>
__device__ __forceinline__ void wave_sort_dpp(uint32_t (&v)[C], uint32_t lane) {
    SortStage<C, (uint32_t)C * 64u>::run(v, lane);
}

}  // namespace rala_hip
