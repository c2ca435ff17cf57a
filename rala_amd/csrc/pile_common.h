// Pieces shared by the position-space pile kernels (pile_kernels.hip,
// pile_repeats_kernel.hip): packed uint16 max, slope-region lists and the
// reference's region resolution / narrowing (pile.cpp:131-256) over an LDS or
// HBM image of the pile.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "geom.h"

namespace rala_hip {
namespace {

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b) {
    u16x2 x = __builtin_bit_cast(u16x2, a), y = __builtin_bit_cast(u16x2, b);
    u16x2 r = __builtin_elementwise_max(x, y);
    return __builtin_bit_cast(uint32_t, r);
}

// ---- slope regions (serial, one lane) --------------------------------------
struct RegionList {
    uint32_t* key;    // first << 1 | is_up
    uint32_t* last;
    uint32_t n;
    uint32_t cap;
    bool overflow;
};

__device__ __forceinline__ void rl_push(RegionList& R, uint32_t key, uint32_t last) {
    if (R.n >= R.cap) { R.overflow = true; return; }
    R.key[R.n] = key;
    R.last[R.n] = last;
    ++R.n;
}

// insertion sort by (key, last); lists are short and nearly sorted
__device__ void rl_sort(RegionList& R) {
    for (uint32_t i = 1; i < R.n; ++i) {
        const uint32_t k = R.key[i], l = R.last[i];
        uint32_t j = i;
        while (j > 0 && (R.key[j - 1] > k || (R.key[j - 1] == k && R.last[j - 1] > l))) {
            R.key[j] = R.key[j - 1];
            R.last[j] = R.last[j - 1];
            --j;
        }
        R.key[j] = k;
        R.last[j] = l;
    }
}

// pile.cpp:131-256: resolve overlapping neighbours, then narrow (up, down) pairs.
// d[j] is the coverage at read position j.
template <class DataPtr>
__device__ void resolve_and_narrow(RegionList& R, DataPtr d, double q) {
    if (R.n == 0) return;
    for (;;) {
        rl_sort(R);
        bool changed = false;
        for (uint32_t i = 0; i + 1 < R.n; ++i) {
            if (R.last[i] < (R.key[i + 1] >> 1)) continue;
            if (R.key[i] & 1) {
                const uint32_t s = R.key[i] >> 1;
                const uint32_t e = umin(R.last[i], R.last[i + 1]);
                // flag j in [s, e) with d[j]*q < max d(j, e]; descending sweep
                int32_t m = d[e];
                bool open = false;
                uint32_t lo = 0, hi = 0;
                for (uint32_t j = e; j-- > s;) {
                    const uint32_t v = d[j];
                    if ((double)v * q < (double)m) {
                        if (open && j + 1 == lo) {
                            lo = j;
                        } else {
                            if (open) rl_push(R, lo << 1 | 1, hi);
                            open = true;
                            lo = hi = j;
                        }
                    }
                    m = max(m, (int32_t)v);
                }
                if (open) rl_push(R, lo << 1 | 1, hi);
                R.key[i] = e << 1 | 1;
            } else {
                if (R.last[i] == (R.key[i + 1] >> 1)) continue;
                const uint32_t s = umax(R.key[i] >> 1, R.key[i + 1] >> 1);
                const uint32_t e = R.last[i];
                int32_t m = -1;
                bool open = false;
                uint32_t lo = 0, hi = 0;
                for (uint32_t j = s; j <= e; ++j) {
                    const uint32_t v = d[j];
                    if (m >= 0 && (double)v * q < (double)m) {
                        if (open && j == hi + 1) {
                            hi = j;
                        } else {
                            if (open) rl_push(R, lo << 1, hi);
                            open = true;
                            lo = hi = j;
                        }
                    }
                    m = max(m, (int32_t)v);
                }
                if (open) rl_push(R, lo << 1, hi);
                R.last[i] = s;
            }
            changed = true;
            break;
        }
        if (!changed || R.overflow) break;
    }
    for (uint32_t i = 0; i + 1 < R.n; ++i) {
        if (!(R.key[i] & 1) || (R.key[i + 1] & 1)) continue;
        const uint32_t b = R.last[i];
        const uint32_t e = R.key[i + 1] >> 1;
        if ((uint32_t)(e - b) > kSlopeWindow) continue;
        uint32_t m = 0;
        for (uint32_t j = b + 1; j < e; ++j) m = umax(m, d[j]);
        const uint32_t u_first = R.key[i] >> 1;
        uint32_t last_ok = u_first;
        for (uint32_t j = u_first; j <= b; ++j) {
            if ((double)m > (double)d[j] * q) last_ok = j;
        }
        uint32_t first_ok = R.last[i + 1];
        for (uint32_t j = e; j <= R.last[i + 1]; ++j) {
            if ((double)m > (double)d[j] * q) { first_ok = j; break; }
        }
        R.last[i] = last_ok;
        R.key[i + 1] = first_ok << 1;
    }
}

struct PadView {
    const uint16_t* p;   // already offset by kPadL
    __device__ uint32_t operator[](uint32_t j) const { return p[j]; }
};

// Where the slope-region lists and the raw intervals of one read live: in LDS at the sizes nearly every pile
// needs, or - for the reads that outgrow those (the reference keeps them in vectors, pile.cpp:66, 359, 448) - in a
// per-workgroup stretch of global memory that the host grows until the read fits.
struct ListSpace {
    uint32_t* rfirst;     // 4 lists x cap_reg: first positions of the flag runs (down / up per threshold)
    uint32_t* rlast;      // 4 lists x cap_reg
    uint32_t* reg;        // 2 thresholds x (key, last) x cap_list: the lists resolve_and_narrow works on
    uint32_t* iv;         // 2 kinds x (in first, in second, out first, out second) x cap_raw
    uint8_t* gone;        // 2 kinds x cap_raw
    uint32_t cap_reg, cap_list, cap_raw;
    static __host__ __device__ constexpr uint64_t words(uint32_t cap_reg, uint32_t cap_list, uint32_t cap_raw) {
        return 8ull * cap_reg + 4ull * cap_list + 8ull * cap_raw + (2ull * cap_raw + 3) / 4;
    }
    __device__ void carve(uint32_t* base, uint32_t cr, uint32_t cl, uint32_t cw) {
        cap_reg = cr; cap_list = cl; cap_raw = cw;
        rfirst = base;
        rlast = rfirst + 4 * (size_t)cr;
        reg = rlast + 4 * (size_t)cr;
        iv = reg + 4 * (size_t)cl;
        gone = (uint8_t*)(iv + 8 * (size_t)cw);
    }
};

// One item per flagged lane appended in lane order (all lanes of the wavefront call); items beyond cap are
// counted but not stored.  Returns the new count.
__device__ __forceinline__ uint32_t wave_append2(bool flag, uint32_t count, uint32_t cap, uint32_t* a, uint32_t va,
                                                 uint32_t* b, uint32_t vb) {
    const uint64_t m = __ballot(flag);
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t at = count + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    if (flag && at < cap) { a[at] = va; b[at] = vb; }
    return count + (uint32_t)__popcll(m);
}

// intervalMerge (pile.cpp:31-52, geom.h: interval_merge) by a whole wavefront: i in input order; the sweep over
// j takes 64 intervals at a time - the first one that overlaps the growing i is absorbed, the ones in front of
// it have been compared with i as it stood when their turn came, the ones behind it are compared again.  The
// same result as the serial sweep, element for element.  Called by all lanes; returns the new count.
__device__ uint32_t interval_merge_wave(uint32_t* first, uint32_t* second, uint32_t n, uint8_t* gone, uint32_t* out_first,
                                        uint32_t* out_second) {
    const uint32_t lane = threadIdx.x & 63;
    for (uint32_t j = lane; j < n; j += 64) gone[j] = 0;
    __threadfence_block();
    uint32_t m = 0;
    for (uint32_t i = 0; i < n; ++i) {
        if (gone[i]) continue;
        uint32_t F = first[i], S = second[i];
        for (uint32_t base = 0; base < n; base += 64) {
            const uint32_t j = base + lane;
            bool open = j < n && j != i && gone[j < n ? j : 0] == 0;
            const uint32_t fj = j < n ? first[j] : 0u, sj = j < n ? second[j] : 0u;
            for (;;) {
                const uint64_t hit = __ballot(open && F < sj && S > fj);
                if (!hit) break;
                const int l0 = __ffsll((unsigned long long)hit) - 1;
                F = umin(F, (uint32_t)__shfl((int)fj, l0, 64));
                S = umax(S, (uint32_t)__shfl((int)sj, l0, 64));
                if ((int)lane == l0) gone[j] = 1;
                if ((int)lane <= l0) open = false;
            }
        }
        if (lane == 0) {
            first[i] = F; second[i] = S;
            out_first[m] = F; out_second[m] = S;
        }
        ++m;
        __threadfence_block();
    }
    return m;
}


}  // namespace
// hist[bin] += 1 for every active lane, one LDS atomic per distinct bin among the lanes of a
// wavefront: neighbouring positions of a pile mostly carry the same value, and 64 atomics on one
// address would be served one after the other.  Called by all lanes of the wavefront.
__device__ __forceinline__ void hist_add(uint32_t* hist, uint32_t bin, bool active) {
    const int lane = (int)(threadIdx.x & 63);
    uint64_t todo = __ballot(active);
    while (todo) {
        const int leader = __ffsll((unsigned long long)todo) - 1;
        const uint32_t lb = (uint32_t)__shfl((int)bin, leader, 64);
        const uint64_t same = __ballot(active && bin == lb);
        if (lane == leader) atomicAdd(&hist[lb], (uint32_t)__popcll(same));
        todo &= ~same;
    }
}


}  // namespace rala_hip
