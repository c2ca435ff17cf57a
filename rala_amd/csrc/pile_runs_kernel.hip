// Pile-o-gram construction and annotation in RUN space, one wavefront per read.
//
// A pile is a step function: coverage only changes at bound events.  The E events of a read
// give R <= E + 1 runs (start, value): through a bitmap of the positions that carry an event
// (reads of up to 16384 / 32768 bases: run index = popcount, +-1 per event into its run's slot,
// prefix sum) or, for longer reads, sorted (bitonic sort in registers, DPP / cross-lane shuffles,
// wave_sort.h) and swept.  Every per-base loop of the reference then becomes a loop over runs:
//   * Pile::add_layers        bitmap + popcounts + wave prefix sum   O(E)
//   * Pile::find_valid_region  streaks of runs with value >= 4      O(R)
//   * Pile::shrink             zero the runs outside the streak; the pile is
//                              expanded once, 16 B per lane, straight to HBM, right
//                              behind the runs so that the stores drain behind the rest
//   * Pile::find_median        histogram over (value, length)       O(R)
//   * Pile::find_slopes flags  within a run the window maximum only has to be
//       compared with ONE threshold, so the flagged positions of a run are a
//       prefix (down) and a suffix (up) of it, bounded by the nearest run to
//       the left / right whose value exceeds the threshold: O(runs per window)
//   * region lists are merged / tested wave-parallel; the reference's serial
//       resolution, narrowing, pit and hill loops run on one lane only for the
//       few piles whose regions actually interact, reading coverage through a
//       run cursor
// Every read starts in the instantiation that fits its length class and its event
// count (pipeline.hip: both are known before the first kernel runs); what still
// does not fit - region lists, event caps - is appended to an overflow list for the
// next kernel in the chain (larger kCap, then the position-space kernel of
// pile_kernels.hip); the chain needs no host synchronisation.
//
// Reference behaviour followed: rvaser/rala src/pile.cpp:64-455.
#include <hip/hip_runtime.h>

#include <stdlib.h>

#include <type_traits>

#include "device_utils.h"
#include "geom.h"
#include "kernels.h"
#include "wave_sort.h"

namespace rala_hip {

namespace {

constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr uint16_t kNone16 = 0xFFFFu;

// The workgroup is ONE wavefront.  Its LDS instructions execute in program order, so lanes that
// exchange data through LDS need no hardware wait - only the compiler must keep the accesses in
// order.  __syncthreads() would also drain every outstanding global store (s_waitcnt vmcnt(0)
// in front of each barrier): after the expansion that parks the wave until its whole pile row
// has reached memory.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <class RsT>
struct RunCursorT {
    const RsT* rs;          // run starts, rs[R] = n (16 bits where no read is longer than 16384 bases)
    const uint16_t* rv;     // run values
    const uint16_t* idx;    // run containing position m << shift
    uint32_t shift;
    uint32_t k;             // cached run
    __device__ uint32_t run_of(uint32_t j) {
        uint32_t c = k;
        if (j < rs[c] || j >= rs[c + 1]) {
            c = idx[j >> shift];
            while (rs[c + 1] <= j) ++c;
            k = c;
        }
        return c;
    }
    __device__ uint32_t operator[](uint32_t j) { return rv[run_of(j)]; }
};
typedef RunCursorT<uint32_t> RunCursor;

struct RegionList {
    uint32_t* key;      // first << 1 | is_up
    uint32_t* last;
    uint32_t n, cap;
    bool overflow;
};

__device__ __forceinline__ void rl_push(RegionList& R, uint32_t key, uint32_t last) {
    if (R.n >= R.cap) { R.overflow = true; return; }
    R.key[R.n] = key;
    R.last[R.n] = last;
    ++R.n;
}

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

// pile.cpp:131-220 (one lane; only reached when two regions overlap).  The
// reference walks positions; coverage is constant inside a run and a position is
// never flagged against its own run (v * q >= v), so whole clipped runs are
// flagged or not together: the walk is over runs.
template <class RsT>
__device__ void resolve_serial(RegionList& R, RunCursorT<RsT>& d, double q) {
    if (R.n == 0) return;
    const RsT* rs = d.rs;
    const uint16_t* rv = d.rv;
    for (;;) {
        rl_sort(R);
        bool changed = false;
        for (uint32_t i = 0; i + 1 < R.n; ++i) {
            if (R.last[i] < (R.key[i + 1] >> 1)) continue;
            if (R.key[i] & 1) {
                // positions j in [s, e) with d[j] * q < max d(j, e]   (pile.cpp:141-174)
                const uint32_t s = R.key[i] >> 1;
                const uint32_t e = umin(R.last[i], R.last[i + 1]);
                uint32_t k = d.run_of(e);
                int32_t m = rv[k];
                bool open = false;
                uint32_t lo = 0, hi = 0;
                while (k > 0 && rs[k] > s) {
                    --k;
                    const uint32_t v = rv[k];
                    const uint32_t c_lo = umax(rs[k], s), c_hi = rs[k + 1] - 1;     // clipped run, c_hi < e
                    if ((double)v * q < (double)m) {
                        if (open && c_hi + 1 == lo) {
                            lo = c_lo;
                        } else {
                            if (open) rl_push(R, lo << 1 | 1, hi);
                            open = true;
                            lo = c_lo; hi = c_hi;
                        }
                    }
                    m = max(m, (int32_t)v);
                }
                if (open) rl_push(R, lo << 1 | 1, hi);
                R.key[i] = e << 1 | 1;
            } else {
                if (R.last[i] == (R.key[i + 1] >> 1)) continue;
                // positions j in (s, e] with d[j] * q < max d[s, j)   (pile.cpp:176-211)
                const uint32_t s = umax(R.key[i] >> 1, R.key[i + 1] >> 1);
                const uint32_t e = R.last[i];
                uint32_t k = d.run_of(s);
                int32_t m = rv[k];
                bool open = false;
                uint32_t lo = 0, hi = 0;
                while (rs[k + 1] <= e) {
                    ++k;
                    const uint32_t v = rv[k];
                    const uint32_t c_lo = rs[k], c_hi = umin(rs[k + 1] - 1, e);
                    if ((double)v * q < (double)m) {
                        if (open && c_lo == hi + 1) {
                            hi = c_hi;
                        } else {
                            if (open) rl_push(R, lo << 1, hi);
                            open = true;
                            lo = c_lo; hi = c_hi;
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
}

// max of the coverage over positions [a, b] (a <= b), by runs
template <class RsT>
__device__ uint32_t range_max_runs(RunCursorT<RsT>& d, uint32_t a, uint32_t b) {
    uint32_t k = d.run_of(a);
    uint32_t m = d.rv[k];
    while (d.rs[k + 1] <= b) {
        ++k;
        m = umax(m, d.rv[k]);
    }
    return m;
}

// pile.cpp:222-256 (one lane; only reached when an up region is followed by a
// down region within the window), by runs
template <class RsT>
__device__ void narrow_serial(RegionList& R, RunCursorT<RsT>& d, double q) {
    const RsT* rs = d.rs;
    const uint16_t* rv = d.rv;
    for (uint32_t i = 0; i + 1 < R.n; ++i) {
        if (!(R.key[i] & 1) || (R.key[i + 1] & 1)) continue;
        const uint32_t b = R.last[i];
        const uint32_t e = R.key[i + 1] >> 1;
        if ((uint32_t)(e - b) > kSlopeWindow) continue;
        const uint32_t m = (b + 1 < e) ? range_max_runs(d, b + 1, e - 1) : 0u;
        const uint32_t u_first = R.key[i] >> 1;
        // last j in [u_first, b] with m > d[j] * q
        uint32_t last_ok = u_first;
        {
            uint32_t k = d.run_of(b);
            for (;;) {
                if ((double)m > (double)rv[k] * q) { last_ok = umin(b, rs[k + 1] - 1); break; }
                if (k == 0 || rs[k] <= u_first) break;
                --k;
            }
            if (last_ok < u_first) last_ok = u_first;
        }
        // first j in [e, w_last] with m > d[j] * q
        const uint32_t w_last = R.last[i + 1];
        uint32_t first_ok = w_last;
        {
            uint32_t k = d.run_of(e);
            for (;;) {
                if ((double)m > (double)rv[k] * q) { first_ok = umax(e, rs[k]); break; }
                if (rs[k + 1] > w_last) break;
                ++k;
            }
        }
        R.last[i] = last_ok;
        R.key[i + 1] = first_ok << 1;
    }
}

// resolve_serial with all lanes of the wavefront (lists of up to 64 regions; walks of up to 64
// runs - longer ones, which need an event every 13 bases over a whole window, go to the serial
// version on lane 0).  One lane sorting a short list by insertion and looking for the first
// overlapping pair through LDS, every access a round trip, took a fifth of the merge phase.
//   * sort: every lane ranks its region among all of them (the others' keys are broadcast reads);
//   * the first pair that overlaps: one ballot;
//   * the walk over the runs between s and e: one run per lane, the running maximum of the serial
//     walk is an exclusive prefix maximum in walk order, maximal groups of consecutive flagged
//     runs come from the ballot of the flags and are appended by their first lanes.
// Returns false if the list overflowed (the read is handed on).  Uniform control flow.
template <class RsT>
__device__ bool resolve_wave(uint32_t* key, uint32_t* last, uint32_t& n, uint32_t cap, const RsT* rs,
                             const uint16_t* rv, const uint16_t* idx, uint32_t shift, double q, uint32_t lane,
                             uint32_t* word) {
    auto run_at = [&](uint32_t pos) {
        uint32_t c = idx[pos >> shift];
        while (rs[c + 1] <= pos) ++c;
        return c;
    };
    for (;;) {
        {
            const uint32_t k_me = lane < n ? key[lane] : 0u, l_me = lane < n ? last[lane] : 0u;
            uint32_t rank = 0;
            for (uint32_t y = 0; y < n; ++y) {
                const uint32_t ky = key[y], ly = last[y];
                rank += (ky < k_me || (ky == k_me && (ly < l_me || (ly == l_me && y < lane)))) ? 1u : 0u;
            }
            wave_sync();
            if (lane < n) { key[rank] = k_me; last[rank] = l_me; }
            wave_sync();
        }
        bool c = false;
        if (lane + 1 < n) {
            const uint32_t ki = key[lane], li = last[lane], kn = key[lane + 1];
            c = li >= (kn >> 1) && ((ki & 1) || li != (kn >> 1));
        }
        const uint64_t pairs = __builtin_amdgcn_ballot_w64(c);
        if (!pairs) return true;
        const uint32_t i = (uint32_t)__builtin_ctzll(pairs);
        const uint32_t ki = key[i], li = last[i], kn = key[i + 1], ln = last[i + 1];
        const bool up = ki & 1;
        // positions [s, e]; the serial walk starts at the run of e (up: towards s) or of s (down: towards e)
        const uint32_t s = up ? ki >> 1 : umax(ki >> 1, kn >> 1);
        const uint32_t e = up ? umin(li, ln) : li;
        const uint32_t k_s = run_at(s), k_e = run_at(e);
        const uint32_t steps = k_e - k_s;                      // runs the walk visits (it never looks at its first run)
        if (steps > 64) {
            // (cannot be decided here: the serial version takes the whole list over)
            uint32_t* flag = word;                               // a scratch word of the caller's
            if (lane == 0) {
                RunCursorT<RsT> d{rs, rv, idx, shift, 0};
                RegionList R;
                R.key = key; R.last = last; R.n = n; R.cap = cap; R.overflow = false;
                resolve_serial(R, d, q);
                flag[0] = R.n | (R.overflow ? 0x80000000u : 0u);
            }
            wave_sync();
            const uint32_t f = flag[0];
            n = f & 0x7FFFFFFFu;
            return !(f >> 31);
        }
        const uint32_t k_first = up ? k_e : k_s;                  // its value opens the running maximum
        const bool mine = lane < steps;
        const uint32_t k = up ? k_e - 1 - lane : k_s + 1 + lane;  // run of this lane in walk order
        const uint32_t v = mine ? rv[k] : 0u;
        const uint32_t incl = wave_scan_incl(v, OpMax());
        uint32_t before = lane_above(incl);
        if (lane == 0) before = 0;
        const uint32_t m = umax((uint32_t)rv[k_first], before);
        const bool f = mine && (double)v * q < (double)m;
        const uint32_t c_lo = mine ? (up ? umax(rs[k], s) : rs[k]) : 0u;
        const uint32_t c_hi = mine ? (up ? rs[k + 1] - 1 : umin(rs[k + 1] - 1, e)) : 0u;
        const uint64_t F = __builtin_amdgcn_ballot_w64(f);
        const uint64_t starts = F & ~(F << 1);                    // first lane of every group, in walk order
        const uint32_t groups = (uint32_t)__popcll(starts);
        if (n + groups > cap) return false;
        // the group of a first lane ends in front of the next unflagged lane
        const uint64_t rest = lane < 63 ? ~F >> (lane + 1) : ~0ull;
        const uint32_t t_end = rest ? lane + (uint32_t)__builtin_ctzll(rest) : 63u;
        const uint32_t lo_end = (uint32_t)__shfl((int)c_lo, (int)umin(t_end, 63u), 64);
        const uint32_t hi_end = (uint32_t)__shfl((int)c_hi, (int)umin(t_end, 63u), 64);
        if ((starts >> lane) & 1) {
            const uint32_t at = n + (uint32_t)__popcll(starts & ((1ull << lane) - 1ull));
            // up: the walk goes down, the group's last lane holds its lowest position; down: the other way round
            key[at] = up ? (lo_end << 1 | 1u) : (c_lo << 1);
            last[at] = up ? c_hi : hi_end;
        }
        if (lane == 0) {
            if (up) key[i] = e << 1 | 1u;
            else last[i] = s;
        }
        n += groups;
        wave_sync();
    }
}

template <int C>
__device__ __forceinline__ void load_sort_store(const uint32_t* __restrict__ gev, uint32_t n_ev, uint32_t n,
                                                uint32_t* ev, uint32_t lane) {
    uint32_t v[C];
#pragma unroll
    for (int t = 0; t < C; ++t) {
        const uint32_t e = (uint32_t)t * 64u + lane;
        uint32_t b = kNone;
        if (e < n_ev) {
            b = gev[e];
            if ((b >> 1) > n) b = kNone;        // outside the read (undefined in the reference)
        }
        v[t] = b;
    }
    wave_sort_dpp<C>(v, lane);
#pragma unroll
    for (int t = 0; t < C; ++t) ev[(uint32_t)t * 64u + lane] = v[t];
}

// A 32-bit word holds two positions; when the value changes at position x (1 .. 7) of a group of
// 8, word q takes the new value in the halves at or behind x: all of it for x <= 2q, the upper half
// for x == 2q + 1, nothing beyond.
__device__ __forceinline__ uint32_t change_mask(uint32_t x, uint32_t q) {
    return x <= 2 * q ? 0xFFFFFFFFu : x == 2 * q + 1 ? 0xFFFF0000u : 0u;
}

// The same as byte selectors for v_perm_b32(new, old, selector): bytes 1, 0 = the old value's, bytes 5, 4 = the new one's.
__device__ __forceinline__ uint32_t change_select(uint32_t x, uint32_t q) {
    const uint32_t lo = x != 0 && x <= 2 * q ? 0x0504u : 0x0100u, hi = x != 0 && x <= 2 * q + 1 ? 0x0504u : 0x0100u;
    return lo | (hi << 16);
}

// (a & mask) | (b & ~mask) in one instruction
__device__ __forceinline__ uint32_t bitfield_insert(uint32_t mask, uint32_t a, uint32_t b) {
    uint32_t d;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(d) : "v"(mask), "v"(a), "v"(b));
    return d;
}

// a 16-bit value in a 32-bit register whose upper half is nobody's business (no instruction to extend it)
__device__ __forceinline__ uint32_t low_half_only(uint16_t x) {
    typedef uint16_t u16x2_t __attribute__((ext_vector_type(2)));
    u16x2_t t;
    t.x = x;                                // (t.y stays what it is)
    return __builtin_bit_cast(uint32_t, t);
}

// index of the lowest set bit, 0xFFFFFFFF for zero (v_ffbl_b32's own answer: no select around it)
__device__ __forceinline__ uint32_t ffbl_or_minus1(uint32_t v) {
    uint32_t d;
    asm("v_ffbl_b32 %0, %1" : "=v"(d) : "v"(v));
    return d;
}

// 16-byte store to base + off with the base in scalar registers and a 32-bit lane offset.  The
// compiler prefers a 64-bit address per lane: two more registers per store in flight, and at this
// kernel's register budget that meant spill reloads inside the store loop - a scratch load is a
// vector memory operation, and the wait for it is a wait for every row store issued so far.
// (RALA_ROW_STORE_MOD: cache-policy bits of the row stores, for measurements - "sc1", "sc0 sc1", "nt"; round 5)
#ifndef RALA_ROW_STORE_MOD
#define RALA_ROW_STORE_MOD ""
#endif
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// (kMod, measurements inside one process - bits 16 - 17 of a kernel variant: 1 non-temporal, 2 sc1, 3 sc0 sc1)
template <int kMod = 0>
__device__ __forceinline__ void store16(const char* base, uint32_t off, const uint4& v) {
    const u32x4 d = {v.x, v.y, v.z, v.w};
    if constexpr (kMod == 1) { asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" : : "v"(off), "v"(d), "s"(base) : "memory"); return; }
    if constexpr (kMod == 2) { asm volatile("global_store_dwordx4 %0, %1, %2 sc1\n\ts_nop 1" : : "v"(off), "v"(d), "s"(base) : "memory"); return; }
    if constexpr (kMod == 3) { asm volatile("global_store_dwordx4 %0, %1, %2 sc0 sc1\n\ts_nop 1" : : "v"(off), "v"(d), "s"(base) : "memory"); return; }
    // (Measured and not kept: the `nt` modifier - streaming stores gave 1 % at C3 inside one gpurun call and 3 %
    // more write traffic at the memory; for the bucketing's scattered 8-byte stores, which need the cache to
    // merge them, it cost 40 %.)
    // (s_nop: a store of more than 8 bytes reads its data registers over the following cycles; the
    // compiler's hazard recogniser keeps vector writes to them away from its own stores, not from this one)
    asm volatile("global_store_dwordx4 %0, %1, %2 " RALA_ROW_STORE_MOD "\n\ts_nop 1" : : "v"(off), "v"(d), "s"(base) : "memory");
}
// (the same with an immediate offset: the lane offset is then a loop constant)
template <int kImm, int kMod = 0>
__device__ __forceinline__ void store16_imm(const char* base, uint32_t off, const uint4& v) {
    const u32x4 d = {v.x, v.y, v.z, v.w};
    if constexpr (kMod == 1) { asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3 nt\n\ts_nop 1" : : "v"(off), "v"(d), "s"(base), "n"(kImm) : "memory"); return; }
    if constexpr (kMod == 2) { asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3 sc1\n\ts_nop 1" : : "v"(off), "v"(d), "s"(base), "n"(kImm) : "memory"); return; }
    if constexpr (kMod == 3) { asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3 sc0 sc1\n\ts_nop 1" : : "v"(off), "v"(d), "s"(base), "n"(kImm) : "memory"); return; }
    asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3 " RALA_ROW_STORE_MOD "\n\ts_nop 1" : : "v"(off), "v"(d), "s"(base), "n"(kImm) : "memory");
}

// kShort: the first kernel of the chain takes only reads of up to 16384 bases (the others go to the
// next one): run starts in 16 bits, no sorted path, lists a little shorter, the last phases' lists inside the
// region of the first phases' bitmap - 5 568 bytes of LDS, seven wavefronts per SIMD instead of five.
// kBases: the reads the bitmap of run starts covers (one bit per position + a 16-bit prefix per word);
// 16384 everywhere but in the chain's second kernel, the short layout for reads of up to 32768 bases
// (9 728 bytes of LDS, four wavefronts per SIMD).
template <uint32_t kCap, bool kShort = false, uint32_t kBases_ = 16384>
struct Layout {
    static constexpr uint32_t kBases = kBases_;
    static constexpr uint32_t kBmWords = kBases / 32;
    static constexpr uint32_t kXbitmap = kBmWords + kBmWords / 2;
    typedef typename std::conditional<kShort, uint16_t, uint32_t>::type rs_t;
    static constexpr bool kShortLayout = kShort;
    // list capacities: a read that needs more goes to the next kernel of the chain
    // three instantiations: 512 events (almost every read), 1024 (high coverage), 2048 (the rest)
    static constexpr uint32_t kMaxReg = kShort ? (kCap <= 512 ? 14 : 28) : kCap <= 512 ? 16 : kCap <= 1024 ? 32 : 64;   // regions per (q, kind) list
    static constexpr uint32_t kMaxRaw = kCap <= 512 ? 8 : kCap <= 1024 ? 16 : 32;    // pits / hills before the merge
    static constexpr uint32_t kArr = kCap + 4;                  // entries per run-indexed array
    static constexpr uint32_t kIdx = kShort ? 128 : kCap <= 512 ? 256 : 512;   // entries of the position -> run index
    // runs that survive the slope filter (k and four uint16 offsets each) + block maxima
    static constexpr uint32_t kSurv = kCap <= 512 ? 192 : kCap <= 1024 ? 384 : kArr;
    static constexpr uint32_t kBm8 = kArr / 8 + 8;
    static constexpr uint32_t kSlopeWords = (5 * kSurv + 1) / 2 + kBm8;
    // X: events (sort) -> bitmap + prefix -> group counts -> histograms -> slope survivors
    static constexpr uint32_t X = 0;
    static constexpr uint32_t kXmin = kCap > kXbitmap ? kCap : kXbitmap;
    // (short layout: the region lists - 8 * kMaxReg words and 4 counts - lie inside X behind the slope survivors; at 512 events
    // the bitmap's size leaves that room anyway)
    static constexpr uint32_t kXlists = kShort ? kSlopeWords + 8 * kMaxReg + 4 : kSlopeWords;
    static constexpr uint32_t kX = kXlists > kXmin ? kXlists : kXmin;
    static constexpr uint32_t RS = X + kX;                      // run starts (+ sentinel)
    static constexpr uint32_t RV = RS + (kShort ? kArr / 2 : kArr);   // run values, uint16 (kArr / 2 words)
    static constexpr uint32_t IDX = RV + kArr / 2;              // kIdx uint16
    // The short layout (seven wavefronts per SIMD: 5 632 B) keeps the lists of the last phases inside X, which is
    // free by then: the region lists behind the slope survivors (written while those are read), the merged
    // regions and the interval scratch at its start (written when the survivors are done with); the per-run
    // sums of the first phase in the place of the run values they become.  What is left behind the position
    // index: the expansion's mask table and the scratch words.
    static constexpr uint32_t XT = ((IDX + kIdx / 2) + 3) & ~3u;   // (short layout) 32 words: 8 masks of the expansion
    static constexpr uint32_t RF = kShort ? X + kSlopeWords : IDX + kIdx / 2;   // 4 lists x kMaxReg firsts
    static constexpr uint32_t RL = RF + 4 * kMaxReg;            // 4 lists x kMaxReg lasts
    static constexpr uint32_t RC = RL + 4 * kMaxReg;            // 4 counts
    static constexpr uint32_t REG = kShort ? X : RC + 4;        // 2 x (key, last) x 2 * kMaxReg
    static constexpr uint32_t IV = REG + 8 * kMaxReg;           // 2 x 4 x kMaxRaw
    static constexpr uint32_t GONE = IV + 8 * kMaxRaw;          // 2 x kMaxRaw bytes
    static constexpr uint32_t CAND = GONE + (2 * kMaxRaw) / 4;  // kMaxRaw hill candidates (i << 16 | j)
    static constexpr uint32_t SEL = kShort ? XT + 32 : CAND + kMaxRaw;   // 12 words
    static constexpr uint32_t WORDS = SEL + 12;
    static constexpr uint32_t DELTA = kShort ? RV : RF;         // per-run sums of +-1 (first phase), two per word
    static_assert(!kShort || (RC + 4 <= X + kX && CAND + kMaxRaw <= X + kSlopeWords), "the last phases' lists inside X");
    // scratch of the expansion: a list of 64 noted groups and a table of 8 masks.  The list takes the position
    // index's place where that is computed behind the expansion (kShort).
    static constexpr uint32_t XLIST = kShort ? IDX : RF;
    static constexpr uint32_t XTABLE = kShort ? XT : ((RF + 64) + 3) & ~3u;
    static_assert(kX >= kCap && kX >= kIdx && kX >= 768, "shared region too small");
    static_assert(kIdx / 2 >= 64 && SEL >= XTABLE + 32, "scratch of the expansion: 64 noted groups, 8 masks");
    static_assert(kBases == 16384 || (kShort && (kBases == 32768 || kBases == 65536)), "bitmap sizes in use");
    static_assert(!kShort || kBases != 16384 || kCap > 512 || WORDS * 4 <= 5632, "seven wavefronts per SIMD: 28 workgroups in 160 KB, 512-byte granules");
    static_assert(!kShort || kBases > 32768 || WORDS * 4 <= 10240, "four wavefronts per SIMD: 16 workgroups in 160 KB");
    static_assert(!kShort || WORDS * 4 <= 16384, "ten workgroups in 160 KB");
};



// Expansion of a pile of up to L::kBases positions (groups of 8 = 16 bytes) from the bitmap of
// run starts: bm (one bit per position that starts a run, bit n set for the padding behind the
// last base), pref[w] = run starts before word w, rv = run values (rv[R] = 0).
//
// A lane owns byte (lane & 3) of bitmap word (group >> 2), so its masks are loop constants.  Six
// groups in seven hold one value in all 8 places, nearly all of the rest two (one run starts inside,
// at place x = 1 .. 7): both are ONE code path - the value at the group's first place, the next run's
// value, and 16 bytes of byte selectors "places at or behind x take the new value" fetched from an eight-entry LDS
// table (x = 0: no place), four v_perm_b32 (round 4; before: a mask and four bit-field inserts over two splatted
// values - three vector instructions more per group, the same time: what bounds the kernel is not its instruction
// count, section 4 of DESIGN.md).  Only a group in which two or more runs start (one in a hundred) is
// noted in an LDS list; the noted groups are walked one per lane, once per read, and stored from
// there - a dozen scattered 16-byte stores per read, where patching EVERY changed group that way
// (one in seven) once cost as much as all the dense stores together.  Every byte of the row is
// stored exactly once.  (Before: every changed group went through the list, a walk and a table of
// finished groups, once per chunk of 256 groups - 700 of the expansion's 900 vector instructions
// per read.)
template <class L, int kMod = 0>
__device__ __forceinline__ void expand_from_bitmap(uint32_t* sm, const uint16_t* rv, uint16_t* pile, uint64_t row_off,
                                                   uint32_t nv, uint32_t lane, bool store) {
    const uint32_t* bm = sm + L::X;
    const uint16_t* pref = (const uint16_t*)(bm + L::kBmWords);
    const uint16_t* rvm1 = rv - 1;                      // indexed by run + 1 (= run starts at or before)
    uint32_t* list = sm + L::XLIST;                     // the region lists are not in use yet
    uint4* masks = (uint4*)(sm + L::XTABLE);            // 8 entries, 16-byte aligned
    // the row address is the same in every lane: keep it in scalar registers, 32-bit lane offsets
    // (the builtin returns int: without the casts the low half would be sign-extended over the high one)
    const uint64_t off = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(row_off >> 32)) << 32) |
                         (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)row_off);
    char* base = (char*)(pile + off);
    if (lane < 8) {
        // word q holds places 2q and 2q + 1: all ones when x <= 2q, the upper half when x == 2q + 1
        masks[lane] = make_uint4(change_select(lane, 0), change_select(lane, 1), change_select(lane, 2), change_select(lane, 3));
    }
    wave_sync();
    const uint32_t sh1 = ((lane & 3u) << 3) + 1u;
    const uint32_t below = (1u << sh1) - 1u;            // the bits at or before the group's first position
    const uint32_t w_lane = lane >> 2;
    const uint32_t lane16 = lane * 16u;
    uint32_t cnt = 0;                                   // noted groups (the same in every lane)
    // the noted groups, one per lane: walk the changes, store
    auto flush = [&]() {
        wave_sync();
        if (lane < cnt) {
            const uint32_t e = list[lane];
            uint32_t k = e & 0xFFFu;
            uint32_t inner = (e >> 12) & 0x7Fu;
            const uint32_t g = e >> 19;
            const uint32_t v0 = rvm1[k];
            const uint32_t vv = v0 | (v0 << 16);
            uint32_t w0 = vv, w1 = vv, w2 = vv, w3 = vv;
            do {
                const uint32_t x = (uint32_t)__builtin_ctz(inner) + 1u;     // position 1 .. 7 inside the group
                inner &= inner - 1;
                const uint32_t nvv = rvm1[++k];
                const uint32_t f = nvv | (nvv << 16);
                w0 = bitfield_insert(change_mask(x, 0), f, w0);
                w1 = bitfield_insert(change_mask(x, 1), f, w1);
                w2 = bitfield_insert(change_mask(x, 2), f, w2);
                w3 = bitfield_insert(change_mask(x, 3), f, w3);
            } while (inner);
            if (store) store16<kMod>(base, g * 16u, make_uint4(w0, w1, w2, w3));
        }
        wave_sync();
        cnt = 0;
    };
    static_assert(L::kArr <= 4096 && L::kBases <= 65536, "noted group = run (12 bits) | starts (7) | group (13)");
    auto chunk = [&](uint32_t g0, auto full_tag) {
        constexpr bool kFull = decltype(full_tag)::value;
        const uint32_t voff = lane16 + g0 * 16u;              // the stores' lane offset: once per four groups
        // (Measured, round 4, no gain: the LDS reads of two groups under way before the first value is looked at instead of
        // read - wait - combine - store group after group: 4.11 - 4.14 ms either way, docs/history/gpurun/r4_ahead.sh.  What the
        // kernel's rate hangs on at a wavefront's start are its round trips to MEMORY - kPlain -, not those to the LDS.)
        uint32_t bits[4], kq[4];
#pragma unroll
        for (uint32_t u = 0; u < 4; ++u) {
            const uint32_t w = (g0 >> 2) + 16u * u + w_lane;        // < kBmWords for every group of the pile
            bits[u] = bm[w];
            kq[u] = pref[w];
        }
#pragma unroll
        for (uint32_t u = 0; u < 4; ++u) {
            const uint32_t k = kq[u] + (uint32_t)__popc(bits[u] & below);      // bit 0 of the bitmap is set: >= 1
            // the value at the group's first place and the next run's (beyond the pile: some LDS word).  (Measured and
            // dropped: both in ONE two-byte aligned 32-bit read - the kernel 4.40 -> 4.85 ms, an LDS word read across
            // its alignment is not one access.)
            // (the byte selectors below take bytes 0, 1 of each only: no extension of the 16-bit LDS reads - the compiler puts a
            // v_and behind a zero-extending one)
#ifdef RALA_EXPAND_EXTEND           // (measurements: the reads zero-extended, as before)
            const uint32_t v0 = rvm1[k], v1 = rvm1[k + 1];
#else
            const uint32_t v0 = low_half_only(rvm1[k]), v1 = low_half_only(rvm1[k + 1]);
#endif
            const uint32_t b7 = (bits[u] >> sh1) & 0x7Fu;                      // run starts at positions 1 .. 7
            const uint32_t gl = 64u * u + lane;
            const bool in = kFull || g0 + gl < nv;
            const bool multi = in && (b7 & (b7 - 1u)) != 0;
            // x = place of the (first) run start inside, 0 = none: find-first-bit gives -1 for none
            const uint32_t x = ffbl_or_minus1(b7) + 1u;
            const uint4 m = masks[x];
            const uint4 out = make_uint4(__builtin_amdgcn_perm(v1, v0, m.x), __builtin_amdgcn_perm(v1, v0, m.y),
                                         __builtin_amdgcn_perm(v1, v0, m.z), __builtin_amdgcn_perm(v1, v0, m.w));
            const uint64_t noted = __builtin_amdgcn_ballot_w64(multi);
            if (noted) {
                const uint32_t more = (uint32_t)__popcll(noted);
                if (cnt + more > 64u) flush();
                if (multi) {
                    const uint32_t slot = cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(noted >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)noted, 0u));
                    list[slot] = k | (b7 << 12) | ((g0 + gl) << 19);
                }
                cnt += more;
            }
            if (in && !multi && store) {
                if (u == 0) store16_imm<0, kMod>(base, voff, out);
                else if (u == 1) store16_imm<1024, kMod>(base, voff, out);
                else if (u == 2) store16_imm<2048, kMod>(base, voff, out);
                else store16_imm<3072, kMod>(base, voff, out);
            }
        }
    };
    uint32_t g0 = 0;
    for (; g0 + 256 <= nv; g0 += 256) chunk(g0, std::true_type());
    if (g0 < nv) chunk(g0, std::false_type());
    if (cnt) flush();
    wave_sync();                                        // the list and the masks lie where others follow
}

}  // namespace

// kDiag: the diagnostic instantiation honours PileArgs::stop_after (per-phase counter runs,
// docs/history/gpurun/gpurun_pmc.sh); the product instantiation carries none of those branches.
//
// kSens: the second, "sensitive" pass (rala -s; reference graph.cpp:917-1026) in the same run space.
// The coverage of a read is then the sum over its primary bound events (still in the slots the
// first pass bucketed them into) and the bounds of the sensitive overlaps that target it
// (PileArgs::sens_off / sens_ev); the valid region is given (begin / end as the chimera stage
// left them), so two more bits in the bitmap of run starts make it a whole number of runs.
//   kSens == 1  Pile::add_layers + Pile::find_median for the targets (pile.cpp:261-297): the
//               runs, the order statistics, the expansion of the new coverage to HBM
//   kSens == 2  Pile::find_repetitive_hills (pile.cpp:500-566) for the members of connected
//               components: slope regions at q = 1.42, every (up, later down) pair tested over
//               the runs between them, intervalMerge, clamp.  Neither reads nor writes the row.
// A read that does not fit (events, length, lists) goes to overflow_list and from there to the
// position-space kernel of pile_repeats_kernel.hip.
// kOne: the launch has one workgroup per item (the first kernel of the chain): no loop over the
// items, so nothing is hoisted out of one and kept in registers for the whole kernel.  (A zero
// vector hoisted that way was spilled and reloaded right behind the row stores: a scratch load, and
// with it a wait for every one of them.)
// kBases > 16384: the short layout with a bigger bitmap, for the reads the first kernel handed on.
// kWaves (with kOne): reads per workgroup, one wavefront each.  The wavefronts of a workgroup share nothing - every one has
// its own stretch of LDS and synchronises with itself only; what they have in common is their place: the rows of
// neighbouring reads are written from one compute unit (tools/fill_bench4.hip: four rows per workgroup fill at 5.4 TB/s
// where one row per workgroup fills at 5.0).
// kVar: measurement variants of the product instantiation, chosen per launch (PileArgs::variant, option "debug_pile_variant") so
// that two builds of the kernel are compared INSIDE one process, on the same allocations, step by step - a library built with
// -DRALA_PILE_AB carries them (round 6: two processes with the same binary differed by 12 % in this kernel's time, two binaries in
// alternating processes by less).  What is kept are the variants behind profiles/r06_c3_pile_sensitivity.txt: bit 0 the loop over the
// items as it was, bit 1 the events by ordinary loads, bit 2 the arguments fetched where they are first used (8195 = bits 0, 1, 13 =
// round 5's kernel); bits 8 - 10 work ADDED behind the expansion, for the sensitivity of the kernel's time to each kind of it - 1: 128
// independent vector instructions, 2: 128 scalar ones, 3: 1024 cycles asleep, 4: 32 LDS reads and their wait, 5: 512 vector
// instructions; bit 13 the reads as launched instead of XCD ranges; bits 16 - 17 the row stores' cache policy (1 nt, 2 sc1, 3 sc0 sc1).
// The others round 6 measured - s_nop pads, sleeps and work in front of the first load, one / four reads per workgroup, priorities,
// non-temporal event loads, the XCC_ID probe, the expansion's look-ups one group ahead, XCDs taking chunks in turn - are a patch:
// docs/history/experiments/r6_pile_kernel_variants.patch (their results: docs/history/r6_pile_kernel_notebook.md).
template <uint32_t kCap, bool kDiag, int kSens, bool kOne = false, uint32_t kBases = 16384, uint32_t kWaves = 1, bool kPlain = false, uint32_t kVar = 0>
__global__ __launch_bounds__(64 * kWaves, kOne ? 7 : kBases > 32768 ? 2 : kBases > 16384 ? 4 : 5) void pile_runs_kernel(PileArgs A, uint32_t* overflow_list, uint32_t* overflow_count) {
    static_assert(kWaves == 1 || kOne, "several reads per workgroup: the first kernel of a chain only");
    static_assert(kSens == 0 || !kDiag, "the sensitive pass has no diagnostic instantiation");
    static_assert(!kOne || kCap <= 512, "the short layout belongs to the first kernel of a chain");
    static_assert(kBases == 16384 || !kOne, "the bigger bitmap belongs to the chain's second kernel");
    // (round 5: the sensitive pass's cap-1024 kernels too - their reads have at most 16384 bases; 14 296 -> 9 840 B of LDS, sixteen
    // workgroups per compute unit instead of eleven, and at C5 this is the kernel nearly every target starts in)
    constexpr bool kXcdRanges = kOne && kPlain && !(kVar & 8192u);
    // The rows' stores (kVar bits 16 - 17): NON-TEMPORAL where the rows lie in mapped chunks (the launch below picks the
    // instantiation), plain otherwise.  History: with the reads as launched `nt` was 18 % slower (round 5: the L2 was what put
    // neighbouring rows' lines together); on top of the XCD ranges it was 3 % faster where one hipMalloc had put the rows well
    // and 4 % slower where it had not (3.70 against 3.82, 4.26 against 4.11) - not a default; with the rows in chunks of 1 GB
    // (pipeline.hip), where the placement is the good one every time: **3.51 - 3.57 ms against 3.71 - 3.80** in eight processes on
    // three boxes, C5 15.2 against 16.2 - the default there.
    constexpr int kRowStoreMod = (int)((kVar >> 16) & 3u);
    constexpr bool kSingleItem = kOne && kWaves > 1 && !(kVar & 1u);
    constexpr bool kSingleItemArgs = kSingleItem && !(kVar & 4u);
    constexpr bool kBufferEvents = kSens == 0 && !(kVar & 2u);
    constexpr bool kShort = kOne || kBases > 16384 || (kSens != 0 && kCap == 1024);     // reads of up to kBases bases only, 16-bit run starts
    constexpr uint32_t kMaxBases = kBases > 65535 ? 65535 : kBases;     // rs[R] = n in 16 bits
    typedef Layout<kCap, kShort, kBases> L;
    typedef typename L::rs_t rs_t;
    constexpr uint32_t kMaxReg = L::kMaxReg, kMaxRaw = L::kMaxRaw;
    __shared__ __align__(16) uint32_t sm_all[kWaves * L::WORDS];
    const uint32_t wave_in_group = kWaves == 1 ? 0u : (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint32_t* sm = sm_all + wave_in_group * L::WORDS;
    uint32_t* ev = sm + L::X;
    rs_t* rs = (rs_t*)(sm + L::RS);
    uint16_t* rv = (uint16_t*)(sm + L::RV);
    uint16_t* idx = (uint16_t*)(sm + L::IDX);
    uint32_t* sel = sm + L::SEL;
    // (one workgroup per item: launch_pile_runs / launch_pile_sens start these instantiations only without a count in device memory -
    // one dependent round trip less at every wavefront's start)
    const uint32_t n_items = kOne ? A.n_items : A.n_items_dev ? *A.n_items_dev : A.n_items;
    if constexpr (kPlain && kSingleItemArgs) {
        // Without a loop over the items the compiler fetches every argument where it is first used: six dependent round trips to the
        // scalar cache in front of the events' loads.  Named here, what a wavefront's first steps need comes in ONE.
        asm volatile("" : : "s"(A.read_len), "s"(A.pile_off), "s"(A.ev_off), "s"(A.ev), "s"(A.ev_shift), "s"(A.skip_dense), "s"(overflow_list),
                     "s"(overflow_count), "s"(A.pile), "s"(A.error));
    }

    // diagnostics: 77 = everything but the row stores; 100 + k = leave after phase k, without them
    const uint32_t stop_k = A.stop_after >= 100 ? A.stop_after % 100 : A.stop_after;
    const bool row_stores = !kDiag || !(A.stop_after == 77 || A.stop_after >= 100);
#define RUN_STOP(k)                                                        \
    if (kDiag && stop_k == (k)) {                                          \
        if (lane == 0) A.alive[A.order ? A.order[item] : item] = 0;        \
        wave_sync();                                                       \
        continue;                                                          \
    }
    // Workgroup i runs on XCD i % 8.  A grid smaller than the number of items (a multiple of 8) gives every
    // XCD one contiguous eighth of the items (tools/fill_bench3.hip: rows written side by side by one XCD
    // stream out at 6.2 TB/s instead of 5.2 TB/s - in a fill kernel; this kernel does not notice.  Round 4, the same
    // with one workgroup per read - workgroup i takes read (i % 8) * n / 8 + i / 8 - on the build that runs at the
    // fill kernel's 5.0 TB/s: C3 4.26 against 4.16 ms, C5 17.4 against 17.7, docs/history/gpurun/r4_xcd.sh; not kept).
    uint32_t item_first = blockIdx.x * kWaves + wave_in_group, item_end = n_items, item_step = gridDim.x * kWaves;
    if (kOne && kWaves == 1 && gridDim.x % 8u == 0 && gridDim.x < n_items) {
        const uint32_t per = (n_items + 7u) / 8u, xcd = blockIdx.x % 8u;
        item_first = xcd * per + blockIdx.x / 8u;
        item_end = umin(n_items, (xcd + 1u) * per);
        item_step = gridDim.x / 8u;
    }
    // The product instantiation (round 6): every XCD writes ONE contiguous range of the rows - workgroup i runs on XCD i % 8 and takes
    // the reads (i % 8) * per + kWaves * (i / 8) + wavefront, per = the reads of an XCD (a multiple of kWaves), 8 * per / kWaves
    // workgroups.  A kernel that only fills the same rows goes from 5.2 to 6.2 TB/s that way (tools/fill_bench3.hip); this kernel, whose
    // floor IS that fill rate (4.065 ms = 5.13 TB/s whatever is added to or taken from its instructions - tools/gpurun/r6_ab_inproc.sh),
    // from 4.07 - 4.3 ms to 3.8 - 3.9 at C3, measured variant against variant inside one process.  Round 4 had
    // tried the mapping with one read per workgroup on a build that sat above that floor for other reasons and saw nothing.
    // (kVar bit 13: the reads as launched - workgroup i takes 2 i and 2 i + 1 - for measurements.)
    if constexpr (kXcdRanges) {
        const uint32_t per = ((n_items + 8u * kWaves - 1u) / (8u * kWaves)) * kWaves, xcd = blockIdx.x & 7u, j = (blockIdx.x >> 3) * kWaves + wave_in_group;
        item_first = xcd * per + j;
        item_end = j < per ? umin(n_items, (xcd + 1u) * per) : 0u;
    }
    // (Round 4 carried a persistent variant of the first kernel here - as many workgroups as the chip holds, the next read's
    // events requested one read ahead by inline assembly into registers the compiler did not know of, the annotation stores
    // parked in LDS, so that nothing was ever waited for behind a row store: the same time as one workgroup per read at every
    // grid size, DESIGN.md section 4.  Removed in round 5: product code that buys nothing and depends on the register
    // allocator's habits should not stay.)
    // (kOne with several reads per workgroup - the product instantiation - is launched with one wavefront per item: said here in a
    // way the compiler sees, the loop is none, and nothing "loop invariant" - the LDS offsets of a wavefront's stretch, every
    // pointer of the arguments - is computed in front of it and kept in scalar registers for the whole kernel: round 5's 56
    // scalar spills = 56 v_writelane at every wavefront's start and a v_readlane wherever one was used, all of them vector
    // instructions of a kernel that is bound by their issue)
    for (uint32_t item = item_first; item < item_end; item = kSingleItem ? item_end : item + item_step) {
        // (The lane id is made opaque once per item: what is derived from it is then no loop invariant.  The
        // compiler used to hoist such values out of the loop and keep them in registers for the whole kernel -
        // 96 VGPRs and spills with a loop over items, 70 without one.)
        uint32_t lane;
        asm volatile("v_mov_b32 %0, %1" : "=v"(lane) : "v"(kWaves == 1 ? threadIdx.x : threadIdx.x & 63u));
        uint32_t r, n, n_ev_p;
        uint64_t row_off;
        const uint32_t* __restrict__ rev = nullptr;
        if constexpr (kPlain) {
            // the reads as they come, their events in the CSR: four loads that do not wait for each other (through the
            // branches of the general case below they were four round trips in a row)
            r = item;
            // (Measured and dropped: the same four values by the scalar unit - s_load_dword / dwordx2 and one wait - 4.6 - 4.8 ms
            // against 4.0 - 4.1, docs/history/gpurun/r4_scalar.sh: lines other kernels have just written are not in its cache.)
            const uint32_t e0 = A.ev_off[r], e1 = A.ev_off[r + 1];
            n = A.read_len[r];
            row_off = A.pile_off[r];
            n_ev_p = (e1 - e0) << A.ev_shift;
            rev = A.ev + ((size_t)e0 << A.ev_shift);
        } else {
            r = A.order ? A.order[item] : item;
            n = A.read_len[r];
            row_off = kSens == 2 ? 0 : A.pile_off[r];
            n_ev_p = A.ev_cnt ? umin(A.ev_cnt[r], A.ev_stride) : (A.ev_off[r + 1] - A.ev_off[r]) << A.ev_shift;
            rev = A.ev_cnt ? A.ev + (size_t)r * A.ev_stride : A.ev + ((size_t)A.ev_off[r] << A.ev_shift);
        }
        uint32_t n_ev = n_ev_p;
        const uint32_t* __restrict__ sev = nullptr;
        uint32_t given_b = 0, given_e = 0;
        if constexpr (kSens != 0) {
            const uint32_t s0 = A.sens_off[r];
            sev = A.sens_ev + s0;
            n_ev += A.sens_off[r + 1] - s0;
            given_b = A.begin[r];
            given_e = A.end[r];
        }
        // the first kernel of the first pass: longer reads start in their own length class's kernel
        // (pipeline.hip), no hand-over through the list
        if constexpr (kPlain) {
            // (one test over everything that was loaded: a branch on the length alone would have the offsets' loads wait
            // behind it)
            const uint32_t not_mine = (uint32_t)(n > kMaxBases) | (uint32_t)(A.skip_dense != 0 && n_ev > kRunEventCap) | (uint32_t)(row_off == ~0ull);
            if (not_mine) continue;
        }
        if (kOne && kSens == 0 && n > kMaxBases) continue;
        // (the two region marks of the sensitive pass may add two runs)
        if (kSens == 0 && kCap <= kRunEventCap && A.skip_dense && n_ev > kRunEventCap) continue;       // listed beforehand
        if (n_ev > (kSens ? kCap - 2 : kCap) || (kSens != 0 && (n > kMaxBases || given_e <= given_b)) || (kShort && n > kMaxBases)) {
            if (lane == 0) overflow_list[atomicAdd(overflow_count, 1u)] = r;
            continue;
        }

        // ---- 1 + 2. runs (start, value mod 2^16) ---------------------------------------------
        // Reads of up to 16384 bases (bitmap = 512 words) need no sort: one bit per position
        // that carries an event, the number of set bits up to a position is the index of its
        // run, so every event adds its +-1 straight into its run's slot (LDS atomics) and a
        // prefix sum over the slots gives the coverage.  Longer reads (and the big
        // instantiation) sort their events and sweep, as the reference does.
        uint32_t R;
        constexpr uint32_t kBitmapBases = kBases;
        constexpr uint32_t kV = L::kBmWords / 256;      // 16-byte vectors of the bitmap per lane
        // (the sensitive pass runs on the bitmap path whatever its event cap - its reads have at most kBases bases, and the marks of
        // the given region go into the bitmap; with 2048 events the events wait in 32 registers)
        const bool bitmap_path = kShort || ((kCap <= 1024 || kSens != 0) && n <= kBitmapBases);
        if (bitmap_path) {
            uint32_t* bm = sm + L::X;
            uint16_t* pref = (uint16_t*)(bm + kBitmapBases / 32);
            // per-run sums of +-1, two per word, biased by 0x8000 so that a subtraction never
            // borrows from the neighbour (at most kCap events meet at one position)
            // (short layout: in the place of the run values: run k's sum is the 16 bits that rv[k] takes)
            uint32_t* delta = sm + L::DELTA;
            static_assert((kCap > 1024 && kSens == 0) || kShort || L::SEL - L::RF >= L::kArr / 2 + 2, "scratch for the per-run sums");
#pragma unroll
            for (uint32_t j = 0; j < kV; ++j) ((uint4*)bm)[lane + 64 * j] = make_uint4(0, 0, 0, 0);
            for (uint32_t k = lane; 2 * k < n_ev + 5; k += 64) delta[k] = 0x80008000u;
            // all loads first, unconditionally (clamped index), so that they are in flight together:
            // a load per predicated block would be waited for one by one
            uint32_t evr[kCap / 64];
            const uint32_t e_last = n_ev ? n_ev - 1 : 0;
            if constexpr (kBufferEvents) {
                // (round 6) the events through a buffer descriptor over exactly this read's slots: one lane offset and eight
                // immediate ones instead of a clamped index and a 64-bit address per load, and nothing to mark behind the last
                // event (a load beyond the descriptor's size answers 0; the loops below ask the lane's index, not the value) -
                // 56 vector instructions of a kernel that is bound by their issue became one
                const __amdgpu_buffer_rsrc_t slots = __builtin_amdgcn_make_buffer_rsrc((void*)rev, 0, (int)(n_ev * 4u), 0x00020000);
                const int lane4 = (int)(lane * 4u);
#pragma unroll
                for (uint32_t t = 0; t < kCap / 64; ++t) evr[t] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(slots, lane4 + (int)(t * 256u), 0, 0);
            } else if constexpr (kSens == 0) {
#pragma unroll
                for (uint32_t t = 0; t < kCap / 64; ++t) evr[t] = rev[umin(t * 64 + lane, e_last)];
#pragma unroll
                for (uint32_t t = 0; t < kCap / 64; ++t) {
                    if (t * 64 + lane >= n_ev) evr[t] = kNone;
                }
            } else {
                // the primary events, then the bounds of the sensitive overlaps
#pragma unroll
                for (uint32_t t = 0; t < kCap / 64; ++t) {
                    const uint32_t i = t * 64 + lane;
                    evr[t] = i < n_ev_p ? rev[i] : i < n_ev ? sev[i - n_ev_p] : kNone;
                }
            }
            wave_sync();
#pragma unroll
            for (uint32_t t = 0; t < kCap / 64; ++t) {
                if (t * 64 >= n_ev) break;
                const uint32_t pos = evr[t] >> 1;
                if ((kBufferEvents ? t * 64 + lane < n_ev : evr[t] != kNone) && pos < n) atomicOr(&bm[pos >> 5], 1u << (pos & 31));
            }
            if (lane == 0) atomicOr(&bm[0], 1u);             // position 0 always starts a run
            if constexpr (kSens != 0) {
                // the given valid region starts and ends on a run boundary
                if (lane == 1 && given_b < n) atomicOr(&bm[given_b >> 5], 1u << (given_b & 31));
                if (lane == 2 && given_e < n) atomicOr(&bm[given_e >> 5], 1u << (given_e & 31));
            }
            wave_sync();
            RUN_STOP(41)
            {
                // a lane owns 4 * kV consecutive words of the bitmap
                uint32_t c[4 * kV];
                uint32_t tot = 0;
#pragma unroll
                for (uint32_t j = 0; j < kV; ++j) {
                    const uint4 x = ((const uint4*)bm)[kV * lane + j];
                    c[4 * j] = (uint32_t)__popc(x.x); c[4 * j + 1] = (uint32_t)__popc(x.y);
                    c[4 * j + 2] = (uint32_t)__popc(x.z); c[4 * j + 3] = (uint32_t)__popc(x.w);
                    tot += c[4 * j] + c[4 * j + 1] + c[4 * j + 2] + c[4 * j + 3];
                }
                const uint32_t incl = wave_scan_incl(tot, OpAdd());
                uint32_t run = incl - tot;
                uint32_t pk[2 * kV];
#pragma unroll
                for (uint32_t q = 0; q < 2 * kV; ++q) {
                    const uint32_t lo = run; run += c[2 * q];
                    const uint32_t hi = run; run += c[2 * q + 1];
                    pk[q] = lo | (hi << 16);
                }
#pragma unroll
                for (uint32_t j = 0; j < kV / 2; ++j)
                    ((uint4*)pref)[(kV / 2) * lane + j] = make_uint4(pk[4 * j], pk[4 * j + 1], pk[4 * j + 2], pk[4 * j + 3]);
                R = read_lane63(incl);
            }
            wave_sync();
            RUN_STOP(42)
#pragma unroll
            for (uint32_t t = 0; t < kCap / 64; ++t) {
                if (t * 64 >= n_ev) break;
                const uint32_t pos = evr[t] >> 1;
                if ((kBufferEvents ? t * 64 + lane < n_ev : evr[t] != kNone) && pos < n) {
                    const uint32_t w = pos >> 5;
                    const uint32_t k = pref[w] + (uint32_t)__popc(bm[w] & ((2u << (pos & 31)) - 1u)) - 1u;
                    rs[k] = pos;
                    const uint32_t one = 1u << (16u * (k & 1u));
                    atomicAdd(&delta[k >> 1], (evr[t] & 1u) ? 0u - one : one);
                }
            }
            if (lane == 0) { rs[0] = 0; rs[R] = n; rs[R + 1] = n; }
            if constexpr (kSens != 0) {
                if ((lane == 1 || lane == 2) && (lane == 1 ? given_b : given_e) < n) {
                    const uint32_t pos = lane == 1 ? given_b : given_e;
                    const uint32_t w = pos >> 5;
                    rs[pref[w] + (uint32_t)__popc(bm[w] & ((2u << (pos & 31)) - 1u)) - 1u] = pos;
                }
            }
            wave_sync();
            RUN_STOP(43)
            {
                const uint32_t cc = (R + 63) / 64;
                const uint32_t lo = umin(R, lane * cc), hi = umin(R, lo + cc);
                int32_t sum = 0;
                auto delta_of = [&](uint32_t k) {
                    return (int32_t)((delta[k >> 1] >> (16u * (k & 1u))) & 0xFFFFu) - 0x8000;
                };
                for (uint32_t k = lo; k < hi; ++k) sum += delta_of(k);
                int32_t cov = wave_scan_incl(sum, OpAdd()) - sum;
                for (uint32_t k = lo; k < hi; ++k) {
                    cov += delta_of(k);
                    rv[k] = (uint16_t)cov;
                }
            }
        } else {
            // events sorted ascending (value = pos << 1 | is_end) -> LDS
            uint32_t P = 64;
            while (P < n_ev) P <<= 1;
            if (P <= 512) {
                const uint32_t* gev = rev;
                if (P == 64) load_sort_store<1>(gev, n_ev, n, ev, lane);
                else if (P == 128) load_sort_store<2>(gev, n_ev, n, ev, lane);
                else if (P == 256) load_sort_store<4>(gev, n_ev, n, ev, lane);
                else load_sort_store<8>(gev, n_ev, n, ev, lane);
                wave_sync();
            } else {
                for (uint32_t k = lane; k < P; k += 64) {
                    uint32_t b = kNone;
                    if (k < n_ev) {
                        b = rev[k];
                        if ((b >> 1) > n) b = kNone;
                    }
                    ev[k] = b;
                }
                wave_sync();
                for (uint32_t k = 2; k <= P; k <<= 1) {
                    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                        for (uint32_t t = lane; t < P / 2; t += 64) {
                            const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                            const uint32_t l = i | j;
                            const uint32_t a = ev[i], b = ev[l];
                            const bool asc = (i & k) == 0;
                            if ((a > b) == asc) { ev[i] = b; ev[l] = a; }
                        }
                        wave_sync();
                    }
                }
            }
            // prefix sum of +-1 -> runs
            {
                const uint32_t c = P / 64;                   // events per lane, contiguous
                const uint32_t lo = lane * c;
                int32_t s = 0;
                uint32_t nb = 0;
                for (uint32_t k = lo; k < lo + c; ++k) {
                    const uint32_t b = ev[k];
                    if (b == kNone) break;
                    s += (b & 1) ? -1 : 1;
                    const uint32_t nx = (k + 1 < P) ? ev[k + 1] : kNone;
                    if ((b >> 1) < n && (nx == kNone || (nx >> 1) != (b >> 1))) ++nb;
                }
                const int32_t s_incl = wave_scan_incl(s, OpAdd());
                const uint32_t b_incl = wave_scan_incl(nb, OpAdd());
                const uint32_t first = ev[0];
                const uint32_t has_init = (first == kNone || (first >> 1) > 0) ? 1u : 0u;
                int32_t cov = s_incl - s;
                uint32_t w = has_init + b_incl - nb;
                for (uint32_t k = lo; k < lo + c; ++k) {
                    const uint32_t b = ev[k];
                    if (b == kNone) break;
                    cov += (b & 1) ? -1 : 1;
                    const uint32_t nx = (k + 1 < P) ? ev[k + 1] : kNone;
                    if ((b >> 1) < n && (nx == kNone || (nx >> 1) != (b >> 1))) {
                        rs[w] = b >> 1;
                        rv[w] = (uint16_t)cov;
                        ++w;
                    }
                }
                R = has_init + read_lane63(b_incl);
                if (lane == 0) {
                    if (has_init) { rs[0] = 0; rv[0] = 0; }
                    rs[R] = n;
                    rs[R + 1] = n;
                }
            }
        }
        wave_sync();
        RUN_STOP(22)

        // ---- 3. first longest streak of runs with value >= 4 ---------------------------
        uint32_t B, E, kB, kE;
        if constexpr (kSens != 0) {
            // the region is given; the marks made it a whole number of runs
            const uint32_t* bm = sm + L::X;
            const uint16_t* pref = (const uint16_t*)(bm + kBitmapBases / 32);
            auto run_at = [&](uint32_t pos) {
                const uint32_t w = pos >> 5;
                return pref[w] + (uint32_t)__popc(bm[w] & ((2u << (pos & 31)) - 1u)) - 1u;
            };
            B = given_b;
            E = umin(given_e, n);
            kB = B < n ? run_at(B) : R;
            kE = E < n ? run_at(E) : R;
        } else {
            const uint32_t c = (R + 63) / 64;
            const uint32_t lo = umin(R, lane * c), hi = umin(R, lo + c);
            uint32_t bad = 0;
            for (uint32_t k = lo; k < hi; ++k) if (rv[k] < kMinCoverage) bad = k + 1;
            uint32_t st = wave_scan_incl(bad, OpMax());
            st = lane_above(st);
            if (lane == 0) st = 0;
            uint64_t best = 0;
            uint32_t best_kb = 0, best_ke = 0;
            for (uint32_t k = lo; k < hi; ++k) {
                if (rv[k] < kMinCoverage) {
                    st = k + 1;
                } else if (k + 1 == R || rv[k + 1] < kMinCoverage) {
                    const uint32_t start = rs[st];
                    const uint64_t cand = ((uint64_t)(rs[k + 1] - start) << 32) | (uint32_t)(~start);
                    if (cand > best) { best = cand; best_kb = st; best_ke = k + 1; }
                }
            }
            const uint64_t g = wave_reduce(best, OpMax());
            if (best == g && g != 0) { sel[0] = best_kb; sel[1] = best_ke; }
            wave_sync();
            const uint32_t len = (uint32_t)(g >> 32);
            B = len ? ~(uint32_t)g : 0;
            E = B + len;
            kB = len ? sel[0] : 0;
            kE = len ? sel[1] : 0;
        }
        if (kSens == 0 && E - B < kMinRegion) {
            if (lane == 0) {
                A.alive[r] = 0;
                A.begin[r] = 0; A.end[r] = 0; A.median[r] = 0; A.p10[r] = 0;
                A.n_pits[r] = 0; A.n_hills[r] = 0; A.iv_slot[r] = kNone;
            }
            wave_sync();
            continue;
        }
        RUN_STOP(23)

        // ---- 4. Pile::shrink: zero outside; position index --------------------------------
        for (uint32_t k = lane; k < R; k += 64) {
            if (k < kB || k >= kE) rv[k] = 0;
        }
        // idx[g] = run that contains position g << shift
        uint32_t shift = 5;
        while ((n >> shift) >= L::kIdx) ++shift;
        const uint32_t ng = ((n - 1) >> shift) + 1;
        // (from the bitmap of run starts and its per-word prefix, which are still there: position
        // m << shift is the first bit of word m << (shift - 5))
        auto index_from_bitmap = [&]() {
            const uint32_t* bm = sm + L::X;
            const uint16_t* pref = (const uint16_t*)(bm + kBitmapBases / 32);
            for (uint32_t m = lane; m < ng; m += 64) {
                const uint32_t w = m << (shift - 5);
                idx[m] = (uint16_t)(pref[w] + (bm[w] & 1u) - 1u);
            }
        };
        if (bitmap_path) {
            // (short layout: behind the expansion, whose list lies there - unless there is no expansion)
            if (!kShort || kSens == 2) index_from_bitmap();
        } else {
            // number of runs j >= 1 that start at or before it: histogram of
            // ceil(start / 2^shift) over the runs, then a prefix sum
            uint32_t* gcnt = sm + L::X;                 // the sorted events are no longer needed
            for (uint32_t m = lane; m < ng; m += 64) gcnt[m] = 0;
            wave_sync();
            const uint32_t round_up = (1u << shift) - 1u;
            for (uint32_t j = 1 + lane; j < R; j += 64) {
                const uint32_t g = (rs[j] + round_up) >> shift;
                if (g < ng) atomicAdd(&gcnt[g], 1u);
            }
            wave_sync();
            const uint32_t c = (ng + 63) / 64;
            const uint32_t lo = umin(ng, lane * c), hi = umin(ng, lo + c);
            uint32_t sum = 0;
            for (uint32_t m = lo; m < hi; ++m) sum += gcnt[m];
            uint32_t run = wave_scan_incl(sum, OpAdd()) - sum;
            for (uint32_t m = lo; m < hi; ++m) {
                run += gcnt[m];
                idx[m] = (uint16_t)run;
            }
        }
        wave_sync();
        RUN_STOP(24)

        // ---- 5. expansion to HBM (Pile::add_layers' result after Pile::shrink), EARLY: the 20 KB of
        // a row take microseconds to drain; issued here they drain behind the annotation phases
        // instead of holding the wavefront's slot, LDS and registers at the end of the kernel.
        // Two things keep the compiler from parking the wave on `s_waitcnt vmcnt(0)` in front of
        // the next phase: every global load of this read is issued before this point (the row
        // offset at the top, not here) and explicitly waited for, so no load is pending on any
        // path when the stores start; stores themselves leave nothing to wait for.  (Reads that
        // are handed on to the next kernel of the chain are written again there; the 4 % of the
        // reads that allocate pool entries wait for their stores at that returning atomic.)
        // 16 bytes (8 positions) per lane and store, consecutive lanes -> consecutive addresses
        // (1 KiB per wave instruction); reads of up to 16384 positions go through
        // expand_from_bitmap on the bitmap of run starts that built the runs.  The sorted path,
        // per segment of 16384 positions: a bitmap with one bit per position that starts a run,
        // and per 32-bit word the number of run starts before it; the run of position p is then
        // pref[w] + popcount(bits up to p) - no searching, and the byte of the bitmap that
        // belongs to a lane's 8 positions says where (if anywhere) the value changes.
        if constexpr (kSens != 2) {
            __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0), nothing else
            constexpr uint32_t kSeg = 16384, kBmWords = kSeg / 32;     // (sorted path: segments of the first size)
            uint32_t* bm = sm + L::X;
            uint16_t* pref = (uint16_t*)(bm + kBmWords);
            uint4* dst = (uint4*)(A.pile + row_off);
            const uint32_t nv = (n + 7) / 8;
            if (lane == 0) rv[R] = 0;                       // padding behind the last base (rs[R] = n)
            uint32_t kbase = 0;                             // run that contains the segment's first position
            if (bitmap_path) {
                // one more bit for the padding behind the last base: run R, value 0
                if (lane == 0 && n < kBases) atomicOr(&bm[n >> 5], 1u << (n & 31));
                wave_sync();
                expand_from_bitmap<L, kRowStoreMod>(sm, rv, A.pile, row_off, nv, lane, row_stores);
                if (kShort) index_from_bitmap();
            } else
            for (uint32_t s0 = 0; s0 < nv * 8; s0 += kSeg) {       // sorted path: long reads, many events
                {
                    wave_sync();
                    ((uint4*)bm)[lane] = make_uint4(0, 0, 0, 0);
                    ((uint4*)bm)[lane + 64] = make_uint4(0, 0, 0, 0);
                    wave_sync();
                    for (uint32_t k = 1 + lane; k <= R; k += 64) {
                        const uint32_t st = rs[k] - s0;         // starts are distinct positions
                        if (st < kSeg && rs[k] > 0) atomicOr(&bm[st >> 5], 1u << (st & 31));
                    }
                    wave_sync();
                    {
                        const uint4 a = ((const uint4*)bm)[2 * lane], b = ((const uint4*)bm)[2 * lane + 1];
                        const uint32_t c[8] = {(uint32_t)__popc(a.x), (uint32_t)__popc(a.y), (uint32_t)__popc(a.z),
                                               (uint32_t)__popc(a.w), (uint32_t)__popc(b.x), (uint32_t)__popc(b.y),
                                               (uint32_t)__popc(b.z), (uint32_t)__popc(b.w)};
                        const uint32_t tot = c[0] + c[1] + c[2] + c[3] + c[4] + c[5] + c[6] + c[7];
                        const uint32_t incl = wave_scan_incl(tot, OpAdd());
                        uint32_t run = kbase + incl - tot;
                        uint32_t pk[4];
#pragma unroll
                        for (int x = 0; x < 4; ++x) {
                            const uint32_t lo = run; run += c[2 * x];
                            const uint32_t hi = run; run += c[2 * x + 1];
                            pk[x] = lo | (hi << 16);
                        }
                        ((uint4*)pref)[lane] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                        kbase += read_lane63(incl);
                    }
                    wave_sync();
                }
                const uint32_t g_lo = s0 / 8, g_hi = umin(nv, (s0 + kSeg) / 8);
                for (uint32_t g0 = g_lo + lane; g0 < g_hi; g0 += 256) {
                    uint32_t bits[4], k[4], v[4];
#pragma unroll
                    for (uint32_t u = 0; u < 4; ++u) {
                        const uint32_t g = umin(g0 + 64 * u, g_hi - 1);
                        const uint32_t w = (g * 8 - s0) >> 5;
                        bits[u] = bm[w];
                        k[u] = pref[w];
                    }
#pragma unroll
                    for (uint32_t u = 0; u < 4; ++u) {
                        const uint32_t g = umin(g0 + 64 * u, g_hi - 1);
                        const uint32_t sh = (g * 8) & 31;
                        k[u] += (uint32_t)__popc(bits[u] & ((2u << sh) - 1u));
                        bits[u] = (bits[u] >> (sh + 1)) & 0x7Fu;        // starts at positions 1 .. 7 of the group
                        v[u] = rv[k[u]];
                    }
#pragma unroll
                    for (uint32_t u = 0; u < 4; ++u) {
                        const uint32_t g = g0 + 64 * u;
                        if (g >= g_hi) break;
                        const uint32_t vv = v[u] | (v[u] << 16);
                        uint32_t w0 = vv, w1 = vv, w2 = vv, w3 = vv;
                        uint32_t inner = bits[u], kk = k[u];
                        while (inner) {
                            const uint32_t x = (uint32_t)__ffs((int)inner);     // position 1 .. 7 inside the group
                            inner &= inner - 1;
                            const uint32_t nvv = rv[++kk];
                            const uint32_t f = nvv | (nvv << 16);
                            w0 = bitfield_insert(change_mask(x, 0), f, w0);
                            w1 = bitfield_insert(change_mask(x, 1), f, w1);
                            w2 = bitfield_insert(change_mask(x, 2), f, w2);
                            w3 = bitfield_insert(change_mask(x, 3), f, w3);
                        }
                        if (row_stores) dst[g] = make_uint4(w0, w1, w2, w3);
                    }
                }
            }
        }
        wave_sync();
        if constexpr (((kVar >> 8) & 7u) != 0) {
            constexpr uint32_t kKind = (kVar >> 8) & 7u;
            uint32_t x0 = lane, x1 = lane + 1, x2 = lane + 2, x3 = lane + 3;
            if constexpr (kKind == 1 || kKind == 5) {
                asm volatile(".rept %4\n\tv_add_u32 %0, 1, %0\n\tv_add_u32 %1, 1, %1\n\tv_add_u32 %2, 1, %2\n\tv_add_u32 %3, 1, %3\n\t.endr"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "n"(kKind == 1 ? 32 : 128));
            } else if constexpr (kKind == 2) {
                uint32_t s0 = n, s1 = R, s2 = n_ev, s3 = r;
                asm volatile(".rept 32\n\ts_add_u32 %0, %0, 1\n\ts_add_u32 %1, %1, 1\n\ts_add_u32 %2, %2, 1\n\ts_add_u32 %3, %3, 1\n\t.endr"
                             : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
                x0 += s0 + s1 + s2 + s3;
            } else if constexpr (kKind == 3) {
                asm volatile("s_sleep 16" : : : "memory");
            } else if constexpr (kKind == 4) {
                uint32_t acc = 0;
                for (uint32_t t = 0; t < 32; ++t) acc += ((volatile uint32_t*)sm)[(lane + 64u * t) & 511u];
                x0 += acc;
            }
            // (the results must be wanted: a store that never happens)
            if (x0 + x1 + x2 + x3 == 0x12345u && n == 0xFFFFFFFFu) A.begin[r] = x0;
        }

        RUN_STOP(31)

        // ---- 6. order statistics over (value, length) of the runs in [kB, kE) --------------
        uint32_t med, p10;
        if constexpr (kSens == 2) {
            med = A.median[r];
            p10 = A.p10[r];
        } else {
            uint32_t* hist = sm + L::X;
            // coverage below 256 everywhere (the rule): one histogram of 256 bins answers both
            const uint32_t m = E - B;
            const uint32_t k1 = m / 2, k2 = m / 10;
            ((uint4*)hist)[lane] = make_uint4(0, 0, 0, 0);
            wave_sync();
            bool big = false;
            for (uint32_t k = kB + lane; k < kE; k += 64) {
                const uint32_t v = rv[k];
                if (v < 256) atomicAdd(&hist[v], rs[k + 1] - rs[k]);
                else big = true;
            }
            const bool any_big = __builtin_amdgcn_ballot_w64(big) != 0;
            wave_sync();
            if (!any_big) {
                const uint4 c4 = ((const uint4*)hist)[lane];
                const uint32_t c[4] = {c4.x, c4.y, c4.z, c4.w};
                const uint32_t incl = wave_scan_incl(c[0] + c[1] + c[2] + c[3], OpAdd());
                uint32_t before = incl - (c[0] + c[1] + c[2] + c[3]);
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    if (k1 >= before && k1 < before + c[b]) sel[6] = 4 * lane + b;
                    if (k2 >= before && k2 < before + c[b]) sel[7] = 4 * lane + b;
                    before += c[b];
                }
                wave_sync();
                med = sel[6];
                p10 = sel[7];
            } else {
            for (uint32_t j = lane; j < 768; j += 64) hist[j] = 0;
            wave_sync();
            for (uint32_t k = kB + lane; k < kE; k += 64) atomicAdd(&hist[rv[k] >> 8], rs[k + 1] - rs[k]);
            wave_sync();
            {
                const uint32_t c0 = hist[4 * lane], c1 = hist[4 * lane + 1], c2 = hist[4 * lane + 2],
                               c3 = hist[4 * lane + 3];
                const uint32_t incl = wave_scan_incl(c0 + c1 + c2 + c3, OpAdd());
                uint32_t before = incl - (c0 + c1 + c2 + c3);
                const uint32_t c[4] = {c0, c1, c2, c3};
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    if (k1 >= before && k1 < before + c[b]) { sel[2] = 4 * lane + b; sel[3] = k1 - before; }
                    if (k2 >= before && k2 < before + c[b]) { sel[4] = 4 * lane + b; sel[5] = k2 - before; }
                    before += c[b];
                }
            }
            wave_sync();
            const uint32_t h1 = sel[2], h2 = sel[4];
            for (uint32_t k = kB + lane; k < kE; k += 64) {
                const uint32_t v = rv[k], len = rs[k + 1] - rs[k];
                if ((v >> 8) == h1) atomicAdd(&hist[256 + (v & 255)], len);
                if ((v >> 8) == h2) atomicAdd(&hist[512 + (v & 255)], len);
            }
            wave_sync();
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                const uint32_t* hh = hist + 256 + 256 * w;
                const uint32_t kk = sel[3 + 2 * w];
                const uint32_t c0 = hh[4 * lane], c1 = hh[4 * lane + 1], c2 = hh[4 * lane + 2], c3 = hh[4 * lane + 3];
                const uint32_t incl = wave_scan_incl(c0 + c1 + c2 + c3, OpAdd());
                uint32_t before = incl - (c0 + c1 + c2 + c3);
                const uint32_t c[4] = {c0, c1, c2, c3};
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    if (kk >= before && kk < before + c[b]) sel[6 + w] = 4 * lane + b;
                    before += c[b];
                }
            }
            wave_sync();
            med = (h1 << 8) | sel[6];
            p10 = (h2 << 8) | sel[7];
            }
        }
        wave_sync();
        RUN_STOP(25)

        if constexpr (kSens == 1) {
            if (lane == 0) { A.median[r] = (uint16_t)med; A.p10[r] = (uint16_t)p10; }
        } else {
        // ---- 7. slope flags per run: a flagged prefix (down) and suffix (up) ----------------
        // down(i), i in run k  <=>  some run j < k with value > t(v_k) reaches into
        // [i-847, i-1]  <=>  i <= end_j + 846 for the nearest such j;   up(i) likewise
        // with the nearest j > k: i >= start_j - 847.   t(v) = int32(v * q)  (pile.cpp:94)
        // stored as uint16 offsets (<= 846): down = last flagged - run start, up = run end - 1 -
        // first flagged; kNone16 = nothing flagged
        // Most runs have no neighbour within the window that exceeds even the q = 1.3
        // threshold.  bm8[b] = max of the 8 runs of block b; a conservative run range of the
        // window comes from the position index; runs whose bound stays below the threshold have
        // nothing flagged, the others ("survivors") are compacted into a list - their run
        // index and four offsets are all that is kept - so that the neighbour scans below run
        // with all lanes busy.
        constexpr uint32_t kSurv = L::kSurv;
        uint16_t* surv = (uint16_t*)(sm + L::X);
        uint16_t* d13 = surv + kSurv;
        uint16_t* u13 = d13 + kSurv;
        uint16_t* d182 = u13 + kSurv;
        uint16_t* u182 = d182 + kSurv;
        uint32_t* bm8 = sm + L::X + (5 * kSurv + 1) / 2;
        for (uint32_t b = lane; b * 8 < R; b += 64) {
            uint32_t m = 0;
#pragma unroll
            for (uint32_t x = 0; x < 8; ++x) m = umax(m, (b * 8 + x) < R ? rv[b * 8 + x] : 0u);
            bm8[b] = m;
        }
        wave_sync();
        RUN_STOP(44)
        uint32_t n_surv = 0;
        for (uint32_t k0 = 0; k0 < R; k0 += 64) {
            const uint32_t k = k0 + lane;
            bool need = false;
            uint32_t sides = 0;         // bit 0 / 1: something left of the run may exceed t13 / t182; bits 2, 3: right
            if (k < R) {
                const uint32_t v = rv[k];
                const uint32_t sk = rs[k], ek = rs[k + 1];
                // == int32(v * 1.3) (sensitive pass: v * 1.42) for every uint16 v
                const int32_t t13 = kSens == 2 ? (int32_t)(v * 142u / 100u) : (int32_t)(v * 13u / 10u);
                const int32_t t182 = kSens == 2 ? t13 : (int32_t)(v * 182u / 100u);
                const uint32_t gl = (sk >= 847u ? sk - 847u : 0u) >> shift;
                const uint32_t gr = ((ek + 846u) >> shift) + 1u;
                const uint32_t jl = idx[gl];
                const uint32_t jr = gr < ng ? idx[gr] : R - 1;
                // bounds for the two sides apart (the run's own block counts for both): the scans
                // below then stop at the first neighbour above t13 unless one above t182 can exist
                const uint32_t own = k >> 3;
                int32_t ub_l = 0, ub_r = 0;
                for (uint32_t b = jl >> 3; b <= own; b += 4) {
                    const int32_t m0 = (int32_t)bm8[b], m1 = (int32_t)bm8[umin(b + 1, own)],
                                  m2 = (int32_t)bm8[umin(b + 2, own)], m3 = (int32_t)bm8[umin(b + 3, own)];
                    ub_l = max(max(ub_l, m0), max(m1, max(m2, m3)));
                }
                for (uint32_t b = own; b <= (jr >> 3); b += 4) {
                    const uint32_t last = jr >> 3;
                    const int32_t m0 = (int32_t)bm8[b], m1 = (int32_t)bm8[umin(b + 1, last)],
                                  m2 = (int32_t)bm8[umin(b + 2, last)], m3 = (int32_t)bm8[umin(b + 3, last)];
                    ub_r = max(max(ub_r, m0), max(m1, max(m2, m3)));
                }
                sides = (ub_l > t13 ? 1u : 0u) | (ub_l > t182 ? 2u : 0u) | (ub_r > t13 ? 4u : 0u) | (ub_r > t182 ? 8u : 0u);
                need = (sides & 5u) != 0;
            }
            const uint64_t m = __ballot(need);
            if (need) {
                const uint32_t w = n_surv + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                if (w < kSurv) surv[w] = (uint16_t)(k | sides << 12);
            }
            n_surv += (uint32_t)__popcll(m);
        }
        wave_sync();
        RUN_STOP(45)
        if (n_surv > kSurv) {
            // more flagged runs than this instantiation keeps: hand the read on
            if (lane == 0) overflow_list[atomicAdd(overflow_count, 1u)] = r;
            wave_sync();
            continue;
        }
        static_assert(L::kArr <= 4096, "run index and side flags share 16 bits");
        for (uint32_t j = lane; j < n_surv; j += 64) {
            const uint32_t k = surv[j] & 0xFFFu, sides = surv[j] >> 12;
            surv[j] = (uint16_t)k;
            const uint32_t v = rv[k];
            const uint32_t sk = rs[k], ek = rs[k + 1];
            // int32(v * q) of pile.cpp:94 in integers: exact for every 16-bit v (tools/threshold_check.c)
            // (the sensitive pass has one threshold, q = 1.42: both pairs of lists hold the same)
            const int32_t t13 = kSens == 2 ? (int32_t)(v * 142u / 100u) : (int32_t)(v * 13u / 10u);
            const int32_t t182 = kSens == 2 ? t13 : (int32_t)(v * 182u / 100u);
            uint32_t dl13 = kNone, dl182 = kNone, ur13 = kNone, ur182 = kNone;
            {
                bool done = !(sides & 1u);
                for (uint32_t j0 = k; j0 > 0 && !done;) {          // neighbours in batches of 4
                    uint32_t ej[4];
                    int32_t vj[4];
#pragma unroll
                    for (uint32_t u = 0; u < 4; ++u) {
                        const uint32_t j = j0 > u ? j0 - 1 - u : 0;
                        ej[u] = rs[j + 1];
                        vj[u] = rv[j];
                    }
#pragma unroll
                    for (uint32_t u = 0; u < 4; ++u) {
                        if (done || j0 <= u) { done = true; break; }
                        if (ej[u] + 846u < sk) { done = true; break; }
                        if (dl13 == kNone && vj[u] > t13) {
                            dl13 = umin(ek - 1, ej[u] + 846u);
                            if (!(sides & 2u)) { done = true; break; }
                        }
                        if (vj[u] > t182) { dl182 = umin(ek - 1, ej[u] + 846u); done = true; break; }
                    }
                    j0 = j0 > 4 ? j0 - 4 : 0;
                }
            }
            {
                bool done = !(sides & 4u);
                for (uint32_t j0 = k + 1; j0 < R && !done; j0 += 4) {
                    uint32_t sj[4];
                    int32_t vj[4];
#pragma unroll
                    for (uint32_t u = 0; u < 4; ++u) {
                        const uint32_t j = umin(j0 + u, R - 1);
                        sj[u] = rs[j];
                        vj[u] = rv[j];
                    }
#pragma unroll
                    for (uint32_t u = 0; u < 4; ++u) {
                        if (done || j0 + u >= R) { done = true; break; }
                        if (sj[u] > ek + 846u) { done = true; break; }
                        const uint32_t lim = umax(sk, sj[u] >= 847u ? sj[u] - 847u : 0u);
                        if (ur13 == kNone && vj[u] > t13) {
                            ur13 = lim;
                            if (!(sides & 8u)) { done = true; break; }
                        }
                        if (vj[u] > t182) { ur182 = lim; done = true; break; }
                    }
                }
            }
            d13[j] = dl13 == kNone ? kNone16 : (uint16_t)(dl13 - sk);
            u13[j] = ur13 == kNone ? kNone16 : (uint16_t)(ek - 1 - ur13);
            d182[j] = dl182 == kNone ? kNone16 : (uint16_t)(dl182 - sk);
            u182[j] = ur182 == kNone ? kNone16 : (uint16_t)(ek - 1 - ur182);
        }
        wave_sync();
        RUN_STOP(26)

        // ---- 8. maximal unions of touching intervals -> regions (first, last) ----------------
        // the two thresholds of a kind (down / up) in one pass: they share the run's bounds and
        // its neighbourhood on the survivor list
#pragma unroll 1
        for (uint32_t kind = 0; kind < 2; ++kind) {
            const bool is_up = kind != 0;
            constexpr uint32_t kThr = kSens == 2 ? 1 : 2;
            uint32_t ns[2] = {0, 0}, ne[2] = {0, 0};
            // a run that is not on the survivor list has nothing flagged: neighbours are the
            // adjacent list entries, if they are the adjacent runs
            for (uint32_t j0 = 0; j0 < n_surv; j0 += 64) {
                const uint32_t j = j0 + lane;
                const bool in = j < n_surv;
                uint32_t k = 0, sk = 0, ek = 0, s_prev = 0, e_next = 0;
                bool adj_prev = false, adj_next = false;
                if (in) {
                    k = surv[j];
                    sk = rs[k]; ek = rs[k + 1];
                    adj_prev = j > 0 && surv[j - 1] + 1u == k;
                    adj_next = j + 1 < n_surv && surv[j + 1] == k + 1u;
                    s_prev = adj_prev ? rs[k - 1] : 0u;
                    e_next = adj_next ? rs[k + 2] : 0u;
                }
#pragma unroll
                for (uint32_t t = 0; t < kThr; ++t) {
                    const uint32_t w = 2 * t + kind;
                    const uint16_t* iv = (w == 0) ? d13 : (w == 1) ? u13 : (w == 2) ? d182 : u182;
                    uint32_t* rf = sm + L::RF + w * kMaxReg;
                    uint32_t* rl = sm + L::RL + w * kMaxReg;
                    bool st = false, en = false;
                    uint32_t fv = 0, lv = 0;
                    const uint32_t me_iv = in ? iv[j] : kNone16;
                    if (me_iv != kNone16) {
                        const bool has_prev = adj_prev && iv[j - 1] != kNone16;
                        const bool has_next = adj_next && iv[j + 1] != kNone16;
                        if (!is_up) {
                            // interval [sk, sk + iv[j]]
                            const uint32_t last_k = sk + me_iv;
                            const bool prev_joins = has_prev && s_prev + iv[j - 1] == sk - 1;
                            const bool next_joins = last_k == ek - 1 && has_next;
                            st = !prev_joins; en = !next_joins;
                            fv = sk; lv = last_k;
                        } else {
                            // interval [ek - 1 - iv[j], ek - 1]
                            const uint32_t first_k = ek - 1 - me_iv;
                            const bool prev_joins = first_k == sk && has_prev;
                            const bool next_joins = has_next && e_next - 1 - iv[j + 1] == ek;
                            st = !prev_joins; en = !next_joins;
                            fv = first_k; lv = ek - 1;
                        }
                    }
                    const uint64_t ms = __ballot(st), me = __ballot(en);
                    const uint64_t below = (1ull << lane) - 1ull;
                    if (st) {
                        const uint32_t p = ns[t] + __popcll(ms & below);
                        if (p < kMaxReg) rf[p] = fv;
                    }
                    if (en) {
                        const uint32_t p = ne[t] + __popcll(me & below);
                        if (p < kMaxReg) rl[p] = lv;
                    }
                    ns[t] += __popcll(ms);
                    ne[t] += __popcll(me);
                }
            }
            if (lane == 0) {
                sm[L::RC + kind] = ns[0];
                if (kThr == 2) sm[L::RC + 2 + kind] = ns[1];
            }
        }
        wave_sync();
        RUN_STOP(27)

        // ---- 9. per q: merged region list; resolve / narrow only when regions interact;
        //         pits (q = 1.82) and hills (q = 1.3) ---------------------------------------------
        bool any_overflow = false;
#pragma unroll 1
        for (uint32_t which = 0; which < (kSens == 2 ? 1u : 2u); ++which) {   // 0: q = 1.3 hills, 1: q = 1.82 pits
            const double q = kSens == 2 ? 1.42 : which ? 1.82 : 1.3;
            const uint32_t nd = sm[L::RC + 2 * which], nu = sm[L::RC + 2 * which + 1];
            uint32_t* key = sm + L::REG + which * 4 * kMaxReg;
            uint32_t* last = key + 2 * kMaxReg;
            uint32_t* ivf = sm + L::IV + which * 4 * kMaxRaw;
            uint32_t* ivs = ivf + kMaxRaw;
            uint32_t* of = ivs + kMaxRaw;
            uint32_t* os = of + kMaxRaw;
            uint8_t* gone = (uint8_t*)(sm + L::GONE) + which * kMaxRaw;
            if (nd > kMaxReg || nu > kMaxReg) {
                any_overflow = true;
                if (lane == 0) sel[8 + which] = 0;
                continue;
            }
            if (nd == 0 || nu == 0) {
                // a pit is a (down, up) pair, a hill an (up, later down) pair, and regions of one
                // kind do not touch each other: nothing to resolve, nothing to find
                if (lane == 0) sel[8 + which] = 0;
                continue;
            }
            if (kSens == 0 && nd == 1 && nu == 1) {
                // the usual pile: one up region where the coverage rises at its left end, one down
                // region at its right end, far apart.  Nothing overlaps or lies within the window
                // (pile.cpp:136,177,224-231), no (down, up) pair for a pit (:357-362), and the one hill
                // candidate fails the distance test (:415-418)
                const uint32_t up_first = sm[L::RF + (2 * which + 1) * kMaxReg], up_last = sm[L::RL + (2 * which + 1) * kMaxReg];
                const uint32_t down_first = sm[L::RF + (2 * which) * kMaxReg];
                if (up_first < down_first && down_first > up_last && down_first - up_last > kSlopeWindow) {
                    if (lane == 0) sel[8 + which] = 0;
                    continue;
                }
            }
            const uint32_t nr = nd + nu;
            // merge the two sorted lists by rank (keys differ in the low bit, so no ties)
            {
                const uint32_t* df = sm + L::RF + (2 * which) * kMaxReg;
                const uint32_t* dl = sm + L::RL + (2 * which) * kMaxReg;
                const uint32_t* uf = df + kMaxReg;
                const uint32_t* ul = dl + kMaxReg;
                for (uint32_t x = lane; x < nr; x += 64) {
                    const bool is_up = x >= nd;
                    const uint32_t a = is_up ? x - nd : x;
                    const uint32_t kx = is_up ? (uf[a] << 1 | 1u) : (df[a] << 1);
                    const uint32_t lx = is_up ? ul[a] : dl[a];
                    uint32_t rank = a;
                    if (is_up) { for (uint32_t y = 0; y < nd; ++y) rank += (df[y] << 1) < kx; }
                    else       { for (uint32_t y = 0; y < nu; ++y) rank += (uf[y] << 1 | 1u) < kx; }
                    key[rank] = kx;
                    last[rank] = lx;
                }
            }
            wave_sync();
            // do neighbours overlap (pile.cpp:136,177)?  is an (up, down) pair within the window (:224-231)?
            bool need_resolve = false, need_narrow = false;
            for (uint32_t i0 = 0; i0 + 1 < nr; i0 += 64) {
                const uint32_t i = i0 + lane;
                bool c1 = false, c2 = false;
                if (i + 1 < nr) {
                    const uint32_t ki = key[i], li = last[i], kn = key[i + 1];
                    c1 = li >= (kn >> 1) && ((ki & 1) || li != (kn >> 1));
                    c2 = (ki & 1) && !(kn & 1) && (uint32_t)((kn >> 1) - li) <= kSlopeWindow;
                }
                need_resolve |= __ballot(c1) != 0;
                need_narrow |= __ballot(c2) != 0;
            }
            uint32_t n_reg = nr;
            if (need_resolve || need_narrow) {
                bool fits = true;
                uint32_t n_now = nr;
                constexpr bool kWaveResolve = 2 * kMaxReg <= 64;     // one region per lane
                if (kWaveResolve && need_resolve) fits = resolve_wave(key, last, n_now, 2 * kMaxReg, rs, rv, idx, shift, q, lane, sel + 4);
                if (lane == 0) {
                    RunCursorT<rs_t> dv{rs, rv, idx, shift, 0};
                    RegionList Rg;
                    Rg.key = key; Rg.last = last; Rg.n = n_now; Rg.cap = 2 * kMaxReg; Rg.overflow = !fits;
                    if (!kWaveResolve && need_resolve) resolve_serial(Rg, dv, q);
                    if (!Rg.overflow) narrow_serial(Rg, dv, q);
                    sel[2] = Rg.n;                       // (the order statistics are done with sel[2 ..])
                    sel[3] = Rg.overflow ? 1u : 0u;
                }
                wave_sync();
                n_reg = sel[2];
                if (sel[3]) {
                    any_overflow = true;
                    if (lane == 0) sel[8 + which] = 0;
                    wave_sync();
                    continue;
                }
            }
            if constexpr (kSens == 2) {
                // repeat hills (pile.cpp:500-566): every (up i, later down j) whose centres are
                // within 0.84 of the region; the positions between them are walked by runs, all
                // lanes on one pair at a time
                constexpr uint32_t kMaxRep = 2 * kMaxRaw;            // both halves of the interval scratch
                uint32_t* rf = sm + L::IV;
                uint32_t* rsd = rf + kMaxRep;
                uint32_t dm = A.dataset_median[r];
                if ((double)med > 1.42 * (double)dm) dm = umax(dm, p10);       // pile.cpp:503-505
                const uint32_t floor_v = (uint32_t)((double)dm * 1.42);
                const double lim = 0.84 * (double)(E - B);
                auto run_at = [&](uint32_t pos) {
                    uint32_t c = idx[pos >> shift];
                    while (rs[c + 1] <= pos) ++c;
                    return c;
                };
                uint32_t n_hit = 0;
                for (uint32_t i = 0; i + 1 < n_reg; ++i) {
                    const uint32_t ki = key[i];
                    if (!(ki & 1)) continue;
                    const uint32_t u_first = ki >> 1, u_last = last[i];
                    const uint32_t mid_u = (u_first + u_last) / 2;
                    for (uint32_t j = i + 1; j < n_reg; ++j) {
                        const uint32_t kj = key[j];
                        if (kj & 1) continue;
                        const uint32_t w_first = kj >> 1, w_last = last[j];
                        const uint32_t mid_w = (w_first + w_last) / 2;
                        if ((double)(uint32_t)(mid_w - mid_u) > lim) continue;
                        uint32_t valid = 0, found = 0;
                        if (u_last + 1 < w_first) {
                            const uint32_t ka = run_at(u_last), kb = run_at(w_first);
                            const uint32_t peak = (uint32_t)(1.42 * (double)umax(rv[ka], rv[kb]));
                            // runs that reach into (u_last, w_first)
                            for (uint32_t k = ka + lane; k <= kb; k += 64) {
                                const uint32_t lo = umax(rs[k], u_last + 1), hi = umin(rs[k + 1], w_first);
                                if (lo < hi) {
                                    const uint32_t v = rv[k];
                                    if (v > floor_v) valid += hi - lo;
                                    found |= v > peak ? 1u : 0u;
                                }
                            }
                            valid = wave_reduce(valid, OpAdd());
                            found = wave_reduce(found, OpMax());
                        }
                        if (found && !((double)valid < 0.9 * (double)(uint32_t)(w_first - u_last))) {
                            if (n_hit < kMaxRep && lane == 0) {
                                rf[n_hit] = (uint32_t)((double)u_last - 0.336 * (double)(uint32_t)(u_last - u_first));
                                rsd[n_hit] = (uint32_t)((double)w_first + 0.336 * (double)(uint32_t)(w_last - w_first));
                            }
                            ++n_hit;
                        }
                    }
                }
                if (n_hit > kMaxRep) {
                    any_overflow = true;
                    continue;
                }
                wave_sync();
                if (lane == 0) {
                    sel[8] = n_hit ? interval_merge(rf, rsd, n_hit, (uint8_t*)(sm + L::GONE), rsd + kMaxRep, rsd + 2 * kMaxRep) : 0u;
                }
                wave_sync();
                continue;
            }
            // raw pits: adjacent (down, up) (pile.cpp:357-362); raw hill candidates: every
            // (up i, later down j) that passes the cheap tests of pile.cpp:415-418
            uint32_t n_raw = 0;
            if (which) {
                for (uint32_t i0 = 0; i0 + 1 < n_reg; i0 += 64) {
                    const uint32_t i = i0 + lane;
                    const bool c = i + 1 < n_reg && !(key[i] & 1) && (key[i + 1] & 1);
                    const uint64_t m = __ballot(c);
                    if (c) {
                        const uint32_t p = n_raw + __popcll(m & ((1ull << lane) - 1ull));
                        if (p < kMaxRaw) { ivf[p] = key[i] >> 1; ivs[p] = last[i + 1]; }
                    }
                    n_raw += __popcll(m);
                }
            } else {
                uint32_t* cand = sm + L::CAND;
                const double span = (double)(E - B);
                const double lo_lim = 0.05 * span + (double)B;
                const double hi_lim = 0.95 * span + (double)B;
                const uint32_t np = n_reg * n_reg;
                for (uint32_t p0 = 0; p0 < np; p0 += 64) {
                    const uint32_t p = p0 + lane;
                    bool c = false;
                    uint32_t i = 0, j = 0;
                    if (p < np) {
                        i = p / n_reg; j = p - i * n_reg;
                        if (j > i && (key[i] & 1) && !(key[j] & 1)) {
                            const uint32_t u_first = key[i] >> 1, u_last = last[i];
                            const uint32_t w_first = key[j] >> 1, w_last = last[j];
                            c = !((double)u_first < lo_lim || (double)w_last > hi_lim ||
                                  (uint32_t)(w_first - u_last) > 840u);
                        }
                    }
                    const uint64_t m = __ballot(c);
                    if (c) {
                        const uint32_t w = n_raw + __popcll(m & ((1ull << lane) - 1ull));
                        if (w < kMaxRaw) cand[w] = i << 16 | j;
                    }
                    n_raw += __popcll(m);
                }
            }
            if (n_raw > kMaxRaw) {
                any_overflow = true;
                if (lane == 0) sel[8 + which] = 0;
                continue;
            }
            if (n_raw) {
                wave_sync();
                if (lane == 0) {
                    uint32_t cnt = n_raw;
                    if (!which) {
                        // peak test and fuzz of pile.cpp:421-448, candidates in (i, j) order
                        RunCursorT<rs_t> dv{rs, rv, idx, shift, 0};
                        const uint32_t* cand = sm + L::CAND;
                        cnt = 0;
                        for (uint32_t c = 0; c < n_raw; ++c) {
                            const uint32_t i = cand[c] >> 16, j = cand[c] & 0xFFFFu;
                            const uint32_t u_first = key[i] >> 1, u_last = last[i];
                            const uint32_t w_first = key[j] >> 1, w_last = last[j];
                            const uint32_t pk = (uint32_t)(1.3 * (double)umax(dv[u_last], dv[w_first]));
                            const bool found = u_last + 1 < w_first && range_max_runs(dv, u_last + 1, w_first - 1) > pk;
                            if (!found) continue;
                            ivf[cnt] = (uint32_t)(u_first - B) > kHillFuzz ? u_first - kHillFuzz : B;
                            ivs[cnt] = (uint32_t)(E - w_last) > kHillFuzz ? w_last + kHillFuzz : E;
                            ++cnt;
                        }
                    }
                    sel[8 + which] = interval_merge(ivf, ivs, cnt, gone, of, os);
                }
            } else if (lane == 0) {
                sel[8 + which] = 0;
            }
            wave_sync();
        }
        wave_sync();
        RUN_STOP(28)

        // ---- 10. publish ----------------------------------------------------------------------------
        if (any_overflow) {
            // more regions / raw intervals than this instantiation keeps: hand the read on
            if (lane == 0) overflow_list[atomicAdd(overflow_count, 1u)] = r;
            wave_sync();
            continue;
        }
        if constexpr (kSens == 2) {
            if (lane == 0) {
                constexpr uint32_t kMaxRep = 2 * kMaxRaw;
                const uint32_t* of = sm + L::IV + 2 * kMaxRep;
                const uint32_t* os = of + kMaxRep;
                uint32_t cnt = sel[8], slot = kNone;
                if (cnt) {
                    slot = atomicAdd(A.rep_pool_count, cnt);
                    if (slot + cnt > A.rep_pool_cap) {
                        atomicOr(A.error, kErrPoolCapacity);
                        slot = kNone; cnt = 0;
                    } else {
                        for (uint32_t k = 0; k < cnt; ++k) {
                            Interval iv;
                            iv.first = umax(B, of[k]);                  // pile.cpp:560-563
                            iv.second = umin(E, os[k]);
                            iv.aux = 0;
                            A.rep_pool[slot + k] = iv;
                        }
                    }
                }
                A.n_rep[r] = cnt;
                A.rep_slot[r] = slot;
            }
        } else if (lane == 0) {
            const uint32_t nh = sel[8], np = sel[9];
            uint32_t err = 0;
            uint32_t slot = kNone;
            uint32_t wp = err ? 0 : np, wh = err ? 0 : nh;
            if (wp + wh) {
                slot = atomicAdd(A.pool_count, wp + wh);
                if (slot + wp + wh > A.pool_cap) {
                    err |= kErrPoolCapacity;
                    slot = kNone; wp = wh = 0;
                } else {
                    RunCursorT<rs_t> dv{rs, rv, idx, shift, 0};
                    const uint32_t* pf = sm + L::IV + 4 * kMaxRaw + 2 * kMaxRaw;
                    const uint32_t* ps = pf + kMaxRaw;
                    const uint32_t* hf = sm + L::IV + 2 * kMaxRaw;
                    const uint32_t* hs = hf + kMaxRaw;
#pragma clang loop vectorize(disable) unroll(disable)
                    for (uint32_t k = 0; k < wp; ++k) {
                        uint32_t mn = 0xFFFFu;
                        const uint32_t ka = dv.run_of(pf[k]), kb = dv.run_of(ps[k]);
                        for (uint32_t x = ka; x <= kb; ++x) mn = umin(mn, rv[x]);
                        Interval iv; iv.first = pf[k]; iv.second = ps[k]; iv.aux = mn;
                        A.pool[slot + k] = iv;
                    }
                    // (vectorised and unrolled, this loop - a few iterations for one read in twenty-five - cost the kernel
                    // a spilled zero vector, and its reload a wait for the row stores in every item)
#pragma clang loop vectorize(disable) unroll(disable)
                    for (uint32_t k = 0; k < wh; ++k) {
                        Interval iv; iv.first = hf[k]; iv.second = hs[k]; iv.aux = 0;
                        A.pool[slot + wp + k] = iv;
                    }
                }
            }
            A.alive[r] = 1;
            A.begin[r] = B; A.end[r] = E;
            A.median[r] = (uint16_t)med; A.p10[r] = (uint16_t)p10;
            A.n_pits[r] = wp; A.n_hills[r] = wh;
            A.iv_slot[r] = slot;
            if (err) atomicOr(A.error, err);
        }
        }
        wave_sync();
        wave_sync();
    }
#undef RUN_STOP
}

// reads with more events than the cap-512 kernels take: listed straight from the bucket counts, so
// that their (latency-bound) kernels run beside the first kernel and not behind it
// (eight reads per thread, one add to the list's counter per workgroup: an add per wavefront that holds a dense read was 30 000
// adds to one word at C5, 0.38 ms beside the pile kernel)
constexpr uint32_t kDenseListPer = 8;
__global__ __launch_bounds__(256) void pile_dense_list_kernel(PileArgs A, uint32_t n_reads, uint32_t* list, uint32_t* count) {
    __shared__ uint32_t s_cnt, s_base;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    bool dense[kDenseListPer];
    uint32_t slot[kDenseListPer];
#pragma unroll
    for (uint32_t u = 0; u < kDenseListPer; ++u) {
        const uint32_t r = (blockIdx.x * kDenseListPer + u) * 256 + threadIdx.x;
        dense[u] = false;
        if (r < n_reads) {
            const uint32_t n_ev = A.ev_cnt ? umin(A.ev_cnt[r], A.ev_stride) : (A.ev_off[r + 1] - A.ev_off[r]) << A.ev_shift;
            dense[u] = n_ev > kRunEventCap;
        }
    }
#pragma unroll
    for (uint32_t u = 0; u < kDenseListPer; ++u) {
        const uint64_t m = __builtin_amdgcn_ballot_w64(dense[u]);
        uint32_t base = 0;
        if (m) {
            if (lane == 0) base = atomicAdd(&s_cnt, (uint32_t)__popcll(m));
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        }
        slot[u] = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    }
    __syncthreads();
    if (threadIdx.x == 0) s_base = s_cnt ? atomicAdd(count, s_cnt) : 0u;
    __syncthreads();
#pragma unroll
    for (uint32_t u = 0; u < kDenseListPer; ++u) {
        if (dense[u]) list[s_base + slot[u]] = (blockIdx.x * kDenseListPer + u) * 256 + threadIdx.x;
    }
}

void launch_pile_dense_list(const PileArgs& args, uint32_t n_reads, uint32_t* list, uint32_t* count, hipStream_t stream) {
    if (n_reads == 0) return;
    hipLaunchKernelGGL(pile_dense_list_kernel, dim3((n_reads + 256 * kDenseListPer - 1) / (256 * kDenseListPer)), dim3(256), 0, stream, args,
                       n_reads, list, count);
}

// (An EIGHTH wavefront per SIMD, round 4, timing only - the 384-event instantiation at 5 056 B and 64 registers over the
// 99.7 % of the C3 reads it can take, the 512-event kernel over the same reads: 4.37 - 4.6 ms against 4.04 - 4.09,
// docs/history/gpurun/r4_probe8.sh.  Seven is the kernel's optimum, not a limit worth lifting.)
// Reads per workgroup in the first kernel (one wavefront each).  Rounds 2 - 3: one (two and four measured no gain: 4.80 /
// 4.83 against 4.74 ms).  Round 4, once the rows start on cache-line boundaries and nothing else runs beside the kernel: two
// 4.07 - 4.10 ms, one 4.16, four 4.23 (docs/history/gpurun/r4_waves2.sh) - two.
constexpr uint32_t kPileWavesPerGroup = 2;

void launch_pile_runs(const PileArgs& args, uint32_t grid, int tier, uint32_t* overflow_list, uint32_t* overflow_count,
                      hipStream_t stream) {
    if (grid == 0) return;
    // diagnostics: extra dynamic LDS lowers the occupancy (sensitivity experiments)
    static const uint32_t extra_lds = getenv("RALA_PILE_EXTRA_LDS") ? (uint32_t)atoi(getenv("RALA_PILE_EXTRA_LDS")) : 0u;
    const bool diag = args.stop_after != 99;
#define RALA_LAUNCH_RUNS(cap, lds)                                                                                      \
    do {                                                                                                                \
        if (diag) hipLaunchKernelGGL((pile_runs_kernel<cap, true, 0>), dim3(grid), dim3(64), lds, stream, args,         \
                                     overflow_list, overflow_count);                                                    \
        else hipLaunchKernelGGL((pile_runs_kernel<cap, false, 0>), dim3(grid), dim3(64), lds, stream, args,             \
                                overflow_list, overflow_count);                                                         \
    } while (0)
    if (tier == 0 && grid >= args.n_items && !args.n_items_dev) {
        // the first kernel of the chain: one workgroup per item.  (Measured and not kept, RALA_PILE_PERSIST=<grid>:
        // persistent wavefronts, each looping over a share of the items - 6144 of them, what the chip holds at
        // once, take 6.5 ms where a million short-lived ones take 4.8; 12288: 5.4 ms.  A wavefront that goes on
        // to its next item waits for that item's first loads alone; a fresh workgroup's start overlaps with the
        // others' work.  The SQ counters' 74 % slot occupancy is not the dispatcher's doing.)
        static const uint32_t persist = getenv("RALA_PILE_PERSIST") ? (uint32_t)atoi(getenv("RALA_PILE_PERSIST")) : 0u;
        const uint32_t g = persist && persist < grid ? persist / 8u * 8u : grid;
        // reads per workgroup (one wavefront each): RALA_PILE_WAVES = 1, 2 or 4
        static const uint32_t waves = getenv("RALA_PILE_WAVES") ? (uint32_t)atoi(getenv("RALA_PILE_WAVES")) : kPileWavesPerGroup;
        if (diag) hipLaunchKernelGGL((pile_runs_kernel<kRunEventCap, true, 0, true>), dim3(g), dim3(64), extra_lds, stream,
                                     args, overflow_list, overflow_count);
        else if (waves == 4 && !persist) hipLaunchKernelGGL((pile_runs_kernel<kRunEventCap, false, 0, true, 16384, 4>), dim3((grid + 3) / 4), dim3(256),
                                                            extra_lds, stream, args, overflow_list, overflow_count);
        else if (waves == 2 && !persist && !args.order && !args.ev_cnt && !getenv("RALA_PILE_NOT_PLAIN")) {
#define RALA_LAUNCH_PRODUCT(var)                                                                                                  \
            hipLaunchKernelGGL((pile_runs_kernel<kRunEventCap, false, 0, true, 16384, 2, true, var>),                            \
                               dim3(((var) & 8192u) ? (grid + 1) / 2 : 8u * ((args.n_items + 15u) / 16u)), dim3(128),            \
                               extra_lds, stream, args, overflow_list, overflow_count)
            // the product's choice: non-temporal row stores (65536) where the rows lie in mapped chunks, plain ones (0) where one
            // hipMalloc holds them; debug_pile_variant = 262144 asks for the plain stores whatever holds the rows
            constexpr uint32_t kNt = 65536u, kForcePlain = 262144u;
            uint32_t variant = args.variant == 0 && args.rows_chunked ? kNt : args.variant == kForcePlain ? 0u : args.variant;
#ifdef RALA_PILE_AB
#ifndef RALA_PILE_AB_CASES          // (-DRALA_PILE_AB_CASES="X(8) X(11)": the variants a measurement build carries beside 0 and 65536)
#define RALA_PILE_AB_CASES X(1) X(2) X(3) X(4)
#endif
            switch (variant) {
#define X(v) case v: RALA_LAUNCH_PRODUCT(v); break;
                RALA_PILE_AB_CASES
#undef X
                case kNt: RALA_LAUNCH_PRODUCT(65536); break;
                default: RALA_LAUNCH_PRODUCT(0); break;
            }
#else
            if (variant == kNt) RALA_LAUNCH_PRODUCT(65536);
            else RALA_LAUNCH_PRODUCT(0);
#endif
#undef RALA_LAUNCH_PRODUCT
        }
        else if (waves == 2 && !persist) hipLaunchKernelGGL((pile_runs_kernel<kRunEventCap, false, 0, true, 16384, 2>), dim3((grid + 1) / 2), dim3(128),
                                                            extra_lds, stream, args, overflow_list, overflow_count);
        else hipLaunchKernelGGL((pile_runs_kernel<kRunEventCap, false, 0, true>), dim3(g), dim3(64), extra_lds, stream,
                                args, overflow_list, overflow_count);
    } else if (tier == 3) {
        // reads of 16385 .. 32768 bases: the short layout with a bitmap twice the size
        if (diag) hipLaunchKernelGGL((pile_runs_kernel<kRunEventCap, true, 0, false, 32768>), dim3(grid), dim3(64), 0, stream,
                                     args, overflow_list, overflow_count);
        else hipLaunchKernelGGL((pile_runs_kernel<kRunEventCap, false, 0, false, 32768>), dim3(grid), dim3(64), 0, stream,
                                args, overflow_list, overflow_count);
    } else if (tier == 0 || tier == 4) RALA_LAUNCH_RUNS(kRunEventCap, extra_lds);
    else if (tier == 1) RALA_LAUNCH_RUNS(kRunEventCapMid, 0);
    else RALA_LAUNCH_RUNS(kRunEventCapBig, 0);
#undef RALA_LAUNCH_RUNS
}

void launch_pile_sens(const PileArgs& args, uint32_t grid, int tier, int mode, uint32_t* overflow_list,
                      uint32_t* overflow_count, hipStream_t stream) {
    if (grid == 0) return;
#define RALA_LAUNCH_SENS(cap, m)                                                                                        \
    hipLaunchKernelGGL((pile_runs_kernel<cap, false, m>), dim3(grid), dim3(64), 0, stream, args, overflow_list,         \
                       overflow_count)
    if (tier == 0 && grid >= args.n_items && !args.n_items_dev) {
        // one workgroup per read: no loop, short layout (the sensitive pass takes reads of up to 16384 bases anyway)
        if (mode == 1) hipLaunchKernelGGL((pile_runs_kernel<kRunEventCap, false, 1, true>), dim3(grid), dim3(64), 0, stream, args,
                                          overflow_list, overflow_count);
        else hipLaunchKernelGGL((pile_runs_kernel<kRunEventCap, false, 2, true>), dim3(grid), dim3(64), 0, stream, args,
                                overflow_list, overflow_count);
    } else if (tier == 3) {
        // reads of up to 32768 bases: the short layout with the bigger bitmap, as in the first pass
        if (mode == 1) hipLaunchKernelGGL((pile_runs_kernel<kRunEventCap, false, 1, false, 32768>), dim3(grid), dim3(64), 0, stream,
                                          args, overflow_list, overflow_count);
        else hipLaunchKernelGGL((pile_runs_kernel<kRunEventCap, false, 2, false, 32768>), dim3(grid), dim3(64), 0, stream, args,
                                overflow_list, overflow_count);
    } else if (tier == 0) {
        if (mode == 1) RALA_LAUNCH_SENS(kRunEventCap, 1);
        else RALA_LAUNCH_SENS(kRunEventCap, 2);
    } else if (tier == 2) {
        if (mode == 1) RALA_LAUNCH_SENS(kRunEventCapBig, 1);
        else RALA_LAUNCH_SENS(kRunEventCapBig, 2);
    } else {
        if (mode == 1) RALA_LAUNCH_SENS(kRunEventCapMid, 1);
        else RALA_LAUNCH_SENS(kRunEventCapMid, 2);
    }
#undef RALA_LAUNCH_SENS
}

}  // namespace rala_hip
