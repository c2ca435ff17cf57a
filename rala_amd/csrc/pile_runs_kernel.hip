// Pile-o-gram construction and annotation in RUN space, one wavefront per read.
//
// A pile is a step function: coverage only changes at bound events.  With the
// E events of a read sorted (LDS bitonic sort), the prefix sum of +-1 gives
// R <= E + 1 runs (start, value).  Every per-base loop of the reference then
// becomes a loop over runs:
//   * Pile::add_layers        sort + wave prefix sum               O(E log^2 E)
//   * Pile::find_valid_region  streaks of runs with value >= 4      O(R)
//   * Pile::shrink             zero the runs outside the streak; the pile is
//                              expanded once, 16 B per lane, straight to HBM
//   * Pile::find_median        radix select over (value, length)    O(R)
//   * Pile::find_slopes flags  within a run the window maximum only has to be
//       compared with ONE threshold, so the flagged positions of a run are a
//       prefix (down) and a suffix (up) of it, bounded by the nearest run to
//       the left / right whose value exceeds the threshold: O(runs per window)
//   * region resolution / narrowing / pits / hills: the reference's loops,
//       reading coverage through a run cursor (one lane; regions are few)
// Reads with more than kCap events are appended to an overflow list and
// processed by the position-space kernel (pile_kernels.hip).
//
// Reference behaviour followed: rvaser/rala src/pile.cpp:64-455.
#include <hip/hip_runtime.h>

#include "device_utils.h"
#include "geom.h"
#include "kernels.h"

namespace rala_hip {

namespace {

constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr uint32_t kMaxReg = 64;       // regions per slope list
constexpr uint32_t kMaxRaw = 32;       // pits / hills before the merge
constexpr uint32_t kIdx = 512;         // entries of the position -> run index

__device__ __forceinline__ void wave_sync() { __syncthreads(); }   // workgroup == one wavefront

struct RunCursor {
    const uint32_t* rs;     // run starts, rs[R] = n
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

struct RegionList {
    uint32_t* key;
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

// pile.cpp:131-256 (same procedure as pile_kernels.hip, coverage through the cursor)
__device__ void resolve_and_narrow(RegionList& R, RunCursor& d, double q) {
    if (R.n == 0) return;
    for (;;) {
        rl_sort(R);
        bool changed = false;
        for (uint32_t i = 0; i + 1 < R.n; ++i) {
            if (R.last[i] < (R.key[i + 1] >> 1)) continue;
            if (R.key[i] & 1) {
                const uint32_t s = R.key[i] >> 1;
                const uint32_t e = umin(R.last[i], R.last[i + 1]);
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

template <uint32_t kCap>
struct Layout {
    static constexpr uint32_t kArr = kCap + 4;                  // words per run-indexed array
    static constexpr uint32_t A = 0;                            // events (sort) -> down end, q = 1.3
    static constexpr uint32_t RS = A + kArr;                    // run starts (+ sentinel)
    static constexpr uint32_t D = RS + kArr;                    // histograms (D..F) -> up start 1.3
    static constexpr uint32_t E = D + kArr;                     // down end 1.82
    static constexpr uint32_t F = E + kArr;                     // up start 1.82
    static constexpr uint32_t RV = F + kArr;                    // run values, uint16 (kArr / 2 words)
    static constexpr uint32_t IDX = RV + kArr / 2;              // kIdx uint16
    static constexpr uint32_t RF = IDX + kIdx / 2;              // 4 lists x kMaxReg firsts
    static constexpr uint32_t RL = RF + 4 * kMaxReg;            // 4 lists x kMaxReg lasts
    static constexpr uint32_t RC = RL + 4 * kMaxReg;            // 4 counts
    static constexpr uint32_t REG = RC + 4;                     // 2 x (key, last) x 2 * kMaxReg
    static constexpr uint32_t IV = REG + 8 * kMaxReg;           // 2 x 4 x kMaxRaw
    static constexpr uint32_t GONE = IV + 8 * kMaxRaw;          // 2 x kMaxRaw bytes
    static constexpr uint32_t SEL = GONE + (2 * kMaxRaw) / 4;   // 16 words
    static constexpr uint32_t WORDS = SEL + 16;
    static_assert(3 * kArr >= 768, "histograms must fit in arrays D..F");
};

}  // namespace

template <uint32_t kCap>
__global__ __launch_bounds__(64) void pile_runs_kernel(PileArgs A, uint32_t* overflow_list, uint32_t* overflow_count) {
    typedef Layout<kCap> L;
    __shared__ __align__(16) uint32_t sm[L::WORDS];
    const uint32_t lane = threadIdx.x;
    uint32_t* ev = sm + L::A;
    uint32_t* rs = sm + L::RS;
    uint16_t* rv = (uint16_t*)(sm + L::RV);
    uint16_t* idx = (uint16_t*)(sm + L::IDX);
    uint32_t* sel = sm + L::SEL;

    for (uint32_t item = blockIdx.x; item < A.n_items; item += gridDim.x) {
        const uint32_t r = A.order ? A.order[item] : item;
        const uint32_t n = A.read_len[r];
        const uint32_t e0 = A.ev_off[r];
        const uint32_t n_ev = A.ev_off[r + 1] - e0;
        if (n_ev > kCap) {
            if (lane == 0) overflow_list[atomicAdd(overflow_count, 1u)] = r;
            continue;
        }

        // ---- 1. events -> LDS, sorted ascending (value = pos << 1 | is_end) -------
        uint32_t P = 64;
        while (P < n_ev) P <<= 1;
        for (uint32_t k = lane; k < P; k += 64) {
            uint32_t b = kNone;
            if (k < n_ev) {
                b = A.ev[e0 + k];
                if ((b >> 1) > n) b = kNone;        // outside the read (undefined in the reference)
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

        // ---- 2. prefix sum of +-1 -> runs (start, value mod 2^16) --------------------
        uint32_t R;
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
            R = has_init + (uint32_t)__shfl((int)b_incl, 63, 64);
            if (lane == 0) {
                if (has_init) { rs[0] = 0; rv[0] = 0; }
                rs[R] = n;
                rs[R + 1] = n;
            }
        }
        wave_sync();

        // ---- 3. first longest streak of runs with value >= 4 ---------------------------
        uint32_t B, E, kB, kE;
        {
            const uint32_t c = (R + 63) / 64;
            const uint32_t lo = umin(R, lane * c), hi = umin(R, lo + c);
            uint32_t bad = 0;
            for (uint32_t k = lo; k < hi; ++k) if (rv[k] < kMinCoverage) bad = k + 1;
            uint32_t st = wave_scan_incl(bad, OpMax());
            st = shfl_up_t(st, 1);
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
        if (E - B < kMinRegion) {
            if (lane == 0) {
                A.alive[r] = 0;
                A.begin[r] = 0; A.end[r] = 0; A.median[r] = 0; A.p10[r] = 0;
                A.n_pits[r] = 0; A.n_hills[r] = 0; A.iv_slot[r] = kNone;
            }
            wave_sync();
            continue;
        }

        // ---- 4. Pile::shrink: zero outside; position index; expand to HBM -----------------
        for (uint32_t k = lane; k < R; k += 64) {
            if (k < kB || k >= kE) rv[k] = 0;
        }
        uint32_t shift = 6;
        while ((n >> shift) >= kIdx) ++shift;
        wave_sync();
        for (uint32_t m = lane; m <= (n - 1) >> shift; m += 64) {
            const uint32_t p = m << shift;
            uint32_t a = 0, b = R - 1;                  // last run with start <= p
            while (a < b) {
                const uint32_t mid = (a + b + 1) >> 1;
                if (rs[mid] <= p) a = mid; else b = mid - 1;
            }
            idx[m] = (uint16_t)a;
        }
        wave_sync();
        {
            uint4* dst = (uint4*)(A.pile + A.pile_off[r]);
            const uint32_t nv = (n + 7) / 8;
            for (uint32_t g = lane; g < nv; g += 64) {
                const uint32_t p = g * 8;
                uint32_t k = idx[p >> shift];
                uint32_t nxt = rs[k + 1];
                while (nxt <= p) { ++k; nxt = rs[k + 1]; }
                uint32_t v = rv[k];
                uint32_t w[4];
#pragma unroll
                for (int x = 0; x < 8; ++x) {
                    const uint32_t q = p + x;
                    while (nxt <= q && k + 1 < R) { ++k; nxt = rs[k + 1]; v = rv[k]; }
                    const uint32_t val = q < n ? v : 0u;
                    if (x & 1) w[x >> 1] |= val << 16; else w[x >> 1] = val;
                }
                dst[g] = make_uint4(w[0], w[1], w[2], w[3]);
            }
        }

        // ---- 5. order statistics over (value, length) of the runs in [kB, kE) --------------
        uint32_t med, p10;
        {
            uint32_t* hist = sm + L::D;
            for (uint32_t j = lane; j < 768; j += 64) hist[j] = 0;
            wave_sync();
            for (uint32_t k = kB + lane; k < kE; k += 64) atomicAdd(&hist[rv[k] >> 8], rs[k + 1] - rs[k]);
            wave_sync();
            const uint32_t m = E - B;
            const uint32_t k1 = m / 2, k2 = m / 10;
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
        wave_sync();

        // ---- 6. slope flags per run: a flagged prefix (down) and suffix (up) ----------------
        // down(i), i in run k  <=>  some run j < k with value > t(v_k) reaches into
        // [i-847, i-1]  <=>  i <= end_j + 846 for the nearest such j;   up(i) likewise
        // with the nearest j > k: i >= start_j - 847.   t(v) = int32(v * q)  (pile.cpp:94)
        uint32_t* d13 = sm + L::A;
        uint32_t* u13 = sm + L::D;
        uint32_t* d182 = sm + L::E;
        uint32_t* u182 = sm + L::F;
        for (uint32_t k = lane; k < R; k += 64) {
            const uint32_t v = rv[k];
            const uint32_t sk = rs[k], ek = rs[k + 1];
            const int32_t t13 = (int32_t)((double)v * 1.3), t182 = (int32_t)((double)v * 1.82);
            uint32_t dl13 = kNone, dl182 = kNone, ur13 = kNone, ur182 = kNone;
            for (uint32_t j = k; j-- > 0;) {
                const uint32_t ej = rs[j + 1];
                if (ej + 846u < sk) break;
                const int32_t vj = rv[j];
                if (dl13 == kNone && vj > t13) dl13 = umin(ek - 1, ej + 846u);
                if (vj > t182) { dl182 = umin(ek - 1, ej + 846u); break; }
            }
            for (uint32_t j = k + 1; j < R; ++j) {
                const uint32_t sj = rs[j];
                if (sj > ek + 846u) break;
                const int32_t vj = rv[j];
                if (ur13 == kNone && vj > t13) ur13 = umax(sk, sj >= 847u ? sj - 847u : 0u);
                if (vj > t182) { ur182 = umax(sk, sj >= 847u ? sj - 847u : 0u); break; }
            }
            d13[k] = dl13; u13[k] = ur13; d182[k] = dl182; u182[k] = ur182;
        }
        wave_sync();

        // ---- 7. maximal unions of touching intervals -> regions (first, last) ----------------
#pragma unroll 1
        for (uint32_t w = 0; w < 4; ++w) {
            const uint32_t* iv = (w == 0) ? d13 : (w == 1) ? u13 : (w == 2) ? d182 : u182;
            const bool is_up = w & 1;
            uint32_t* rf = sm + L::RF + w * kMaxReg;
            uint32_t* rl = sm + L::RL + w * kMaxReg;
            uint32_t ns = 0, ne = 0;
            for (uint32_t k0 = 0; k0 < R; k0 += 64) {
                const uint32_t k = k0 + lane;
                bool st = false, en = false;
                uint32_t fv = 0, lv = 0;
                if (k < R && iv[k] != kNone) {
                    const uint32_t sk = rs[k], ek = rs[k + 1];
                    if (!is_up) {
                        // interval [sk, iv[k]]
                        const bool prev_joins = k > 0 && iv[k - 1] != kNone && iv[k - 1] == sk - 1;
                        const bool next_joins = iv[k] == ek - 1 && k + 1 < R && iv[k + 1] != kNone;
                        st = !prev_joins; en = !next_joins;
                        fv = sk; lv = iv[k];
                    } else {
                        // interval [iv[k], ek - 1]
                        const bool prev_joins = iv[k] == sk && k > 0 && iv[k - 1] != kNone;
                        const bool next_joins = k + 1 < R && iv[k + 1] != kNone && iv[k + 1] == ek;
                        st = !prev_joins; en = !next_joins;
                        fv = iv[k]; lv = ek - 1;
                    }
                }
                const uint64_t ms = __ballot(st), me = __ballot(en);
                const uint64_t below = (1ull << lane) - 1ull;
                if (st) {
                    const uint32_t p = ns + __popcll(ms & below);
                    if (p < kMaxReg) rf[p] = fv;
                }
                if (en) {
                    const uint32_t p = ne + __popcll(me & below);
                    if (p < kMaxReg) rl[p] = lv;
                }
                ns += __popcll(ms);
                ne += __popcll(me);
            }
            if (lane == 0) sm[L::RC + w] = ns;
        }
        wave_sync();

        // ---- 8. resolve, pits (q = 1.82), hills (q = 1.3); lanes 0 and 1 -----------------------
        if (lane < 2) {
            const uint32_t which = lane;                    // 0: hills, 1: pits
            const double q = which ? 1.82 : 1.3;
            RunCursor dv{rs, rv, idx, shift, 0};
            RegionList Rg;
            Rg.key = sm + L::REG + which * 4 * kMaxReg;
            Rg.last = Rg.key + 2 * kMaxReg;
            Rg.n = 0; Rg.cap = 2 * kMaxReg; Rg.overflow = false;
            const uint32_t nd = sm[L::RC + 2 * which], nu = sm[L::RC + 2 * which + 1];
            if (nd > kMaxReg || nu > kMaxReg) Rg.overflow = true;
            if (!Rg.overflow) {
                const uint32_t* df = sm + L::RF + (2 * which) * kMaxReg;
                const uint32_t* dl = sm + L::RL + (2 * which) * kMaxReg;
                const uint32_t* uf = df + kMaxReg;
                const uint32_t* ul = dl + kMaxReg;
                for (uint32_t k = 0; k < nd; ++k) rl_push(Rg, df[k] << 1, dl[k]);
                for (uint32_t k = 0; k < nu; ++k) rl_push(Rg, uf[k] << 1 | 1, ul[k]);
                resolve_and_narrow(Rg, dv, q);
            }
            uint32_t* ivf = sm + L::IV + which * 4 * kMaxRaw;
            uint32_t* ivs = ivf + kMaxRaw;
            uint32_t* of = ivs + kMaxRaw;
            uint32_t* os = of + kMaxRaw;
            uint8_t* gone = (uint8_t*)(sm + L::GONE) + which * kMaxRaw;
            uint32_t cnt = 0;
            bool ovf = Rg.overflow;
            if (!ovf && Rg.n) {
                if (which) {
                    for (uint32_t i = 0; i + 1 < Rg.n; ++i) {
                        if (!(Rg.key[i] & 1) && (Rg.key[i + 1] & 1)) {
                            if (cnt >= kMaxRaw) { ovf = true; break; }
                            ivf[cnt] = Rg.key[i] >> 1;
                            ivs[cnt] = Rg.last[i + 1];
                            ++cnt;
                        }
                    }
                } else {
                    const double span = (double)(E - B);
                    const double lo_lim = 0.05 * span + (double)B;
                    const double hi_lim = 0.95 * span + (double)B;
                    for (uint32_t i = 0; i + 1 < Rg.n && !ovf; ++i) {
                        if (!(Rg.key[i] & 1)) continue;
                        const uint32_t u_first = Rg.key[i] >> 1, u_last = Rg.last[i];
                        for (uint32_t j = i + 1; j < Rg.n; ++j) {
                            if (Rg.key[j] & 1) continue;
                            const uint32_t w_first = Rg.key[j] >> 1, w_last = Rg.last[j];
                            if ((double)u_first < lo_lim || (double)w_last > hi_lim ||
                                (uint32_t)(w_first - u_last) > 840u) {
                                continue;
                            }
                            const uint32_t pk = (uint32_t)(1.3 * (double)umax(dv[u_last], dv[w_first]));
                            bool found = false;
                            for (uint32_t x = u_last + 1; x < w_first; ++x) {
                                if (dv[x] > pk) { found = true; break; }
                            }
                            if (!found) continue;
                            if (cnt >= kMaxRaw) { ovf = true; break; }
                            ivf[cnt] = (uint32_t)(u_first - B) > kHillFuzz ? u_first - kHillFuzz : B;
                            ivs[cnt] = (uint32_t)(E - w_last) > kHillFuzz ? w_last + kHillFuzz : E;
                            ++cnt;
                        }
                    }
                }
                if (!ovf) cnt = interval_merge(ivf, ivs, cnt, gone, of, os);
            }
            sel[8 + which] = ovf ? 0 : cnt;
            sel[10 + which] = ovf ? 1u : 0u;
        }
        wave_sync();

        // ---- 9. publish ----------------------------------------------------------------------------
        if (lane == 0) {
            const uint32_t nh = sel[8], np = sel[9];
            uint32_t err = (sel[10] | sel[11]) ? kErrRegionCapacity : 0;
            uint32_t slot = kNone;
            uint32_t wp = err ? 0 : np, wh = err ? 0 : nh;
            if (wp + wh) {
                slot = atomicAdd(A.pool_count, wp + wh);
                if (slot + wp + wh > A.pool_cap) {
                    err |= kErrPoolCapacity;
                    slot = kNone; wp = wh = 0;
                } else {
                    RunCursor dv{rs, rv, idx, shift, 0};
                    const uint32_t* pf = sm + L::IV + 4 * kMaxRaw + 2 * kMaxRaw;
                    const uint32_t* ps = pf + kMaxRaw;
                    const uint32_t* hf = sm + L::IV + 2 * kMaxRaw;
                    const uint32_t* hs = hf + kMaxRaw;
                    for (uint32_t k = 0; k < wp; ++k) {
                        uint32_t mn = 0xFFFFu;
                        const uint32_t ka = dv.run_of(pf[k]), kb = dv.run_of(ps[k]);
                        for (uint32_t x = ka; x <= kb; ++x) mn = umin(mn, rv[x]);
                        Interval iv; iv.first = pf[k]; iv.second = ps[k]; iv.aux = mn;
                        A.pool[slot + k] = iv;
                    }
                    for (uint32_t k = 0; k < wh; ++k) {
                        Interval iv; iv.first = hf[k]; iv.second = hs[k]; iv.aux = 0;
                        A.pool[slot + wp + k] = iv;
                    }
                }
            }
            A.alive[r] = 1;
            A.begin[r] = B; A.end[r] = E;
            A.median[r] = (uint16_t)med; A.p10[r] = (uint16_t)p10;
            A.n_pits[r] = (uint8_t)wp; A.n_hills[r] = (uint8_t)wh;
            A.iv_slot[r] = slot;
            if (err) atomicOr(A.error, err);
        }
        wave_sync();
    }
}

void launch_pile_runs(const PileArgs& args, uint32_t grid, uint32_t* overflow_list, uint32_t* overflow_count,
                      hipStream_t stream) {
    if (grid == 0) return;
    hipLaunchKernelGGL(pile_runs_kernel<kRunEventCap>, dim3(grid), dim3(64), 0, stream, args, overflow_list,
                       overflow_count);
}

}  // namespace rala_hip
