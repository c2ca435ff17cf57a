// Pile-o-gram construction and annotation on gfx950.
//
// One workgroup (256 threads = 4 wavefronts) owns one read.  The read's
// coverage vector ("pile", one uint16 per base) is built and kept in LDS:
//   1. bound events of the read (CSR bucket) -> LDS difference array (ds_add)
//   2. workgroup prefix sum -> uint16 coverage          [Pile::add_layers]
//   3. first longest run with coverage >= 4             [Pile::find_valid_region]
//   4. zero outside the run, stream the pile to HBM     [Pile::shrink]
//   5. median / p10 by two-level LDS radix histogram    [Pile::find_median]
//   6. window-847 maxima by doubling (packed u16 max)   [Pile::find_slopes]
//   7. slope flags for q = 1.3 and 1.82 -> ballot bitmasks -> regions
//   8. region resolution, pits and hills on two lanes   [find_chimeric_pits/hills]
// Reads too long for LDS use the same code on a per-workgroup HBM slab.
//
// Reference behaviour followed: rvaser/rala src/pile.cpp:64-455 (see
// DESIGN.md for the per-step mapping).  Integer results are bit-exact; the
// only floating point is IEEE double multiply/compare, compiled with
// -ffp-contract=off.
#include <hip/hip_runtime.h>

#include "device_utils.h"
#include "geom.h"
#include "kernels.h"
#include "pile_common.h"

namespace rala_hip {

namespace {

constexpr int kBlock = 256;

// Scratch words (uint32) behind the three big arrays.  The lists of the last phases (ListSpace) at the sizes
// nearly every pile needs; a read that outgrows them runs again with the lists in global memory.
constexpr uint32_t kMaxRegions = 192;   // per flag-run list; the two lists of a threshold together while they are resolved
constexpr uint32_t kMaxRawIv = 64;      // pits / hills before the merge
constexpr uint32_t SC_TMP = 0;                         // 16 words: scans / reductions (as u64 x 8)
constexpr uint32_t SC_HIST = SC_TMP + 16;              // 3 x 256 words
constexpr uint32_t SC_SEL = SC_HIST + 768;             // 8 words
constexpr uint32_t SC_RCOUNT = SC_SEL + 8;             // 4 words: runs per mask
constexpr uint32_t SC_OUT = SC_RCOUNT + 4;             // 8 words: n_hills, n_pits, error bits, pool slot
constexpr uint32_t SC_LISTS = SC_OUT + 8;
constexpr uint32_t SC_WORDS = SC_LISTS + (uint32_t)ListSpace::words(kMaxRegions, kMaxRegions, kMaxRawIv);

}  // namespace

uint32_t pile_lds_bytes(uint32_t lw) { return 3u * lw * 2u + SC_WORDS * 4u; }
uint32_t pile_lw_for(uint32_t n) { return (kPadL + n + 848u + 7u) & ~7u; }
uint64_t pile_big_words(uint32_t cap_reg, uint32_t cap_list, uint32_t cap_raw) { return ListSpace::words(cap_reg, cap_list, cap_raw); }

// kLds: big arrays in LDS (dynamic shared memory) or in a per-workgroup HBM slab.
// kBig: the region lists and raw intervals in global memory (A.big_space), at the sizes the host chose.
template <bool kLds, bool kBig>
__global__ __launch_bounds__(kBlock) void pile_build_annotate(PileArgs A) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x;
    const uint32_t LW = A.lw;

    uint16_t* P;
    uint32_t* sc;
    if constexpr (kLds) {
        P = (uint16_t*)smem;
        sc = (uint32_t*)(smem + 3u * LW * 2u);
    } else {
        P = A.slab + (size_t)blockIdx.x * 3u * LW;
        sc = (uint32_t*)smem;
    }
    uint16_t* MA = P + LW;
    uint16_t* MB = MA + LW;
    int32_t* diff = (int32_t*)MA;           // (n + 1) int32 <= 2 * LW uint16
    ListSpace S;
    if constexpr (kBig) {
        S.carve(A.big_space + (size_t)blockIdx.x * ListSpace::words(A.big_cap_reg, A.big_cap_list, A.big_cap_raw), A.big_cap_reg,
                A.big_cap_list, A.big_cap_raw);
    } else {
        S.carve(sc + SC_LISTS, kMaxRegions, kMaxRegions, kMaxRawIv);
    }
    uint64_t* tmp64 = (uint64_t*)(sc + SC_TMP);
    uint32_t* tmp32 = sc + SC_TMP;

#define RALA_STOP(k)                                   \
    if (A.stop_after == (k)) {                         \
        if (tid == 0) A.alive[A.order[item]] = 0;      \
        __syncthreads();                               \
        continue;                                      \
    }
    const uint32_t n_items = A.n_items_dev ? *A.n_items_dev : A.n_items;
    for (uint32_t item = blockIdx.x; item < n_items; item += gridDim.x) {
        const uint32_t r = A.order[item];
        const uint32_t n = A.read_len[r];
        const uint16_t* D = P + kPadL;      // D[j] = coverage at position j

        // ---- 0. clear ----------------------------------------------------
        {
            uint32_t* P32 = (uint32_t*)P;
            for (uint32_t j = tid; j < LW / 2; j += kBlock) P32[j] = 0;
            for (uint32_t j = tid; j <= n; j += kBlock) diff[j] = 0;
            if (tid < 8) sc[SC_OUT + tid] = 0;
        }
        __syncthreads();

        RALA_STOP(0)
        // ---- 1. bound events -> difference array --------------------------
        {
            const uint32_t n_ev = A.ev_cnt ? umin(A.ev_cnt[r], A.ev_stride) : (A.ev_off[r + 1] - A.ev_off[r]) << A.ev_shift;
            const uint32_t* __restrict__ rev = A.ev_cnt ? A.ev + (size_t)r * A.ev_stride : A.ev + ((size_t)A.ev_off[r] << A.ev_shift);
            for (uint32_t k = tid; k < n_ev; k += kBlock) {
                const uint32_t b = rev[k];
                const uint32_t pos = b >> 1;
                if (pos <= n) atomicAdd(&diff[pos], (b & 1) ? -1 : 1);
            }
        }
        __syncthreads();

        RALA_STOP(1)
        // ---- 2. prefix sum -> coverage (mod 2^16), chunk per thread ---------
        const uint32_t C = ((n + kBlock - 1) / kBlock) | 1u;    // odd: conflict-free strides
        const uint32_t lo = umin(n, (uint32_t)tid * C), hi = umin(n, lo + C);
        {
            int32_t s = 0;
            for (uint32_t j = lo; j < hi; ++j) s += diff[j];
            int32_t total;
            int32_t run = block_scan_excl<kBlock>(s, OpAdd(), (int32_t)0, (int32_t*)tmp32, total);
            uint16_t* Dw = P + kPadL;
            if (A.add_to_existing) {
                const uint16_t* old = A.pile + A.pile_off[r];
                for (uint32_t j = lo; j < hi; ++j) {
                    run += diff[j];
                    Dw[j] = (uint16_t)(old[j] + (uint32_t)run);
                }
            } else {
                for (uint32_t j = lo; j < hi; ++j) {
                    run += diff[j];
                    Dw[j] = (uint16_t)run;
                }
            }
        }
        __syncthreads();

        RALA_STOP(2)
        // ---- 3. first longest run with coverage >= 4 ------------------------
        uint32_t B, E;
        {
            uint32_t bad = 0;                       // 1 + last position < 4 in my chunk
            for (uint32_t j = lo; j < hi; ++j) {
                if (D[j] < kMinCoverage) bad = j + 1;
            }
            uint32_t tot;
            uint32_t rs = block_scan_excl<kBlock>(bad, OpMax(), 0u, tmp32, tot);
            uint64_t best = 0;                      // (len << 32) | ~start
            for (uint32_t j = lo; j < hi; ++j) {
                if (D[j] < kMinCoverage) {
                    rs = j + 1;
                } else if (j + 1 == n || D[j + 1] < kMinCoverage) {
                    const uint64_t cand = ((uint64_t)(j + 1 - rs) << 32) | (uint32_t)(~rs);
                    if (cand > best) best = cand;
                }
            }
            best = block_reduce<kBlock>(best, OpMax(), (uint64_t)0, tmp64);
            const uint32_t len = (uint32_t)(best >> 32);
            B = len ? ~(uint32_t)best : 0;
            E = B + len;
        }
        if (E - B < kMinRegion) {                   // Pile::shrink fails -> pile dropped
            if (tid == 0) {
                A.alive[r] = 0;
                A.begin[r] = 0; A.end[r] = 0; A.median[r] = 0; A.p10[r] = 0;
                A.n_pits[r] = 0; A.n_hills[r] = 0; A.iv_slot[r] = 0xFFFFFFFFu;
            }
            __syncthreads();
            continue;
        }

        RALA_STOP(3)
        // ---- 4. zero outside [B, E), stream the pile to HBM -----------------
        {
            uint16_t* Dw = P + kPadL;
            for (uint32_t j = tid; j < B; j += kBlock) Dw[j] = 0;
            for (uint32_t j = E + tid; j < n; j += kBlock) Dw[j] = 0;
        }
        __syncthreads();
        {
            uint4* dst = (uint4*)(A.pile + A.pile_off[r]);
            const uint4* src = (const uint4*)(P + kPadL);
            const uint32_t nv = (n + 7) / 8;        // pile rows are padded to 8 elements
            for (uint32_t j = tid; j < nv; j += kBlock) dst[j] = src[j];
        }

        RALA_STOP(4)
        // ---- 5. order statistics of [B, E) -----------------------------------
        uint32_t med, p10;
        {
            uint32_t* hist = sc + SC_HIST;
            uint32_t* sel = sc + SC_SEL;
            for (uint32_t j = tid; j < 768; j += kBlock) hist[j] = 0;
            __syncthreads();
            for (uint32_t j0 = B; j0 < E; j0 += kBlock) {
                const uint32_t j = j0 + tid;
                const bool in = j < E;
                hist_add(hist, in ? (uint32_t)(D[j] >> 8) : 0u, in);
            }
            __syncthreads();
            const uint32_t m = E - B;
            const uint32_t k1 = m / 2, k2 = m / 10;
            if (tid < 64) {
                // lane owns 4 consecutive bins
                const uint32_t c0 = hist[4 * tid], c1 = hist[4 * tid + 1], c2 = hist[4 * tid + 2], c3 = hist[4 * tid + 3];
                const uint32_t incl = wave_scan_incl(c0 + c1 + c2 + c3, OpAdd());
                uint32_t before = incl - (c0 + c1 + c2 + c3);
                const uint32_t c[4] = {c0, c1, c2, c3};
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    if (k1 >= before && k1 < before + c[b]) { sel[0] = 4 * tid + b; sel[1] = k1 - before; }
                    if (k2 >= before && k2 < before + c[b]) { sel[2] = 4 * tid + b; sel[3] = k2 - before; }
                    before += c[b];
                }
            }
            __syncthreads();
            const uint32_t h1 = sel[0], h2 = sel[2];
            for (uint32_t j0 = B; j0 < E; j0 += kBlock) {
                const uint32_t j = j0 + tid;
                const bool in = j < E;
                const uint32_t v = in ? D[j] : 0u;
                hist_add(hist + 256, v & 255, in && (v >> 8) == h1);
                hist_add(hist + 512, v & 255, in && (v >> 8) == h2);
            }
            __syncthreads();
            if (tid < 128) {
                const int w = tid >> 6, l = tid & 63;
                const uint32_t* hh = hist + 256 + 256 * w;
                const uint32_t kk = sel[1 + 2 * w];
                const uint32_t c0 = hh[4 * l], c1 = hh[4 * l + 1], c2 = hh[4 * l + 2], c3 = hh[4 * l + 3];
                const uint32_t incl = wave_scan_incl(c0 + c1 + c2 + c3, OpAdd());
                uint32_t before = incl - (c0 + c1 + c2 + c3);
                const uint32_t c[4] = {c0, c1, c2, c3};
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    if (kk >= before && kk < before + c[b]) sel[4 + w] = 4 * l + b;
                    before += c[b];
                }
            }
            __syncthreads();
            med = (h1 << 8) | sel[4];
            p10 = (h2 << 8) | sel[5];
        }

        RALA_STOP(5)
        // ---- 6. window maxima: M512[j] = max P[j .. j+511] by doubling --------
        {
            const uint32_t W = LW / 2;              // packed pairs
            const uint32_t* s32 = (const uint32_t*)P;
            uint32_t* d32 = (uint32_t*)MA;
            // s = 1: pair (2j, 2j+1) needs P[2j+2]
            for (uint32_t j = tid; j < W; j += kBlock) {
                const uint32_t w0 = s32[j];
                const uint32_t w1 = (j + 1 < W) ? s32[j + 1] : 0u;
                d32[j] = pk_max_u16(w0, (w0 >> 16) | (w1 << 16));
            }
            __syncthreads();
            uint32_t* a = (uint32_t*)MA;
            uint32_t* b = (uint32_t*)MB;
            for (uint32_t s = 2; s <= 256; s <<= 1) {
                const uint32_t h = s / 2;
                for (uint32_t j = tid; j < W; j += kBlock) {
                    const uint32_t w1 = (j + h < W) ? a[j + h] : 0u;
                    b[j] = pk_max_u16(a[j], w1);
                }
                __syncthreads();
                uint32_t* t = a; a = b; b = t;
            }
            // eight swaps: the result is back in MA, MB is free
        }

        RALA_STOP(6)
        // ---- 7. slope flags -> bitmasks --------------------------------------
        const uint32_t nw = (n + 63) / 64;
        uint64_t* mask = (uint64_t*)MB;             // 4 x nw words: dn1.3, up1.3, dn1.82, up1.82
        {
            const uint16_t* M = MA + kPadL;         // M[x] = max D[x .. x+511], x may be negative
            for (uint32_t i = tid; i < nw * 64; i += kBlock) {
                bool d13 = false, u13 = false, d182 = false, u182 = false;
                if (i < n) {
                    const int32_t v = D[i];
                    const int32_t lm = max((int32_t)M[(int32_t)i - 847], (int32_t)M[(int32_t)i - 512]);
                    const int32_t rm = max((int32_t)M[i + 1], (int32_t)M[i + 336]);
                    const int32_t t13 = (int32_t)((double)v * 1.3);
                    const int32_t t182 = (int32_t)((double)v * 1.82);
                    const bool nf = i != 0, nl = i != n - 1;
                    d13 = nf && lm > t13;  u13 = nl && rm > t13;
                    d182 = nf && lm > t182; u182 = nl && rm > t182;
                }
                const uint64_t b0 = __ballot(d13), b1 = __ballot(u13), b2 = __ballot(d182), b3 = __ballot(u182);
                if ((tid & 63) == 0) {
                    const uint32_t w = i >> 6;
                    mask[w] = b0; mask[nw + w] = b1; mask[2 * nw + w] = b2; mask[3 * nw + w] = b3;
                }
            }
        }
        __syncthreads();

        RALA_STOP(7)
        // ---- 8. runs of set bits -> (first, last); wave w owns mask w ---------
        {
            const int w = wave_id(), l = lane_id();
            const uint64_t* mk = mask + (size_t)w * nw;
            uint32_t* rf = S.rfirst + (size_t)w * S.cap_reg;
            uint32_t* rl = S.rlast + (size_t)w * S.cap_reg;
            uint32_t base_s = 0, base_e = 0;
            for (uint32_t w0 = 0; w0 < nw; w0 += 64) {
                const uint32_t x = w0 + l;
                uint64_t m = 0, starts = 0, ends = 0;
                if (x < nw) {
                    m = mk[x];
                    const uint64_t prev = x > 0 ? (mk[x - 1] >> 63) : 0;
                    const uint64_t next = x + 1 < nw ? (mk[x + 1] & 1) : 0;
                    starts = m & ~((m << 1) | prev);
                    ends = m & ~((m >> 1) | (next << 63));
                }
                const uint32_t cs = __popcll(starts), ce = __popcll(ends);
                const uint32_t is = wave_scan_incl(cs, OpAdd()), ie = wave_scan_incl(ce, OpAdd());
                uint32_t ps = base_s + is - cs, pe = base_e + ie - ce;
                while (starts) {
                    const uint32_t bit = __ffsll((unsigned long long)starts) - 1;
                    starts &= starts - 1;
                    if (ps < S.cap_reg) rf[ps] = x * 64 + bit;
                    ++ps;
                }
                while (ends) {
                    const uint32_t bit = __ffsll((unsigned long long)ends) - 1;
                    ends &= ends - 1;
                    if (pe < S.cap_reg) rl[pe] = x * 64 + bit;
                    ++pe;
                }
                base_s += __shfl((int)is, 63, 64);
                base_e += __shfl((int)ie, 63, 64);
            }
            if (l == 0) sc[SC_RCOUNT + w] = base_s;
        }
        __threadfence_block();
        __syncthreads();

        RALA_STOP(8)
        // ---- 9. wave 0: q = 1.3 -> hills, wave 1: q = 1.82 -> pits --------------------
        // The regions are resolved by one lane (pile.cpp:131-256 is a serial procedure); pits and hill
        // candidates are found and merged by the whole wavefront, in the reference's order.
        if (wave_id() < 2) {
            const int which = wave_id();                    // 0: q = 1.3 hills, 1: q = 1.82 pits
            const int l = lane_id();
            const double q = which ? 1.82 : 1.3;
            RegionList R;
            R.key = S.reg + (size_t)which * 2 * S.cap_list;
            R.last = R.key + S.cap_list;
            R.n = 0; R.cap = S.cap_list; R.overflow = false;
            const uint32_t nd = sc[SC_RCOUNT + 2 * which], nu = sc[SC_RCOUNT + 2 * which + 1];
            uint32_t err = 0;
            if (nd > S.cap_reg || nu > S.cap_reg || nd + nu > S.cap_list) err = kErrRegionCapacity;
            if (!kBig && A.force_big && A.big_list) err = kErrRegionCapacity;      // tests: as if the lists had overflowed
            if (!err && l == 0) {
                const uint32_t* df = S.rfirst + (size_t)(2 * which) * S.cap_reg;
                const uint32_t* dl = S.rlast + (size_t)(2 * which) * S.cap_reg;
                const uint32_t* uf = df + S.cap_reg;
                const uint32_t* ul = dl + S.cap_reg;
                for (uint32_t k = 0; k < nd; ++k) rl_push(R, df[k] << 1, dl[k]);
                for (uint32_t k = 0; k < nu; ++k) rl_push(R, uf[k] << 1 | 1, ul[k]);
                PadView dv{D};
                resolve_and_narrow(R, dv, q);
            }
            __threadfence_block();
            const uint32_t n_reg = (uint32_t)__shfl((int)R.n, 0, 64);
            if (__shfl((int)R.overflow, 0, 64)) err |= kErrRegionCapacity;
            uint32_t* ivf = S.iv + (size_t)which * 4 * S.cap_raw;
            uint32_t* ivs = ivf + S.cap_raw;
            uint32_t* of = ivs + S.cap_raw;
            uint32_t* os = of + S.cap_raw;
            uint8_t* gone = S.gone + (size_t)which * S.cap_raw;
            uint32_t cnt = 0;
            if (!err && n_reg) {
                if (which) {
                    // pile.cpp:357-363: adjacent (down, up) -> pit
                    for (uint32_t i0 = 0; i0 + 1 < n_reg; i0 += 64) {
                        const uint32_t i = i0 + l;
                        const bool in = i + 1 < n_reg;
                        const uint32_t k0 = in ? R.key[i] : 1u, k1 = in ? R.key[i + 1] : 0u;
                        const bool pit = in && !(k0 & 1) && (k1 & 1);
                        cnt = wave_append2(pit, cnt, S.cap_raw, ivf, k0 >> 1, ivs, pit ? R.last[i + 1] : 0u);
                    }
                } else {
                    // pile.cpp:411-451: every (up, later down) pair
                    const double span = (double)(E - B);
                    const double lo_lim = 0.05 * span + (double)B;
                    const double hi_lim = 0.95 * span + (double)B;
                    for (uint32_t i = 0; i + 1 < n_reg; ++i) {
                        const uint32_t ki = R.key[i];
                        if (!(ki & 1)) continue;
                        const uint32_t u_first = ki >> 1, u_last = R.last[i];
                        if ((double)u_first < lo_lim) continue;
                        for (uint32_t j0 = i + 1; j0 < n_reg; j0 += 64) {
                            const uint32_t j = j0 + l;
                            bool hill = j < n_reg;
                            uint32_t w_last = 0;
                            if (hill) {
                                const uint32_t kj = R.key[j];
                                const uint32_t w_first = kj >> 1;
                                w_last = R.last[j];
                                hill = !(kj & 1) && !((double)w_last > hi_lim) && !((uint32_t)(w_first - u_last) > 840u);
                                if (hill) {
                                    const uint32_t pk = (uint32_t)(1.3 * (double)umax(D[u_last], D[w_first]));
                                    bool found = false;
                                    for (uint32_t x = u_last + 1; x < w_first; ++x) {
                                        if (D[x] > pk) { found = true; break; }
                                    }
                                    hill = found;
                                }
                            }
                            cnt = wave_append2(hill, cnt, S.cap_raw, ivf, (uint32_t)(u_first - B) > kHillFuzz ? u_first - kHillFuzz : B,
                                               ivs, (uint32_t)(E - w_last) > kHillFuzz ? w_last + kHillFuzz : E);
                        }
                    }
                }
                if (cnt > S.cap_raw) {
                    err |= kErrRawCapacity;
                } else {
                    __threadfence_block();
                    cnt = interval_merge_wave(ivf, ivs, cnt, gone, of, os);
                }
            }
            if (l == 0) {
                sc[SC_OUT + which] = err ? 0 : cnt;
                if (err) atomicOr(&sc[SC_OUT + 2], err);
            }
        }
        __threadfence_block();
        __syncthreads();

        RALA_STOP(9)
        // ---- 10. publish ------------------------------------------------------
        {
            const uint32_t nh = sc[SC_OUT + 0], np = sc[SC_OUT + 1];
            const uint32_t lists_err = sc[SC_OUT + 2];
            if (lists_err && A.big_list) {
                // the lists of this read do not fit: once more, with (larger) lists in global memory
                if (tid == 0) {
                    A.big_list[atomicAdd(A.big_count, 1u)] = r;
                    if (kBig) atomicOr(A.error, lists_err);      // which of the lists the host has to grow
                }
                __syncthreads();
                continue;
            }
            if (tid == 0) {
                uint32_t slot = 0xFFFFFFFFu;
                if (!lists_err && np + nh) {
                    slot = atomicAdd(A.pool_count, np + nh);
                    if (slot + np + nh > A.pool_cap || slot + np + nh < slot) {
                        atomicOr(A.error, (uint32_t)kErrPoolCapacity);
                        slot = 0xFFFFFFFFu;
                    }
                }
                sc[SC_OUT + 3] = slot;
                if (lists_err) atomicOr(A.error, lists_err);
            }
            __syncthreads();
            const uint32_t slot = sc[SC_OUT + 3];
            const uint32_t wp = slot == 0xFFFFFFFFu ? 0 : np, wh = slot == 0xFFFFFFFFu ? 0 : nh;
            {
                const uint32_t* pf = S.iv + (size_t)4 * S.cap_raw + 2 * S.cap_raw;   // pits: merged out
                const uint32_t* ps = pf + S.cap_raw;
                const uint32_t* hf = S.iv + (size_t)2 * S.cap_raw;                   // hills: merged out
                const uint32_t* hs = hf + S.cap_raw;
                for (uint32_t k = tid; k < wp; k += kBlock) {
                    uint32_t mn = 0xFFFFu;
                    for (uint32_t x = pf[k]; x <= ps[k]; ++x) mn = umin(mn, D[x]);
                    Interval iv; iv.first = pf[k]; iv.second = ps[k]; iv.aux = mn;
                    A.pool[slot + k] = iv;
                }
                for (uint32_t k = tid; k < wh; k += kBlock) {
                    Interval iv; iv.first = hf[k]; iv.second = hs[k]; iv.aux = 0;
                    A.pool[slot + wp + k] = iv;
                }
            }
            if (tid == 0) {
                A.alive[r] = 1;
                A.begin[r] = B; A.end[r] = E;
                A.median[r] = (uint16_t)med; A.p10[r] = (uint16_t)p10;
                A.n_pits[r] = wp; A.n_hills[r] = wh;
                A.iv_slot[r] = slot;
            }
        }
        __syncthreads();
    }
}

void launch_pile_build_annotate(const PileArgs& args, uint32_t grid, bool in_lds, hipStream_t stream) {
    if (grid == 0) return;
    const bool big = args.big_space != nullptr;
    if (in_lds) {
        const uint32_t bytes = pile_lds_bytes(args.lw);
        if (big) {
            hipFuncSetAttribute((const void*)pile_build_annotate<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
            hipLaunchKernelGGL((pile_build_annotate<true, true>), dim3(grid), dim3(kBlock), bytes, stream, args);
        } else {
            hipFuncSetAttribute((const void*)pile_build_annotate<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
            hipLaunchKernelGGL((pile_build_annotate<true, false>), dim3(grid), dim3(kBlock), bytes, stream, args);
        }
    } else if (big) {
        hipLaunchKernelGGL((pile_build_annotate<false, true>), dim3(grid), dim3(kBlock), SC_WORDS * 4u, stream, args);
    } else {
        hipLaunchKernelGGL((pile_build_annotate<false, false>), dim3(grid), dim3(kBlock), SC_WORDS * 4u, stream, args);
    }
}

}  // namespace rala_hip
