// Second, "sensitive" pass over the piles (rala -s):
//   mode 1  Pile::add_layers on top of the existing coverage + Pile::find_median
//           for the target reads of the sensitive overlaps
//           (reference graph.cpp:917-969, pile.cpp:261-297)
//   mode 2  Pile::find_repetitive_hills for every read of a connected component
//           (graph.cpp:1006-1026, pile.cpp:500-566): find_slopes(1.42), every
//           (up, later down) pair tested, intervalMerge, clamp
// Position space: the existing pile row is loaded into LDS (zeroed outside the
// current valid region, which is what Pile::shrink left behind, pile.cpp:311-318)
// and written back.  One workgroup (256 threads) per read; long reads use an HBM slab.
#include <hip/hip_runtime.h>

#include "device_utils.h"
#include "geom.h"
#include "kernels.h"
#include "pile_common.h"

namespace rala_hip {

namespace {

constexpr int kBlock = 256;
// the lists of mode 2 at the sizes nearly every pile needs (LDS); a read that outgrows them runs again with the
// lists in global memory (RepeatArgs::big_space), grown until it fits
constexpr uint32_t kMaxRegions = 192;      // per flag-run list; twice that while the two lists are resolved
constexpr uint32_t kMaxRawIv = 64;

constexpr uint32_t SC_TMP = 0;                                 // 16 words
constexpr uint32_t SC_HIST = SC_TMP + 16;                      // 768 words
constexpr uint32_t SC_SEL = SC_HIST + 768;                     // 16 words
constexpr uint32_t SC_RCOUNT = SC_SEL + 16;                    // 2 words (+2 pad)
constexpr uint32_t SC_LISTS = SC_RCOUNT + 4;

// mode 2's lists: flag runs (2 x cap_reg firsts, lasts), the resolved list (key, last: cap_list each), raw hills
// (in first, in second, out first, out second: cap_raw each), gone bytes
struct RepLists {
    uint32_t *rfirst, *rlast, *key, *last, *iv;
    uint8_t* gone;
    uint32_t cap_reg, cap_list, cap_raw;
    static __host__ __device__ constexpr uint64_t words(uint32_t cr, uint32_t cl, uint32_t cw) {
        return 4ull * cr + 2ull * cl + 4ull * cw + (cw + 3ull) / 4;
    }
    __device__ void carve(uint32_t* base, uint32_t cr, uint32_t cl, uint32_t cw) {
        cap_reg = cr; cap_list = cl; cap_raw = cw;
        rfirst = base;
        rlast = rfirst + 2 * (size_t)cr;
        key = rlast + 2 * (size_t)cr;
        last = key + cl;
        iv = last + cl;
        gone = (uint8_t*)(iv + 4 * (size_t)cw);
    }
};
constexpr uint32_t SC_WORDS = SC_LISTS + (uint32_t)RepLists::words(kMaxRegions, 2 * kMaxRegions, kMaxRawIv);

}  // namespace

uint32_t repeats_lds_bytes(uint32_t lw) { return 3u * lw * 2u + SC_WORDS * 4u; }
uint64_t repeats_big_words(uint32_t cap_reg, uint32_t cap_list, uint32_t cap_raw) { return RepLists::words(cap_reg, cap_list, cap_raw); }

template <bool kLds, int kMode, bool kBig = false>
__global__ __launch_bounds__(kBlock) void pile_repeats_kernel(RepeatArgs A) {
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
    int32_t* diff = (int32_t*)MA;
    uint32_t* tmp32 = sc + SC_TMP;
    RepLists S;
    if constexpr (kBig) {
        S.carve(A.big_space + (size_t)blockIdx.x * RepLists::words(A.big_cap_reg, A.big_cap_list, A.big_cap_raw), A.big_cap_reg,
                A.big_cap_list, A.big_cap_raw);
    } else {
        S.carve(sc + SC_LISTS, kMaxRegions, 2 * kMaxRegions, kMaxRawIv);
    }

    for (uint32_t item = blockIdx.x; item < A.n_items; item += gridDim.x) {
        const uint32_t r = A.order[item];
        const uint32_t n = A.read_len[r];
        const uint32_t B = A.begin[r], E = A.end[r];
        const uint16_t* D = P + kPadL;
        uint16_t* Dw = P + kPadL;

        // ---- load the row, zero outside [B, E) ------------------------------------
        {
            uint32_t* P32 = (uint32_t*)P;
            for (uint32_t j = tid; j < LW / 2; j += kBlock) P32[j] = 0;
            if constexpr (kMode == 1) {
                for (uint32_t j = tid; j <= n; j += kBlock) diff[j] = 0;
            }
        }
        __syncthreads();
        {
            const uint16_t* row = A.pile + A.pile_off[r];
            for (uint32_t j = B + tid; j < E; j += kBlock) Dw[j] = row[j];
        }
        __syncthreads();

        if constexpr (kMode == 1) {
            // ---- Pile::add_layers on top (pile.cpp:274-297) ---------------------------
            const uint32_t e0 = A.ev_off[r], e1 = A.ev_off[r + 1];
            for (uint32_t k = e0 + tid; k < e1; k += kBlock) {
                const uint32_t b = A.ev[k];
                const uint32_t pos = b >> 1;
                if (pos <= n) atomicAdd(&diff[pos], (b & 1) ? -1 : 1);
            }
            __syncthreads();
            const uint32_t C = ((n + kBlock - 1) / kBlock) | 1u;
            const uint32_t lo = umin(n, (uint32_t)tid * C), hi = umin(n, lo + C);
            int32_t s = 0;
            for (uint32_t j = lo; j < hi; ++j) s += diff[j];
            int32_t total;
            int32_t run = block_scan_excl<kBlock>(s, OpAdd(), (int32_t)0, (int32_t*)tmp32, total);
            for (uint32_t j = lo; j < hi; ++j) {
                run += diff[j];
                Dw[j] = (uint16_t)(Dw[j] + (uint32_t)run);
            }
            __syncthreads();
        }
        // write the row back (mode 1: new coverage; mode 2: zeroes outside the region)
        {
            uint4* dst = (uint4*)(A.pile + A.pile_off[r]);
            const uint4* src = (const uint4*)(P + kPadL);
            const uint32_t nv = (n + 7) / 8;
            for (uint32_t j = tid; j < nv; j += kBlock) dst[j] = src[j];
        }

        if constexpr (kMode == 1) {
            // ---- Pile::find_median (pile.cpp:261-272) ------------------------------------
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
            if (tid == 0) {
                A.median[r] = (uint16_t)((h1 << 8) | sel[4]);
                A.p10[r] = (uint16_t)((h2 << 8) | sel[5]);
            }
            __syncthreads();
        } else {
            // ---- window maxima by doubling, slope flags for q = 1.42 ----------------------
            {
                const uint32_t W = LW / 2;
                const uint32_t* s32 = (const uint32_t*)P;
                uint32_t* d32 = (uint32_t*)MA;
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
            }
            const uint32_t nw = (n + 63) / 64;
            uint64_t* mask = (uint64_t*)MB;             // 2 x nw words: down, up
            {
                const uint16_t* M = MA + kPadL;
                for (uint32_t i = tid; i < nw * 64; i += kBlock) {
                    bool dn = false, up = false;
                    if (i < n) {
                        const int32_t v = D[i];
                        const int32_t lm = max((int32_t)M[(int32_t)i - 847], (int32_t)M[(int32_t)i - 512]);
                        const int32_t rm = max((int32_t)M[i + 1], (int32_t)M[i + 336]);
                        const int32_t t = (int32_t)((double)v * 1.42);
                        dn = i != 0 && lm > t;
                        up = i != n - 1 && rm > t;
                    }
                    const uint64_t b0 = __ballot(dn), b1 = __ballot(up);
                    if ((tid & 63) == 0) {
                        mask[i >> 6] = b0;
                        mask[nw + (i >> 6)] = b1;
                    }
                }
            }
            __syncthreads();
            if (wave_id() < 2) {
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

            // ---- regions resolved on one lane (pile.cpp:131-256 is a serial procedure) -----------------
            uint32_t* key = S.key;
            uint32_t* last = S.last;
            uint32_t* sel = sc + SC_SEL;
            // dataset median adjusted by the pile's own statistics (pile.cpp:503-505)
            uint32_t dm = A.dataset_median[r];
            if ((double)A.median[r] > 1.42 * (double)dm) dm = umax(dm, A.p10[r]);
            if (tid == 0) {
                RegionList R;
                R.key = key; R.last = last; R.n = 0; R.cap = S.cap_list; R.overflow = false;
                const uint32_t nd = sc[SC_RCOUNT], nu = sc[SC_RCOUNT + 1];
                bool ovf = nd > S.cap_reg || nu > S.cap_reg || (!kBig && A.force_big && A.big_list);       // (tests)
                if (!ovf) {
                    const uint32_t* df = S.rfirst;
                    const uint32_t* dl = S.rlast;
                    const uint32_t* uf = df + S.cap_reg;
                    const uint32_t* ul = dl + S.cap_reg;
                    for (uint32_t k = 0; k < nd; ++k) rl_push(R, df[k] << 1, dl[k]);
                    for (uint32_t k = 0; k < nu; ++k) rl_push(R, uf[k] << 1 | 1, ul[k]);
                    PadView dv{D};
                    resolve_and_narrow(R, dv, 1.42);
                    ovf = R.overflow;
                }
                sel[8] = ovf ? 0u : R.n;
                sel[9] = ovf ? (uint32_t)kErrRegionCapacity : 0u;
                sel[10] = 0;        // raw hills
            }
            __threadfence_block();
            __syncthreads();
            // ---- every (up, later down) pair in the reference's order (pile.cpp:515-556); the workgroup walks the
            // positions between the two regions together
            const uint32_t n_reg = sel[8];
            uint32_t* ivf = S.iv;
            uint32_t* ivs = ivf + S.cap_raw;
            uint32_t* of = ivs + S.cap_raw;
            uint32_t* os = of + S.cap_raw;
            const double lim = 0.84 * (double)(E - B);
            const uint32_t floor_v = (uint32_t)((double)dm * 1.42);
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
                    const uint32_t peak = (uint32_t)(1.42 * (double)umax(D[u_last], D[w_first]));
                    uint32_t valid = 0, found = 0;
                    for (uint32_t x = u_last + 1 + tid; x < w_first; x += kBlock) {
                        const uint32_t v = D[x];
                        valid += v > floor_v;
                        found |= v > peak;
                    }
                    valid = block_reduce<kBlock>(valid, OpAdd(), 0u, tmp32);
                    found = block_reduce<kBlock>(found, OpMax(), 0u, tmp32);
                    if (tid == 0 && found && !((double)valid < 0.9 * (double)(uint32_t)(w_first - u_last))) {
                        const uint32_t k = sel[10];
                        if (k < S.cap_raw) {
                            ivf[k] = (uint32_t)((double)u_last - 0.336 * (double)(uint32_t)(u_last - u_first));
                            ivs[k] = (uint32_t)((double)w_first + 0.336 * (double)(uint32_t)(w_last - w_first));
                        }
                        sel[10] = k + 1;
                    }
                }
            }
            __threadfence_block();
            __syncthreads();
            if (wave_id() == 0) {
                uint32_t err = sel[9];
                const uint32_t n_raw = sel[10];
                if (n_raw > S.cap_raw) err |= kErrRawCapacity;
                uint32_t cnt = 0;
                if (!err) cnt = interval_merge_wave(ivf, ivs, n_raw, S.gone, of, os);
                if (tid == 0) {
                    uint32_t slot = 0xFFFFFFFFu;
                    if (err && A.big_list) {
                        // the lists of this read do not fit: once more, with (larger) lists in global memory
                        A.big_list[atomicAdd(A.big_count, 1u)] = r;
                        if (!kBig) err = 0;      // (kBig: which of the lists the host has to grow)
                        cnt = 0;
                    } else if (cnt) {
                        slot = atomicAdd(A.pool_count, cnt);
                        if (slot + cnt > A.pool_cap || slot + cnt < slot) {
                            err |= kErrPoolCapacity;
                            slot = 0xFFFFFFFFu; cnt = 0;
                        } else {
                            for (uint32_t k = 0; k < cnt; ++k) {
                                Interval iv;
                                iv.first = umax(B, of[k]);          // pile.cpp:560-563
                                iv.second = umin(E, os[k]);
                                iv.aux = 0;
                                A.pool[slot + k] = iv;
                            }
                        }
                    }
                    A.n_rep[r] = cnt;
                    A.rep_slot[r] = slot;
                    if (err) atomicOr(A.error, err);
                }
            }
            __syncthreads();
        }
    }
}

void launch_pile_repeats(const RepeatArgs& args, uint32_t grid, bool in_lds, int mode, hipStream_t stream) {
    if (grid == 0) return;
    const uint32_t bytes = in_lds ? repeats_lds_bytes(args.lw) : SC_WORDS * 4u;
    const bool big = mode == 2 && args.big_space != nullptr;
#define RALA_REPEATS(lds, m, b)                                                                                          \
    do {                                                                                                                 \
        if (lds) hipFuncSetAttribute((const void*)pile_repeats_kernel<lds, m, b>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                     (int)bytes);                                                                        \
        hipLaunchKernelGGL((pile_repeats_kernel<lds, m, b>), dim3(grid), dim3(kBlock), bytes, stream, args);             \
    } while (0)
    if (in_lds) {
        if (mode == 1) RALA_REPEATS(true, 1, false);
        else if (big) RALA_REPEATS(true, 2, true);
        else RALA_REPEATS(true, 2, false);
    } else {
        if (mode == 1) RALA_REPEATS(false, 1, false);
        else if (big) RALA_REPEATS(false, 2, true);
        else RALA_REPEATS(false, 2, false);
    }
#undef RALA_REPEATS
}

}  // namespace rala_hip
