// Graph::preprocess (chimeras) and graph construction on device-resident survivor lists.
//
// After the second overlap pass a few per cent of the overlaps are left (state 1 =
// "overlaps", state 2 = "internals").  The reference then iterates
//   break over hills -> re-trim -> { components, component median, break over pits,
//   re-trim, promote internals that became dovetails } until no overlap dies
//   -> in-order containment removal -> nodes / edges
// (src/graph.cpp:699-880, 553-632).  Here every per-item step is a kernel over the
// list (items are never moved: dead items keep state 0, promoted internals get state 3
// and the round in which they were promoted, which fixes their place in the reference's
// list order); the in-order containment removal is the same fixed point as in the second
// pass; only the per-component medians are computed on the host (from the labels).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "device_utils.h"
#include "geom.h"
#include "kernels.h"
#include "scan_pass.h"

namespace rala_hip {

namespace {

constexpr int kBlock = 256;
constexpr uint32_t kInf = 0xFFFFFFFFu;

inline dim3 grid_for(uint64_t n) { return dim3((unsigned)((n + kBlock - 1) / kBlock)); }

// Pile::shrink without the data part (pile.cpp:299-322); marks the read dirty
__device__ __forceinline__ bool dev_shrink(const TailReads& R, uint32_t r, uint32_t b, uint32_t e) {
    if (b > e || e - b < kMinRegion) { R.dirty[r] = 1; return false; }
    if (R.begin[r] != b || R.end[r] != e) R.dirty[r] = 1;
    R.begin[r] = b;
    R.end[r] = e;
    return true;
}

// Pile::break_over_chimeric_hills (pile.cpp:471-498), one thread per read
__global__ __launch_bounds__(kBlock) void break_hills_kernel(TailReads R, uint32_t n_reads) {
    const uint32_t r = blockIdx.x * kBlock + threadIdx.x;
    if (r >= n_reads || !R.alive[r] || R.n_hills[r] == 0) return;
    const Interval* hills = R.pool + R.iv_slot[r] + R.n_pits0[r];
    // a hill that more than three overlaps span is no chimera (pile.cpp:480)
    const Piece keep = longest_piece(R.begin[r], R.end[r], R.n_hills[r],
                                     [&](uint32_t i, uint32_t& f, uint32_t& s) { f = hills[i].first; s = hills[i].second; },
                                     [&](uint32_t i) { return hills[i].aux <= 3; });
    if (!dev_shrink(R, r, keep.begin, keep.end)) R.alive[r] = 0;
    R.n_hills[r] = 0;
}

// Pile::break_over_chimeric_pits (pile.cpp:366-402) for the reads of a component; the pile
// kernel recorded the minimum coverage inside each pit (data * 1.84 <= median is monotone)
// gate (may be null): the count of overlaps the round before dropped - a round enqueued ahead of the host's
// look at that count does nothing when it was zero (the reference's loop has ended there, graph.cpp:826-828)
__global__ __launch_bounds__(kBlock) void break_pits_kernel(TailReads R, const uint32_t* __restrict__ alive_reads,
                                                            const uint8_t* __restrict__ touched,
                                                            const uint16_t* __restrict__ comp_median, uint32_t n_alive,
                                                            const uint32_t* gate) {
    if (gate && *gate == 0) return;
    const uint32_t q = blockIdx.x * kBlock + threadIdx.x;
    if (q >= n_alive || !touched[q]) return;
    const uint32_t r = alive_reads[q];
    if (!R.alive[r] || R.n_pits[r] == 0) return;
    Interval* pits = R.pool + R.iv_slot[r];
    const double med = (double)comp_median[q];
    uint32_t w = 0;
    // a pit is real when some coverage inside it is at most median / 1.84 (pile.cpp:370: its minimum decides); the others
    // stay on the list, in place
    const Piece keep = longest_piece(R.begin[r], R.end[r], R.n_pits[r],
                                     [&](uint32_t k, uint32_t& f, uint32_t& s) { f = pits[k].first; s = pits[k].second; },
                                     [&](uint32_t k) {
                                         if ((double)pits[k].aux * 1.84 <= med) return true;
                                         pits[w++] = pits[k];
                                         return false;
                                     });
    R.n_pits[r] = w;
    if (!dev_shrink(R, r, keep.begin, keep.end)) R.alive[r] = 0;
}

__device__ __forceinline__ Coords item_coords(const TailList& L, uint32_t k) {
    Coords c;
    c.a_begin = L.a_begin[k]; c.a_end = L.a_end[k]; c.b_begin = L.b_begin[k]; c.b_end = L.b_end[k];
    c.length = L.length[k];
    return c;
}

// Overlap::trim for the items that touch a dirty read (graph.cpp:722-736, 801-824): dropped
// overlaps are counted (the loop continues while any died), internals that became dovetails
// are promoted
__global__ __launch_bounds__(kBlock) void retrim_kernel(TailList L, TailReads R, uint32_t promote, uint32_t round,
                                                        uint32_t* dropped, const uint32_t* gate) {
    if (gate && *gate == 0) return;
    const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
    bool died = false;
    if (k < L.n) {
        const uint8_t st = L.state[k];
        if (st != 0) {
            const uint32_t a = L.a[k], b = L.b[k];
            // untouched items are still trimmed (and internals still kX); an internal whose type went
            // stale in the first re-trim (no promotion there, graph.cpp:730-736) is looked at again
            const bool stale_internal = promote && st == 2 && L.type[k] == 255;
            if ((R.dirty[a] | R.dirty[b]) || stale_internal) {
                Coords c = item_coords(L, k);
                const uint32_t strand = L.strand[k];
                const bool ok = R.alive[a] && R.alive[b] && ovl_trim(c, strand, R.begin[a], R.end[a], R.begin[b], R.end[b]);
                if (!ok) {
                    L.state[k] = 0;
                    died = st != 2;
                } else {
                    L.a_begin[k] = c.a_begin; L.a_end[k] = c.a_end; L.b_begin[k] = c.b_begin; L.b_end[k] = c.b_end;
                    L.length[k] = c.length;
                    uint8_t t = 255;
                    if (st == 2 && promote) {
                        t = (uint8_t)ovl_type(c, strand, R.begin[a], R.end[a], R.begin[b], R.end[b]);
                        if (t == kTypeAB || t == kTypeBA) {
                            L.state[k] = 3;
                            L.round[k] = (uint8_t)round;
                        }
                    }
                    L.type[k] = t;
                }
            }
        }
    }
    // *dropped != 0 is all the loop asks (graph.cpp:826: "while any overlap died").  A hundred thousand
    // dropped overlaps adding one by one to the same word took 150 us of the first round at C3 - adds to
    // one word cost about 10 ns apiece; a plain store of the same value from whoever saw one costs nothing.
    if (__ballot(died) != 0 && (threadIdx.x & 63) == 0) *dropped = 1u;
}

// edges of the component graph in rank space; dead items become self loops; every read its own component
// (label[v] = v, v < n_labels: the start of the hooking rounds)
__global__ __launch_bounds__(kBlock) void cc_edges_kernel(TailList L, const uint32_t* __restrict__ rank,
                                                          uint32_t* __restrict__ edges, uint8_t* touched,
                                                          uint32_t* __restrict__ label, uint32_t n_labels) {
    const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
    if (k < n_labels) label[k] = k;
    if (k >= L.n) return;
    const uint8_t st = L.state[k];
    uint32_t x = 0, y = 0;
    if (st == 1 || st == 3) {
        x = rank[L.a[k]]; y = rank[L.b[k]];
        touched[x] = 1; touched[y] = 1;
    }
    edges[2 * k] = x;
    edges[2 * k + 1] = y;
}

// stale types (coordinates or a region moved) before the containment scans
__global__ __launch_bounds__(kBlock) void refresh_types_kernel(TailList L, TailReads R) {
    const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= L.n || L.state[k] == 0 || L.type[k] != 255) return;
    const uint32_t a = L.a[k], b = L.b[k];
    if (!R.alive[a] || !R.alive[b]) return;
    const Coords c = item_coords(L, k);
    L.type[k] = (uint8_t)ovl_type(c, L.strand[k], R.begin[a], R.end[a], R.begin[b], R.end[b]);
}

// position of an item in the reference's list: overlaps in order, then the promoted
// internals by (round, index); internals are scanned separately, by index
__device__ __forceinline__ uint32_t item_key(const TailList& L, uint32_t k, uint8_t st) {
    return st == 3 ? L.n * (1u + L.round[k]) + k : k;
}

// In-order containment removal without the chimera guard (graph.cpp:831-866), both scans (the overlaps
// with the promoted internals, then the internals), without a look from the host.  It is the same fixed
// point as in the second pass - an item that contains a read deletes it unless its own container was
// deleted earlier in the list:
//     X_r[t] = min { key : killer (key, t, keeper) with X_(r-1)[keeper] > key },   X_0 = "never",
// until X_r = X_(r-1) - but by now only the containments that the re-trimming created are left (C3: 27 k +
// 96 k killers among 1.26 M items), and nearly all of them have a keeper that no killer targets: what they
// do does not depend on the round.  So:
//   collect   the killers {key, target, keeper} of both classes (first class from the front of the list,
//             second from its end) and a mark per targeted read;
//   reduce    (per class) a killer whose keeper is no target goes into base[target] = min key, for good;
//             the others - the conditional ones, a few thousand - are listed again;
//   rounds    (per class) ONE workgroup iterates over the conditional killers (fixed_point_kernels.hip);
//             then base[] = what settled;
//   apply     drops the containments of both classes and the internals behind a death, kill the reads.
// The second scan sees the first one's deaths as base0[] (a read is gone when alive[] = 0 or base0[] is
// set): alive[] itself changes in the last kernel only.  (Before: per round a kernel over all items and one
// over all reads, twelve rounds, a look from the host and two fills of 4 MB per scan.)
// Values that other wavefronts write with atomics are read past the vector cache.
struct TailKillers {
    uint32_t *key, *target, *keeper;            // all killers: class 0 at [0, count[0]), class 1 at (n - 1 - count[1], n - 1]
    uint32_t *c_key, *c_target, *c_keeper;      // the conditional ones of the class at hand, from the front
    uint32_t* count;                            // [0], [1] killers per class; [2], [3] conditional killers per class
    uint32_t* error;                            // set when a fixed point did not settle
    uint8_t* mark[2];                           // per class: the read is the target of a killer
    uint32_t* base[2];                          // per class: all ones at the start, the read's death at the end
};

__device__ __forceinline__ uint32_t ld_past_l1(const uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
constexpr int kContainBlock = 1024;
constexpr uint32_t kCollectPer = 4;             // items per thread of the collecting kernel

// Append to a list from a whole workgroup with ONE add to the list's counter (adds to one word cost about
// 10 ns apiece wherever they come from).  Call from all threads; slot = place in the list or kInf.
// s_cnt / s_base: LDS words of the caller, s_cnt zeroed and synchronised before the first note().
struct BlockAppend {
    uint32_t* s_cnt;
    uint32_t* s_base;
    __device__ uint32_t note(bool mine) const {               // place inside the workgroup's batch
        const uint32_t lane = threadIdx.x & 63u;
        const uint64_t m = __ballot(mine);
        uint32_t base = 0;
        if (m) {
            const uint32_t leader = (uint32_t)__ffsll((long long)m) - 1u;
            if (lane == leader) base = atomicAdd(s_cnt, (uint32_t)__popcll(m));
            base = (uint32_t)__shfl((int)base, (int)leader);
        }
        return mine ? base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)) : kInf;
    }
    __device__ void reserve(uint32_t* counter) const {        // all threads; afterwards *s_base is the batch's start
        __syncthreads();
        if (threadIdx.x == 0) *s_base = *s_cnt ? atomicAdd(counter, *s_cnt) : 0u;
        __syncthreads();
    }
};

// (types that went stale when coordinates or a region moved are refreshed on the way - refresh_types_kernel's work)
__global__ __launch_bounds__(kContainBlock) void tail_contain_collect_kernel(TailList L, TailReads R, const uint8_t* __restrict__ alive,
                                                                             TailKillers K) {
    __shared__ uint32_t s_cnt[2], s_base[2];
    if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const BlockAppend app[2] = {{&s_cnt[0], &s_base[0]}, {&s_cnt[1], &s_base[1]}};
    uint32_t key[kCollectPer], target[kCollectPer], keeper[kCollectPer], slot[kCollectPer], cls[kCollectPer];
#pragma unroll
    for (uint32_t u = 0; u < kCollectPer; ++u) {
        const uint32_t k = (blockIdx.x * kCollectPer + u) * kContainBlock + threadIdx.x;
        bool killer = false;
        cls[u] = 0; key[u] = 0; target[u] = 0; keeper[u] = 0;
        if (k < L.n) {
            const uint8_t st = L.state[k];
            uint8_t t = L.type[k];
            if (st != 0 && t == 255) {
                const uint32_t a = L.a[k], b = L.b[k];
                if (alive[a] && alive[b]) {
                    const Coords c = item_coords(L, k);
                    t = (uint8_t)ovl_type(c, L.strand[k], R.begin[a], R.end[a], R.begin[b], R.end[b]);
                    L.type[k] = t;
                }
            }
            if (st != 0 && (t == kTypeA || t == kTypeB)) {
                const uint32_t a = L.a[k], b = L.b[k];
                if (alive[a] && alive[b]) {
                    killer = true;
                    cls[u] = st == 2 ? 1u : 0u;
                    target[u] = t == kTypeA ? b : a;
                    keeper[u] = t == kTypeA ? a : b;
                    key[u] = cls[u] ? k : item_key(L, k, st);
                }
            }
        }
        const uint32_t s0 = app[0].note(killer && cls[u] == 0), s1 = app[1].note(killer && cls[u] == 1);
        slot[u] = killer ? (cls[u] ? s1 : s0) : kInf;
    }
    app[0].reserve(&K.count[0]);
    app[1].reserve(&K.count[1]);
#pragma unroll
    for (uint32_t u = 0; u < kCollectPer; ++u) {
        if (slot[u] == kInf) continue;
        const uint32_t at = cls[u] ? L.n - 1u - (s_base[1] + slot[u]) : s_base[0] + slot[u];
        K.key[at] = key[u]; K.target[at] = target[u]; K.keeper[at] = keeper[u];
        K.mark[cls[u]][target[u]] = 1;
    }
}

// a read is out of the second scan when it was gone before the scans or the first scan deleted it
__device__ __forceinline__ bool gone_before_second_scan(const uint8_t* alive, const uint32_t* base0, uint32_t r) {
    return !alive[r] || base0[r] != kInf;
}

template <int kClass>
__global__ __launch_bounds__(kBlock) void tail_contain_reduce_kernel(TailKillers K, uint32_t n_items, const uint8_t* __restrict__ alive) {
    __shared__ uint32_t s_cnt, s_base;
    const BlockAppend app = {&s_cnt, &s_base};
    const uint32_t n = K.count[kClass];
    uint32_t* base = K.base[kClass];
    for (uint32_t i0 = blockIdx.x * kBlock; i0 < n; i0 += gridDim.x * kBlock) {
        if (threadIdx.x == 0) s_cnt = 0;
        __syncthreads();
        const uint32_t i = i0 + threadIdx.x;
        bool conditional = false;
        uint32_t key = 0, target = 0, keeper = 0;
        if (i < n) {
            const uint32_t at = kClass ? n_items - 1u - i : i;
            key = K.key[at]; target = K.target[at]; keeper = K.keeper[at];
            const bool ok = kClass == 0 || !(gone_before_second_scan(alive, K.base[0], target) ||
                                             gone_before_second_scan(alive, K.base[0], keeper));
            if (ok) {
                if (K.mark[kClass][keeper]) conditional = true;
                else if (ld_past_l1(base + target) > key) atomicMin(base + target, key);
            }
        }
        const uint32_t slot = app.note(conditional);
        app.reserve(&K.count[2 + kClass]);
        if (conditional) {
            const uint32_t w = s_base + slot;
            K.c_key[w] = key; K.c_target[w] = target; K.c_keeper[w] = keeper;
        }
        __syncthreads();
    }
}

// after the scans: the containments of both classes are dropped (they either deleted a read or had lost
// one), and the internals that the loop reached when one of their reads was already gone
__global__ __launch_bounds__(kBlock) void tail_contain_apply_kernel(TailList L, const uint8_t* __restrict__ alive,
                                                                    const uint32_t* __restrict__ base0,
                                                                    const uint32_t* __restrict__ base1) {
    const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= L.n) return;
    const uint8_t st = L.state[k];
    if (st == 0) return;
    const uint8_t t = L.type[k];
    bool drop = t == kTypeA || t == kTypeB;
    if (st == 2 && !drop) {
        const uint32_t a = L.a[k], b = L.b[k];
        drop = gone_before_second_scan(alive, base0, a) || gone_before_second_scan(alive, base0, b) || base1[a] < k || base1[b] < k;
    }
    if (drop) L.state[k] = 0;
}

// ... and then the reads the scans deleted
__global__ __launch_bounds__(kBlock) void tail_contain_kill_kernel(TailKillers K, uint32_t n_items, uint8_t* alive) {
    const uint32_t n0 = K.count[0], n1 = K.count[1];
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n0 + n1; i += gridDim.x * kBlock) {
        const uint32_t target = K.target[i < n0 ? i : n_items - 1u - (i - n0)];
        if (K.base[0][target] != kInf || K.base[1][target] != kInf) alive[target] = 0;
    }
}

// start of the tail: list states, nothing dirty, the pit counts the pile kernel wrote (one launch for
// what used to be a kernel, a fill and a copy)
__global__ __launch_bounds__(kBlock) void tail_init_kernel(uint8_t* state, uint8_t* round, uint32_t n0, uint32_t m,
                                                           uint8_t* dirty, uint32_t* n_pits0, const uint32_t* n_pits,
                                                           uint32_t n_reads, uint32_t* base2, uint8_t* mark2, uint32_t* map, uint32_t* zero22) {
    const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
    if (k < 22) zero22[k] = 0;
    if (k < n_reads) {
        base2[k] = kInf; base2[n_reads + k] = kInf; map[k] = kInf;
        mark2[k] = 0; mark2[n_reads + k] = 0;
    }
    if (k < m) {
        state[k] = k < n0 ? 1 : 2;
        round[k] = 0;
    }
    if (k < n_reads) {
        dirty[k] = 0;
        n_pits0[k] = n_pits[k];
    }
}

// ---- scans with their producers and consumers inside (scan_pass.h) ---------------------------------
// rank[r] / alive_reads[rank] of the reads that are alive (the component graph lives on ranks)
struct RankPass {
    const uint8_t* alive;
    uint32_t* rank;
    uint32_t* alive_reads;
    __device__ uint64_t value(uint32_t r) const { return alive[r] ? 1ull : 0ull; }
    __device__ void place(uint32_t r, uint64_t v, uint64_t before) const {
        rank[r] = v ? (uint32_t)before : kInf;
        if (v) alive_reads[(uint32_t)before] = r;
    }
    __device__ void total(uint64_t) const {}
};

// nodes: two per surviving read, in read order (graph.cpp:553-574); *n_final = surviving reads
struct NodePass {
    const uint8_t* alive;
    uint32_t* node_rank;
    uint32_t* node_read;
    uint32_t* n_final;
    __device__ uint64_t value(uint32_t r) const { return alive[r] ? 1ull : 0ull; }
    __device__ void place(uint32_t r, uint64_t v, uint64_t before) const {
        node_rank[r] = (uint32_t)before;
        if (v) {
            node_read[2 * (uint32_t)before] = r;
            node_read[2 * (uint32_t)before + 1] = r;
        }
    }
    __device__ void total(uint64_t sum) const { *n_final = (uint32_t)sum; }
};

// One segment of the final overlap list (the originals, or the internals promoted in one round; list
// order = the reference's, graph.cpp:813-823) AND its edges: an item that is kept knows its place among
// the kept ones and, if it is a dovetail, among the dovetails (two counters in one scan: low 31 bits
// kept, high bits dovetails), so the two edges of graph.cpp:576-632 are written by the same launch.
// base[0 .. 1] = kept items / dovetails of the segments in front; updated for the next segment.
struct SegmentPass {
    TailList L;
    TailReads R;
    uint32_t want_state, want_round;
    uint32_t* base;
    uint32_t* kept_item;
    const uint32_t* node_rank;
    uint32_t *e_src, *e_dst, *e_len;
    __device__ uint64_t value(uint32_t k) const {
        const uint8_t st = L.state[k];
        if (st != want_state || (st == 3 && L.round[k] != want_round)) return 0ull;
        const uint32_t a = L.a[k], b = L.b[k];
        if (!R.alive[a] || !R.alive[b]) return 0ull;
        const Coords c = item_coords(L, k);
        const uint32_t t = ovl_type(c, L.strand[k], R.begin[a], R.end[a], R.begin[b], R.end[b]);
        L.type[k] = (uint8_t)t;
        return 1ull | ((t == kTypeAB || t == kTypeBA) ? 1ull << 31 : 0ull);
    }
    __device__ void place(uint32_t k, uint64_t v, uint64_t before) const {
        if (!v) return;
        const uint32_t kept_before = base[0] + (uint32_t)(before & 0x7FFFFFFFull);
        kept_item[kept_before] = k;
        if (!(v >> 31)) return;
        const uint32_t w = 2u * (base[1] + (uint32_t)(before >> 31));
        const uint32_t a = L.a[k], b = L.b[k];
        const Coords c = item_coords(L, k);
        EdgePair e;
        ovl_edges(c, L.strand[k], L.type[k], 2u * node_rank[a], 2u * node_rank[b], R.begin[a], R.end[a], R.begin[b], R.end[b], e);
        e_src[w] = e.src0; e_dst[w] = e.dst0; e_len[w] = e.len0;
        e_src[w + 1] = e.src1; e_dst[w + 1] = e.dst1; e_len[w + 1] = e.len1;
    }
    // (runs in the last tile, after every other tile has read base[] for its own items?  No: other tiles may
    // still be placing.  The next segment's base is therefore written to next[], not over base[].)
    uint32_t* next;
    __device__ void total(uint64_t sum) const {
        next[0] = base[0] + (uint32_t)(sum & 0x7FFFFFFFull);
        next[1] = base[1] + (uint32_t)(sum >> 31);
    }
};

// exclusive scan of plain values (row offsets of a CSR from its row lengths); out[n] = the sum
struct OffsetsPass {
    const uint32_t* in;
    uint32_t* out;
    uint32_t* copy;         // a second copy of the offsets (a fill cursor), may be null
    uint32_t n;
    __device__ uint64_t value(uint32_t i) const { return in[i]; }
    __device__ void place(uint32_t i, uint64_t, uint64_t before) const {
        out[i] = (uint32_t)before;
        if (copy) copy[i] = (uint32_t)before;
    }
    __device__ void total(uint64_t sum) const { out[n] = (uint32_t)sum; }
};

// two exclusive scans side by side (the second pass' chunk counts of surviving overlaps and internals): 31 bits each in the
// scan's 62 - a context holds fewer than 2^30 overlaps (rala_hip_set_overlaps), so neither sum reaches 2^31;
// out[n] = the sums, which also go to totals[0 .. 1]
struct PairOffsetsPass {
    const uint32_t *in0, *in1;
    uint32_t *out0, *out1, *totals;
    uint32_t n;
    __device__ uint64_t value(uint32_t i) const { return (uint64_t)in0[i] | (uint64_t)in1[i] << 31; }
    __device__ void place(uint32_t i, uint64_t, uint64_t before) const {
        out0[i] = (uint32_t)(before & 0x7FFFFFFFull);
        out1[i] = (uint32_t)(before >> 31);
    }
    __device__ void total(uint64_t sum) const {
        out0[n] = totals[0] = (uint32_t)(sum & 0x7FFFFFFFull);
        out1[n] = totals[1] = (uint32_t)(sum >> 31);
    }
};

}  // namespace

void launch_break_hills(const TailReads& R, uint32_t n_reads, hipStream_t s) {
    if (n_reads) hipLaunchKernelGGL(break_hills_kernel, grid_for(n_reads), dim3(kBlock), 0, s, R, n_reads);
}
void launch_break_pits(const TailReads& R, const uint32_t* alive_reads, const uint8_t* touched, const uint16_t* comp_median,
                       uint32_t n_alive, hipStream_t s, const uint32_t* gate) {
    if (n_alive) {
        hipLaunchKernelGGL(break_pits_kernel, grid_for(n_alive), dim3(kBlock), 0, s, R, alive_reads, touched, comp_median,
                           n_alive, gate);
    }
}
void launch_retrim(const TailList& L, const TailReads& R, uint32_t promote, uint32_t round, uint32_t* dropped,
                   hipStream_t s, const uint32_t* gate) {
    if (L.n) hipLaunchKernelGGL(retrim_kernel, grid_for(L.n), dim3(kBlock), 0, s, L, R, promote, round, dropped, gate);
}
void launch_cc_edges(const TailList& L, const uint32_t* rank, uint32_t* edges, uint8_t* touched, uint32_t* label, uint32_t n_labels,
                     hipStream_t s) {
    const uint32_t n = std::max<uint32_t>(L.n, n_labels);
    if (n) hipLaunchKernelGGL(cc_edges_kernel, grid_for(n), dim3(kBlock), 0, s, L, rank, edges, touched, label, n_labels);
}
void launch_refresh_types(const TailList& L, const TailReads& R, hipStream_t s) {
    if (L.n) hipLaunchKernelGGL(refresh_types_kernel, grid_for(L.n), dim3(kBlock), 0, s, L, R);
}
hipError_t launch_tail_contain(const TailList& L, const TailReads& R, uint8_t* alive, uint32_t* const lists[6], uint32_t* zeroed21, uint32_t* const work[4],
                               uint32_t* base2, uint8_t* mark2, uint32_t* map, uint32_t* pack, uint32_t n_reads, uint32_t lds_limit, hipStream_t s) {
    if (!L.n) return hipSuccess;
    TailKillers K;
    K.key = lists[0]; K.target = lists[1]; K.keeper = lists[2];
    K.c_key = lists[3]; K.c_target = lists[4]; K.c_keeper = lists[5];
    K.error = zeroed21; K.count = zeroed21 + 1;
    K.mark[0] = mark2; K.mark[1] = mark2 + n_reads;
    K.base[0] = base2; K.base[1] = base2 + n_reads;
    const uint32_t per_block = kCollectPer * kContainBlock;
    hipLaunchKernelGGL(tail_contain_collect_kernel, dim3((L.n + per_block - 1) / per_block), dim3(kContainBlock), 0, s, L, R,
                       (const uint8_t*)alive, K);
    for (int c = 0; c < 2; ++c) {
        if (c == 0) hipLaunchKernelGGL(tail_contain_reduce_kernel<0>, dim3(128), dim3(kBlock), 0, s, K, L.n, (const uint8_t*)alive);
        else hipLaunchKernelGGL(tail_contain_reduce_kernel<1>, dim3(128), dim3(kBlock), 0, s, K, L.n, (const uint8_t*)alive);
        const FixedPointList conditional = {K.c_key, K.c_target, K.c_keeper, K.count + 2 + c, lds_limit};
        const hipError_t e = launch_fixed_point_finish(conditional, K.base[c], map, pack, work, zeroed21 + 5 + 8 * c, K.error, nullptr, s);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(tail_contain_apply_kernel, grid_for(L.n), dim3(kBlock), 0, s, L, (const uint8_t*)alive,
                       (const uint32_t*)K.base[0], (const uint32_t*)K.base[1]);
    hipLaunchKernelGGL(tail_contain_kill_kernel, dim3(128), dim3(kBlock), 0, s, K, L.n, alive);
    return hipGetLastError();
}
void launch_tail_init(const TailList& L, uint32_t n0, const TailReads& R, uint32_t* n_pits0, uint32_t n_reads, uint32_t* base2,
                      uint8_t* mark2, uint32_t* map, uint32_t* zero22, hipStream_t s) {
    const uint32_t n = std::max<uint32_t>(std::max<uint32_t>(L.n, n_reads), 22u);
    hipLaunchKernelGGL(tail_init_kernel, grid_for(n), dim3(kBlock), 0, s, L.state, L.round, n0, L.n, R.dirty, n_pits0, R.n_pits, n_reads,
                       base2, mark2, map, zero22);
}
bool launch_rank_pass(const uint8_t* alive, uint32_t* rank, uint32_t* alive_reads, uint32_t n_reads, ScanSpace& space, hipStream_t s) {
    return launch_scan_pass(n_reads, RankPass{alive, rank, alive_reads}, space, s);
}
bool launch_node_pass(const uint8_t* alive, uint32_t* node_rank, uint32_t* node_read, uint32_t* n_final, uint32_t n_reads,
                      ScanSpace& space, hipStream_t s) {
    return launch_scan_pass(n_reads, NodePass{alive, node_rank, node_read, n_final}, space, s);
}
bool launch_segment_pass(const TailList& L, const TailReads& R, uint32_t want_state, uint32_t want_round, uint32_t* base,
                         uint32_t* next, uint32_t* kept_item, const uint32_t* node_rank, uint32_t* e_src, uint32_t* e_dst,
                         uint32_t* e_len, ScanSpace& space, hipStream_t s) {
    SegmentPass f;
    f.L = L; f.R = R; f.want_state = want_state; f.want_round = want_round; f.base = base; f.kept_item = kept_item;
    f.node_rank = node_rank; f.e_src = e_src; f.e_dst = e_dst; f.e_len = e_len; f.next = next;
    return launch_scan_pass(L.n, f, space, s);
}
bool launch_pair_offsets_pass(const uint32_t* in0, const uint32_t* in1, uint32_t* out0, uint32_t* out1, uint32_t* totals, uint32_t n,
                              ScanSpace& space, hipStream_t s) {
    return launch_scan_pass(n, PairOffsetsPass{in0, in1, out0, out1, totals, n}, space, s);
}
bool launch_offsets_pass(const uint32_t* in, uint32_t* out, uint32_t* copy, uint32_t n, ScanSpace& space, hipStream_t s) {
    return launch_scan_pass(n, OffsetsPass{in, out, copy, n}, space, s);
}

namespace {
__global__ __launch_bounds__(256) void count_zero_u8_kernel(const uint8_t* __restrict__ x, uint32_t n, uint32_t* out) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const uint64_t m = __ballot(i < n && x[i] == 0);
    if (m && (threadIdx.x & 63) == 0) atomicAdd(out, (uint32_t)__popcll(m));
}
}  // namespace
void launch_count_zero_u8(const uint8_t* x, uint32_t n, uint32_t* out, hipStream_t s) {
    if (n) hipLaunchKernelGGL(count_zero_u8_kernel, dim3((n + 255) / 256), dim3(256), 0, s, x, n, out);
}

}  // namespace rala_hip
