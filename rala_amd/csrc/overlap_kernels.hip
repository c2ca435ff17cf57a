// Per-overlap kernels: duplicate removal, bound bucketing (CSR by read),
// trim/type classification, the in-order containment ("death") scan as a
// parallel fixed point, hill span counters and survivor gathering.
//
// Reference behaviour followed (rvaser/rala src/graph.cpp):
//   remove_duplicate_overlaps :273-307   store_overlap_bounds :311-326
//   construct, pass 2         :443-518   (Overlap::trim / ::type in geom.h)
#include <hip/hip_runtime.h>
#include <algorithm>

#include "device_utils.h"
#include "geom.h"
#include "kernels.h"

namespace rala_hip {

namespace {

constexpr int kBlock = 256;
constexpr uint32_t kInf = 0xFFFFFFFFu;

// One thread per overlap.  Within a maximal run of equal a_id (records whose
// names do not resolve are skipped and do not break a run), per b_id exactly
// the last occurrence of the greatest length stays valid; self overlaps are
// invalid (SURVEY B-T2 closed form of graph.cpp:273-307).
// A run whose targets are strictly increasing cannot hold two overlaps of the same pair
// (the usual case: overlappers emit each query's hits ordered by target).  This pass flags
// the queries that own a run where that order is broken, or a record that does not resolve;
// only their overlaps need the full comparison below.
__global__ __launch_bounds__(kBlock) void dedupe_mark_kernel(OvlSoA o, uint32_t n_reads, uint8_t* __restrict__ suspect) {
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= o.n || i == 0) return;
    const uint32_t a = o.a_id[i], b = o.b_id[i];
    const uint32_t pa = o.a_id[i - 1], pb = o.b_id[i - 1];
    const bool ok = a < n_reads && b < n_reads, pok = pa < n_reads && pb < n_reads;
    if (ok && pok) {
        if (pa == a && b <= pb) suspect[a] = 1;
    } else {
        // an unresolved record hides the order of its neighbours: flag both queries
        if (a < n_reads) suspect[a] = 1;
        if (pa < n_reads) suspect[pa] = 1;
    }
}

// the full comparison for overlap i of a marked query (a, b resolve, a != b)
__device__ __forceinline__ bool dedupe_survives(const OvlSoA& o, uint32_t n_reads, uint64_t i, uint32_t a, uint32_t b) {
    const uint32_t len = o.length[i];
    bool ok = true;
    // neighbours in batches of four: the loads do not depend on the loop exit
    {
        bool done = false;
        for (uint64_t j0 = i; j0 > 0 && !done;) {     // earlier members of the run
            uint32_t aj[4], bj[4];
            uint64_t jj[4];
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) {
                jj[u] = j0 > u ? j0 - 1 - u : 0;
                aj[u] = o.a_id[jj[u]];
                bj[u] = o.b_id[jj[u]];
            }
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) {
                if (done || j0 <= u) { done = true; break; }
                if (aj[u] >= n_reads || bj[u] >= n_reads) continue;
                if (aj[u] != a) { done = true; break; }
                if (bj[u] == b && o.length[jj[u]] > len) { ok = false; done = true; break; }
            }
            j0 = j0 > 4 ? j0 - 4 : 0;
        }
    }
    if (ok) {
        bool done = false;
        for (uint64_t j0 = i + 1; j0 < o.n && !done; j0 += 4) {      // later members
            uint32_t aj[4], bj[4];
            uint64_t jj[4];
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) {
                jj[u] = j0 + u < o.n ? j0 + u : o.n - 1;
                aj[u] = o.a_id[jj[u]];
                bj[u] = o.b_id[jj[u]];
            }
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) {
                if (done || j0 + u >= o.n) { done = true; break; }
                if (aj[u] >= n_reads || bj[u] >= n_reads) continue;
                if (aj[u] != a) { done = true; break; }
                if (bj[u] == b && o.length[jj[u]] >= len) { ok = false; done = true; break; }
            }
        }
    }
    return ok;
}

__global__ __launch_bounds__(kBlock) void dedupe_kernel(OvlSoA o, uint32_t n_reads, const uint8_t* __restrict__ suspect,
                                                        uint8_t* __restrict__ valid) {
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= o.n) return;
    const uint32_t a = o.a_id[i], b = o.b_id[i];
    if (a >= n_reads || b >= n_reads || a == b) {
        valid[i] = 0;
        return;
    }
    if (!suspect[a]) {          // every run of this query has strictly increasing targets
        valid[i] = 1;
        return;
    }
    valid[i] = dedupe_survives(o, n_reads, i, a, b) ? 1 : 0;
}

// behind the counting pass that marked queries and wrote the validity bytes of everybody else
// (group_count_dedupe_kernel, bucket_kernels.hip).  The list was given up (its counter beyond its capacity): every overlap
// of a marked query.
__global__ __launch_bounds__(kBlock) void dedupe_fix_kernel(OvlSoA o, uint32_t n_reads, const uint8_t* __restrict__ suspect,
                                                            const uint32_t* __restrict__ list_count, uint32_t list_cap,
                                                            uint8_t* __restrict__ valid) {
    if (*list_count <= list_cap) return;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < o.n; i += (uint64_t)gridDim.x * kBlock) {
        const uint32_t a = o.a_id[i], b = o.b_id[i];
        if (a >= n_reads || b >= n_reads || a == b || !suspect[a]) continue;
        valid[i] = dedupe_survives(o, n_reads, i, a, b) ? 1 : 0;
    }
}

// The usual case: a wavefront per listed mark (position i, query a).  The run of query a around position i - records that do
// not resolve neither start nor end a run (graph.cpp:343-350) - is found by the wavefront, 64 positions at a time to either
// side; every member of the run then takes the full comparison.  A run with several marks is redone as often: the answers
// are the same.
__global__ __launch_bounds__(kBlock) void dedupe_fix_list_kernel(OvlSoA o, uint32_t n_reads, const uint32_t* __restrict__ list_pos,
                                                                 const uint32_t* __restrict__ list_query,
                                                                 const uint32_t* __restrict__ list_count, uint32_t list_cap,
                                                                 uint8_t* __restrict__ valid) {
    const uint32_t n_list = *list_count;
    if (n_list == 0 || n_list > list_cap) return;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t n_waves = gridDim.x * (kBlock / 64);
    for (uint32_t e = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); e < n_list; e += n_waves) {
        const uint64_t at = list_pos[e];
        const uint32_t a = list_query[e];
        auto breaks = [&](uint64_t j) {             // a resolved record of another query
            const uint32_t aj = o.a_id[j], bj = o.b_id[j];
            return aj < n_reads && bj < n_reads && aj != a;
        };
        uint64_t lo = at, hi = at + 1;              // the run lies within [lo, hi)
        for (;;) {                                  // to the left: positions lo - 1 - lane
            if (lo == 0) break;
            const bool there = lo > lane;
            const bool stop = !there || breaks(lo - 1 - lane);
            const uint64_t m = __ballot(stop);
            if (m) { lo -= (uint32_t)__ffsll((unsigned long long)m) - 1u; break; }
            lo -= 64;
        }
        for (;;) {                                  // to the right: positions hi + lane
            if (hi >= o.n) break;
            const bool there = hi + lane < o.n;
            const bool stop = !there || breaks(hi + lane);
            const uint64_t m = __ballot(stop);
            if (m) { hi += (uint32_t)__ffsll((unsigned long long)m) - 1u; break; }
            hi += 64;
        }
        for (uint64_t j = lo + lane; j < hi; j += 64) {
            const uint32_t aj = o.a_id[j], bj = o.b_id[j];
            if (aj >= n_reads || bj >= n_reads || aj != a || aj == bj) continue;
            valid[j] = dedupe_survives(o, n_reads, j, aj, bj) ? 1 : 0;
        }
    }
}

// Every resolvable overlap (valid or not) contributes two bounds to each of
// its reads (graph.cpp:311-326).  Overlap files are grouped by query, so the
// lanes of a wavefront mostly share a_id: one atomic per segment of equal a_id
// (leader = first lane of the segment, found with a ballot), one per lane for
// the scattered target side.
__device__ __forceinline__ uint32_t segment_of(uint32_t key, bool active, uint32_t lane, uint32_t& leader) {
    // lanes are consecutive overlaps; a segment = maximal run of active lanes with equal key
    const uint32_t prev = (uint32_t)__shfl_up((int)key, 1, 64);
    const bool prev_active = __shfl_up((int)active, 1, 64) != 0;
    const bool head = active && (lane == 0 || !prev_active || prev != key);
    const uint64_t heads = __ballot(head);
    const uint64_t act = __ballot(active);
    // leader: highest head at or below this lane
    const uint64_t below = heads & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
    leader = below ? 63u - (uint32_t)__clzll((long long)below) : lane;
    // segment length (for the leader): up to the next head or the first inactive lane
    const uint64_t after = (lane == 63) ? 0ull : ((heads | ~act) >> (lane + 1));
    const uint32_t len = after ? (uint32_t)__ffsll((unsigned long long)after) : (64u - lane);
    return head ? len : 0u;
}

// Fixed-slot bucketing: read r owns ev_fixed[r * stride .. + stride); the counting atomic hands
// out the position inside the slot and the bounds are stored right away (8-byte stores).  The
// kernel runs at the rate of the random atomics (about 25 G/s on MI355X whatever their scope,
// tools/atomic_bench.hip); the stores ride along.
__global__ __launch_bounds__(kBlock) void bucket_fixed_kernel(OvlSoA o, uint32_t n_reads, uint32_t stride,
                                                             uint32_t* counts, uint32_t* __restrict__ ev_fixed,
                                                             uint32_t* over) {
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63;
    uint32_t a = kInf, b = kInf;
    if (i < o.n) { a = o.a_id[i]; b = o.b_id[i]; }
    const bool ok = a < n_reads && b < n_reads;
    uint32_t leader;
    const uint32_t seg = segment_of(a, ok, lane, leader);
    uint32_t base = 0;
    if (seg) base = atomicAdd(&counts[a], 2u * seg);
    base = (uint32_t)__shfl((int)base, (int)leader, 64);
    if (!ok) return;
    const uint32_t pa = base + 2u * (lane - leader);
    const uint32_t pb = atomicAdd(&counts[b], 2u);
    if (pa + 2u <= stride) {
        *(uint2*)(ev_fixed + (size_t)a * stride + pa) = make_uint2((o.a_begin[i] + 15u) << 1, ((o.a_end[i] - 15u) << 1) | 1u);
    }
    if (pb + 2u <= stride) {
        *(uint2*)(ev_fixed + (size_t)b * stride + pb) = make_uint2((o.b_begin[i] + 15u) << 1, ((o.b_end[i] - 15u) << 1) | 1u);
    }
    if (pa + 2u > stride || pb + 2u > stride) *over = 1u;
}

// Counting pass.  The value an atomic add returns is a unique slot inside the read's bucket,
// so it is kept (rank_a / rank_b, 4 B per overlap and side) and the scatter pass needs no
// atomics at all.
__global__ __launch_bounds__(kBlock) void count_bounds_kernel(OvlSoA o, uint32_t n_reads, uint32_t* counts,
                                                             uint32_t* __restrict__ rank_a,
                                                             uint32_t* __restrict__ rank_b) {
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63;
    uint32_t a = kInf, b = kInf;
    if (i < o.n) { a = o.a_id[i]; b = o.b_id[i]; }
    const bool ok = a < n_reads && b < n_reads;
    uint32_t leader;
    const uint32_t seg = segment_of(a, ok, lane, leader);
    uint32_t base = 0;
    if (seg) base = atomicAdd(&counts[a], 2u * seg);
    base = (uint32_t)__shfl((int)base, (int)leader, 64);
    if (!ok) return;
    rank_a[i] = base + 2u * (lane - leader);
    rank_b[i] = atomicAdd(&counts[b], 2u);
}

__global__ __launch_bounds__(kBlock) void scatter_bounds_kernel(OvlSoA o, uint32_t n_reads,
                                                                const uint32_t* __restrict__ ev_off,
                                                                const uint32_t* __restrict__ rank_a,
                                                                const uint32_t* __restrict__ rank_b,
                                                                uint32_t* __restrict__ ev) {
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= o.n) return;
    const uint32_t a = o.a_id[i], b = o.b_id[i];
    if (a >= n_reads || b >= n_reads) return;
    // bucket offsets and ranks are even: 8-byte stores
    const uint32_t pa = ev_off[a] + rank_a[i];
    *(uint2*)(ev + pa) = make_uint2((o.a_begin[i] + 15u) << 1, ((o.a_end[i] - 15u) << 1) | 1u);
    const uint32_t pb = ev_off[b] + rank_b[i];
    *(uint2*)(ev + pb) = make_uint2((o.b_begin[i] + 15u) << 1, ((o.b_end[i] - 15u) << 1) | 1u);
}

// Multi-GPU: the four bounds of overlap i as 8-byte tuples {read, bound} at 4i .. 4i+3; the read
// is ~0u for records that do not resolve.
__global__ __launch_bounds__(kBlock) void emit_tuples_kernel(OvlSoA o, uint32_t n_reads, uint2* __restrict__ tuples) {
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= o.n) return;
    const uint32_t a = o.a_id[i], b = o.b_id[i];
    const bool ok = a < n_reads && b < n_reads;
    const uint32_t ra = ok ? a : kInf, rb = ok ? b : kInf;
    uint4* out = (uint4*)(tuples + 4 * i);
    out[0] = make_uint4(ra, (o.a_begin[i] + 15u) << 1, ra, ((o.a_end[i] - 15u) << 1) | 1u);
    out[1] = make_uint4(rb, (o.b_begin[i] + 15u) << 1, rb, ((o.b_end[i] - 15u) << 1) | 1u);
}

// Multi-GPU: tuples grouped by the owner of the read (read % world; local read = read / world).
// A workgroup owns a chunk of kBucketChunk consecutive overlaps; wavefronts 0/1 take the
// query side, 2/3 the target side, every lane emitting the begin and the end bound of its
// overlap next to each other (so the owner's bucketing sees runs of equal reads).  Bucket
// sizes are accumulated in LDS: pass 0 adds them to the global per-owner counters, pass 1
// reserves the chunk's range of every bucket with one atomic per owner and scatters.  The
// order inside a bucket is irrelevant (coverage is additive).
constexpr uint32_t kBucketChunk = 2048;

template <class F>
__device__ __forceinline__ void for_each_owner(uint32_t owner, F f) {
    // wave-uniform loop over the distinct owners present in the wavefront
    uint64_t todo = __ballot(owner != kInf);
    while (todo) {
        const uint32_t first = (uint32_t)__ffsll((unsigned long long)todo) - 1;
        const uint32_t p = (uint32_t)__shfl((int)owner, (int)first, 64);
        const uint64_t m = __ballot(owner == p);
        f(p, m);
        todo &= ~m;
    }
}

// kRecords: ONE 8-byte element per overlap side instead of two - a bound record {local read : 22 | begin : 21 | end : 21}
// (the raw coordinates, what lies beyond 2^21 - 2 stays there: outside every read this format is used for)
constexpr uint32_t kRecordCoordBits = 21;
constexpr uint32_t kRecordCoordMax = (1u << kRecordCoordBits) - 1u;
__device__ __forceinline__ uint64_t bound_record(uint32_t local, uint32_t begin, uint32_t end) {
    return (uint64_t)local << (2 * kRecordCoordBits) | (uint64_t)(begin < kRecordCoordMax ? begin : kRecordCoordMax) << kRecordCoordBits |
           (uint64_t)(end < kRecordCoordMax ? end : kRecordCoordMax);
}

template <bool kRecords>
__global__ __launch_bounds__(kBlock) void bucket_tuples_kernel(OvlSoA o, uint32_t n_reads, uint32_t world, uint32_t pass,
                                                               uint32_t* counters, uint2* __restrict__ tuples) {
    constexpr uint32_t kEach = kRecords ? 1u : 2u;                 // elements per overlap side
    static_assert(kBlock == 256, "four wavefronts per workgroup");
    __shared__ uint32_t s_cnt[64], s_base[64];
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = threadIdx.x >> 6;
    const bool side_b = wave >= 2;
    const uint64_t first = (uint64_t)blockIdx.x * kBucketChunk + (wave & 1u) * 64u;
    if (threadIdx.x < 64) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    auto owner_of = [&](uint64_t i, uint32_t& read) {
        read = kInf;
        if (i < o.n) {
            const uint32_t a = o.a_id[i], b = o.b_id[i];
            if (a < n_reads && b < n_reads) read = side_b ? b : a;
        }
        return read == kInf ? kInf : read % world;
    };
    for (uint32_t k = 0; k < kBucketChunk; k += 128) {
        uint32_t read;
        const uint32_t owner = owner_of(first + k + lane, read);
        for_each_owner(owner, [&](uint32_t p, uint64_t m) {
            if (lane == 0) atomicAdd(&s_cnt[p], kEach * (uint32_t)__popcll(m));
        });
    }
    __syncthreads();
    if (pass == 0) {
        if (threadIdx.x < world && s_cnt[threadIdx.x]) atomicAdd(&counters[threadIdx.x], s_cnt[threadIdx.x]);
        return;
    }
    if (threadIdx.x < world) {
        const uint32_t c = s_cnt[threadIdx.x];
        s_base[threadIdx.x] = c ? atomicAdd(&counters[threadIdx.x], c) : 0u;
    }
    __syncthreads();
    if (threadIdx.x < 64) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t k = 0; k < kBucketChunk; k += 128) {
        const uint64_t i = first + k + lane;
        uint32_t read;
        const uint32_t owner = owner_of(i, read);
        for_each_owner(owner, [&](uint32_t p, uint64_t m) {
            uint32_t off = 0;
            if (lane == 0) off = atomicAdd(&s_cnt[p], kEach * (uint32_t)__popcll(m));
            off = (uint32_t)__shfl((int)off, 0, 64);
            if (owner == p) {
                const uint32_t w = s_base[p] + off + kEach * (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                const uint32_t lo = side_b ? o.b_begin[i] : o.a_begin[i];
                const uint32_t hi = side_b ? o.b_end[i] : o.a_end[i];
                const uint32_t local = read / world;
                if (kRecords) *(uint64_t*)(tuples + w) = bound_record(local, lo, hi);
                else *(uint4*)(tuples + w) = make_uint4(local, (lo + 15u) << 1, local, ((hi - 15u) << 1) | 1u);
            }
        });
    }
}

// fixed-slot bucketing of (read, bound) tuples (multi-GPU owners): as bucket_fixed_kernel
__global__ __launch_bounds__(kBlock) void bucket_fixed_tuples_kernel(const uint2* __restrict__ tuples, uint64_t n,
                                                                     uint32_t n_reads, uint32_t stride, uint32_t* counts,
                                                                     uint32_t* __restrict__ ev_fixed, uint32_t* over) {
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63;
    const uint2 t = i < n ? tuples[i] : make_uint2(kInf, 0u);
    const uint32_t r = t.x;
    uint32_t leader;
    const uint32_t seg = segment_of(r, r < n_reads, lane, leader);
    uint32_t base = 0;
    if (seg) base = atomicAdd(&counts[r], seg);
    base = (uint32_t)__shfl((int)base, (int)leader, 64);
    if (r >= n_reads) return;
    const uint32_t p = base + (lane - leader);
    if (p < stride) ev_fixed[(size_t)r * stride + p] = t.y;
    else *over = 1u;
}

__global__ __launch_bounds__(kBlock) void count_tuples_kernel(const uint2* __restrict__ tuples, uint64_t n,
                                                              uint32_t n_reads, uint32_t* counts) {
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t r = i < n ? tuples[i].x : kInf;
    uint32_t leader;
    const uint32_t seg = segment_of(r, r < n_reads, lane, leader);
    if (seg) atomicAdd(&counts[r], seg);
}

__global__ __launch_bounds__(kBlock) void scatter_tuples_kernel(const uint2* __restrict__ tuples, uint64_t n,
                                                                uint32_t n_reads, uint32_t* cursor,
                                                                uint32_t* __restrict__ ev) {
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63;
    const uint2 t = i < n ? tuples[i] : make_uint2(kInf, 0u);
    const uint32_t r = t.x;
    uint32_t leader;
    const uint32_t seg = segment_of(r, r < n_reads, lane, leader);
    uint32_t base = 0;
    if (seg) base = atomicAdd(&cursor[r], seg);
    base = (uint32_t)__shfl((int)base, (int)leader, 64);
    if (r < n_reads) ev[base + (lane - leader)] = t.y;
}

__device__ __forceinline__ Coords load_coords(const OvlSoA& o, uint64_t i) {
    Coords c;
    c.a_begin = o.a_begin[i]; c.a_end = o.a_end[i];
    c.b_begin = o.b_begin[i]; c.b_end = o.b_end[i];
    c.length = o.length[i];
    return c;
}

// ---- second overlap pass (graph.cpp:443-518), static part -----------------------------------
// Per-read state in two tables:
//   * rec[r]  = {begin, n_hills, n_pits, first pool slot}, 16 bytes - read only by
//               the few overlaps that touch a read with chimeric hills (their span counters);
//   * crec[r] = what trim / type / the containment rule need of a read - valid region, "has a chimeric
//               region" (the container guard of graph.cpp:469-480), "has hills" - in ONE word when no read
//               is longer than 32767 bases (two otherwise).  Every overlap looks up both of its reads: two
//               random accesses into a table of 4 MB per million reads, which stays in each XCD's L2; the
//               16-byte records (16 MB) did not, and every look-up moved a whole line from further out.
// Positions in the overlap file are 1-based from here on (o.base + i + 1), so that 0 is a death value of
// its own: "gone before the first overlap" = a read that find_valid_region dropped.
template <bool kSmall>
struct CRec;
template <>
struct CRec<true> {
    typedef uint32_t word;
    static __device__ __forceinline__ word pack(uint32_t b, uint32_t e, bool alive, bool chimeric, bool hills) {
        return alive ? b | e << 15 | (chimeric ? 1u << 30 : 0u) | (hills ? 1u << 31 : 0u) : 0u;
    }
    word w;
    __device__ __forceinline__ uint32_t begin() const { return w & 0x7FFFu; }
    __device__ __forceinline__ uint32_t end() const { return (w >> 15) & 0x7FFFu; }
    __device__ __forceinline__ bool alive() const { return end() != 0; }
    __device__ __forceinline__ bool chimeric() const { return (w >> 30) & 1u; }
    __device__ __forceinline__ bool hills() const { return w >> 31; }
};
template <>
struct CRec<false> {
    typedef uint2 word;
    static __device__ __forceinline__ word pack(uint32_t b, uint32_t e, bool alive, bool chimeric, bool hills) {
        return alive ? make_uint2(b, e | (chimeric ? 1u << 30 : 0u) | (hills ? 1u << 31 : 0u)) : make_uint2(0u, 0u);
    }
    word w;
    __device__ __forceinline__ uint32_t begin() const { return w.x; }
    __device__ __forceinline__ uint32_t end() const { return w.y & 0x3FFFFFFFu; }
    __device__ __forceinline__ bool alive() const { return end() != 0; }
    __device__ __forceinline__ bool chimeric() const { return (w.y >> 30) & 1u; }
    __device__ __forceinline__ bool hills() const { return w.y >> 31; }
};

constexpr uint32_t kFateSurvives = 1, kFateHills = 2;      // fate[r]: the read outlives the containment scan / has chimeric hills

// one thread per read: both tables, the reads that are gone already (death value 0)
template <bool kSmall>
__global__ __launch_bounds__(kBlock) void pack_reads_kernel(ReadState rs, uint32_t n, uint4* __restrict__ rec,
                                                            typename CRec<kSmall>::word* __restrict__ crec,
                                                            uint32_t* __restrict__ sure) {
    const uint32_t r = blockIdx.x * kBlock + threadIdx.x;
    if (r >= n) return;
    const bool alive = rs.alive[r] != 0;
    const uint32_t np = rs.n_pits[r], nh = rs.n_hills[r];
    const uint32_t b = rs.begin[r], e = rs.end[r];
    rec[r] = make_uint4(b, nh, np, rs.iv_slot[r]);
    crec[r] = CRec<kSmall>::pack(b, e, alive, (np | nh) != 0, nh != 0);
    if (!alive) sure[r] = 0u;
}
__device__ __forceinline__ uint32_t rec_pits(const uint4& r) { return r.z; }
__device__ __forceinline__ uint32_t rec_hills(const uint4& r) { return r.y; }

// A workgroup looks at a chunk of kClassifyChunk consecutive overlaps (8 per thread).
constexpr uint32_t kClassifyChunk = 2048;

// trim + type of overlap i against the pass-1 piles; what it would do: 0 nothing, 1 delete a, 2 delete b
// (graph.cpp:464-483: a contained read goes unless its container has a chimeric region).  false = the
// overlap is dropped (duplicate, a read gone, trim failed).
template <bool kSmall>
__device__ __forceinline__ bool classify_one(const OvlSoA& o, uint64_t i, const CRec<kSmall>& ra, const CRec<kSmall>& rb,
                                             Coords& c, uint32_t& strand, uint32_t& type, uint32_t& kills) {
    if (!ra.alive() || !rb.alive()) return false;
    c = load_coords(o, i);
    strand = o.strand[i];
    if (!ovl_trim(c, strand, ra.begin(), ra.end(), rb.begin(), rb.end())) return false;
    type = ovl_type(c, strand, ra.begin(), ra.end(), rb.begin(), rb.end());
    kills = type == kTypeB && !rb.chimeric() ? 1u : type == kTypeA && !ra.chimeric() ? 2u : 0u;
    return true;
}

// The overlaps that would delete a read go to the killer list {position, target, keeper}: slots inside
// the chunk through an LDS counter (one add per wavefront and iteration), one global add per workgroup.
// The list order is irrelevant (the fixed point takes minima).  Nothing else is kept of this pass: who
// survives is decided once the deaths are known, by the few overlaps whose reads both live
// (survivor_masks_kernel), which classify themselves again.
// a column value that is read once: a streaming load that should not displace the per-read tables in the L2
// (RALA_STREAM_PLAIN: ordinary loads, for measurements)
template <class T>
__device__ __forceinline__ T stream_load(const T* p) {
#ifdef RALA_STREAM_PLAIN
    return *p;
#else
    return __builtin_nontemporal_load(p);
#endif
}
// four consecutive words of a column (p on a 16-byte boundary)
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void stream_load4(const uint32_t* p, uint32_t (&out)[4]) {
#ifdef RALA_STREAM_PLAIN
    const u32x4_t v = *(const u32x4_t*)p;
#else
    const u32x4_t v = __builtin_nontemporal_load((const u32x4_t*)p);
#endif
    out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
}

// trim + type from values that are already in registers (see classify_one)
template <bool kSmall>
__device__ __forceinline__ bool classify_loaded(Coords& c, uint32_t strand, const CRec<kSmall>& ra, const CRec<kSmall>& rb,
                                                uint32_t& type, uint32_t& kills) {
    if (!ra.alive() || !rb.alive()) return false;
    if (!ovl_trim(c, strand, ra.begin(), ra.end(), rb.begin(), rb.end())) return false;
    type = ovl_type(c, strand, ra.begin(), ra.end(), rb.begin(), rb.end());
    kills = type == kTypeB && !rb.chimeric() ? 1u : type == kTypeA && !ra.chimeric() ? 2u : 0u;
    return true;
}

// The loads of an overlap depend on each other (validity -> ids -> the reads' records -> coordinates); taken
// overlap by overlap, eight per thread, a wavefront spent its life waiting for one round trip after the
// other (0.9 ms at C3 for 1.5 GB).  Here a thread issues ALL loads of its overlaps that do not depend on
// data first, unconditionally (columns, at clamped indices), then all record look-ups, and only then
// computes: two dependent round trips per kPer overlaps instead of four per overlap.
// kVec (round 5): a thread takes FOUR CONSECUTIVE overlaps per group instead of four that lie a workgroup's width apart, so
// that every column comes by one 16-byte load per lane (and the two byte columns by one word) - eight vector-memory
// instructions per four overlaps instead of thirty-two.  The kernel moves 26 bytes per overlap and ran at 2 TB/s: what it
// was short of were not bytes but requests.  Needs the columns on 16-byte boundaries (launch_classify looks; a slice of a
// sharded run that starts in the middle of its columns takes the other instantiation).
template <bool kSmall, bool kVec>
__global__ __launch_bounds__(kBlock) void classify_kernel(OvlSoA o, uint32_t n_reads, const uint8_t* __restrict__ valid,
                                                          const typename CRec<kSmall>::word* __restrict__ crec,
                                                          KillList kl, uint32_t* lo) {
    __shared__ uint32_t s_cnt, s_base;
    constexpr uint32_t kPer = kClassifyChunk / kBlock;
    constexpr uint32_t kHalf = kPer / 2;
    static_assert(kHalf == 4, "a group is one 16-byte vector of every column");
    const uint32_t lane = threadIdx.x & 63;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    uint32_t slot[kPer], tgt[kPer], kpr[kPer];
    uint32_t any = 0;
    const uint64_t last = o.n - 1;
    // the overlap a thread's item u is
    auto item = [&](uint32_t u) -> uint64_t {
        return kVec ? (uint64_t)blockIdx.x * kClassifyChunk + (u / kHalf) * (kHalf * kBlock) + kHalf * threadIdx.x + u % kHalf
                    : (uint64_t)blockIdx.x * kClassifyChunk + u * kBlock + threadIdx.x;
    };
    // Two halves of four overlaps per thread, software-pipelined (round 5): the second half's columns are requested while the
    // first half's records are on their way - three dependent round trips per wavefront instead of four at six wavefronts per
    // SIMD instead of seven (77 registers): 0.603 -> 0.577 ms at C3, 4.70 -> 4.67 at C5 (docs/history/gpurun/r5_classify_pipe.sh).
    struct Half {
        uint32_t a[kHalf], b[kHalf], st[kHalf];
        Coords c[kHalf];
        bool ok[kHalf];
        typename CRec<kSmall>::word wa[kHalf], wb[kHalf];
    };
    auto load_half = [&](Half& H, uint32_t h) {
        bool loaded = false;
        if constexpr (kVec) {
            const uint64_t i0 = item(h);
            if (i0 + kHalf <= o.n) {
                loaded = true;
                const uint32_t vw = stream_load((const uint32_t*)(valid + i0)), sw = stream_load((const uint32_t*)(o.strand + i0));
                uint32_t xa[4], xb[4], xab[4], xae[4], xbb[4], xbe[4];
                stream_load4(o.a_id + i0, xa); stream_load4(o.b_id + i0, xb);
                stream_load4(o.a_begin + i0, xab); stream_load4(o.a_end + i0, xae);
                stream_load4(o.b_begin + i0, xbb); stream_load4(o.b_end + i0, xbe);
#pragma unroll
                for (uint32_t v = 0; v < kHalf; ++v) {
                    H.ok[v] = ((vw >> (8 * v)) & 0xFFu) != 0;
                    H.a[v] = xa[v]; H.b[v] = xb[v];
                    H.c[v].a_begin = xab[v]; H.c[v].a_end = xae[v]; H.c[v].b_begin = xbb[v]; H.c[v].b_end = xbe[v];
                    H.c[v].length = 0;
                    H.st[v] = (sw >> (8 * v)) & 0xFFu;
                }
            }
        }
        if (!loaded) {
#pragma unroll
            for (uint32_t v = 0; v < kHalf; ++v) {
                const uint64_t i = item(h + v);
                const uint64_t j = i < o.n ? i : last;
                H.ok[v] = i < o.n && stream_load(valid + j);
                H.a[v] = stream_load(o.a_id + j); H.b[v] = stream_load(o.b_id + j);
                H.c[v].a_begin = stream_load(o.a_begin + j); H.c[v].a_end = stream_load(o.a_end + j);
                H.c[v].b_begin = stream_load(o.b_begin + j); H.c[v].b_end = stream_load(o.b_end + j);
                H.c[v].length = 0;
                H.st[v] = stream_load(o.strand + j);
            }
        }
    };
    auto look_half = [&](Half& H) {
#pragma unroll
        for (uint32_t v = 0; v < kHalf; ++v) {
            H.wa[v] = crec[H.a[v] < n_reads ? H.a[v] : 0u];
            H.wb[v] = crec[H.b[v] < n_reads ? H.b[v] : 0u];
        }
    };
    auto compute_half = [&](Half& H, uint32_t h) {
#pragma unroll
        for (uint32_t v = 0; v < kHalf; ++v) {
            const uint32_t u = h + v;
            uint32_t kills = 0, t = 0;
            CRec<kSmall> ra, rb;
            ra.w = H.wa[v]; rb.w = H.wb[v];
            if (!(H.ok[v] && classify_loaded<kSmall>(H.c[v], H.st[v], ra, rb, t, kills))) kills = 0;
            tgt[u] = kills == 1 ? H.a[v] : H.b[v];
            kpr[u] = kills == 1 ? H.b[v] : H.a[v];
            const uint64_t m = __ballot(kills != 0);
            uint32_t base = 0;
            if (m) {
                if (lane == 0) base = atomicAdd(&s_cnt, (uint32_t)__popcll(m));
                base = (uint32_t)__shfl((int)base, 0, 64);
            }
            slot[u] = kills ? base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)) : 0xFFFFFFFFu;
            any |= kills;
        }
    };
    {
        Half H0, H1;
#ifdef RALA_CLASSIFY_TWO_TRIPS      // (measurement: everything requested up front)
        load_half(H0, 0);
        load_half(H1, kHalf);
        look_half(H0);
        look_half(H1);
        compute_half(H0, 0);
        compute_half(H1, kHalf);
#else
        load_half(H0, 0);
        look_half(H0);
        load_half(H1, kHalf);
        compute_half(H0, 0);
        look_half(H1);
        compute_half(H1, kHalf);
#endif
    }
    __syncthreads();
    if (threadIdx.x == 0) s_base = s_cnt ? atomicAdd(kl.count, s_cnt) : 0u;
    __syncthreads();
    if (!any) return;
    const uint32_t base = s_base;
#pragma unroll
    for (uint32_t u = 0; u < kPer; ++u) {
        if (slot[u] == 0xFFFFFFFFu) continue;
        const uint64_t i = item(u);
        const uint32_t w = base + slot[u];
        const uint32_t pos = (uint32_t)(o.base + i) + 1u;  // 1-based position in the whole file (multi-GPU: slices)
        kl.ovl[w] = pos;
        kl.target[w] = tgt[u];
        kl.keeper[w] = kpr[u];
        // the fixed point's first lower bound, everybody's first killer (death_lower_kernel over the
        // whole list: 0.13 ms of atomics at C3 - here they run beside this kernel's loads)
        uint32_t* d = &lo[tgt[u]];
        if (__hip_atomic_load(d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > pos) atomicMin(d, pos);
    }
}

// In-order containment removal (graph.cpp:464-483) as a fixed point:
//     death[r] = min { i : killer i targets r and death[keeper(i)] > i }
// The map is antitone, so iterating it from "nobody dies" gives lower and upper bounds in turn.
// With a lower bound lo[] and an upper bound up[] a killer i is decided as soon as
// lo[keeper] > i (it kills for sure) or up[keeper] <= i (its keeper is gone for sure):
//   * sure killers feed sure[target] = min i  (an upper bound of death[target]),
//   * killers that cannot beat sure[target], and the dead ones, leave the list,
//   * the rest stay undecided and are looked at again with the next, tighter bounds
//     (lo' = min(sure, undecided), up' = sure).
// The undecided set shrinks quickly; when it is empty, death = sure.

// lo[target] = min(lo[target], i) over the list (a coherent load skips hopeless atomics)
__global__ __launch_bounds__(kBlock) void death_lower_kernel(KillList kl, uint32_t* lo) {
    const uint32_t n = *kl.count;
    for (uint32_t k = blockIdx.x * kBlock + threadIdx.x; k < n; k += gridDim.x * kBlock) {
        const uint32_t i = kl.ovl[k];
        uint32_t* d = &lo[kl.target[k]];
        if (__hip_atomic_load(d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > i) atomicMin(d, i);
    }
}

// up == nullptr: the first round, every upper bound is still "never" (one random load less for
// each of the 1.4 M killers of that round at C3)
__global__ __launch_bounds__(kBlock) void death_decide_kernel(KillList in, const uint32_t* __restrict__ lo,
                                                              const uint32_t* __restrict__ up, uint32_t* sure,
                                                              KillList out) {
    __shared__ uint32_t s_cnt, s_base;
    const uint32_t n = *in.count;
    const uint32_t lane = threadIdx.x & 63;
    // Grid-stride over chunks of kPer items per thread so that the append is aggregated per chunk: ONE add
    // to the list's counter per workgroup and chunk (adds to one word cost about 10 ns apiece wherever they
    // come from - with a chunk of 256 the 12 000 adds of the first round were a third of its time).
    constexpr uint32_t kPer = 8;
    for (uint32_t k0 = blockIdx.x * kBlock * kPer; k0 < n; k0 += gridDim.x * kBlock * kPer) {
        if (threadIdx.x == 0) s_cnt = 0;
        __syncthreads();
        uint32_t iv[kPer], tv[kPer], kv[kPer], slot[kPer];
#pragma unroll
        for (uint32_t u = 0; u < kPer; ++u) {
            const uint32_t k = k0 + u * kBlock + threadIdx.x;
            const uint32_t kk = k < n ? k : n - 1;
            iv[u] = in.ovl[kk]; tv[u] = in.target[kk]; kv[u] = in.keeper[kk];
        }
        // (Measured and not kept, round 5: the three look-ups of all eight items together instead of each behind the test before
        // it - the same at C3, 1.28 -> 1.54 ms at C5: the tests save more look-ups than the round trips cost.)
#pragma unroll
        for (uint32_t u = 0; u < kPer; ++u) {
            const uint32_t k = k0 + u * kBlock + threadIdx.x;
            bool keep = false;
            if (k < n) {
                const uint32_t i = iv[u], t = tv[u], kp = kv[u];
                const uint32_t best = __hip_atomic_load(&sure[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (i < best) {
                    if (lo[kp] > i) atomicMin(&sure[t], i);
                    else if (up == nullptr || up[kp] > i) keep = true;
                }
            }
            const uint64_t m = __ballot(keep);
            uint32_t base = 0;
            if (m) {
                if (lane == 0) base = atomicAdd(&s_cnt, (uint32_t)__popcll(m));
                base = (uint32_t)__shfl((int)base, 0, 64);
            }
            slot[u] = keep ? base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)) : 0xFFFFFFFFu;
        }
        __syncthreads();
        if (threadIdx.x == 0) s_base = s_cnt ? atomicAdd(out.count, s_cnt) : 0u;
        __syncthreads();
#pragma unroll
        for (uint32_t u = 0; u < kPer; ++u) {
            if (slot[u] != 0xFFFFFFFFu) {
                const uint32_t w = s_base + slot[u];
                out.ovl[w] = iv[u]; out.target[w] = tv[u]; out.keeper[w] = kv[u];
            }
        }
        __syncthreads();
    }
}

// (x is the older of the two: it becomes the next round's output and is reset for it here - one
// launch instead of a fill per round)
__global__ __launch_bounds__(kBlock) void death_diff_kernel(uint32_t* __restrict__ x,
                                                            const uint32_t* __restrict__ y, uint32_t n,
                                                            uint32_t* changed) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const bool d = x[i] != y[i];
    x[i] = 0xFFFFFFFFu;
    if (d) *changed = 1u;       // benign race: every writer stores the same value
}

// bounds for the next round of the containment fixed point: up = lo = sure (one pass instead of
// two device-to-device copies)
// (+ the list's length for the round count, when the host does not look at it: *status = all ones - *count)
__global__ __launch_bounds__(kBlock) void death_tighten_kernel(const uint32_t* __restrict__ sure, uint32_t* __restrict__ up,
                                                               uint32_t* __restrict__ lo, uint32_t n, const uint32_t* count,
                                                               uint32_t* status) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i == 0 && status) *status = 0xFFFFFFFFu - *count;
    if (i >= n) return;
    const uint32_t v = sure[i];
    up[i] = v;
    lo[i] = v;
}

// Liveness, hill span counters (Pile::check_chimeric_hills, pile.cpp:457-469, including the begin_
// double count) and the survivors, once death[] is final.  An overlap at (1-based) position p is live
// when neither of its reads died before p (death >= p; a read that was gone from the start has death 0).
// What nearly every overlap needs of that is ONE BYTE per read (fate[]: "never dies", "has hills"; 1 MB
// per million reads, which does stay in the L2 - random look-ups into the 4-byte death values came from
// the memory-side cache, a whole line each: 0.5 ms at C3): only an overlap that touches a read with hills
// and is live (counters), or whose reads BOTH never die (a survivor unless it is a containment that
// deleted nobody because its container is chimeric - those stay out, graph.cpp:485-514), looks at the
// death values, loads its coordinates and classifies itself again.  Per 64 overlaps one mask word of the survivors
// that go on as overlaps and one of those that go on as internals (type kX); per chunk their numbers.
// (lanes of ONE wavefront exchanging data through LDS: program order is enough for the hardware, the
// fences keep the compiler from reordering)
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// (Measured, docs/history/gpurun/r3_probe_masks.sh at C3: the three columns alone stream in 0.11 ms, the two fate
// bytes add 0.22 ms, the candidates 0.11 ms.  The target's byte is a random access that misses the vector
// L1; a ONE-bit table "never dies or has hills" in LDS - 125 KB per million reads, one workgroup of 1024
// threads per compute unit - takes nine look-ups in ten off the memory path and came out at the same
// 0.43 ms, persistent or not: sixteen wavefronts per compute unit hide less than the thirty-two of this
// kernel.  Not kept.)
// One overlap in forty needs more than the two bytes, and a wavefront that takes them as they come has
// such a lane in four iterations out of five: every one of its eight iterations then walks the long path
// (death values -> records -> coordinates -> hill intervals, one round trip after the other) for one or
// two lanes - 0.52 ms at C3, whatever the tables' size.  So the candidates of a wavefront's 512 overlaps
// are noted in LDS first and then worked on side by side, one per lane: the long path once per
// wavefront instead of eight times.
template <bool kSmall>
__global__ __launch_bounds__(kBlock) void survivor_masks_kernel(OvlSoA o, uint32_t n_reads, const uint8_t* __restrict__ valid,
                                                                const uint8_t* __restrict__ fate,
                                                                const uint32_t* __restrict__ death,
                                                                const typename CRec<kSmall>::word* __restrict__ crec,
                                                                const uint4* __restrict__ rec, Interval* pool,
                                                                uint64_t* __restrict__ mask_ov, uint64_t* __restrict__ mask_in,
                                                                uint32_t* __restrict__ chunk_ov,
                                                                uint32_t* __restrict__ chunk_in, uint32_t probe) {
    constexpr uint32_t kPer = kClassifyChunk / kBlock, kWaves = kBlock / 64;
    __shared__ uint32_t s_ov, s_in;
    __shared__ uint16_t s_list[kWaves][kPer * 64];
    __shared__ unsigned long long s_mask[kWaves][2][kPer];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) { s_ov = 0; s_in = 0; }
    if (lane < 2 * kPer) s_mask[wave][lane / kPer][lane % kPer] = 0ull;
    __syncthreads();
    const uint64_t last = o.n - 1;
    const uint64_t below = (1ull << lane) - 1ull;
    // (all independent loads first, then the look-ups that depend on them - see classify_kernel)
    uint32_t a[kPer], b[kPer];
    bool ok[kPer];
#pragma unroll
    for (uint32_t u = 0; u < kPer; ++u) {
        const uint64_t i = (uint64_t)blockIdx.x * kClassifyChunk + u * kBlock + threadIdx.x;
        const uint64_t j = i < o.n ? i : last;
        ok[u] = i < o.n && stream_load(valid + j);
        // (ordinary loads: the candidates come back for their ids - 0.430 -> 0.408 ms for this stage at C3, docs/history/gpurun/r5_stream.sh;
        // classify_kernel, which does not come back, is 8 % slower with them)
        a[u] = o.a_id[j]; b[u] = o.b_id[j];
    }
    // The query's death first (the file is grouped by query: neighbouring lanes ask for the same word): an
    // overlap behind its query's death is not live, whatever the target - and most reads die early in their
    // own run of the file, so only one overlap in five goes on to the target's byte, the random access that
    // misses the vector L1 (0.22 of this kernel's 0.44 ms at C3 when every overlap asked for it).
    uint32_t fa[kPer], fb[kPer];
#pragma unroll
    for (uint32_t u = 0; u < kPer; ++u) {
        const uint32_t ra = a[u] < n_reads ? a[u] : 0u;
        fa[u] = probe & 2u ? a[u] & 3u : fate[ra];
        const uint64_t i = (uint64_t)blockIdx.x * kClassifyChunk + u * kBlock + threadIdx.x;
        ok[u] = ok[u] && death[ra] >= (uint32_t)(o.base + i) + 1u;
    }
#pragma unroll
    for (uint32_t u = 0; u < kPer; ++u) {
        fb[u] = !ok[u] ? 0u : probe & 2u ? b[u] & 3u : fate[b[u] < n_reads ? b[u] : 0u];
    }
    uint32_t cnt = 0;           // candidates of this wavefront (the same in every lane)
#pragma unroll
    for (uint32_t u = 0; u < kPer; ++u) {
        const bool both = (fa[u] & fb[u] & kFateSurvives) != 0;
        const bool hills = ((fa[u] | fb[u]) & kFateHills) != 0;
        const bool cand = ok[u] && (both | hills);
        const uint64_t m = __ballot(cand);
        if (cand) s_list[wave][cnt + (uint32_t)__popcll(m & below)] = (uint16_t)(u * 64u + lane);
        cnt += (uint32_t)__popcll(m);
    }
    wave_lds_sync();
    if (probe & 1u) cnt = 0;
    for (uint32_t k = lane; k < cnt; k += 64) {
        const uint32_t e = s_list[wave][k];
        const uint32_t u = e >> 6, l = e & 63u;
        const uint64_t i = (uint64_t)blockIdx.x * kClassifyChunk + u * kBlock + wave * 64u + l;
        const uint32_t ia = o.a_id[i], ib = o.b_id[i];
        const uint32_t at = (uint32_t)(o.base + i) + 1u;
#ifdef RALA_CANDIDATES_STEP_BY_STEP     // (measurement: deaths -> records -> coordinates, a round trip each)
        const uint32_t da = death[ia], db = death[ib];
        if (!(da >= at && db >= at)) continue;
        const bool both = da == kInf && db == kInf;
        CRec<kSmall> ra, rb;
        ra.w = crec[ia]; rb.w = crec[ib];
        Coords c;
        uint32_t st, t, kills;
        if (!classify_one<kSmall>(o, i, ra, rb, c, st, t, kills)) continue;
#else
        // (everything a candidate may need behind its ids in ONE round trip - deaths, records, coordinates, strand; the empty asm
        // keeps the compiler from sinking the loads behind the tests that would make them round trips of their own)
        uint32_t da = death[ia], db = death[ib];
        CRec<kSmall> ra, rb;
        ra.w = crec[ia]; rb.w = crec[ib];
        Coords c;
        c.a_begin = o.a_begin[i]; c.a_end = o.a_end[i]; c.b_begin = o.b_begin[i]; c.b_end = o.b_end[i];
        c.length = 0;
        uint32_t st = o.strand[i], t = 0, kills = 0;
        asm volatile("" : "+v"(da), "+v"(db), "+v"(c.a_begin), "+v"(c.a_end), "+v"(c.b_begin), "+v"(c.b_end), "+v"(st));
        if (!(da >= at && db >= at)) continue;
        const bool both = da == kInf && db == kInf;
        if (!classify_loaded<kSmall>(c, st, ra, rb, t, kills)) continue;
#endif
        if (ra.hills() | rb.hills()) {
            const uint4 xa = rec[ia], xb = rec[ib];
            const uint32_t nha = rec_hills(xa), nhb = rec_hills(xb);
            if (nha) {
                Interval* h = pool + xa.w + rec_pits(xa);
                const uint32_t x = xa.x + c.a_begin, y = xa.x + c.a_end;
                for (uint32_t q = 0; q < nha; ++q) {
                    if (x < h[q].first && y > h[q].second) atomicAdd(&h[q].aux, 1u);
                }
            }
            if (nhb) {
                Interval* h = pool + xb.w + rec_pits(xb);
                const uint32_t x = xb.x + c.b_begin, y = xb.x + c.b_end;
                for (uint32_t q = 0; q < nhb; ++q) {
                    if (x < h[q].first && y > h[q].second) atomicAdd(&h[q].aux, 1u);
                }
            }
        }
        if (both && !kills) atomicOr(&s_mask[wave][t == kTypeX ? 1 : 0][u], 1ull << l);
    }
    wave_lds_sync();
    uint32_t n_ov = 0, n_in = 0;
    if (lane < kPer) {
        const uint64_t m_ov = s_mask[wave][0][lane], m_in = s_mask[wave][1][lane];
        const uint64_t word = (uint64_t)blockIdx.x * (kClassifyChunk / 64) + lane * kWaves + wave;
        if (word * 64 < o.n) { mask_ov[word] = m_ov; mask_in[word] = m_in; }
        n_ov = (uint32_t)__popcll(m_ov);
        n_in = (uint32_t)__popcll(m_in);
    }
    n_ov = wave_reduce(n_ov, OpAdd());
    n_in = wave_reduce(n_in, OpAdd());
    if (lane == 0) { atomicAdd(&s_ov, n_ov); atomicAdd(&s_in, n_in); }
    __syncthreads();
    if (threadIdx.x == 0) { chunk_ov[blockIdx.x] = s_ov; chunk_in[blockIdx.x] = s_in; }
}

// the reads that the containment scan deleted; the number of reads that are left
__global__ __launch_bounds__(kBlock) void apply_death_kernel(const uint32_t* __restrict__ death, uint8_t* alive,
                                                             const uint32_t* __restrict__ n_hills, uint8_t* __restrict__ fate,
                                                             uint32_t n, uint32_t* n_alive) {
    // (few workgroups, one add each: adds to ONE word cost about 10 ns apiece wherever they come from -
    // one per wavefront of a million reads was 0.18 ms)
    __shared__ uint32_t tmp[kBlock / 64 + 1];
    uint32_t mine = 0;
    for (uint32_t r = blockIdx.x * kBlock + threadIdx.x; r < n; r += gridDim.x * kBlock) {
        const bool lives = death[r] == kInf;
        if (!lives) alive[r] = 0;
        fate[r] = (uint8_t)((lives ? kFateSurvives : 0u) | (n_hills[r] ? kFateHills : 0u));
        mine += lives ? 1u : 0u;
    }
    mine = block_reduce<kBlock>(mine, OpAdd(), 0u, tmp);
    if (threadIdx.x == 0 && mine) atomicAdd(n_alive, mine);
}

// Survivors into dense arrays in file order: overlaps from slot 0, internals from slot n_ov_total (trim
// re-applied against the pass-1 piles).  The masks say who; a survivor's place comes from the scanned
// chunk counts and the popcounts of the mask words in front of it.
template <bool kSmall>
__global__ __launch_bounds__(kBlock) void gather_kernel(OvlSoA o, const uint64_t* __restrict__ mask_ov,
                                                        const uint64_t* __restrict__ mask_in,
                                                        const typename CRec<kSmall>::word* __restrict__ crec,
                                                        const uint32_t* __restrict__ chunk_ov_off,
                                                        const uint32_t* __restrict__ chunk_in_off,
                                                        uint32_t n_ov_total, Survivors out) {
    constexpr uint32_t kPer = kClassifyChunk / kBlock, kWaves = kBlock / 64;
    __shared__ uint32_t t_ov[kPer * kWaves + 1], t_in[kPer * kWaves + 1];
    __shared__ unsigned long long s_m[kWaves][2][kPer];
    __shared__ uint16_t s_list[kWaves][kPer * 64];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t any = 0;
    if (lane < kPer) {
        const uint64_t word = (uint64_t)blockIdx.x * (kClassifyChunk / 64) + lane * kWaves + wave;
        const bool in = word * 64 < o.n;
        const uint64_t mo = in ? mask_ov[word] : 0ull, mi = in ? mask_in[word] : 0ull;
        s_m[wave][0][lane] = mo; s_m[wave][1][lane] = mi;
        t_ov[lane * kWaves + wave] = (uint32_t)__popcll(mo);
        t_in[lane * kWaves + wave] = (uint32_t)__popcll(mi);
        any = mo | mi;
    }
    any = __ballot(any != 0);
    __syncthreads();
    if (threadIdx.x < 2) {
        uint32_t* t = threadIdx.x ? t_in : t_ov;
        uint32_t run = threadIdx.x ? n_ov_total + chunk_in_off[blockIdx.x] : chunk_ov_off[blockIdx.x];
        for (uint32_t q = 0; q < kPer * kWaves; ++q) { const uint32_t v = t[q]; t[q] = run; run += v; }
    }
    __syncthreads();
    if (!any) return;
    // the wavefront's survivors side by side, one per lane (see survivor_masks_kernel)
    const uint64_t below = (1ull << lane) - 1ull;
    uint32_t cnt = 0;
#pragma unroll
    for (uint32_t u = 0; u < kPer; ++u) {
        const uint64_t m = s_m[wave][0][u] | s_m[wave][1][u];
        if ((m >> lane) & 1ull) s_list[wave][cnt + (uint32_t)__popcll(m & below)] = (uint16_t)(u * 64u + lane);
        cnt += (uint32_t)__popcll(m);
    }
    wave_lds_sync();
    for (uint32_t k = lane; k < cnt; k += 64) {
        const uint32_t e = s_list[wave][k];
        const uint32_t u = e >> 6, l = e & 63u;
        const uint64_t mo = s_m[wave][0][u], mi = s_m[wave][1][u];
        const uint64_t before = (1ull << l) - 1ull;
        const bool is_in = (mi >> l) & 1ull;
        const uint32_t p = is_in ? t_in[u * kWaves + wave] + (uint32_t)__popcll(mi & before)
                                 : t_ov[u * kWaves + wave] + (uint32_t)__popcll(mo & before);
        const uint64_t i = (uint64_t)blockIdx.x * kClassifyChunk + u * kBlock + wave * 64u + l;
        const uint32_t a = o.a_id[i], b = o.b_id[i];
        CRec<kSmall> ra, rb;
        ra.w = crec[a]; rb.w = crec[b];
        Coords c;
        uint32_t st = 0, t = 0, kills = 0;
#ifdef RALA_CANDIDATES_STEP_BY_STEP
        (void)classify_one<kSmall>(o, i, ra, rb, c, st, t, kills);
#else
        // (the coordinates with the records, not behind them)
        c.a_begin = o.a_begin[i]; c.a_end = o.a_end[i]; c.b_begin = o.b_begin[i]; c.b_end = o.b_end[i];
        c.length = 0;
        st = o.strand[i];
        asm volatile("" : "+v"(c.a_begin), "+v"(c.a_end), "+v"(c.b_begin), "+v"(c.b_end), "+v"(st));
        (void)classify_loaded<kSmall>(c, st, ra, rb, t, kills);
#endif
        out.src[p] = (uint32_t)(o.base + i);
        out.a_id[p] = a; out.b_id[p] = b;
        out.a_begin[p] = c.a_begin; out.a_end[p] = c.a_end;
        out.b_begin[p] = c.b_begin; out.b_end[p] = c.b_end;
        out.length[p] = c.length;
        out.strand[p] = (uint8_t)st;
        out.type[p] = (uint8_t)t;
    }
}

// sharded runs: 0xFFFFFFFF - (undecided killers of this rank); the all-reduce (min) leaves the
// largest list's size
__global__ void death_status_kernel(const uint32_t* count, uint32_t* status) { *status = 0xFFFFFFFFu - *count; }

// Hill span counters of a sharded run: every rank counted the overlaps of its slice into its own
// copy of the pool.  mode 0: dense[slot of hill] = counter (pits stay 0 in the pre-zeroed array);
// mode 1: counters <- dense (after the all-reduce).
__global__ __launch_bounds__(kBlock) void hill_counts_kernel(ReadState rs, uint32_t n_reads, uint32_t* dense, uint32_t mode) {
    const uint32_t r = blockIdx.x * kBlock + threadIdx.x;
    if (r >= n_reads) return;
    const uint32_t nh = rs.n_hills[r];
    if (nh == 0 || rs.iv_slot[r] == kInf) return;
    const uint32_t first = rs.iv_slot[r] + rs.n_pits[r];
    for (uint32_t q = 0; q < nh; ++q) {
        if (mode == 0) dense[first + q] = rs.pool[first + q].aux;
        else rs.pool[first + q].aux = dense[first + q];
    }
}

// survivor blocks of all ranks (packed_list_view layout) -> the tail's lists: overlaps of rank
// 0, 1, ... in front, internals of rank 0, 1, ... behind them (= file order in both)
__global__ __launch_bounds__(kBlock) void unpack_lists_kernel(const uint8_t* blocks, ListBlocks lb, Survivors out) {
    const uint32_t p = blockIdx.y;
    const uint32_t m = lb.n0[p] + lb.n1[p];
    const uint8_t* block = blocks + lb.block_off[p];
    const uint32_t* c = (const uint32_t*)block;
    for (uint32_t j = blockIdx.x * kBlock + threadIdx.x; j < m; j += gridDim.x * kBlock) {
        const uint32_t d = j < lb.n0[p] ? lb.dst0[p] + j : lb.dst1[p] + (j - lb.n0[p]);
        out.src[d] = c[j]; out.a_id[d] = c[(size_t)m + j]; out.b_id[d] = c[2 * (size_t)m + j];
        out.a_begin[d] = c[3 * (size_t)m + j]; out.a_end[d] = c[4 * (size_t)m + j];
        out.b_begin[d] = c[5 * (size_t)m + j]; out.b_end[d] = c[6 * (size_t)m + j];
        out.length[d] = c[7 * (size_t)m + j];
        out.strand[d] = block[32 * (size_t)m + j];
        out.type[d] = block[33 * (size_t)m + j];
    }
}

inline dim3 grid_for(uint64_t n) { return dim3((unsigned)((n + kBlock - 1) / kBlock)); }

}  // namespace

void launch_dedupe(const OvlSoA& o, uint32_t n_reads, uint8_t* suspect, uint8_t* valid, hipStream_t s) {
    if (!o.n) return;
    (void)hipMemsetAsync(suspect, 0, n_reads, s);
    hipLaunchKernelGGL(dedupe_mark_kernel, grid_for(o.n), dim3(kBlock), 0, s, o, n_reads, suspect);
    hipLaunchKernelGGL(dedupe_kernel, grid_for(o.n), dim3(kBlock), 0, s, o, n_reads, (const uint8_t*)suspect, valid);
}
void launch_dedupe_fix(const OvlSoA& o, uint32_t n_reads, const BucketDedupe& d, hipStream_t s) {
    if (!o.n) return;
    // (the lengths are on the device: both kernels look, the one that is not meant leaves at once)
    const uint32_t grid_list = (uint32_t)std::min<uint64_t>(4096, ((uint64_t)d.list_cap + kBlock / 64 - 1) / (kBlock / 64));
    hipLaunchKernelGGL(dedupe_fix_list_kernel, dim3(std::max<uint32_t>(grid_list, 1)), dim3(kBlock), 0, s, o, n_reads, (const uint32_t*)d.list_pos,
                       (const uint32_t*)d.list_query, (const uint32_t*)d.list_count, d.list_cap, d.valid);
    const uint32_t grid = (uint32_t)std::min<uint64_t>(2048, (o.n + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(dedupe_fix_kernel, dim3(grid), dim3(kBlock), 0, s, o, n_reads, (const uint8_t*)d.suspect, (const uint32_t*)d.list_count,
                       d.list_cap, d.valid);
}
void launch_bucket_fixed(const OvlSoA& o, uint32_t n_reads, uint32_t stride, uint32_t* counts, uint32_t* ev_fixed,
                         uint32_t* over, hipStream_t s) {
    if (o.n) {
        hipLaunchKernelGGL(bucket_fixed_kernel, grid_for(o.n), dim3(kBlock), 0, s, o, n_reads, stride, counts, ev_fixed,
                           over);
    }
}
void launch_bucket_fixed_tuples(const uint2* tuples, uint64_t n, uint32_t n_reads,
                                uint32_t stride, uint32_t* counts, uint32_t* ev_fixed, uint32_t* over, hipStream_t s) {
    if (n) {
        hipLaunchKernelGGL(bucket_fixed_tuples_kernel, grid_for(n), dim3(kBlock), 0, s, tuples, n, n_reads, stride,
                           counts, ev_fixed, over);
    }
}
void launch_count_bounds(const OvlSoA& o, uint32_t n_reads, uint32_t* counts, uint32_t* rank_a, uint32_t* rank_b,
                         hipStream_t s) {
    if (o.n) hipLaunchKernelGGL(count_bounds_kernel, grid_for(o.n), dim3(kBlock), 0, s, o, n_reads, counts, rank_a, rank_b);
}
void launch_scatter_bounds(const OvlSoA& o, uint32_t n_reads, const uint32_t* ev_off, const uint32_t* rank_a,
                           const uint32_t* rank_b, uint32_t* ev, hipStream_t s) {
    if (o.n) {
        hipLaunchKernelGGL(scatter_bounds_kernel, grid_for(o.n), dim3(kBlock), 0, s, o, n_reads, ev_off, rank_a, rank_b, ev);
    }
}
void launch_emit_tuples(const OvlSoA& o, uint32_t n_reads, uint2* tuples, hipStream_t s) {
    if (o.n) hipLaunchKernelGGL(emit_tuples_kernel, grid_for(o.n), dim3(kBlock), 0, s, o, n_reads, tuples);
}
// the owners' buckets side by side: cursor[p] = what lies in front of bucket p (between the two passes of bucket_tuples_kernel)
__global__ void owner_offsets_kernel(const uint32_t* __restrict__ count, uint32_t world, uint32_t* __restrict__ cursor) {
    uint32_t acc = 0;
    for (uint32_t p = 0; p < world; ++p) { cursor[p] = acc; acc += count[p]; }
}
void launch_owner_offsets(const uint32_t* count, uint32_t world, uint32_t* cursor, hipStream_t s) {
    hipLaunchKernelGGL(owner_offsets_kernel, dim3(1), dim3(1), 0, s, count, world, cursor);
}
void launch_bucket_tuples(const OvlSoA& o, uint32_t n_reads, uint32_t world, uint32_t pass, uint32_t* counters,
                          uint2* tuples, hipStream_t s, bool records) {
    if (!o.n) return;
    const dim3 grid((uint32_t)((o.n + kBucketChunk - 1) / kBucketChunk));
    if (records) hipLaunchKernelGGL(bucket_tuples_kernel<true>, grid, dim3(kBlock), 0, s, o, n_reads, world, pass, counters, tuples);
    else hipLaunchKernelGGL(bucket_tuples_kernel<false>, grid, dim3(kBlock), 0, s, o, n_reads, world, pass, counters, tuples);
}

// a bound record as its two tuples (owners whose input does not suit the partitioned bucketing)
__global__ __launch_bounds__(kBlock) void records_to_tuples_kernel(const uint64_t* __restrict__ records, uint64_t n,
                                                                   uint2* __restrict__ tuples) {
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const uint64_t r = records[i];
    const uint32_t local = (uint32_t)(r >> (2 * kRecordCoordBits));
    const uint32_t lo = (uint32_t)(r >> kRecordCoordBits) & kRecordCoordMax, hi = (uint32_t)r & kRecordCoordMax;
    *(uint4*)(tuples + 2 * i) = make_uint4(local, (lo + 15u) << 1, local, ((hi - 15u) << 1) | 1u);
}
void launch_records_to_tuples(const uint64_t* records, uint64_t n, uint2* tuples, hipStream_t s) {
    if (n) hipLaunchKernelGGL(records_to_tuples_kernel, grid_for(n), dim3(kBlock), 0, s, records, n, tuples);
}
void launch_count_tuples(const uint2* tuples, uint64_t n, uint32_t n_reads, uint32_t* counts, hipStream_t s) {
    if (n) hipLaunchKernelGGL(count_tuples_kernel, grid_for(n), dim3(kBlock), 0, s, tuples, n, n_reads, counts);
}
void launch_scatter_tuples(const uint2* tuples, uint64_t n, uint32_t n_reads, uint32_t* cursor, uint32_t* ev,
                           hipStream_t s) {
    if (n) hipLaunchKernelGGL(scatter_tuples_kernel, grid_for(n), dim3(kBlock), 0, s, tuples, n, n_reads, cursor, ev);
}
size_t compact_record_bytes(bool small_records) { return small_records ? 4 : 8; }
void launch_pack_reads(const ReadState& rs, uint32_t n_reads, uint4* rec, void* crec, bool small_records, uint32_t* sure,
                       hipStream_t s) {
    if (!n_reads) return;
    if (small_records) hipLaunchKernelGGL(pack_reads_kernel<true>, grid_for(n_reads), dim3(kBlock), 0, s, rs, n_reads, rec, (uint32_t*)crec, sure);
    else hipLaunchKernelGGL(pack_reads_kernel<false>, grid_for(n_reads), dim3(kBlock), 0, s, rs, n_reads, rec, (uint2*)crec, sure);
}
uint32_t pass2_chunks(uint64_t n_overlaps) { return (uint32_t)((n_overlaps + kClassifyChunk - 1) / kClassifyChunk); }
void launch_classify(const OvlSoA& o, uint32_t n_reads, const uint8_t* valid, const void* crec, bool small_records,
                     const KillList& kl, uint32_t* lo, hipStream_t s) {
    if (!o.n) return;
    const dim3 grid(pass2_chunks(o.n));
    // (16-byte loads want the columns on 16-byte boundaries: a sharded run's slice may start anywhere in them)
    const uintptr_t bits = (uintptr_t)o.a_id | (uintptr_t)o.b_id | (uintptr_t)o.a_begin | (uintptr_t)o.a_end | (uintptr_t)o.b_begin |
                           (uintptr_t)o.b_end | (uintptr_t)o.strand * 4 | (uintptr_t)valid * 4;
    static const bool no_vec = getenv("RALA_CLASSIFY_NO_VEC") != nullptr;       // (measurements)
    const bool vec = (bits & 15u) == 0 && !no_vec;
    if (small_records) {
        if (vec) hipLaunchKernelGGL((classify_kernel<true, true>), grid, dim3(kBlock), 0, s, o, n_reads, valid, (const uint32_t*)crec, kl, lo);
        else hipLaunchKernelGGL((classify_kernel<true, false>), grid, dim3(kBlock), 0, s, o, n_reads, valid, (const uint32_t*)crec, kl, lo);
    } else {
        if (vec) hipLaunchKernelGGL((classify_kernel<false, true>), grid, dim3(kBlock), 0, s, o, n_reads, valid, (const uint2*)crec, kl, lo);
        else hipLaunchKernelGGL((classify_kernel<false, false>), grid, dim3(kBlock), 0, s, o, n_reads, valid, (const uint2*)crec, kl, lo);
    }
}
// the list length lives on the device: a grid sized for what the host knows of it strides over it
// (4096 workgroups that find nothing to do still take 20 us to come and go)
static uint32_t death_grid(uint64_t at_most, uint32_t per_thread = 1) {
    const uint64_t per_block = (uint64_t)kBlock * per_thread;
    if (at_most >= 4096ull * per_block) return 4096;
    return (uint32_t)std::max<uint64_t>(1, (at_most + per_block - 1) / per_block);
}
void launch_death_lower(const KillList& kl, uint32_t* lo, hipStream_t s, uint64_t at_most) {
    hipLaunchKernelGGL(death_lower_kernel, dim3(death_grid(at_most)), dim3(kBlock), 0, s, kl, lo);
}
void launch_death_decide(const KillList& in, const uint32_t* lo, const uint32_t* up, uint32_t* sure, const KillList& out,
                         hipStream_t s, uint64_t at_most) {
    hipLaunchKernelGGL(death_decide_kernel, dim3(death_grid(at_most, 8)), dim3(kBlock), 0, s, in, lo, up, sure, out);
}
void launch_death_diff(uint32_t* older, const uint32_t* newer, uint32_t n, uint32_t* changed, hipStream_t s) {
    if (n) hipLaunchKernelGGL(death_diff_kernel, grid_for(n), dim3(kBlock), 0, s, older, newer, n, changed);
}
void launch_death_tighten(const uint32_t* sure, uint32_t* up, uint32_t* lo, uint32_t n, hipStream_t s, const uint32_t* count,
                          uint32_t* status) {
    if (n) hipLaunchKernelGGL(death_tighten_kernel, grid_for(n), dim3(kBlock), 0, s, sure, up, lo, n, count, status);
}
void launch_survivor_masks(const OvlSoA& o, uint32_t n_reads, const uint8_t* valid, const uint8_t* fate, const uint32_t* death, const void* crec, bool small_records,
                           const uint4* rec, Interval* pool, uint64_t* mask_ov, uint64_t* mask_in, uint32_t* chunk_ov,
                           uint32_t* chunk_in, hipStream_t s) {
    if (!o.n) return;
    const dim3 grid(pass2_chunks(o.n));
    if (small_records) {
        static const uint32_t probe = getenv("RALA_MASKS_PROBE") ? (uint32_t)atoi(getenv("RALA_MASKS_PROBE")) : 0u;   // measurements only
        hipLaunchKernelGGL(survivor_masks_kernel<true>, grid, dim3(kBlock), 0, s, o, n_reads, valid, fate, death, (const uint32_t*)crec, rec, pool,
                           mask_ov, mask_in, chunk_ov, chunk_in, probe);
    } else {
        hipLaunchKernelGGL(survivor_masks_kernel<false>, grid, dim3(kBlock), 0, s, o, n_reads, valid, fate, death, (const uint2*)crec, rec, pool,
                           mask_ov, mask_in, chunk_ov, chunk_in, 0u);
    }
}
void launch_death_status(const uint32_t* count, uint32_t* status, hipStream_t s) {
    hipLaunchKernelGGL(death_status_kernel, dim3(1), dim3(1), 0, s, count, status);
}
void launch_hill_counts(const ReadState& rs, uint32_t n_reads, uint32_t* dense, uint32_t mode, hipStream_t s) {
    if (n_reads) hipLaunchKernelGGL(hill_counts_kernel, grid_for(n_reads), dim3(kBlock), 0, s, rs, n_reads, dense, mode);
}
void launch_unpack_lists(const uint8_t* blocks, const ListBlocks& lb, const Survivors& out, hipStream_t s) {
    uint32_t most = 0;
    for (uint32_t p = 0; p < lb.world; ++p) most = most > lb.n0[p] + lb.n1[p] ? most : lb.n0[p] + lb.n1[p];
    if (most == 0) return;
    const uint32_t gx = (most + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(unpack_lists_kernel, dim3(gx < 1024 ? gx : 1024, lb.world), dim3(kBlock), 0, s, blocks, lb, out);
}
void launch_apply_death(const uint32_t* death, uint8_t* alive, const uint32_t* n_hills, uint8_t* fate, uint32_t n_reads,
                        uint32_t* n_alive, hipStream_t s) {
    if (n_reads) {
        const uint32_t blocks = std::min<uint32_t>(256, (n_reads + kBlock - 1) / kBlock);
        hipLaunchKernelGGL(apply_death_kernel, dim3(blocks), dim3(kBlock), 0, s, death, alive, n_hills, fate, n_reads, n_alive);
    }
}
void launch_gather_survivors(const OvlSoA& o, const uint64_t* mask_ov, const uint64_t* mask_in, const void* crec,
                             bool small_records, const uint32_t* chunk_ov_off, const uint32_t* chunk_in_off, uint32_t n_ov_total,
                             const Survivors& out, hipStream_t s) {
    if (!o.n) return;
    const dim3 grid(pass2_chunks(o.n));
    if (small_records) {
        hipLaunchKernelGGL(gather_kernel<true>, grid, dim3(kBlock), 0, s, o, mask_ov, mask_in, (const uint32_t*)crec, chunk_ov_off,
                           chunk_in_off, n_ov_total, out);
    } else {
        hipLaunchKernelGGL(gather_kernel<false>, grid, dim3(kBlock), 0, s, o, mask_ov, mask_in, (const uint2*)crec, chunk_ov_off,
                           chunk_in_off, n_ov_total, out);
    }
}

}  // namespace rala_hip
