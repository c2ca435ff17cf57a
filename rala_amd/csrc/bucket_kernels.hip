// Bound events by read WITHOUT a memory-side atomic per overlap: the target side is partitioned.
//
// Graph::initialize pushes four bounds per overlap to the lists of its two reads (reference
// src/graph.cpp:311-326).  The query side is cheap: overlap files are grouped by query, so the lanes of a
// wavefront mostly share the query and one counting atomic serves a whole segment of them.  The target
// side is a scatter by read id, and the single-pass kernel (bucket_fixed_kernel, overlap_kernels.hip) pays
// one memory-side atomic and one 32-byte partial write per overlap for it: 21 G/s of each at C3, 0.10 of
// the HBM peak for the 40 bytes it moves (profiles/r03_c3_pmc_bucket_fixed.json).  Here the target side
// goes through two partitioning passes whose writes are whole lines, and the events land in an exact CSR
// (ev_off / ev) that every pile kernel already reads:
//
//   count         target-side records per group of 128 reads: one persistent workgroup per compute unit keeps a
//                 histogram of ALL groups in LDS (31 KB per million reads) over its share of the file
//   layout        one scan over the groups: where every group's records go at level 2, and with that every
//                 partition's (32 groups) at level 1; the table of level-2 tiles, none of which crosses a
//                 partition (one workgroup)
//   l1_scatter    the tile's target records {target & 4095, begin, end} (8 bytes) sorted by partition in LDS,
//                 copied out partition by partition - consecutive lanes write consecutive records; one global
//                 add per (tile, partition) reserves the place.  Also the query-side event counts per read
//                 (one add per wavefront segment).
//   l2_scatter    the same inside every partition of 4096 reads: groups of 128 reads
//   group sums    query-side pairs per group of 128 reads (acount[] from l1_scatter) and, with the groups' record counts,
//                 where every group's rows start (two small kernels)
//   final         one workgroup per group: target-side events per read (LDS histogram), the reads' row offsets (scan over
//                 128 values) -> ev_off, then the group's records at places handed out by LDS counters - 8-byte stores
//                 inside a window of 200 KB that the L2 merges into whole lines
//   query_side    one thread per overlap, as in the single-pass kernel: segment base from one atomic
//
// Sizes are counted, not guessed: files that name every pair once, query = the lower id, make a read a target
// in proportion to its id - the first groups of the file hold next to nothing, the last ones twice the
// average.
//
// The order of a read's events is irrelevant (Pile::add_layers sorts them; the run-space kernel does not
// even need that), so the result equals the other bucketing paths' as a multiset per read.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "device_utils.h"
#include "geom.h"
#include "kernels.h"
#include "scan_pass.h"

namespace rala_hip {

namespace {

constexpr uint32_t kInf = 0xFFFFFFFFu;
constexpr uint32_t kTile = 4096;                // records per workgroup
constexpr uint32_t kBlockP = 512;
constexpr uint32_t kPer = kTile / kBlockP;
constexpr uint32_t kL1Shift = 12;               // 4096 reads per level-1 partition (a sharded run's owners; one GPU: PartGeom below)
constexpr uint32_t kL1Reads = 1u << kL1Shift;
constexpr uint32_t kGroupShift = 7;             // 128 reads per final group
constexpr uint32_t kGroupReads = 1u << kGroupShift;
constexpr uint32_t kGroupsPerPart = kL1Reads / kGroupReads;
constexpr uint32_t kKeyBits = 14;               // the low bits of the read a record keeps: enough for the largest partition
constexpr uint32_t kCoordBits = 25;
constexpr uint32_t kCoordMax = (1u << kCoordBits) - 1u;
static_assert(kKeyBits + 2 * kCoordBits == 64, "a record is 8 bytes");

// One GPU (round 6): the first scatter's bins are partitions of 4096 reads up to a million reads - and of 8192 or 16384 beyond, so
// that there are never more than 256 of them.  A tile of 4096 records spread over C5's 977 partitions wrote runs of four records
// (34 bytes); over 245 partitions of 16384 reads it writes whole lines again, and the second scatter takes the two more bits
// (128 groups per partition instead of 32).  C5's first scatter: 2.45 ms for 6x C3's records where C3 takes 0.31.
struct PartGeom {
    uint32_t shift, gpp, n_part;                // reads per partition = 1 << shift; groups per partition; partitions
};
}  // namespace
uint32_t g_part_shift = 0;                      // (option "debug_part_shift", tests and measurements: 12 .. 14; 0 = by the rule)
uint32_t g_count_window = 0;                    // (option "debug_count_window", tests: groups per counting pass; 0 = what the LDS holds)
namespace {
inline PartGeom part_geom(uint32_t n_reads) {
    uint32_t shift = kL1Shift;
    while (shift < kKeyBits && ((uint64_t)n_reads + (1u << shift) - 1) >> shift > 256) ++shift;
    if (g_part_shift >= kL1Shift && g_part_shift <= kKeyBits) shift = g_part_shift;
    return PartGeom{shift, 1u << (shift - kGroupShift), (uint32_t)(((uint64_t)n_reads + (1u << shift) - 1) >> shift)};
}

// groups of 128 reads whose counts one counting pass keeps in a workgroup's LDS (4 bytes each in 150 KB: 4.9 M reads)
constexpr uint32_t kCountWindow = 150u * 1024u / 4u;
inline uint32_t count_window() { return g_count_window ? g_count_window : kCountWindow; }

__device__ __forceinline__ uint64_t pack_record(uint32_t b, uint32_t begin, uint32_t end) {
    // coordinates beyond 2^25 lie outside every read this path takes (partition_path_fits) and stay outside
    return (uint64_t)(b & ((1u << kKeyBits) - 1u)) << (2 * kCoordBits) | (uint64_t)(begin < kCoordMax ? begin : kCoordMax) << kCoordBits |
           (uint64_t)(end < kCoordMax ? end : kCoordMax);
}
__device__ __forceinline__ uint32_t rec_key(uint64_t r) { return (uint32_t)(r >> (2 * kCoordBits)); }
// shrink: 15 for the primary overlaps (graph.cpp:317-324), 0 for the sensitive ones (graph.cpp:929-933: no +-15)
__device__ __forceinline__ uint2 rec_events(uint64_t r, uint32_t shrink) {
    const uint32_t begin = (uint32_t)(r >> kCoordBits) & kCoordMax, end = (uint32_t)r & kCoordMax;
    return make_uint2((begin + shrink) << 1, ((end - shrink) << 1) | 1u);
}

// lanes are consecutive overlaps; a segment = maximal run of active lanes with equal key.  Returns the
// segment's length in its first lane (0 elsewhere), `leader` = first lane of this lane's segment.
__device__ __forceinline__ uint32_t segment_of(uint32_t key, bool active, uint32_t lane, uint32_t& leader) {
    const uint32_t prev = (uint32_t)__shfl_up((int)key, 1, 64);
    const bool prev_active = __shfl_up((int)active, 1, 64) != 0;
    const bool head = active && (lane == 0 || !prev_active || prev != key);
    const uint64_t heads = __ballot(head);
    const uint64_t act = __ballot(active);
    const uint64_t below = heads & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
    leader = below ? 63u - (uint32_t)__clzll((long long)below) : lane;
    const uint64_t after = (lane == 63) ? 0ull : ((heads | ~act) >> (lane + 1));
    const uint32_t len = after ? (uint32_t)__ffsll((unsigned long long)after) : (64u - lane);
    return head ? len : 0u;
}

// ---- level 1: target >> 12 ------------------------------------------------------------------------
constexpr uint32_t kBlockC = 1024;
constexpr uint32_t kCountPer = 8;
// g_lo, n_groups (round 6): the window of groups this launch counts - all of them where their histogram fits a workgroup's LDS
// (4.9 M reads), otherwise window after window, the ids streamed once per window (count_window)
__global__ __launch_bounds__(kBlockC) void group_count_kernel(OvlSoA o, uint32_t n_reads, uint32_t g_lo, uint32_t n_groups, uint32_t* group_count) {
    extern __shared__ uint32_t s_hist[];
    for (uint32_t g = threadIdx.x; g < n_groups; g += kBlockC) s_hist[g] = 0;
    __syncthreads();
    // this workgroup's share of the file: whole chunks of kBlockC * kC overlaps, interleaved with the others'
    // (the loads in flight are what sets this kernel's rate: 4 per thread and one workgroup per compute unit ran at
    // 3 TB/s)
    constexpr uint32_t kC = kCountPer;
    const uint64_t last = o.n - 1;
    for (uint64_t i0 = (uint64_t)blockIdx.x * kBlockC * kC; i0 < o.n; i0 += (uint64_t)gridDim.x * kBlockC * kC) {
        uint32_t a[kC], b[kC];
#pragma unroll
        for (uint32_t u = 0; u < kC; ++u) {
            const uint64_t i = i0 + u * kBlockC + threadIdx.x;
            const uint64_t j = i < o.n ? i : last;
            a[u] = i < o.n ? __builtin_nontemporal_load(o.a_id + j) : kInf;
            b[u] = __builtin_nontemporal_load(o.b_id + j);
        }
#pragma unroll
        for (uint32_t u = 0; u < kC; ++u) {
            const uint32_t g = (b[u] >> kGroupShift) - g_lo;
            if (a[u] < n_reads && b[u] < n_reads && g < n_groups) atomicAdd(&s_hist[g], 1u);
        }
    }
    __syncthreads();
    for (uint32_t g = threadIdx.x; g < n_groups; g += kBlockC) {
        const uint32_t c = s_hist[g];
        if (c) atomicAdd(&group_count[g_lo + g], c);
    }
}

// The counting pass AND duplicate removal's first pass in one (round 5).  Both read the two id columns of every overlap;
// duplicate removal ran beside the bucketing on a second stream and cost it 0.14 ms at C3, its own two kernels moved the
// ids twice more.  Here a thread takes four CONSECUTIVE overlaps per group (16-byte loads) and, besides the histogram:
//   * marks the queries that own a run whose targets are not strictly increasing, or a record that does not resolve
//     (dedupe_mark_kernel's rule, overlap_kernels.hip; the element in front of a thread's four comes from the lane below);
//   * writes the validity byte that holds for every query that is NOT marked - resolvable and no self overlap
//     (graph.cpp:273-307 on a run of strictly increasing targets) - four of them as one word;
//   * LISTS where it marked (position, query) - staged per trip in LDS, one add to the list's counter per workgroup and trip:
//     dedupe_fix_list_kernel (overlap_kernels.hip) then redoes just the runs around those positions with the full
//     comparison.  (First version of this round: a flag "anybody marked" and a pass over all overlaps behind it - at C3,
//     where repeats make 1 - 2 % of the queries suspect, that pass ran 0.33 ms beside the first scatter and cost it 0.08.)
//     A list that outgrows its room (a file whose runs are not sorted by target at all) leaves its counter beyond the
//     capacity and the pass over all overlaps takes over.
// Needs the id columns on 16-byte boundaries (the caller looks).
__device__ __forceinline__ void load4_ids(const uint32_t* p, uint64_t i0, uint64_t n, uint32_t (&out)[4]) {
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    if (i0 + 4 <= n) {
        const u32x4_t v = __builtin_nontemporal_load((const u32x4_t*)(p + i0));
        out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
    } else {
#pragma unroll
        for (uint32_t e = 0; e < 4; ++e) out[e] = i0 + e < n ? p[i0 + e] : kInf;
    }
}
constexpr uint32_t kCountVec = 2;               // groups of four overlaps per thread and trip
constexpr uint32_t kFlagStage = 512;            // marks a workgroup stages per trip
// kShard: the sender's count of a sharded run (shard_count_kernel's bins: both sides of an overlap, by owner and the owner's
// group) with the same first pass of duplicate removal on the way; n_groups is then world * groups.
template <bool kShard>
__global__ __launch_bounds__(kBlockC) void group_count_dedupe_kernel(OvlSoA o, uint32_t n_reads, uint32_t n_groups, uint32_t* group_count,
                                                                     uint8_t* __restrict__ suspect, uint8_t* __restrict__ valid, uint32_t* list_pos,
                                                                     uint32_t* list_query, uint32_t list_cap, uint32_t* list_count,
                                                                     uint32_t world, uint32_t groups) {
    extern __shared__ uint32_t s_hist[];
    __shared__ uint32_t s_fpos[kFlagStage], s_fq[kFlagStage], s_fcnt, s_fbase;
    for (uint32_t g = threadIdx.x; g < n_groups; g += kBlockC) s_hist[g] = 0;
    if (threadIdx.x == 0) s_fcnt = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    auto mark = [&](uint32_t query, uint64_t at) {
        suspect[query] = 1;
        const uint32_t slot = atomicAdd(&s_fcnt, 1u);
        if (slot < kFlagStage) { s_fpos[slot] = (uint32_t)at; s_fq[slot] = query; }
    };
    constexpr uint64_t kTrip = (uint64_t)kBlockC * 4 * kCountVec;
    for (uint64_t c0 = (uint64_t)blockIdx.x * kTrip; c0 < o.n; c0 += (uint64_t)gridDim.x * kTrip) {
        uint32_t a[kCountVec][4], b[kCountVec][4];
#pragma unroll
        for (uint32_t u = 0; u < kCountVec; ++u) {
            const uint64_t i0 = c0 + ((uint64_t)u * kBlockC + threadIdx.x) * 4;
            load4_ids(o.a_id, i0, o.n, a[u]);
            load4_ids(o.b_id, i0, o.n, b[u]);
        }
#pragma unroll
        for (uint32_t u = 0; u < kCountVec; ++u) {
            const uint64_t i0 = c0 + ((uint64_t)u * kBlockC + threadIdx.x) * 4;
            // the overlap in front of this thread's four: the last one of the lane below, or (first lane of a wavefront) from memory
            uint32_t pa = (uint32_t)__shfl_up((int)a[u][3], 1, 64), pb = (uint32_t)__shfl_up((int)b[u][3], 1, 64);
            if (lane == 0 && i0 > 0 && i0 < o.n) { pa = o.a_id[i0 - 1]; pb = o.b_id[i0 - 1]; }
            bool have_prev = i0 > 0;
            uint32_t word = 0;
#pragma unroll
            for (uint32_t e = 0; e < 4; ++e) {
                const uint32_t x = a[u][e], y = b[u][e];
                const bool in = i0 + e < o.n;
                const bool ok = in && x < n_reads && y < n_reads;
                if (ok) {
                    if constexpr (kShard) {
                        atomicAdd(&s_hist[(x % world) * groups + ((x / world) >> kGroupShift)], 1u);
                        atomicAdd(&s_hist[(y % world) * groups + ((y / world) >> kGroupShift)], 1u);
                    } else {
                        // (the first window of groups - usually all of them; the windows behind it: group_count_kernel)
                        if ((y >> kGroupShift) < n_groups) atomicAdd(&s_hist[y >> kGroupShift], 1u);
                    }
                }
                if (in && have_prev) {
                    const bool pok = pa < n_reads && pb < n_reads;
                    if (ok && pok) {
                        if (pa == x && y <= pb) mark(x, i0 + e);
                    } else {
                        // an unresolved record hides the order of its neighbours: flag both queries
                        if (x < n_reads) mark(x, i0 + e);
                        if (pa < n_reads) mark(pa, i0 + e - 1);
                    }
                }
                if (ok && x != y) word |= 1u << (8 * e);
                pa = x; pb = y; have_prev = true;
            }
            if (i0 + 4 <= o.n) *(uint32_t*)(valid + i0) = word;
            else for (uint32_t e = 0; e < 4 && i0 + e < o.n; ++e) valid[i0 + e] = (uint8_t)(word >> (8 * e));
        }
        // this trip's marks to the list: one add to its counter (a trip with more marks than the stage holds: the list is
        // given up - its counter goes beyond its capacity - and the pass over all overlaps will do the work)
        __syncthreads();
        const uint32_t staged = s_fcnt;
        if (staged) {                           // (the same answer in every thread)
            // (giving up is a bit that stays - kDedupeListGivenUp, beyond every capacity - not an addition that 4096 such trips
            // would carry around the 32-bit counter and back below the capacity; what the trips that do fit add stays below
            // it: at most kFlagStage marks a trip, 2^17 trips in 2^30 overlaps)
            if (threadIdx.x == 0) s_fbase = staged > kFlagStage ? atomicOr(list_count, kDedupeListGivenUp) : atomicAdd(list_count, staged);
            __syncthreads();
            if (staged <= kFlagStage) {
                for (uint32_t k = threadIdx.x; k < staged; k += kBlockC) {
                    const uint32_t at = s_fbase + k;
                    if (at < list_cap) { list_pos[at] = s_fpos[k]; list_query[at] = s_fq[k]; }
                }
            }
            __syncthreads();
            if (threadIdx.x == 0) s_fcnt = 0;
            __syncthreads();
        }
    }
    for (uint32_t g = threadIdx.x; g < n_groups; g += kBlockC) {
        const uint32_t c = s_hist[g];
        if (c) atomicAdd(&group_count[g], c);
    }
}

// From the groups' counts (n_groups of them, padded with empty ones to n_part * kGroupsPerPart): group_base[0 ..
// n] and group_cursor = exclusive prefix; part_cursor[p] = group_base[p * kGroupsPerPart] (a partition's records
// lie where its groups' will); the table of level-2 tiles - tile t covers records tile_lo[t] .. tile_hi[t] of
// partition tile_part[t], *n_tiles of them.  One workgroup.
__global__ __launch_bounds__(1024) void layout_kernel(const uint32_t* __restrict__ group_count, uint32_t n_part, uint32_t gpp,
                                                      uint32_t* __restrict__ group_base, uint32_t* __restrict__ group_cursor,
                                                      uint32_t* __restrict__ part_cursor, uint32_t* __restrict__ tile_part,
                                                      uint32_t* __restrict__ tile_lo, uint32_t* __restrict__ tile_hi, uint32_t* n_tiles) {
    __shared__ uint32_t tmp[1024 / 64 + 1];
    const uint32_t n = n_part * gpp;
    uint32_t carry = 0;
    for (uint32_t g0 = 0; g0 < n; g0 += 1024) {
        const uint32_t g = g0 + threadIdx.x;
        const uint32_t c = g < n ? group_count[g] : 0u;
        uint32_t tot;
        const uint32_t ex = block_scan_excl<1024>(c, OpAdd(), 0u, tmp, tot);
        if (g < n) {
            group_base[g] = carry + ex;
            group_cursor[g] = carry + ex;
            if (g % gpp == 0) part_cursor[g / gpp] = carry + ex;
        }
        carry += tot;
    }
    if (threadIdx.x == 0) group_base[n] = carry;
    __syncthreads();
    uint32_t tile_carry = 0;
    for (uint32_t p0 = 0; p0 < n_part; p0 += 1024) {
        const uint32_t p = p0 + threadIdx.x;
        const uint32_t lo = p < n_part ? group_base[p * gpp] : 0u;
        const uint32_t hi = p < n_part ? group_base[(p + 1u) * gpp] : 0u;
        const uint32_t tiles = (hi - lo + kTile - 1) / kTile;
        uint32_t ttot;
        const uint32_t tex = block_scan_excl<1024>(tiles, OpAdd(), 0u, tmp, ttot);
        for (uint32_t k = 0; k < tiles; ++k) {
            tile_part[tile_carry + tex + k] = p;
            tile_lo[tile_carry + tex + k] = lo + k * kTile;
            tile_hi[tile_carry + tex + k] = umin(hi, lo + (k + 1u) * kTile);
        }
        tile_carry += ttot;
    }
    if (threadIdx.x == 0) *n_tiles = tile_carry;
}

// dynamic LDS: stage[kTile] (8 B), bin_of[kTile] (2 B), hist / off / gbase [n_bins] (4 B each)
struct StageLds {
    uint64_t* stage;
    uint16_t* bin_of;
    uint32_t *hist, *off, *gbase;
    __device__ StageLds(unsigned char* base, uint32_t n_bins) {
        stage = (uint64_t*)base;
        bin_of = (uint16_t*)(stage + kTile);
        hist = (uint32_t*)(bin_of + kTile);
        off = hist + n_bins;
        gbase = off + n_bins;
    }
};
inline size_t stage_lds_bytes(uint32_t n_bins) { return (size_t)kTile * 10 + (size_t)n_bins * 12 + 16; }

// The records of one tile, each with its bin and whether it counts: ranks inside the bins (LDS adds), the
// bins' places in the staging area (scan) and in the output (one add per bin with records), the records
// sorted by bin in LDS, then copied out - consecutive lanes, consecutive addresses.
__device__ __forceinline__ void stage_and_copy(StageLds& L, uint32_t n_bins, const uint64_t* rec, const uint32_t* bin, const bool* in,
                                               uint32_t* cursor, uint64_t* __restrict__ out, uint32_t* tmp) {
    for (uint32_t p = threadIdx.x; p < n_bins; p += kBlockP) L.hist[p] = 0;
    __syncthreads();
    uint32_t rank[kPer];
#pragma unroll
    for (uint32_t u = 0; u < kPer; ++u) rank[u] = in[u] ? atomicAdd(&L.hist[bin[u]], 1u) : 0u;
    __syncthreads();
    uint32_t carry = 0;
    for (uint32_t p0 = 0; p0 < n_bins; p0 += kBlockP) {
        const uint32_t p = p0 + threadIdx.x;
        const uint32_t c = p < n_bins ? L.hist[p] : 0u;
        uint32_t tot;
        const uint32_t ex = block_scan_excl<(int)kBlockP>(c, OpAdd(), 0u, tmp, tot);
        if (p < n_bins) {
            L.off[p] = carry + ex;
            L.gbase[p] = c ? atomicAdd(&cursor[p], c) : 0u;
        }
        carry += tot;
    }
    __syncthreads();
    const uint32_t total = carry;
#pragma unroll
    for (uint32_t u = 0; u < kPer; ++u) {
        if (in[u]) {
            const uint32_t at = L.off[bin[u]] + rank[u];
            L.stage[at] = rec[u];
            L.bin_of[at] = (uint16_t)bin[u];
        }
    }
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < total; j += kBlockP) {
        const uint32_t p = L.bin_of[j];
        out[L.gbase[p] + (j - L.off[p])] = L.stage[j];
    }
}

// + acount[a] = resolvable overlaps of query a (its query-side events / 2)
__global__ __launch_bounds__(kBlockP) void l1_scatter_kernel(OvlSoA o, uint32_t n_reads, uint32_t n_part, uint32_t shift, uint32_t* part_cursor,
                                                             uint64_t* __restrict__ rec1, uint32_t* acount) {
    extern __shared__ __align__(16) unsigned char s_raw[];
    __shared__ uint32_t tmp[kBlockP / 64 + 1];
    StageLds L(s_raw, n_part);
    uint64_t rec[kPer];
    uint32_t bin[kPer];
    bool in[kPer];
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t last = o.n - 1;
#pragma unroll
    for (uint32_t u = 0; u < kPer; ++u) {
        const uint64_t i = (uint64_t)blockIdx.x * kTile + u * kBlockP + threadIdx.x;
        const uint64_t j = i < o.n ? i : last;
        const uint32_t a = o.a_id[j], b = o.b_id[j];
        in[u] = i < o.n && a < n_reads && b < n_reads;
        rec[u] = pack_record(b, o.b_begin[j], o.b_end[j]);
        bin[u] = in[u] ? b >> shift : 0u;
        uint32_t leader;
        const uint32_t seg = segment_of(a, in[u], lane, leader);
        if (seg) atomicAdd(&acount[a], seg);
    }
    stage_and_copy(L, n_part, rec, bin, in, part_cursor, rec1, tmp);
}

// ---- the same two kernels for an owner rank's input: bound records {local read : 22 | begin : 21 | end : 21}, both sides
// of every overlap as records of their own (overlap_kernels.hip: bound_record) -------------------------------
__device__ __forceinline__ uint32_t bound_record_read(uint64_t r) { return (uint32_t)(r >> (2 * kBoundRecordCoordBits)); }

__global__ __launch_bounds__(kBlockC) void group_count_records_kernel(const uint64_t* __restrict__ records, uint64_t n, uint32_t n_reads,
                                                                      uint32_t g_lo, uint32_t n_groups, uint32_t* group_count) {
    extern __shared__ uint32_t s_hist[];
    for (uint32_t g = threadIdx.x; g < n_groups; g += kBlockC) s_hist[g] = 0;
    __syncthreads();
    constexpr uint32_t kC = kCountPer;
    for (uint64_t i0 = (uint64_t)blockIdx.x * kBlockC * kC; i0 < n; i0 += (uint64_t)gridDim.x * kBlockC * kC) {
        uint32_t key[kC];
#pragma unroll
        for (uint32_t u = 0; u < kC; ++u) {
            const uint64_t i = i0 + u * kBlockC + threadIdx.x;
            key[u] = i < n ? bound_record_read(__builtin_nontemporal_load(records + i)) : kInf;
        }
#pragma unroll
        for (uint32_t u = 0; u < kC; ++u) {
            const uint32_t g = (key[u] >> kGroupShift) - g_lo;
            if (key[u] < n_reads && g < n_groups) atomicAdd(&s_hist[g], 1u);
        }
    }
    __syncthreads();
    for (uint32_t g = threadIdx.x; g < n_groups; g += kBlockC) {
        const uint32_t c = s_hist[g];
        if (c) atomicAdd(&group_count[g_lo + g], c);
    }
}

__global__ __launch_bounds__(kBlockP) void l1_scatter_records_kernel(const uint64_t* __restrict__ records, uint64_t n, uint32_t n_reads,
                                                                     uint32_t n_part, uint32_t shift, uint32_t* part_cursor, uint64_t* __restrict__ rec1) {
    extern __shared__ __align__(16) unsigned char s_raw[];
    __shared__ uint32_t tmp[kBlockP / 64 + 1];
    StageLds L(s_raw, n_part);
    uint64_t rec[kPer];
    uint32_t bin[kPer];
    bool in[kPer];
    constexpr uint32_t kMask = (1u << kBoundRecordCoordBits) - 1u;
#pragma unroll
    for (uint32_t u = 0; u < kPer; ++u) {
        const uint64_t i = (uint64_t)blockIdx.x * kTile + u * kBlockP + threadIdx.x;
        const uint64_t r = i < n ? __builtin_nontemporal_load(records + i) : ~0ull;
        const uint32_t key = bound_record_read(r);
        in[u] = i < n && key < n_reads;
        rec[u] = pack_record(key, (uint32_t)(r >> kBoundRecordCoordBits) & kMask, (uint32_t)r & kMask);
        bin[u] = in[u] ? key >> shift : 0u;
    }
    stage_and_copy(L, n_part, rec, bin, in, part_cursor, rec1, tmp);
}

// ---- level 2: inside every partition, groups of 128 reads; tiles that do not cross a partition -------
// (tile_part's top bit: the tile lies behind rec1b - an owner rank's own block, which stays in its send buffer)
constexpr uint32_t kTileOtherBase = 0x80000000u;
__global__ __launch_bounds__(kBlockP) void l2_scatter_kernel(const uint64_t* __restrict__ rec1, const uint64_t* __restrict__ rec1b,
                                                             const uint32_t* __restrict__ tile_part,
                                                             const uint32_t* __restrict__ tile_lo, const uint32_t* __restrict__ tile_hi,
                                                             const uint32_t* __restrict__ n_tiles, uint32_t* group_cursor,
                                                             uint64_t* __restrict__ rec2, uint32_t gpp) {
    extern __shared__ __align__(16) unsigned char s_raw[];
    __shared__ uint32_t tmp[kBlockP / 64 + 1];
    if (blockIdx.x >= *n_tiles) return;
    StageLds L(s_raw, gpp);
    const uint32_t part_mask = gpp * kGroupReads - 1u;      // (a record keeps 14 bits of its read: those of its partition)
    const uint32_t part_word = tile_part[blockIdx.x], lo = tile_lo[blockIdx.x], hi = tile_hi[blockIdx.x];
    const uint32_t part = part_word & ~kTileOtherBase;
    const uint64_t* __restrict__ src = (part_word & kTileOtherBase) ? rec1b : rec1;
    uint64_t rec[kPer];
    uint32_t bin[kPer];
    bool in[kPer];
#pragma unroll
    for (uint32_t u = 0; u < kPer; ++u) {
        const uint32_t j = lo + u * kBlockP + threadIdx.x;
        in[u] = j < hi;
        rec[u] = in[u] ? src[j] : 0ull;
        bin[u] = (rec_key(rec[u]) & part_mask) >> kGroupShift;
    }
    stage_and_copy(L, gpp, rec, bin, in, group_cursor + part * gpp, rec2, tmp);
}

// ---- final: one workgroup per group of 128 reads -------------------------------------------------------
// Where a group's events start follows from what is known once the first scatter is done - the group's records
// (its target side, group_base[]) and its reads' query-side counts (acount[]): two small kernels add those up
// (group_query_sum_kernel, group_event_base_kernel) and the final kernel does the rest per group - the events per
// read (the group's records, counted in LDS), the reads' row offsets (a scan over 128 values), the rows.  (Before:
// a counting kernel over all records, a device-wide scan of the per-read counts, then the writing kernel - the
// records came from memory twice; now the second pass over a group's 50 KB of records finds them in the L2.)
__global__ __launch_bounds__(kGroupReads) void group_query_sum_kernel(const uint32_t* __restrict__ acount, uint32_t n_reads,
                                                                      uint32_t* __restrict__ qsum) {
    __shared__ uint32_t tmp[kGroupReads / 64 + 1];
    const uint32_t r = blockIdx.x * kGroupReads + threadIdx.x;
    const uint32_t v = block_reduce<(int)kGroupReads>(r < n_reads ? acount[r] : 0u, OpAdd(), 0u, tmp);
    if (threadIdx.x == 0) qsum[blockIdx.x] = v;
}

// pair_base[g] = bound PAIRS in front of group g's rows (exclusive prefix of query-side + target-side pairs), [n] = all
__global__ __launch_bounds__(1024) void group_event_base_kernel(const uint32_t* __restrict__ qsum, const uint32_t* __restrict__ group_base,
                                                                uint32_t n_groups, uint32_t* __restrict__ pair_base) {
    __shared__ uint32_t tmp[1024 / 64 + 1];
    uint32_t carry = 0;
    for (uint32_t g0 = 0; g0 < n_groups; g0 += 1024) {
        const uint32_t g = g0 + threadIdx.x;
        const uint32_t c = g < n_groups ? qsum[g] + (group_base[g + 1] - group_base[g]) : 0u;
        uint32_t tot;
        const uint32_t ex = block_scan_excl<1024>(c, OpAdd(), 0u, tmp, tot);
        if (g < n_groups) pair_base[g] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) pair_base[n_groups] = carry;
}

// target side: the group's records, tile by tile, sorted by read in LDS (as bound pairs) and copied out to the
// reads' rows behind their query-side events: consecutive lanes write consecutive pairs of one read.
// (Each record stored straight from the lane that read it - an 8-byte store to one of 12 000 places in a
// window of 200 KB - took 0.75 ms at C3: one request per lane, whatever the L2 merges afterwards.)
__global__ __launch_bounds__(kBlockP) void final_kernel(const uint64_t* __restrict__ rec2, const uint32_t* __restrict__ group_base,
                                                        const uint32_t* __restrict__ pair_base, uint32_t n_reads,
                                                        const uint32_t* __restrict__ acount, uint32_t* __restrict__ ev_off,
                                                        uint32_t* __restrict__ ev, uint32_t shrink, uint32_t ev_shift) {
    extern __shared__ __align__(16) unsigned char s_raw[];
    __shared__ uint32_t tmp[kBlockP / 64 + 1];
    __shared__ uint32_t s_cursor[kGroupReads];          // next free PAIR of every read's row
    __shared__ uint32_t s_cnt[kGroupReads];
    static_assert(kGroupReads <= kBlockP, "one thread per read of the group");
    StageLds L(s_raw, kGroupReads);
    const uint32_t g = blockIdx.x;
    if (threadIdx.x < kGroupReads) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t lo = group_base[g], hi = group_base[g + 1];
    // (the counting pass keeps what the first tile below will want - the same records by the same threads: a group's 6 500 records
    // are two tiles, and the first one need not come from the L2 a second time)
    uint64_t first[kPer];
#pragma unroll
    for (uint32_t u = 0; u < kPer; ++u) {
        const uint32_t j = lo + u * kBlockP + threadIdx.x;
        first[u] = j < hi ? rec2[j] : 0ull;
        if (j < hi) atomicAdd(&s_cnt[rec_key(first[u]) & (kGroupReads - 1u)], 1u);
    }
    for (uint32_t j = lo + kTile + threadIdx.x; j < hi; j += kBlockP) atomicAdd(&s_cnt[rec_key(rec2[j]) & (kGroupReads - 1u)], 1u);
    __syncthreads();
    {
        // the reads' rows: query-side pairs, then target-side pairs; offsets in events (two per pair; ev_shift 0) or in pairs (1)
        const uint32_t r = g * kGroupReads + threadIdx.x;
        const bool mine = threadIdx.x < kGroupReads && r < n_reads;
        const uint32_t q = mine && acount ? acount[r] : 0u;
        const uint32_t pairs = mine ? q + s_cnt[threadIdx.x] : 0u;
        uint32_t tot;
        const uint32_t before = pair_base[g] + block_scan_excl<(int)kBlockP>(pairs, OpAdd(), 0u, tmp, tot);
        if (mine) {
            ev_off[r] = before << (1u - ev_shift);
            s_cursor[threadIdx.x] = before + q;
            if (r == n_reads - 1u) ev_off[n_reads] = (before + pairs) << (1u - ev_shift);
        }
    }
    __syncthreads();
    for (uint32_t j0 = lo; j0 < hi; j0 += kTile) {
        uint64_t pair[kPer];
        uint32_t bin[kPer];
        bool in[kPer];
#pragma unroll
        for (uint32_t u = 0; u < kPer; ++u) {
            const uint32_t j = j0 + u * kBlockP + threadIdx.x;
            in[u] = j < hi;
            const uint64_t rec = j0 == lo ? first[u] : in[u] ? rec2[j] : 0ull;
            const uint2 e = rec_events(rec, shrink);
            pair[u] = (uint64_t)e.x | (uint64_t)e.y << 32;
            bin[u] = rec_key(rec) & (kGroupReads - 1u);
        }
        stage_and_copy(L, kGroupReads, pair, bin, in, s_cursor, (uint64_t*)ev, tmp);
        __syncthreads();
    }
}

// query side: one thread per overlap; the lanes of a wavefront that share the query take their places from
// ONE atomic (the file is grouped by query).  written[] counts what a read has handed out so far - events, like ev_off
// (ev_shift 0), or pairs like it (1).
template <uint32_t kQ>
__global__ __launch_bounds__(256) void query_side_kernel(OvlSoA o, uint32_t n_reads, const uint32_t* __restrict__ ev_off,
                                                         uint32_t* written, uint32_t* __restrict__ ev, uint32_t ev_shift) {
    const uint32_t lane = threadIdx.x & 63;
    uint32_t a[kQ], b[kQ], begin[kQ], end[kQ];
    // (the coordinates with the ids, not behind the add's round trip; kQ overlaps per thread, their loads together)
#pragma unroll
    for (uint32_t u = 0; u < kQ; ++u) {
        const uint64_t i = ((uint64_t)blockIdx.x * kQ + u) * 256 + threadIdx.x;
        a[u] = kInf; b[u] = kInf; begin[u] = 0; end[u] = 0;
        if (i < o.n) { a[u] = o.a_id[i]; b[u] = o.b_id[i]; begin[u] = __builtin_nontemporal_load(o.a_begin + i); end[u] = __builtin_nontemporal_load(o.a_end + i); }
    }
    uint32_t base[kQ], leader[kQ];
    bool ok[kQ];
#pragma unroll
    for (uint32_t u = 0; u < kQ; ++u) {
        ok[u] = a[u] < n_reads && b[u] < n_reads;
        const uint32_t seg = segment_of(a[u], ok[u], lane, leader[u]);
        base[u] = 0;
        if (seg) base[u] = ev_off[a[u]] + atomicAdd(&written[a[u]], (2u >> ev_shift) * seg);
    }
#pragma unroll
    for (uint32_t u = 0; u < kQ; ++u) {
        const uint32_t at = (uint32_t)__shfl((int)base[u], (int)leader[u], 64);
        if (ok[u]) *(uint2*)(ev + ((size_t)at << ev_shift) + 2u * (lane - leader[u])) = make_uint2((begin[u] + 15u) << 1, ((end[u] - 15u) << 1) | 1u);
    }
}


// ---- sharded runs: ONE scatter on the sender, by (owner rank, partition of the owner's reads) ------------------------
// Read r is local read r / P of rank r % P.  A rank used to group the bounds of its slice by owner, ship them, and the
// owner partitioned what it received from level 1 on (count, l1 scatter): the same records scattered twice, counted
// twice.  owner = read % P and partition = (read / P) >> 12 are both functions of the read, so the sender scatters once
// into P * n_part bins (as many as the level-1 bins of one GPU), both sides of every overlap as records
// {local read & 4095 : 12 | begin : 26 | end : 26}; an owner's block is [header: its groups' record counts][its records,
// partition by partition].  The owner adds up the headers (the groups' places at level 2, the table of level-2 tiles over
// the blocks as they lie in the receive buffer - no tile crosses a (sender, partition) segment) and starts at the level-2
// scatter; no pass over the received records counts anything.
__global__ __launch_bounds__(kBlockC) void shard_count_kernel(OvlSoA o, uint32_t n_reads, uint32_t world, uint32_t groups,
                                                              uint32_t n_bins, uint32_t* group_count) {
    extern __shared__ uint32_t s_hist[];
    for (uint32_t g = threadIdx.x; g < n_bins; g += kBlockC) s_hist[g] = 0;
    __syncthreads();
    constexpr uint32_t kC = kCountPer;
    const uint64_t last = o.n - 1;
    for (uint64_t i0 = (uint64_t)blockIdx.x * kBlockC * kC; i0 < o.n; i0 += (uint64_t)gridDim.x * kBlockC * kC) {
        uint32_t a[kC], b[kC];
#pragma unroll
        for (uint32_t u = 0; u < kC; ++u) {
            const uint64_t i = i0 + u * kBlockC + threadIdx.x;
            const uint64_t j = i < o.n ? i : last;
            a[u] = i < o.n ? __builtin_nontemporal_load(o.a_id + j) : kInf;
            b[u] = __builtin_nontemporal_load(o.b_id + j);
        }
#pragma unroll
        for (uint32_t u = 0; u < kC; ++u) {
            if (a[u] < n_reads && b[u] < n_reads) {
                atomicAdd(&s_hist[(a[u] % world) * groups + ((a[u] / world) >> kGroupShift)], 1u);
                atomicAdd(&s_hist[(b[u] % world) * groups + ((b[u] / world) >> kGroupShift)], 1u);
            }
        }
    }
    __syncthreads();
    for (uint32_t g = threadIdx.x; g < n_bins; g += kBlockC) {
        const uint32_t c = s_hist[g];
        if (c) atomicAdd(&group_count[g], c);
    }
}

// The blocks of the send buffer (8-byte words): block p = header (g.header words: the counts of owner p's groups, two per
// word) + owner p's records; part_cursor[p * n_part + q] = where partition q of owner p starts; send_words[p] = the
// block's length.  One workgroup.
__global__ __launch_bounds__(1024) void shard_send_layout_kernel(const uint32_t* __restrict__ group_count, ShardGeometry g,
                                                                 uint64_t* __restrict__ send, uint32_t* __restrict__ part_cursor,
                                                                 uint32_t* __restrict__ send_words) {
    __shared__ uint32_t tmp[1024 / 64 + 1];
    uint32_t block_start = 0;
    for (uint32_t p = 0; p < g.world; ++p) {
        uint32_t* header = (uint32_t*)(send + block_start);
        uint32_t carry = block_start + g.header;
        for (uint32_t g0 = 0; g0 < g.groups; g0 += 1024) {
            const uint32_t k = g0 + threadIdx.x;
            const uint32_t c = k < g.groups ? group_count[p * g.groups + k] : 0u;
            uint32_t tot;
            const uint32_t ex = block_scan_excl<1024>(c, OpAdd(), 0u, tmp, tot);
            if (k < g.groups) {
                header[k] = c;
                if (k % kGroupsPerPart == 0) part_cursor[p * g.n_part + k / kGroupsPerPart] = carry + ex;
            }
            carry += tot;
        }
        if (threadIdx.x == 0) send_words[p] = carry - block_start;
        block_start = carry;
    }
}

// both sides of kTile / 2 overlaps = kTile records, sorted by (owner, partition) in LDS and copied out bin by bin
__global__ __launch_bounds__(kBlockP) void shard_l1_scatter_kernel(OvlSoA o, uint32_t n_reads, uint32_t world, uint32_t n_part,
                                                                   uint32_t* part_cursor, uint64_t* __restrict__ send) {
    extern __shared__ __align__(16) unsigned char s_raw[];
    __shared__ uint32_t tmp[kBlockP / 64 + 1];
    const uint32_t n_bins = world * n_part;
    StageLds L(s_raw, n_bins);
    uint64_t rec[kPer];
    uint32_t bin[kPer];
    bool in[kPer];
    const uint64_t last = o.n - 1;
    static_assert(kPer % 2 == 0, "two records per overlap");
#pragma unroll
    for (uint32_t u = 0; u < kPer / 2; ++u) {
        const uint64_t i = (uint64_t)blockIdx.x * (kTile / 2) + u * kBlockP + threadIdx.x;
        const uint64_t j = i < o.n ? i : last;
        const uint32_t a = o.a_id[j], b = o.b_id[j];
        const bool ok = i < o.n && a < n_reads && b < n_reads;
        const uint32_t la = a / world, lb = b / world;
        in[2 * u] = in[2 * u + 1] = ok;
        rec[2 * u] = pack_record(la, __builtin_nontemporal_load(o.a_begin + j), __builtin_nontemporal_load(o.a_end + j));
        rec[2 * u + 1] = pack_record(lb, __builtin_nontemporal_load(o.b_begin + j), __builtin_nontemporal_load(o.b_end + j));
        bin[2 * u] = ok ? (a - la * world) * n_part + (la >> kL1Shift) : 0u;
        bin[2 * u + 1] = ok ? (b - lb * world) * n_part + (lb >> kL1Shift) : 0u;
    }
    stage_and_copy(L, n_bins, rec, bin, in, part_cursor, send, tmp);
}

// The owner's side.  blocks.off[p]: where rank p's block lies (8-byte words from `base`).  Out: group_base[0 .. groups]
// and group_cursor = exclusive prefix of the groups' record counts summed over the senders; the table of level-2 tiles -
// tile t covers words tile_lo[t] .. tile_hi[t] of `base`, all of partition tile_part[t] and of one sender; *n_tiles.
// pair_pref: world * n_part + 1 words of scratch.  One workgroup.
__global__ __launch_bounds__(1024) void shard_owner_layout_kernel(const uint64_t* __restrict__ base, const uint64_t* __restrict__ base_self,
                                                                  ShardBlocks blocks, ShardGeometry g,
                                                                  uint32_t* __restrict__ group_base, uint32_t* __restrict__ group_cursor,
                                                                  uint32_t* __restrict__ pair_pref, uint32_t* __restrict__ tile_part,
                                                                  uint32_t* __restrict__ tile_lo, uint32_t* __restrict__ tile_hi,
                                                                  uint32_t* n_tiles) {
    __shared__ uint32_t tmp[1024 / 64 + 1];
    auto header_of = [&](uint32_t p) { return (const uint32_t*)((p == blocks.self ? base_self : base) + blocks.off[p]); };
    uint32_t carry = 0;
    for (uint32_t g0 = 0; g0 < g.groups; g0 += 1024) {
        const uint32_t k = g0 + threadIdx.x;
        uint32_t c = 0;
        if (k < g.groups) {
            for (uint32_t p = 0; p < g.world; ++p) c += header_of(p)[k];
        }
        uint32_t tot;
        const uint32_t ex = block_scan_excl<1024>(c, OpAdd(), 0u, tmp, tot);
        if (k < g.groups) { group_base[k] = carry + ex; group_cursor[k] = carry + ex; }
        carry += tot;
    }
    if (threadIdx.x == 0) group_base[g.groups] = carry;
    // (sender, partition) segments: their lengths, an exclusive prefix over all of them
    const uint32_t n_pairs = g.world * g.n_part;
    uint32_t pcarry = 0;
    for (uint32_t j0 = 0; j0 < n_pairs; j0 += 1024) {
        const uint32_t j = j0 + threadIdx.x;
        uint32_t len = 0;
        if (j < n_pairs) {
            const uint32_t* header = header_of(j / g.n_part) + (j % g.n_part) * kGroupsPerPart;
            for (uint32_t k = 0; k < kGroupsPerPart; ++k) len += header[k];
        }
        uint32_t tot;
        const uint32_t ex = block_scan_excl<1024>(len, OpAdd(), 0u, tmp, tot);
        if (j < n_pairs) pair_pref[j] = pcarry + ex;
        pcarry += tot;
    }
    if (threadIdx.x == 0) pair_pref[n_pairs] = pcarry;
    __syncthreads();
    uint32_t tile_carry = 0;
    for (uint32_t j0 = 0; j0 < n_pairs; j0 += 1024) {
        const uint32_t j = j0 + threadIdx.x;
        uint32_t lo = 0, len = 0, tiles = 0;
        if (j < n_pairs) {
            const uint32_t p = j / g.n_part;
            len = pair_pref[j + 1] - pair_pref[j];
            lo = blocks.off[p] + g.header + (pair_pref[j] - pair_pref[p * g.n_part]);
            tiles = (len + kTile - 1) / kTile;
        }
        uint32_t ttot;
        const uint32_t tex = block_scan_excl<1024>(tiles, OpAdd(), 0u, tmp, ttot);
        for (uint32_t k = 0; k < tiles; ++k) {
            tile_part[tile_carry + tex + k] = (j % g.n_part) | (j / g.n_part == blocks.self ? kTileOtherBase : 0u);
            tile_lo[tile_carry + tex + k] = lo + k * kTile;
            tile_hi[tile_carry + tex + k] = umin(lo + len, lo + (k + 1u) * kTile);
        }
        tile_carry += ttot;
    }
    if (threadIdx.x == 0) *n_tiles = tile_carry;
}

}  // namespace

uint32_t partition_count(uint32_t n_reads) { return part_geom(n_reads).n_part; }
uint32_t partition_group_slots(uint32_t n_reads) { const PartGeom G = part_geom(n_reads); return G.n_part * G.gpp + 2; }
size_t partition_records_needed(uint32_t, uint64_t n_overlaps) { return (size_t)n_overlaps + 64; }
size_t partition_tile_slots(uint32_t n_reads, uint64_t n_overlaps) { return (size_t)(n_overlaps / kTile) + partition_count(n_reads) + 4; }
bool partition_path_fits(uint32_t n_reads, uint32_t max_read_len, uint64_t n_overlaps) {
    // coordinates in 25 bits; the histogram of all groups in the LDS of one workgroup; level-1 histograms next to
    // the staging area; the bound PAIRS' positions (two per overlap) in 32 bits; enough overlaps per partition for whole-line copies
    if (n_reads == 0 || n_overlaps == 0) return false;
    const PartGeom G = part_geom(n_reads);
    const uint64_t n_part = G.n_part;
    // (round 6: a histogram of all groups that outgrows the LDS is counted in windows - up to eight passes over the ids)
    return max_read_len < kCoordMax - 32u && n_part * G.gpp <= 8ull * kCountWindow &&
           stage_lds_bytes((uint32_t)n_part) <= 60u * 1024u && 2ull * n_overlaps < 0xFFFFFFF0ull && n_overlaps / n_part >= 512;
}

// Buffers (device): acount, written: n_reads + 2 words each; part_cursor: n_part + 2 words; group: 3 *
// partition_group_slots(n_reads) words (counts, bases, cursors); tiles: 3 * partition_tile_slots words + 1; rec1,
// rec2: partition_records_needed records; ev_off: n_reads + 2; ev: 4 * n_overlaps + 8.  fills: launched here, with
// what the caller has put in.  workgroups: compute units of the device (the counting kernel's persistent workgroups).
namespace {
struct PartitionBuffers {
    uint32_t n_part, n_groups, group_slots, shift, gpp;
    size_t tile_slots;
    uint32_t *group_count, *group_base, *group_cursor, *tile_part, *tile_lo, *tile_hi, *n_tiles;
    PartitionBuffers(uint32_t n_reads, uint64_t n_records, uint32_t* group, uint32_t* tiles) {
        const PartGeom G = part_geom(n_reads);
        n_part = G.n_part; shift = G.shift; gpp = G.gpp;
        n_groups = (n_reads + kGroupReads - 1) / kGroupReads;
        group_slots = partition_group_slots(n_reads);
        tile_slots = partition_tile_slots(n_reads, n_records);
        group_count = group; group_base = group + group_slots; group_cursor = group + 2 * (size_t)group_slots;
        tile_part = tiles; tile_lo = tiles + tile_slots; tile_hi = tiles + 2 * tile_slots; n_tiles = tiles + 3 * tile_slots;
    }
};

hipError_t count_attribute(size_t lds_count) {
    if (lds_count + 8192 > 64 * 1024) { // (per launch: the attribute belongs to the function on the current device; 8 KB: the kernels' static LDS)
        hipError_t e = hipFuncSetAttribute((const void*)group_count_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_count);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)group_count_dedupe_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_count);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)group_count_records_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_count);
        return e;
    }
    return hipSuccess;
}

// level 2 and the rows from the level-1 records
void launch_partition_rest(const PartitionBuffers& B, uint32_t n_reads, const uint32_t* acount, const uint64_t* rec1, uint64_t* rec2,
                           uint32_t* ev_off, uint32_t* ev, hipStream_t s, uint32_t ev_shift, uint32_t shrink = 15u) {
    const uint32_t tiles2 = (uint32_t)(B.tile_slots - 4);               // at least as many as the table can hold
    hipLaunchKernelGGL(l2_scatter_kernel, dim3(tiles2), dim3(kBlockP), stage_lds_bytes(B.gpp), s, rec1, (const uint64_t*)nullptr,
                       (const uint32_t*)B.tile_part, (const uint32_t*)B.tile_lo, (const uint32_t*)B.tile_hi, (const uint32_t*)B.n_tiles,
                       B.group_cursor, rec2, B.gpp);
    // (the groups' counts and cursors have served: their places take the query-side sums and the groups' first pairs)
    uint32_t *qsum = B.group_count, *pair_base = B.group_cursor;
    hipLaunchKernelGGL(group_query_sum_kernel, dim3(B.n_groups), dim3(kGroupReads), 0, s, acount, n_reads, qsum);
    hipLaunchKernelGGL(group_event_base_kernel, dim3(1), dim3(1024), 0, s, (const uint32_t*)qsum, (const uint32_t*)B.group_base, B.n_groups,
                       pair_base);
    hipLaunchKernelGGL(final_kernel, dim3(B.n_groups), dim3(kBlockP), stage_lds_bytes(kGroupReads), s, (const uint64_t*)rec2,
                       (const uint32_t*)B.group_base, (const uint32_t*)pair_base, n_reads, acount, ev_off, ev, shrink, ev_shift);
}
}  // namespace

bool bucket_count_can_dedupe(const OvlSoA& o, const uint8_t* valid) {
    return (((uintptr_t)o.a_id | (uintptr_t)o.b_id) & 15u) == 0 && ((uintptr_t)valid & 3u) == 0 && getenv("RALA_DEDUPE_APART") == nullptr;
}

// dedupe (may be null): the counting pass does duplicate removal's first pass on the way (group_count_dedupe_kernel) -
// suspect: n_reads bytes, cleared here; valid: the validity bytes; list_*: room for list_cap marks and a zeroed counter; the
// caller runs launch_dedupe_fix behind this call's counting pass (the event `counted`, recorded here when given)
hipError_t launch_bucket_partitioned(const OvlSoA& o, uint32_t n_reads, uint32_t* acount, uint32_t* written,
                                     uint32_t* part_cursor, uint32_t* group, uint32_t* tiles, uint64_t* rec1, uint64_t* rec2,
                                     uint32_t* ev_off, uint32_t* ev, uint32_t workgroups, FillList& fills, hipStream_t s,
                                     const BucketDedupe* dedupe, uint32_t ev_shift) {
    const PartitionBuffers B(n_reads, o.n, group, tiles);
    // (with whatever the caller wants cleared at this point)
    fills.add(acount, 0, (size_t)n_reads * 4);
    fills.add(written, 0, (size_t)n_reads * 4);
    fills.add(B.group_count, 0, (size_t)B.group_slots * 4);
    if (dedupe) fills.add(dedupe->suspect, 0, n_reads);
    hipError_t e = fills.launch(s);
    if (e != hipSuccess) return e;
    const uint32_t all_groups = B.n_part * B.gpp, window = std::min(all_groups, count_window());
    const size_t lds_count = (size_t)window * 4;
    e = count_attribute(lds_count);
    if (e != hipSuccess) return e;
    const uint32_t chunks = (uint32_t)((o.n + kBlockC * kCountPer - 1) / (kBlockC * kCountPer));
    auto wait_for = [&](hipEvent_t ev) { return dedupe && ev ? hipStreamWaitEvent(s, ev, 0) : hipSuccess; };
    e = wait_for(dedupe ? dedupe->ids : nullptr);
    if (e != hipSuccess) return e;
    // (two workgroups per compute unit where their histograms fit side by side)
    const uint32_t count_groups = (workgroups ? workgroups : 256u) * (2 * lds_count <= 150u * 1024u ? 2u : 1u);
    if (dedupe) {
        hipLaunchKernelGGL(group_count_dedupe_kernel<false>, dim3(std::min<uint32_t>(count_groups, chunks)), dim3(kBlockC), lds_count, s, o,
                           n_reads, window, B.group_count, dedupe->suspect, dedupe->valid, dedupe->list_pos,
                           dedupe->list_query, dedupe->list_cap, dedupe->list_count, 1u, 0u);
        if (dedupe->counted) {
            e = hipEventRecord(dedupe->counted, s);
            if (e != hipSuccess) return e;
        }
    } else {
        hipLaunchKernelGGL(group_count_kernel, dim3(std::min<uint32_t>(count_groups, chunks)), dim3(kBlockC), lds_count, s, o,
                           n_reads, 0u, window, B.group_count);
    }
    for (uint32_t g_lo = window; g_lo < all_groups; g_lo += window) {       // (more than 4.9 M reads: the other windows of groups)
        hipLaunchKernelGGL(group_count_kernel, dim3(std::min<uint32_t>(count_groups, chunks)), dim3(kBlockC), lds_count, s, o,
                           n_reads, g_lo, std::min(window, all_groups - g_lo), B.group_count);
    }
    hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(1024), 0, s, (const uint32_t*)B.group_count, B.n_part, B.gpp, B.group_base, B.group_cursor,
                       part_cursor, B.tile_part, B.tile_lo, B.tile_hi, B.n_tiles);
    const uint32_t tiles1 = (uint32_t)((o.n + kTile - 1) / kTile);
    e = wait_for(dedupe ? dedupe->b_coords : nullptr);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(l1_scatter_kernel, dim3(tiles1), dim3(kBlockP), stage_lds_bytes(B.n_part), s, o, n_reads, B.n_part, B.shift, part_cursor, rec1,
                       acount);
    launch_partition_rest(B, n_reads, acount, rec1, rec2, ev_off, ev, s, ev_shift);
    // (overlaps per thread: two - 1.27 against 1.31 ms for the stage at C3 in two of three alternations, four: the same as one;
    // RALA_QUERY_PER for the measurement)
    e = wait_for(dedupe ? dedupe->a_coords : nullptr);
    if (e != hipSuccess) return e;
    static const int q_per = getenv("RALA_QUERY_PER") ? atoi(getenv("RALA_QUERY_PER")) : 2;
    if (q_per == 4) hipLaunchKernelGGL(query_side_kernel<4>, dim3((uint32_t)((o.n + 1023) / 1024)), dim3(256), 0, s, o, n_reads, (const uint32_t*)ev_off, written, ev, ev_shift);
    else if (q_per == 2) hipLaunchKernelGGL(query_side_kernel<2>, dim3((uint32_t)((o.n + 511) / 512)), dim3(256), 0, s, o, n_reads, (const uint32_t*)ev_off, written, ev, ev_shift);
    else hipLaunchKernelGGL(query_side_kernel<1>, dim3((uint32_t)((o.n + 255) / 256)), dim3(256), 0, s, o, n_reads, (const uint32_t*)ev_off, written, ev, ev_shift);
    return hipGetLastError();
}

// An owner rank's bound records (both sides of every overlap that touches one of its reads) into the same CSR.  zero_counts:
// n_reads + 2 zeroed words (cleared here, through `fills`): no query side apart from the records.
hipError_t launch_bucket_partitioned_records(const uint64_t* records, uint64_t n, uint32_t n_reads, uint32_t* zero_counts,
                                             uint32_t* part_cursor, uint32_t* group, uint32_t* tiles, uint64_t* rec1, uint64_t* rec2,
                                             uint32_t* ev_off, uint32_t* ev, uint32_t workgroups, FillList& fills, hipStream_t s,
                                             uint32_t shrink, uint32_t ev_shift) {
    const PartitionBuffers B(n_reads, n, group, tiles);
    fills.add(zero_counts, 0, (size_t)n_reads * 4);
    fills.add(B.group_count, 0, (size_t)B.group_slots * 4);
    hipError_t e = fills.launch(s);
    if (e != hipSuccess) return e;
    const uint32_t all_groups = B.n_part * B.gpp, window = std::min(all_groups, count_window());
    const size_t lds_count = (size_t)window * 4;
    e = count_attribute(lds_count);
    if (e != hipSuccess) return e;
    const uint32_t chunks = (uint32_t)((n + kBlockC * kCountPer - 1) / (kBlockC * kCountPer));
    const uint32_t count_groups = (workgroups ? workgroups : 256u) * (2 * lds_count <= 150u * 1024u ? 2u : 1u);
    for (uint32_t g_lo = 0; g_lo < all_groups; g_lo += window) {
        hipLaunchKernelGGL(group_count_records_kernel, dim3(std::min<uint32_t>(count_groups, std::max<uint32_t>(chunks, 1u))), dim3(kBlockC),
                           lds_count, s, records, n, n_reads, g_lo, std::min(window, all_groups - g_lo), B.group_count);
    }
    hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(1024), 0, s, (const uint32_t*)B.group_count, B.n_part, B.gpp, B.group_base, B.group_cursor,
                       part_cursor, B.tile_part, B.tile_lo, B.tile_hi, B.n_tiles);
    const uint32_t tiles1 = (uint32_t)((n + kTile - 1) / kTile);
    if (tiles1) {
        hipLaunchKernelGGL(l1_scatter_records_kernel, dim3(tiles1), dim3(kBlockP), stage_lds_bytes(B.n_part), s, records, n, n_reads,
                           B.n_part, B.shift, part_cursor, rec1);
    }
    launch_partition_rest(B, n_reads, zero_counts, rec1, rec2, ev_off, ev, s, ev_shift, shrink);
    return hipGetLastError();
}

// (n_records: an owner's records - two per overlap that stays on the rank)
bool partition_path_fits_records(uint32_t n_reads, uint32_t max_read_len, uint64_t n_records) {
    return max_read_len < (1u << kBoundRecordCoordBits) - 32u && partition_path_fits(n_reads, max_read_len, (n_records + 1) / 2);
}

// ---- sharded runs (kernels above) --------------------------------------------------------------------------------------
ShardGeometry shard_geometry(uint64_t n_reads, uint32_t world) {
    ShardGeometry g;
    g.world = world;
    const uint64_t most = (n_reads + world - 1) / world;                 // rank 0 owns the most reads
    g.n_part = (uint32_t)std::max<uint64_t>(1, (most + kL1Reads - 1) / kL1Reads);
    g.groups = g.n_part * kGroupsPerPart;
    g.header = g.groups / 2;
    return g;
}
bool shard_path_fits(uint64_t n_reads, uint32_t max_read_len, uint32_t world) {
    // the same answer on every rank: nothing here depends on a rank's slice
    if (n_reads == 0 || world == 0 || world > 64 || n_reads >= 0x7FFFFFFFull) return false;
    const ShardGeometry g = shard_geometry(n_reads, world);
    return max_read_len < kCoordMax - 32u && (uint64_t)g.world * g.groups * 4u <= 150u * 1024u &&
           stage_lds_bytes(g.world * g.n_part) <= 60u * 1024u;
}
size_t shard_send_words(const ShardGeometry& g, uint64_t n_overlaps) { return 2 * (size_t)n_overlaps + (size_t)g.world * g.header + 64; }
size_t shard_tile_slots(const ShardGeometry& g, uint64_t n_records) { return (size_t)(n_records / kTile) + (size_t)g.world * g.n_part + 4; }

// group_count: world * groups words (cleared here, through `fills`); part_cursor: world * n_part words; send:
// shard_send_words(); send_words: world words (the blocks' lengths, for the host)
hipError_t launch_shard_emit(const OvlSoA& o, uint32_t n_reads, const ShardGeometry& g, uint32_t* group_count, uint32_t* part_cursor,
                             uint64_t* send, uint32_t* send_words, uint32_t workgroups, FillList& fills, hipStream_t s,
                             const BucketDedupe* dedupe) {
    const uint32_t n_bins = g.world * g.groups;
    fills.add(group_count, 0, (size_t)n_bins * 4);
    if (dedupe) fills.add(dedupe->suspect, 0, n_reads);
    hipError_t e = fills.launch(s);
    if (e != hipSuccess) return e;
    const size_t lds_count = (size_t)n_bins * 4;
    if (lds_count + 8192 > 64 * 1024) {
        e = hipFuncSetAttribute((const void*)shard_count_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_count);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)group_count_dedupe_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_count);
        if (e != hipSuccess) return e;
    }
    if (o.n) {
        const uint32_t chunks = (uint32_t)((o.n + kBlockC * kCountPer - 1) / (kBlockC * kCountPer));
        const uint32_t count_groups = (workgroups ? workgroups : 256u) * (2 * lds_count <= 150u * 1024u ? 2u : 1u);
        if (dedupe) {
            // (the sender's count with duplicate removal's first pass on the way, as the single GPU's counting pass has it)
            hipLaunchKernelGGL(group_count_dedupe_kernel<true>, dim3(std::min<uint32_t>(count_groups, chunks)), dim3(kBlockC), lds_count, s, o,
                               n_reads, n_bins, group_count, dedupe->suspect, dedupe->valid, dedupe->list_pos, dedupe->list_query,
                               dedupe->list_cap, dedupe->list_count, g.world, g.groups);
            if (dedupe->counted) {
                e = hipEventRecord(dedupe->counted, s);
                if (e != hipSuccess) return e;
            }
        } else {
            hipLaunchKernelGGL(shard_count_kernel, dim3(std::min<uint32_t>(count_groups, chunks)), dim3(kBlockC), lds_count, s, o, n_reads,
                               g.world, g.groups, n_bins, group_count);
        }
    }
    hipLaunchKernelGGL(shard_send_layout_kernel, dim3(1), dim3(1024), 0, s, (const uint32_t*)group_count, g, send, part_cursor, send_words);
    if (o.n) {
        const uint32_t tiles1 = (uint32_t)((o.n + kTile / 2 - 1) / (kTile / 2));
        hipLaunchKernelGGL(shard_l1_scatter_kernel, dim3(tiles1), dim3(kBlockP), stage_lds_bytes(g.world * g.n_part), s, o, n_reads, g.world,
                           g.n_part, part_cursor, send);
    }
    return hipGetLastError();
}

// The owner: the blocks as they lie behind `base` -> the exact CSR of its n_reads_local reads.  group: 3 *
// (g.groups + 2) words + g.world * g.n_part + 2 (bases, cursors, the segments' prefix); tiles: 3 * shard_tile_slots() + 2 words;
// rec2: n_records + 64 records; ev_off: n_reads_local + 2; ev: 2 * n_records + 8.
hipError_t launch_bucket_from_blocks(const uint64_t* base, const uint64_t* base_self, const ShardBlocks& blocks, const ShardGeometry& g,
                                     uint32_t n_reads_local,
                                     uint64_t n_records, uint32_t* group, uint32_t* tiles, uint64_t* rec2, uint32_t* ev_off, uint32_t* ev,
                                     FillList& fills, hipStream_t s, uint32_t ev_shift) {
    hipError_t e = fills.launch(s);
    if (e != hipSuccess) return e;
    const uint32_t slots = g.groups + 2;
    uint32_t *group_base = group, *group_cursor = group + slots, *pair_pref = group + 2 * (size_t)slots;
    const size_t tile_slots = shard_tile_slots(g, n_records);
    uint32_t *tile_part = tiles, *tile_lo = tiles + tile_slots, *tile_hi = tiles + 2 * tile_slots, *n_tiles = tiles + 3 * tile_slots;
    hipLaunchKernelGGL(shard_owner_layout_kernel, dim3(1), dim3(1024), 0, s, base, base_self, blocks, g, group_base, group_cursor, pair_pref, tile_part,
                       tile_lo, tile_hi, n_tiles);
    hipLaunchKernelGGL(l2_scatter_kernel, dim3((uint32_t)(tile_slots - 4)), dim3(kBlockP), stage_lds_bytes(kGroupsPerPart), s, base, base_self,
                       (const uint32_t*)tile_part, (const uint32_t*)tile_lo, (const uint32_t*)tile_hi, (const uint32_t*)n_tiles, group_cursor,
                       rec2, kGroupsPerPart);
    // (every event is a record's: the groups' first pairs are the groups' first records)
    const uint32_t n_groups = (n_reads_local + kGroupReads - 1) / kGroupReads;
    if (n_groups) {
        hipLaunchKernelGGL(final_kernel, dim3(n_groups), dim3(kBlockP), stage_lds_bytes(kGroupReads), s, (const uint64_t*)rec2,
                           (const uint32_t*)group_base, (const uint32_t*)group_base, n_reads_local, (const uint32_t*)nullptr, ev_off, ev, 15u, ev_shift);
    }
    return hipGetLastError();
}
size_t shard_group_words(const ShardGeometry& g) { return 3 * ((size_t)g.groups + 2) + (size_t)g.world * g.n_part + 2; }

}  // namespace rala_hip
