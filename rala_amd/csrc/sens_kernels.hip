// Per-overlap work of Graph::preprocess(overlaps, sensitive overlaps) (rvaser/rala src/graph.cpp:882-1054):
// Overlap::transmute_ + bound emission for the second add_layers (:919-934), the first trim
// (:935-939), and the pass in which sensitive dovetails mark the repeat hills they bridge
// (:1028-1043, Pile::check_repetitive_hills pile.cpp:568-592).
#include <hip/hip_runtime.h>

#include "device_utils.h"
#include "geom.h"
#include "kernels.h"

namespace rala_hip {

namespace {

constexpr int kBlock = 256;
constexpr uint32_t kInf = 0xFFFFFFFFu;

// Overlap::transmute_ (overlap.cpp:84-114): the target side is given in the coordinates of the
// trimmed read, so the pile's begin is added; the two bounds of the target side go out as (read,
// bound) tuples 2i, 2i + 1 - no +-15 here (graph.cpp:929-933).  Records whose names do not
// resolve or whose target did not survive are an error of the call (bit 0 / bit 1 of *error).
__global__ __launch_bounds__(kBlock) void sens_tuples_kernel(OvlSoA o, uint32_t n_reads, const uint32_t* __restrict__ begin,
                                                             const uint8_t* __restrict__ alive,
                                                             uint32_t* __restrict__ tb_begin, uint32_t* __restrict__ tb_end,
                                                             uint2* __restrict__ tuples, uint32_t* error) {
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= o.n) return;
    const uint32_t a = o.a_id[i], b = o.b_id[i];
    uint32_t r = kInf, x = 0, y = 0;
    if (a >= n_reads || b >= n_reads) {
        atomicOr(error, 1u);
    } else if (!alive[b]) {
        atomicOr(error, 2u);
    } else {
        const uint32_t B = begin[b];
        x = o.b_begin[i] + B;
        y = o.b_end[i] + B;
        r = b;
    }
    tb_begin[i] = x; tb_end[i] = y;
    *(uint4*)(tuples + 2 * i) = make_uint4(r, x << 1, r, (y << 1) | 1u);
}

// the same with one bound record {read : 22, begin : 21, end : 21} per overlap for the partitioned bucketing
__global__ __launch_bounds__(kBlock) void sens_records_kernel(OvlSoA o, uint32_t n_reads, const uint32_t* __restrict__ begin,
                                                              const uint8_t* __restrict__ alive,
                                                              uint32_t* __restrict__ tb_begin, uint32_t* __restrict__ tb_end,
                                                              uint64_t* __restrict__ records, uint32_t* error) {
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= o.n) return;
    const uint32_t a = o.a_id[i], b = o.b_id[i];
    uint32_t x = 0, y = 0;
    uint64_t rec = ~0ull;                          // (names no read)
    if (a >= n_reads || b >= n_reads) {
        atomicOr(error, 1u);
    } else if (!alive[b]) {
        atomicOr(error, 2u);
    } else {
        const uint32_t B = begin[b];
        x = o.b_begin[i] + B;
        y = o.b_end[i] + B;
        constexpr uint32_t kMask = (1u << kBoundRecordCoordBits) - 1u;
        rec = (uint64_t)b << (2 * kBoundRecordCoordBits) | (uint64_t)(x < kMask ? x : kMask) << kBoundRecordCoordBits | (uint64_t)(y < kMask ? y : kMask);
    }
    tb_begin[i] = x; tb_end[i] = y;
    records[i] = rec;
}

// first Overlap::trim of every sensitive overlap (graph.cpp:935-939); state 1 = kept
__global__ __launch_bounds__(kBlock) void sens_trim_kernel(OvlSoA o, const uint32_t* __restrict__ tb_begin,
                                                           const uint32_t* __restrict__ tb_end,
                                                           const uint32_t* __restrict__ begin, const uint32_t* __restrict__ end,
                                                           const uint8_t* __restrict__ alive, SensCoords out) {
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= o.n) return;
    const uint32_t a = o.a_id[i], b = o.b_id[i];
    uint8_t st = 0;
    if (alive[a] && alive[b]) {
        Coords c;
        c.a_begin = o.a_begin[i]; c.a_end = o.a_end[i]; c.b_begin = tb_begin[i]; c.b_end = tb_end[i];
        c.length = o.length[i];
        if (ovl_trim(c, o.strand[i], begin[a], end[a], begin[b], end[b])) {
            st = 1;
            out.a_begin[i] = c.a_begin; out.a_end[i] = c.a_end; out.b_begin[i] = c.b_begin; out.b_end[i] = c.b_end;
            out.length[i] = c.length;
        }
    }
    out.state[i] = st;
}

// graph.cpp:1028-1043: trim once more, and a dovetail marks the repeat hills of its target that
// it reaches across (Pile::check_repetitive_hills; every writer stores the same 1)
__global__ __launch_bounds__(kBlock) void sens_bridge_kernel(OvlSoA o, SensCoords sc, const uint32_t* __restrict__ begin,
                                                             const uint32_t* __restrict__ end,
                                                             const uint8_t* __restrict__ alive,
                                                             const uint32_t* __restrict__ n_rep,
                                                             const uint32_t* __restrict__ rep_slot, Interval* rep_pool) {
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= o.n || !sc.state[i]) return;
    const uint32_t a = o.a_id[i], b = o.b_id[i];
    if (!alive[a] || !alive[b]) return;
    const uint32_t nr = n_rep[b];
    if (nr == 0) return;                        // nothing to mark; the overlap itself is not kept
    Coords c;
    c.a_begin = sc.a_begin[i]; c.a_end = sc.a_end[i]; c.b_begin = sc.b_begin[i]; c.b_end = sc.b_end[i];
    c.length = sc.length[i];
    const uint32_t st = o.strand[i];
    const uint32_t Ba = begin[a], Ea = end[a], B = begin[b], E = end[b];
    if (!ovl_trim(c, st, Ba, Ea, B, E)) return;
    const uint32_t t = ovl_type(c, st, Ba, Ea, B, E);
    if (t != kTypeAB && t != kTypeBA) return;
    const uint32_t x = c.b_begin, y = c.b_end;
    Interval* h = rep_pool + rep_slot[b];
    for (uint32_t k = 0; k < nr; ++k) {
        const uint32_t hf = h[k].first, hs = h[k].second;
        if (!(x < hs && hf < y)) continue;
        if ((double)hf < 0.1 * (double)(E - B) + (double)B && (uint32_t)(x - B) < (uint32_t)(E - y)) {
            if (y >= hs + kHillFuzz) h[k].aux = 1;
        } else if ((double)hs > 0.9 * (double)(E - B) + (double)B && (uint32_t)(x - B) > (uint32_t)(E - y)) {
            if (x + kHillFuzz <= hf) h[k].aux = 1;
        }
    }
}

// Pile::is_valid_overlap (pile.cpp:605-630): an overlap that ends inside a bridged repeat hill at
// the edge of the read is not trusted
__device__ __forceinline__ bool pile_valid_overlap(uint32_t x, uint32_t y, uint32_t B, uint32_t E, const Interval* h, uint32_t nr) {
    for (uint32_t k = 0; k < nr; ++k) {
        const uint32_t hf = h[k].first, hs = h[k].second;
        if (!(x < hs && hf < y)) continue;
        if ((double)hf < 0.1 * (double)(E - B) + (double)B) {
            if (y < hs + kHillFuzz && h[k].aux) return false;
        } else if ((double)hs > 0.9 * (double)(E - B) + (double)B) {
            if (x + kHillFuzz > hf && h[k].aux) return false;
        }
    }
    return true;
}

// graph.cpp:1045-1051 on the device list: overlaps (not internals) that fail on either side leave
__global__ __launch_bounds__(kBlock) void sens_filter_kernel(TailList L, const uint32_t* __restrict__ begin,
                                                             const uint32_t* __restrict__ end,
                                                             const uint32_t* __restrict__ n_rep,
                                                             const uint32_t* __restrict__ rep_slot,
                                                             const Interval* __restrict__ rep_pool) {
    const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= L.n) return;
    const uint8_t st = L.state[k];
    if (st != 1 && st != 3) return;
    const uint32_t a = L.a[k], b = L.b[k];
    const uint32_t na = n_rep[a], nb = n_rep[b];
    if ((na | nb) == 0) return;
    bool ok = true;
    if (na) ok = pile_valid_overlap(L.a_begin[k], L.a_end[k], begin[a], end[a], rep_pool + rep_slot[a], na);
    if (ok && nb) ok = pile_valid_overlap(L.b_begin[k], L.b_end[k], begin[b], end[b], rep_pool + rep_slot[b], nb);
    if (!ok) L.state[k] = 0;
}

// the chimera stage is over: overlaps that lost a read are gone (graph.cpp:869-877); internals stay
__global__ __launch_bounds__(kBlock) void finalize_states_kernel(TailList L, const uint8_t* __restrict__ alive) {
    const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= L.n) return;
    const uint8_t st = L.state[k];
    if ((st == 1 || st == 3) && !(alive[L.a[k]] && alive[L.b[k]])) L.state[k] = 0;
}

// component median of every read that has an overlap, by read number (0 elsewhere)
__global__ __launch_bounds__(kBlock) void scatter_component_medians_kernel(const uint32_t* __restrict__ alive_reads,
                                                                          const uint8_t* __restrict__ touched,
                                                                          const uint16_t* __restrict__ cmed, uint32_t n_alive,
                                                                          uint16_t* __restrict__ out) {
    const uint32_t q = blockIdx.x * kBlock + threadIdx.x;
    if (q < n_alive && touched[q]) out[alive_reads[q]] = cmed[q];
}

dim3 grid_for(uint64_t n) { return dim3((uint32_t)((n + kBlock - 1) / kBlock)); }

}  // namespace

void launch_sens_tuples(const OvlSoA& o, uint32_t n_reads, const uint32_t* begin, const uint8_t* alive, uint32_t* tb_begin,
                        uint32_t* tb_end, uint2* tuples, uint32_t* error, hipStream_t s) {
    if (o.n) {
        hipLaunchKernelGGL(sens_tuples_kernel, grid_for(o.n), dim3(kBlock), 0, s, o, n_reads, begin, alive, tb_begin, tb_end,
                           tuples, error);
    }
}
void launch_sens_records(const OvlSoA& o, uint32_t n_reads, const uint32_t* begin, const uint8_t* alive, uint32_t* tb_begin,
                         uint32_t* tb_end, uint64_t* records, uint32_t* error, hipStream_t s) {
    if (o.n) hipLaunchKernelGGL(sens_records_kernel, grid_for(o.n), dim3(kBlock), 0, s, o, n_reads, begin, alive, tb_begin, tb_end, records, error);
}
void launch_sens_trim(const OvlSoA& o, const uint32_t* tb_begin, const uint32_t* tb_end, const uint32_t* begin,
                      const uint32_t* end, const uint8_t* alive, const SensCoords& out, hipStream_t s) {
    if (o.n) hipLaunchKernelGGL(sens_trim_kernel, grid_for(o.n), dim3(kBlock), 0, s, o, tb_begin, tb_end, begin, end, alive, out);
}
void launch_sens_bridge(const OvlSoA& o, const SensCoords& sc, const uint32_t* begin, const uint32_t* end,
                        const uint8_t* alive, const uint32_t* n_rep, const uint32_t* rep_slot, Interval* rep_pool,
                        hipStream_t s) {
    if (o.n) {
        hipLaunchKernelGGL(sens_bridge_kernel, grid_for(o.n), dim3(kBlock), 0, s, o, sc, begin, end, alive, n_rep, rep_slot,
                           rep_pool);
    }
}

void launch_sens_filter(const TailList& L, const uint32_t* begin, const uint32_t* end, const uint32_t* n_rep,
                        const uint32_t* rep_slot, const Interval* rep_pool, hipStream_t s) {
    if (L.n) hipLaunchKernelGGL(sens_filter_kernel, grid_for(L.n), dim3(kBlock), 0, s, L, begin, end, n_rep, rep_slot, rep_pool);
}
namespace {
// what a wavefront's flagged lanes append, in lane order, behind one add to the list's counter
__device__ __forceinline__ void append_flagged(bool flag, uint32_t value, uint32_t* list, uint32_t* count) {
    const uint64_t m = __ballot(flag);
    if (m == 0) return;
    const uint32_t lane = threadIdx.x & 63;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(count, (uint32_t)__popcll(m));
    base = (uint32_t)__shfl((int)base, 0, 64);
    if (flag) list[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = value;
}
// the same for kListPer items per thread behind ONE add per workgroup (adds to one word cost about 10 ns apiece: a wavefront's
// add each, list_targets_kernel took 0.70 ms for 4 M reads at C5, list 0.28 M long)
constexpr uint32_t kListPer = 8;
__device__ __forceinline__ void append_flagged_block(const bool (&flag)[kListPer], const uint32_t (&value)[kListPer], uint32_t* list,
                                                     uint32_t* count) {
    __shared__ uint32_t s_cnt, s_base;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    uint32_t slot[kListPer];
#pragma unroll
    for (uint32_t u = 0; u < kListPer; ++u) {
        const uint64_t m = __ballot(flag[u]);
        uint32_t base = 0;
        if (m) {
            if (lane == 0) base = atomicAdd(&s_cnt, (uint32_t)__popcll(m));
            base = (uint32_t)__shfl((int)base, 0, 64);
        }
        slot[u] = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    }
    __syncthreads();
    if (threadIdx.x == 0) s_base = s_cnt ? atomicAdd(count, s_cnt) : 0u;
    __syncthreads();
#pragma unroll
    for (uint32_t u = 0; u < kListPer; ++u) {
        if (flag[u]) list[s_base + slot[u]] = value[u];
    }
}
// the targets of the sensitive overlaps: the reads that received bounds (graph.cpp:941-953 walks the piles with bounds)
__global__ __launch_bounds__(kBlock) void list_targets_kernel(const uint32_t* __restrict__ off, uint32_t n, uint32_t* list, uint32_t* count) {
    bool flag[kListPer];
    uint32_t value[kListPer];
#pragma unroll
    for (uint32_t u = 0; u < kListPer; ++u) {
        const uint32_t r = (blockIdx.x * kListPer + u) * kBlock + threadIdx.x;
        value[u] = r;
        flag[u] = r < n && off[r + 1] != off[r];
    }
    append_flagged_block(flag, value, list, count);
}
// the members of the components: alive reads with an overlap (graph.cpp:1006-1026); of a sharded run, this rank's (read r
// lives on rank r % world as its read r / world)
__global__ __launch_bounds__(kBlock) void list_members_kernel(const uint32_t* __restrict__ alive_reads, const uint8_t* __restrict__ touched,
                                                              uint32_t n_alive, uint32_t world, uint32_t rank, uint32_t* list,
                                                              uint32_t* count) {
    bool flag[kListPer];
    uint32_t value[kListPer];
#pragma unroll
    for (uint32_t u = 0; u < kListPer; ++u) {
        const uint32_t q = (blockIdx.x * kListPer + u) * kBlock + threadIdx.x;
        const uint32_t r = q < n_alive ? alive_reads[q] : 0u;
        value[u] = r / world;
        flag[u] = q < n_alive && touched[q] && r % world == rank;
    }
    append_flagged_block(flag, value, list, count);
}
// (n_dev: the list's length, on the device; the grid covers the host's bound of it)
__global__ __launch_bounds__(kBlock) void sens_split_kernel(const uint32_t* __restrict__ list, const uint32_t* __restrict__ n_dev, uint32_t bound,
                                                            SensSplitArgs A) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t n = umin(*n_dev, bound);
    uint32_t cls = 5, r = 0;
    if (i < n) {
        r = list[i];
        const uint32_t len = A.read_len[r];
        const uint32_t ev = (A.ev_cnt ? umin(A.ev_cnt[r], A.ev_stride) : (A.ev_off[r + 1] - A.ev_off[r]) << A.ev_shift) + A.sens_off[r + 1] - A.sens_off[r];
        const bool region = A.end[r] > A.begin[r];
        cls = !region ? 4u
            : len <= 16384u && ev <= kRunEventCap - 2u ? 0u
            : len <= 32768u && ev <= kRunEventCap - 2u ? 1u
            : len <= 16384u && ev <= kRunEventCapMid - 2u ? 2u
            : len <= 16384u && ev <= kRunEventCapBig - 2u ? 3u : 4u;
    }
#pragma unroll
    for (uint32_t c = 0; c < 5; ++c) append_flagged(cls == c, r, A.out[c], A.counts + c);
}
}  // namespace

void launch_sens_split(const uint32_t* list, uint32_t bound, const uint32_t* n_dev, const SensSplitArgs& args, hipStream_t s) {
    if (bound) hipLaunchKernelGGL(sens_split_kernel, grid_for(bound), dim3(kBlock), 0, s, list, n_dev, bound, args);
}
namespace {
__global__ void status_clear_kernel(uint32_t* status, uint32_t bits, uint32_t* zero) {
    atomicAnd(status, ~bits);
    if (zero) *zero = 0;
}
}  // namespace
void launch_status_clear(uint32_t* status, uint32_t bits, uint32_t* zero, hipStream_t s) {
    hipLaunchKernelGGL(status_clear_kernel, dim3(1), dim3(1), 0, s, status, bits, zero);
}

void launch_list_targets(const uint32_t* off, uint32_t n, uint32_t* list, uint32_t* count, hipStream_t s) {
    if (n) hipLaunchKernelGGL(list_targets_kernel, dim3((n + kListPer * kBlock - 1) / (kListPer * kBlock)), dim3(kBlock), 0, s, off, n, list, count);
}
void launch_list_members(const uint32_t* alive_reads, const uint8_t* touched, uint32_t n_alive, uint32_t world, uint32_t rank,
                         uint32_t* list, uint32_t* count, hipStream_t s) {
    if (n_alive) hipLaunchKernelGGL(list_members_kernel, dim3((n_alive + kListPer * kBlock - 1) / (kListPer * kBlock)), dim3(kBlock), 0, s, alive_reads, touched, n_alive, world, rank, list, count);
}

void launch_finalize_states(const TailList& L, const uint8_t* alive, hipStream_t s) {
    if (L.n) hipLaunchKernelGGL(finalize_states_kernel, grid_for(L.n), dim3(kBlock), 0, s, L, alive);
}
void launch_scatter_component_medians(const uint32_t* alive_reads, const uint8_t* touched, const uint16_t* cmed, uint32_t n_alive,
                                      uint16_t* out, hipStream_t s) {
    if (n_alive) {
        hipLaunchKernelGGL(scatter_component_medians_kernel, grid_for(n_alive), dim3(kBlock), 0, s, alive_reads, touched, cmed,
                           n_alive, out);
    }
}

}  // namespace rala_hip
