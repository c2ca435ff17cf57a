// Transitive-edge marking on a CSR overlap graph.
//
// Reference: rvaser/rala Graph::remove_transitive_edges (src/graph.cpp:1281-1335).
// For every path a->b->c with an edge a->c present, a->c (the LAST such edge in
// a's out-list, graph.cpp:1291-1293) and its reverse-complement twin are marked
// when len(ab)+len(bc) is within 12 % of len(ac) (comparable(), :26-29, FP64).
// Marked edges keep serving as ab / bc, so the marked set does not depend on
// the visiting order (SURVEY B-T15): one wavefront-independent thread per
// edge a->b probes a's out-list for every successor of b.
#include <hip/hip_runtime.h>

#include "device_utils.h"
#include "geom.h"
#include "kernels.h"

namespace rala_hip {

namespace {

constexpr int kBlock = 256;

__global__ __launch_bounds__(kBlock) void tr_mark_kernel(const uint32_t* __restrict__ row_ptr,
                                                         const uint2* __restrict__ adj,
                                                         const uint32_t* __restrict__ esrc,
                                                         const uint32_t* __restrict__ edst,
                                                         const uint32_t* __restrict__ elen, uint32_t n_nodes,
                                                         uint32_t first_edge, uint32_t n_edges, uint8_t* marks) {
    // edges [first_edge, n_edges): all of them, or one rank's share of a sharded run
    const uint32_t e_ab = first_edge + blockIdx.x * kBlock + threadIdx.x;
    if (e_ab >= n_edges) return;
    const uint32_t a = esrc[e_ab], b = edst[e_ab];
    if (a >= n_nodes || b >= n_nodes) return;          // reported by tr_degree_kernel
    const uint32_t len_ab = elen[e_ab];
    const uint32_t a0 = row_ptr[a], a1 = row_ptr[a + 1];
    const uint32_t b0 = row_ptr[b], b1 = row_ptr[b + 1];
    // (the out-lists hold {edge, its target}: one sequential read per list element instead of a
    // gather through the edge's target for each)
    constexpr uint32_t kHeld = 16;
    if (a1 - a0 <= kHeld) {
        // a's out-list in registers (an overlap graph's nodes have eight or nine successors): every successor of b
        // is then compared with it without a load - 17 instead of 72 loads per edge at C3
        uint2 held[kHeld];
#pragma unroll
        for (uint32_t m = 0; m < kHeld; ++m) held[m] = a0 + m < a1 ? adj[a0 + m] : make_uint2(0u, 0xFFFFFFFFu);
        for (uint32_t k = b0; k < b1; ++k) {
            const uint2 bc = adj[k];
            const uint32_t c = bc.y;
            uint32_t cand = 0xFFFFFFFFu;
#pragma unroll
            for (uint32_t m = 0; m < kHeld; ++m) {
                if (held[m].y == c && (cand == 0xFFFFFFFFu || held[m].x > cand)) cand = held[m].x;
            }
            if (cand == 0xFFFFFFFFu) continue;
            const uint32_t sum = len_ab + elen[bc.x];
            if (comparable((double)sum, (double)elen[cand], 0.12)) {
                marks[cand] = 1;
                marks[cand ^ 1u] = 1;
            }
        }
        return;
    }
    for (uint32_t k = b0; k < b1; ++k) {
        const uint2 bc = adj[k];
        const uint32_t c = bc.y;
        // the candidate is the LAST edge a->c of a's out-list = the one with the highest id
        // (out-lists are in edge-id order in the reference; here the list order is arbitrary)
        uint32_t cand = 0xFFFFFFFFu;
        for (uint32_t m = a0; m < a1; ++m) {
            const uint2 ac = adj[m];
            if (ac.y == c && (cand == 0xFFFFFFFFu || ac.x > cand)) cand = ac.x;
        }
        if (cand == 0xFFFFFFFFu) continue;
        const uint32_t sum = len_ab + elen[bc.x];
        if (comparable((double)sum, (double)elen[cand], 0.12)) {
            marks[cand] = 1;
            marks[cand ^ 1u] = 1;
        }
    }
}

__global__ __launch_bounds__(kBlock) void tr_count_kernel(uint8_t* __restrict__ marks, uint32_t n_edges,
                                                          uint32_t* n_pairs) {
    // (few workgroups, one add each: adds to one word cost about 10 ns apiece)
    __shared__ uint32_t tmp[kBlock / 64 + 1];
    uint32_t v = 0;
    for (uint32_t p = blockIdx.x * kBlock + threadIdx.x; 2 * p + 1 < n_edges; p += gridDim.x * kBlock) {      // pair index
        if (marks[2 * p]) { marks[2 * p] = 1; marks[2 * p + 1] = 1; ++v; }        // (sharded runs summed the ranks' bytes)
    }
    v = block_reduce<kBlock>(v, OpAdd(), 0u, tmp);
    if (threadIdx.x == 0 && v) atomicAdd(n_pairs, v);
}

__global__ __launch_bounds__(kBlock) void tr_degree_kernel(const uint32_t* __restrict__ src,
                                                           const uint32_t* __restrict__ dst, uint32_t n_nodes,
                                                           uint32_t n_edges, uint32_t* deg, uint32_t* bad) {
    const uint32_t e = blockIdx.x * kBlock + threadIdx.x;
    if (e >= n_edges) return;
    const uint32_t a = src[e], b = dst[e];
    if (a >= n_nodes || b >= n_nodes) { *bad = 1; return; }
    atomicAdd(&deg[a], 1u);
}

__global__ __launch_bounds__(kBlock) void tr_fill_kernel(const uint32_t* __restrict__ src, const uint32_t* __restrict__ dst,
                                                         uint32_t n_nodes, uint32_t n_edges, uint32_t* cursor, uint2* adj) {
    const uint32_t e = blockIdx.x * kBlock + threadIdx.x;
    if (e >= n_edges) return;
    const uint32_t a = src[e];
    if (a >= n_nodes) return;
    adj[atomicAdd(&cursor[a], 1u)] = make_uint2(e, dst[e]);
}

// ---- connected components (Graph::preprocess, reference graph.cpp:740-773; only the member
// sets matter: a median per component) -----------------------------------------------------
__global__ __launch_bounds__(kBlock) void cc_init_kernel(uint32_t* label, uint32_t n) {
    const uint32_t v = blockIdx.x * kBlock + threadIdx.x;
    if (v < n) label[v] = v;
}

__device__ __forceinline__ uint32_t cc_root(const uint32_t* label, uint32_t v) {
    uint32_t p = label[v];
    while (p != v) { v = p; p = label[v]; }
    return v;
}

// unite the trees of every edge's endpoints (the larger root under the smaller one; labels only ever decrease)
// sample != 0: only the first two edges of every run of equal first endpoint take part
// (the list is grouped by query read).  Two such rounds connect most of a large component
// at the price of two atomics per node; the full rounds that follow then find almost every
// edge inside one tree already and skip it, instead of hammering a few roots with a million
// atomic minima.
__global__ __launch_bounds__(kBlock) void cc_hook_kernel(const uint32_t* __restrict__ edges, uint32_t n_edges,
                                                         uint32_t sample, uint32_t* label, uint32_t* changed) {
    const uint32_t e = blockIdx.x * kBlock + threadIdx.x;
    bool hooked = false;
    if (e < n_edges) {
        const uint32_t a = edges[2 * e], b = edges[2 * e + 1];
        if (a != b && (!sample || e < 2 || edges[2 * (e - 2)] != a)) {
            // Lock-free union, to the end: the larger root goes under the smaller one; if it was no
            // root any more (somebody hooked it under `old` in the meantime) it now points to the
            // smaller of the two, and what is left to do is the union of those two.  Labels only
            // fall, so this ends; when the kernel ends every edge of the launch lies inside one
            // tree - one full launch is enough, no rounds, no look from the host.
            uint32_t ra = cc_root(label, a), rb = cc_root(label, b);
            while (ra != rb) {
                const uint32_t hi = ra > rb ? ra : rb, lo = ra > rb ? rb : ra;
                // (a coherent look first: no atomic on a node that is no root any more - a few roots
                // would otherwise take a million atomic minima)
                uint32_t old = __hip_atomic_load(&label[hi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (old == hi) {
                    old = atomicMin(&label[hi], lo);
                    hooked = true;
                    if (old == hi) break;
                }
                ra = cc_root(label, old < lo ? old : lo);
                rb = cc_root(label, old < lo ? lo : old);
            }
        }
    }
    if (__ballot(hooked) != 0 && (threadIdx.x & 63) == 0) *changed = 1;
}

__global__ __launch_bounds__(kBlock) void cc_compress_kernel(uint32_t* label, uint32_t n) {
    const uint32_t v = blockIdx.x * kBlock + threadIdx.x;
    if (v >= n) return;
    uint32_t r = label[v];
    while (label[r] != r) r = label[r];
    label[v] = r;
}

// ... and, behind the last round, the components' sizes in reads with an overlap (median_kernels.hip): one add per
// WORKGROUP for the label its first such read has (an overlap graph is one big component and a few small ones: nearly
// all reads share it, and adds to one word take about 10 ns apiece), one apiece for the others
__global__ __launch_bounds__(kBlock) void cc_compress_count_kernel(uint32_t* label, uint32_t n, const uint8_t* __restrict__ touched,
                                                                   uint32_t* size) {
    __shared__ uint32_t s_label[kBlock / 64], s_any[kBlock / 64], s_cnt[kBlock / 64];
    const uint32_t v = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t r = 0;
    if (v < n) {
        r = label[v];
        while (label[r] != r) r = label[r];
        label[v] = r;
    }
    const bool t = v < n && touched[v];
    const uint64_t m = __ballot(t);
    const uint32_t first = (uint32_t)__shfl((int)r, m ? __ffsll((long long)m) - 1 : 0);
    if (lane == 0) { s_any[wave] = m != 0; s_label[wave] = first; }
    __syncthreads();
    uint32_t shared = 0xFFFFFFFFu;
    for (int w = kBlock / 64 - 1; w >= 0; --w) if (s_any[w]) shared = s_label[w];
    if (shared == 0xFFFFFFFFu) return;
    const bool same = t && r == shared;
    const uint64_t ms = __ballot(same);
    if (lane == 0) s_cnt[wave] = (uint32_t)__popcll(ms);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t c = 0;
        for (int w = 0; w < kBlock / 64; ++w) c += s_cnt[w];
        atomicAdd(&size[shared], c);
    }
    if (t && !same) atomicAdd(&size[r], 1u);
}

}  // namespace

void launch_cc_compress_count(uint32_t* label, uint32_t n, const uint8_t* touched, uint32_t* size, hipStream_t s) {
    if (n) hipLaunchKernelGGL(cc_compress_count_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, s, label, n, touched, size);
}

void launch_tr_degree(const uint32_t* src, const uint32_t* dst, uint32_t n_nodes, uint32_t n_edges, uint32_t* deg,
                      uint32_t* bad, hipStream_t s) {
    if (n_edges) {
        hipLaunchKernelGGL(tr_degree_kernel, dim3((n_edges + kBlock - 1) / kBlock), dim3(kBlock), 0, s, src, dst,
                           n_nodes, n_edges, deg, bad);
    }
}
void launch_tr_fill(const uint32_t* src, const uint32_t* dst, uint32_t n_nodes, uint32_t n_edges, uint32_t* cursor, uint32_t* adj,
                    hipStream_t s) {
    if (n_edges) {
        hipLaunchKernelGGL(tr_fill_kernel, dim3((n_edges + kBlock - 1) / kBlock), dim3(kBlock), 0, s, src, dst, n_nodes,
                           n_edges, cursor, (uint2*)adj);
    }
}
void launch_cc_init(uint32_t* label, uint32_t n, hipStream_t s) {
    if (n) hipLaunchKernelGGL(cc_init_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, s, label, n);
}
void launch_cc_hook(const uint32_t* edges, uint32_t n_edges, uint32_t sample, uint32_t* label, uint32_t* changed,
                    hipStream_t s) {
    if (n_edges) {
        hipLaunchKernelGGL(cc_hook_kernel, dim3((n_edges + kBlock - 1) / kBlock), dim3(kBlock), 0, s, edges, n_edges,
                           sample, label, changed);
    }
}
void launch_cc_compress(uint32_t* label, uint32_t n, hipStream_t s) {
    if (n) hipLaunchKernelGGL(cc_compress_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, s, label, n);
}

void launch_tr_mark(const uint32_t* row_ptr, const uint32_t* adj_edge, const uint32_t* edge_src,
                    const uint32_t* edge_dst, const uint32_t* edge_len, uint32_t n_nodes, uint32_t first_edge,
                    uint32_t last_edge, uint8_t* marks, hipStream_t s) {
    if (last_edge <= first_edge) return;
    hipLaunchKernelGGL(tr_mark_kernel, dim3((last_edge - first_edge + kBlock - 1) / kBlock), dim3(kBlock), 0, s, row_ptr,
                       (const uint2*)adj_edge, edge_src, edge_dst, edge_len, n_nodes, first_edge, last_edge, marks);
}

void launch_tr_count(uint8_t* marks, uint32_t n_edges, uint32_t* n_pairs, hipStream_t s) {
    const uint32_t pairs = n_edges / 2;
    if (pairs == 0) return;
    const uint32_t blocks = (pairs + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(tr_count_kernel, dim3(blocks < 128 ? blocks : 128), dim3(kBlock), 0, s, marks, n_edges, n_pairs);
}

}  // namespace rala_hip
