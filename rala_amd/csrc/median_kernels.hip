// Median of the pile medians per connected component (Graph::preprocess, graph.cpp:777-783).
//
// Every read with an overlap needs the element at size / 2 of its component's sorted pile medians
// (the reference's nth_element).  Rounds 1 - 3 sorted keys `label << 16 | median` with rocPRIM (a block
// sort and eight merge passes for 134 k keys: nine launches, twice per C3 step).  Now, in tree:
//   count   (with the components' last compression, tr_kernels.hip) size[label] = reads with an overlap;
//           one add per wavefront for the label its first such read has - an overlap graph is one big
//           component and a few small ones, so nearly all lanes share it - one add apiece for the rest;
//   offsets exclusive scan of size[] (scan_pass.h);
//   fill    the medians, grouped by component: a place from the component's cursor (again one returning
//           add per wavefront for the shared label);
//   select  the element at size / 2 of every component: a workgroup finds the roots among its 1024 labels;
//           components of up to 64 reads are ranked by one wavefront each (every lane counts the values
//           below its own), larger ones by the workgroup with two 256-bin histograms in LDS (high byte,
//           then low byte of the 16-bit medians) - a radix select, no sort;
//   spread  cmed[read] = its root's.
#include <hip/hip_runtime.h>

#include "device_utils.h"
#include "kernels.h"
#include "scan_pass.h"

namespace rala_hip {

namespace {

constexpr int kBlock = 256;
constexpr int kSelectBlock = 1024;

struct MedianSpace {
    uint32_t *size, *cursor, *start;
    uint16_t *vals, *root;
    ScanSpace scan;
    size_t zero_bytes;          // from `size` on: size[], cursor[], the scan's tile states and ticket
};

// layout: size[n] | cursor[n] | scan states (tiles + 1 words of 8 bytes) | ticket (8 bytes) || start[n + 2] | vals[n] | root[n]
inline size_t align16(size_t b) { return (b + 15) & ~(size_t)15; }
inline MedianSpace median_space(void* tmp, uint32_t n) {
    MedianSpace m;
    unsigned char* p = (unsigned char*)tmp;
    m.size = (uint32_t*)p; p += align16((size_t)n * 4);
    m.cursor = (uint32_t*)p; p += align16((size_t)n * 4);
    const size_t tiles = scan_tiles_for(n) + 1;
    m.scan.state = (uint64_t*)p; p += tiles * 8;
    m.scan.ticket = (uint32_t*)p; p += 16;
    m.scan.words = tiles; m.scan.used = 0;
    m.zero_bytes = (size_t)(p - (unsigned char*)tmp);
    m.start = (uint32_t*)p; p += align16((size_t)(n + 2) * 4);
    m.vals = (uint16_t*)p; p += align16((size_t)n * 2);
    m.root = (uint16_t*)p; p += align16((size_t)n * 2);
    return m;
}

// a place for every read with an overlap in its component's stretch of vals[]: one returning add per workgroup for
// the label its first such read has, one apiece for the others
__global__ __launch_bounds__(kBlock) void median_fill_kernel(const uint32_t* __restrict__ label, const uint8_t* __restrict__ touched,
                                                             const uint32_t* __restrict__ reads, const uint16_t* __restrict__ median,
                                                             uint32_t n, const uint32_t* __restrict__ start, uint32_t* cursor,
                                                             uint16_t* __restrict__ vals) {
    __shared__ uint32_t s_label[kBlock / 64], s_any[kBlock / 64], s_cnt[kBlock / 64], s_base;
    const uint32_t v = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const bool t = v < n && touched[v];
    const uint32_t r = t ? label[v] : 0u;
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
        s_base = atomicAdd(&cursor[shared], c);
    }
    __syncthreads();
    if (!t) return;
    uint32_t at;
    if (same) {
        at = s_base + (uint32_t)__popcll(ms & ((1ull << lane) - 1ull));
        for (uint32_t w = 0; w < wave; ++w) at += s_cnt[w];
    } else {
        at = atomicAdd(&cursor[r], 1u);
    }
    vals[start[r] + at] = median[reads[v]];
}

// the element at size / 2 (ascending) of every component whose root is among this workgroup's labels
__global__ __launch_bounds__(kSelectBlock) void median_select_kernel(const uint32_t* __restrict__ label, const uint32_t* __restrict__ size,
                                                                     const uint32_t* __restrict__ start, const uint16_t* __restrict__ vals,
                                                                     uint32_t n, uint16_t* __restrict__ root_median) {
    __shared__ uint32_t s_small[kSelectBlock], s_big[kSelectBlock], s_n[2];
    __shared__ uint32_t hist_w[(kSelectBlock / 64) * 256], s_pick[2];
    __shared__ uint32_t tmp[kSelectBlock / 64 + 1];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (tid < 2) s_n[tid] = 0;
    __syncthreads();
    const uint32_t v = blockIdx.x * kSelectBlock + tid;
    if (v < n && label[v] == v) {
        const uint32_t s = size[v];
        if (s > 64) s_big[atomicAdd(&s_n[1], 1u)] = v;
        else if (s > 0) s_small[atomicAdd(&s_n[0], 1u)] = v;
    }
    __syncthreads();
    // small components: one wavefront each; a lane's value is the wanted one when exactly k values come before it
    // (smaller, or equal with a lower index)
    for (uint32_t j = wave; j < s_n[0]; j += kSelectBlock / 64) {
        const uint32_t r = s_small[j], s = size[r], p = start[r], k = s / 2;
        const uint32_t mine = lane < s ? vals[p + lane] : 0xFFFFFFFFu;
        uint32_t before = 0;
        for (uint32_t o = 0; o < s; ++o) {
            const uint32_t x = (uint32_t)__shfl((int)mine, (int)o);
            before += (x < mine || (x == mine && o < lane)) ? 1u : 0u;
        }
        if (lane < s && before == k) root_median[r] = (uint16_t)mine;
    }
    // larger ones: the whole workgroup, one after the other; two histograms of 256 bins.  Pile medians crowd a few
    // values (the coverage): every wavefront has a histogram of its own (all wavefronts adding to ONE word took 60 us a
    // pass for 134 k values)
    constexpr uint32_t kWaves = kSelectBlock / 64;
    for (uint32_t j = 0; j < s_n[1]; ++j) {
        const uint32_t r = s_big[j], s = size[r], p = start[r];
        uint32_t k = s / 2, high = 0;
        for (int pass = 0; pass < 2; ++pass) {
            __syncthreads();
            for (uint32_t b = tid; b < kWaves * 256; b += kSelectBlock) hist_w[b] = 0;
            __syncthreads();
            uint32_t* mine = hist_w + wave * 256;
            // eight values per 16-byte load from the aligned window around the component's stretch of vals[]
            const uint32_t first = p & ~7u, last = p + s;
            const uint32_t n_groups = (last - first + 7u) / 8u;
            const uint4* groups = (const uint4*)(vals + first);
            constexpr uint32_t kHeld = 4;
            // (a lane keeps the count of the bin it saw last and adds when the bin changes: the high bytes of a
            // component's medians are nearly all the same - adds of all lanes to one LDS word take their turns)
            uint32_t held_bin = 0xFFFFFFFFu, held = 0;
            for (uint32_t g0 = 0; g0 < n_groups; g0 += kSelectBlock * kHeld) {
                uint4 q[kHeld];
#pragma unroll
                for (uint32_t u = 0; u < kHeld; ++u) {
                    const uint32_t g = g0 + u * kSelectBlock + tid;
                    q[u] = g < n_groups ? groups[g] : make_uint4(0, 0, 0, 0);
                }
#pragma unroll
                for (uint32_t u = 0; u < kHeld; ++u) {
                    const uint32_t g = g0 + u * kSelectBlock + tid;
                    const uint32_t w[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
#pragma unroll
                    for (uint32_t e = 0; e < 8; ++e) {
                        const uint32_t idx = first + 8u * g + e;
                        const uint32_t x = (w[e >> 1] >> (16u * (e & 1u))) & 0xFFFFu;
                        const bool in = g < n_groups && idx >= p && idx < last && (pass == 0 || (x >> 8) == high);
                        if (!in) continue;
                        const uint32_t bin = pass == 0 ? x >> 8 : x & 255u;
                        if (bin == held_bin) { ++held; continue; }
                        if (held) atomicAdd(&mine[held_bin], held);
                        held_bin = bin; held = 1;
                    }
                }
            }
            if (held) atomicAdd(&mine[held_bin], held);
            __syncthreads();
            uint32_t c = 0;
            if (tid < 256) for (uint32_t w = 0; w < kWaves; ++w) c += hist_w[w * 256 + tid];
            uint32_t tot;
            const uint32_t ex = block_scan_excl<kSelectBlock>(c, OpAdd(), 0u, tmp, tot);
            if (tid < 256 && k >= ex && k < ex + c) { s_pick[0] = tid; s_pick[1] = k - ex; }
            __syncthreads();
            if (pass == 0) high = s_pick[0];
            k = s_pick[1];
        }
        if (tid == 0) root_median[r] = (uint16_t)(high << 8 | s_pick[0]);
    }
}

__global__ __launch_bounds__(kBlock) void median_spread_kernel(const uint32_t* __restrict__ label, const uint8_t* __restrict__ touched,
                                                               const uint16_t* __restrict__ root_median, uint32_t n,
                                                               uint16_t* __restrict__ cmed) {
    const uint32_t q = blockIdx.x * kBlock + threadIdx.x;
    if (q < n && touched[q]) cmed[q] = root_median[label[q]];
}

}  // namespace

size_t component_median_workspace(uint32_t n) {
    return 2 * align16((size_t)n * 4) + ((size_t)scan_tiles_for(n) + 1) * 8 + 16 + align16((size_t)(n + 2) * 4) +
           2 * align16((size_t)n * 2) + 256;
}

void component_median_clear(void* tmp, uint32_t n, FillList& fills) {
    if (n) fills.add(tmp, 0, median_space(tmp, n).zero_bytes);
}

uint32_t* component_median_sizes(void* tmp, uint32_t n) { return median_space(tmp, n).size; }

hipError_t launch_component_medians(const uint32_t* label, const uint8_t* touched, const uint32_t* alive_reads,
                                    const uint16_t* median, uint32_t n_alive, void* tmp, uint16_t* cmed, hipStream_t s) {
    if (n_alive == 0) return hipSuccess;
    MedianSpace m = median_space(tmp, n_alive);
    if (!launch_offsets_pass(m.size, m.start, nullptr, n_alive, m.scan, s)) return hipErrorOutOfMemory;
    const dim3 grid((n_alive + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(median_fill_kernel, grid, dim3(kBlock), 0, s, label, touched, alive_reads, median, n_alive,
                       (const uint32_t*)m.start, m.cursor, m.vals);
    hipLaunchKernelGGL(median_select_kernel, dim3((n_alive + kSelectBlock - 1) / kSelectBlock), dim3(kSelectBlock), 0, s, label,
                       (const uint32_t*)m.size, (const uint32_t*)m.start, (const uint16_t*)m.vals, n_alive, m.root);
    hipLaunchKernelGGL(median_spread_kernel, grid, dim3(kBlock), 0, s, label, touched, (const uint16_t*)m.root, n_alive, cmed);
    return hipGetLastError();
}

}  // namespace rala_hip
