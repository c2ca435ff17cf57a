// Device-wide exclusive prefix sum (uint32): tile sums -> tile scans, the carry-in of a tile added
// up from the sums in front of it (two launches); beyond 2048 tiles a scan of the tile sums by one
// workgroup in between (three).
#include <hip/hip_runtime.h>

#include "device_utils.h"
#include "kernels.h"

namespace rala_hip {

namespace {

constexpr int kBlock = 256;
constexpr int kItems = 16;                      // per thread
constexpr uint32_t kTile = kBlock * kItems;     // 4096 values per workgroup

__global__ __launch_bounds__(kBlock) void scan_tile_sums(const uint32_t* __restrict__ in, uint64_t n,
                                                         uint32_t* __restrict__ sums) {
    __shared__ uint32_t tmp[kBlock / 64 + 1];
    const uint64_t base = (uint64_t)blockIdx.x * kTile;
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < kItems; ++k) {
        const uint64_t i = base + (uint64_t)k * kBlock + threadIdx.x;
        if (i < n) s += in[i];
    }
    s = block_reduce<kBlock>(s, OpAdd(), 0u, tmp);
    if (threadIdx.x == 0) sums[blockIdx.x] = s;
}

// exclusive scan of the tile sums in place; sums[n_tiles] = grand total
__global__ __launch_bounds__(kBlock) void scan_sums(uint32_t* sums, uint32_t n_tiles) {
    __shared__ uint32_t tmp[kBlock / 64 + 1];
    uint32_t carry = 0;
    for (uint32_t b = 0; b < n_tiles; b += kBlock) {
        const uint32_t i = b + threadIdx.x;
        const uint32_t v = i < n_tiles ? sums[i] : 0;
        uint32_t tot;
        const uint32_t ex = block_scan_excl<kBlock>(v, OpAdd(), 0u, tmp, tot);
        if (i < n_tiles) sums[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) sums[n_tiles] = carry;
}

__global__ __launch_bounds__(kBlock) void scan_tiles(const uint32_t* in, uint32_t* out, uint64_t n,
                                                     const uint32_t* __restrict__ sums, uint32_t n_tiles) {
    __shared__ uint32_t tmp[kBlock / 64 + 1];
    const uint64_t base = (uint64_t)blockIdx.x * kTile;
    uint32_t carry = sums[blockIdx.x];
#pragma unroll 1
    for (int k = 0; k < kItems; ++k) {
        const uint64_t i = base + (uint64_t)k * kBlock + threadIdx.x;
        const uint32_t v = i < n ? in[i] : 0;
        uint32_t tot;
        const uint32_t ex = block_scan_excl<kBlock>(v, OpAdd(), 0u, tmp, tot);
        if (i < n) out[i] = carry + ex;
        carry += tot;
    }
    if (blockIdx.x == n_tiles - 1 && threadIdx.x == 0) out[n] = sums[n_tiles];
}

// The same with the carry taken straight from the raw tile sums (every workgroup adds up the sums of
// the tiles in front of it: at most kFusedTiles values out of the L2) - two launches instead of three
// for the scans of up to 8 M values, which is all of them but the bucketing's fallback path.
constexpr uint32_t kFusedTiles = 2048;

__global__ __launch_bounds__(kBlock) void scan_tiles_fused(const uint32_t* in, uint32_t* out, uint64_t n,
                                                           const uint32_t* __restrict__ sums, uint32_t n_tiles) {
    __shared__ uint32_t tmp[kBlock / 64 + 1];
    uint32_t before = 0;
    for (uint32_t t = threadIdx.x; t < blockIdx.x; t += kBlock) before += sums[t];
    uint32_t carry = block_reduce<kBlock>(before, OpAdd(), 0u, tmp);       // (in every thread, behind a barrier)
    const uint64_t base = (uint64_t)blockIdx.x * kTile;
#pragma unroll 1
    for (int k = 0; k < kItems; ++k) {
        const uint64_t i = base + (uint64_t)k * kBlock + threadIdx.x;
        const uint32_t v = i < n ? in[i] : 0;
        uint32_t tot;
        const uint32_t ex = block_scan_excl<kBlock>(v, OpAdd(), 0u, tmp, tot);
        if (i < n) out[i] = carry + ex;
        carry += tot;
    }
    if (blockIdx.x == n_tiles - 1 && threadIdx.x == 0) out[n] = carry;
}

__global__ void scan_empty(uint32_t* out) { out[0] = 0; }

}  // namespace

size_t scan_workspace_bytes(uint64_t n) { return ((n + kTile - 1) / kTile + 2) * sizeof(uint32_t); }

void launch_exclusive_scan(const uint32_t* in, uint32_t* out, uint64_t n, void* workspace, hipStream_t s) {
    if (n == 0) {
        hipLaunchKernelGGL(scan_empty, dim3(1), dim3(1), 0, s, out);
        return;
    }
    uint32_t* sums = (uint32_t*)workspace;
    const uint32_t n_tiles = (uint32_t)((n + kTile - 1) / kTile);
    hipLaunchKernelGGL(scan_tile_sums, dim3(n_tiles), dim3(kBlock), 0, s, in, n, sums);
    if (n_tiles <= kFusedTiles) {
        hipLaunchKernelGGL(scan_tiles_fused, dim3(n_tiles), dim3(kBlock), 0, s, in, out, n, (const uint32_t*)sums, n_tiles);
        return;
    }
    hipLaunchKernelGGL(scan_sums, dim3(1), dim3(kBlock), 0, s, sums, n_tiles);
    hipLaunchKernelGGL(scan_tiles, dim3(n_tiles), dim3(kBlock), 0, s, in, out, n, (const uint32_t*)sums, n_tiles);
}

}  // namespace rala_hip
