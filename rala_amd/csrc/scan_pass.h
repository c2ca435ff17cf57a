// Single-pass device-wide exclusive scan with the producer and the consumer inside the kernel.
//
// The tail of Graph::preprocess (tail_kernels.hip) and the graph build are chains of "flag the items
// that ..., scan the flags, place the flagged items": three to five launches of 5 us each for a few
// microseconds of work, on a million items.  Here one launch does all of it: a workgroup takes the
// next tile of kScanTile items (a ticket from an atomic counter, so that tiles start in order), asks the
// functor for every item's value, publishes the tile's sum, finds the sum of all tiles in front of it by
// looking back over their published sums (the "decoupled look-back" of single-pass scans: a tile whose
// predecessors have published an inclusive prefix stops there), and hands every item its exclusive
// prefix.  The value is 62 bits wide (what is published is masked to that: a functor that overflows cannot touch the status
// bits), so two counters of up to 31 bits can ride in one scan (kept
// overlaps and dovetails among them; survivors that go on as overlaps and as internals).
//
// State per tile: one 64-bit word {status : 2, value : 62}; the words of a scan must be zero when it
// starts (ScanSpace hands out regions of one buffer that is cleared once per stage).  The ticket counter
// resets itself: the workgroup that draws the last ticket knows every other one has drawn.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_utils.h"

namespace rala_hip {

constexpr uint32_t kScanBlock = 256;
constexpr uint32_t kScanItems = 8;
constexpr uint32_t kScanTile = kScanBlock * kScanItems;

constexpr uint64_t kScanValueMask = (1ull << 62) - 1ull;
constexpr uint64_t kScanAggregate = 1ull << 62;     // the tile's own sum
constexpr uint64_t kScanPrefix = 2ull << 62;        // the sum of this tile and everything in front of it

inline uint32_t scan_tiles_for(uint64_t n) { return (uint32_t)((n + kScanTile - 1) / kScanTile); }

// F: uint64_t value(uint32_t item) for item < n (called once per item);
//    void place(uint32_t item, uint64_t value, uint64_t exclusive_prefix);
//    void total(uint64_t sum) - called by one thread of the last tile.
template <class F>
__global__ __launch_bounds__(kScanBlock) void scan_pass_kernel(uint32_t n, F f, uint64_t* __restrict__ state, uint32_t* ticket) {
    __shared__ uint64_t tmp[kScanBlock / 64 + 1];
    __shared__ uint32_t s_tile;
    __shared__ uint64_t s_before;
    const uint32_t n_tiles = (n + kScanTile - 1) / kScanTile;
    if (threadIdx.x == 0) {
        const uint32_t t = atomicAdd(ticket, 1u);
        s_tile = t;
        if (t + 1 == n_tiles) *ticket = 0;          // every tile has drawn: ready for the next scan
    }
    __syncthreads();
    const uint32_t tile = s_tile;
    // consecutive threads take consecutive items (the functor's loads coalesce); kScanItems rounds per tile
    uint64_t v[kScanItems], ex[kScanItems];
    uint64_t tile_sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < kScanItems; ++k) {
        const uint32_t i = tile * kScanTile + k * kScanBlock + threadIdx.x;
        v[k] = i < n ? f.value(i) : 0ull;
    }
#pragma unroll
    for (uint32_t k = 0; k < kScanItems; ++k) {
        uint64_t tot;
        ex[k] = tile_sum + block_scan_excl<(int)kScanBlock>(v[k], OpAdd(), (uint64_t)0, tmp, tot);
        tile_sum += tot;
    }
    if (threadIdx.x < 64) {
        // One wavefront looks back.  The state words carry everything a later tile needs of an earlier one,
        // so relaxed accesses are enough - an acquire inside the spin would invalidate the caches under the
        // other tiles' loads, once per look.
        const uint32_t lane = threadIdx.x;
        uint64_t before = 0;
        if (tile == 0) {
            if (lane == 0) __hip_atomic_store(&state[0], kScanPrefix | (tile_sum & kScanValueMask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (lane == 0) __hip_atomic_store(&state[tile], kScanAggregate | (tile_sum & kScanValueMask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int64_t first = (int64_t)tile - 1;       // nearest tile not yet accounted for
            for (;;) {
                const int64_t t = first - (int64_t)lane;
                uint64_t w = kScanPrefix;            // (in front of tile 0: an empty prefix)
                if (t >= 0) {
                    do {
                        w = __hip_atomic_load(&state[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } while ((w >> 62) == 0);
                }
                const uint64_t has_prefix = __ballot((w >> 62) == 2);
                // the nearest lane that holds an inclusive prefix ends the walk
                const uint32_t stop = has_prefix ? (uint32_t)__builtin_ctzll(has_prefix) : 63u;
                const uint64_t part = lane <= stop ? (w & kScanValueMask) : 0ull;
                before += wave_reduce(part, OpAdd());
                if (has_prefix) break;
                first -= 64;
            }
            if (lane == 0) __hip_atomic_store(&state[tile], kScanPrefix | ((before + tile_sum) & kScanValueMask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) {
            s_before = before;
            if (tile + 1 == n_tiles) f.total(before + tile_sum);
        }
    }
    __syncthreads();
    const uint64_t before = s_before;
#pragma unroll
    for (uint32_t k = 0; k < kScanItems; ++k) {
        const uint32_t i = tile * kScanTile + k * kScanBlock + threadIdx.x;
        if (i < n) f.place(i, v[k], before + ex[k]);
    }
}

// regions of tile states for the scans of one stage; cleared with one memset per stage
struct ScanSpace {
    uint64_t* state = nullptr;
    uint32_t* ticket = nullptr;
    size_t words = 0, used = 0;
    uint64_t* take(uint64_t n_items) {
        const size_t need = scan_tiles_for(n_items) + 1;
        uint64_t* p = state + used;
        used += need;
        return used <= words ? p : nullptr;
    }
};

template <class F>
inline bool launch_scan_pass(uint32_t n, const F& f, ScanSpace& space, hipStream_t s) {
    if (n == 0) return true;
    uint64_t* st = space.take(n);
    if (!st) return false;
    hipLaunchKernelGGL(scan_pass_kernel<F>, dim3(scan_tiles_for(n)), dim3(kScanBlock), 0, s, n, f, st, space.ticket);
    return true;
}

}  // namespace rala_hip
