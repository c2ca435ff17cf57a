// Checksums of the pile rows where they lie (20 GB at C3, 80 GB at C5): what a test - or a maintainer comparing two builds -
// needs to know about EVERY row without copying it to the host.
//
// Per read: FNV-1a-64 over the bytes of Pile::data() as rala_hip_get_pile_data would hand it out (the values outside the valid
// region that applies count as zero - Pile::shrink, reference src/pile.cpp:311-318; uint16 values, low byte first: the bytes of
// the reference's std::vector<uint16_t> data_, src/pile.hpp:53), the sum of the row inside the region and the sum of the stored
// values outside it.  FNV-1a is a chain of dependent multiplications, so a row is one thread's work; a thread's 16-byte loads walk
// its own row (the cache holds a wavefront's 64 lines between the eight loads that share one).  A million rows take tens of
// milliseconds: a verification pass, not a stage of the path.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace rala_hip {

namespace {

constexpr uint64_t kFnvOffset = 1469598103934665603ull, kFnvPrime = 1099511628211ull;

__global__ __launch_bounds__(64) void pile_row_digest_kernel(const uint16_t* __restrict__ pile, const uint64_t* __restrict__ pile_off,
                                                              const uint32_t* __restrict__ read_len, const uint32_t* __restrict__ begin,
                                                              const uint32_t* __restrict__ end, const uint8_t* __restrict__ alive,
                                                              uint32_t n_rows, uint64_t* __restrict__ fnv, uint64_t* __restrict__ inside,
                                                              uint64_t* __restrict__ outside) {
    const uint32_t j = blockIdx.x * 64u + threadIdx.x;
    if (j >= n_rows) return;
    uint64_t h = 0, in = 0, out = 0;
    if (alive[j] && pile_off[j] != ~0ull) {
        const uint32_t n = read_len[j], B = begin[j], E = end[j];
        const uint4* row = (const uint4*)(pile + pile_off[j]);            // rows start on 128-byte boundaries, padded to 8 values
        h = kFnvOffset;
        for (uint32_t p0 = 0; p0 < n; p0 += 8) {
            const uint4 q = row[p0 >> 3];
            const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (uint32_t e = 0; e < 8; ++e) {
                const uint32_t p = p0 + e;
                if (p >= n) break;
                const uint32_t stored = (w[e >> 1] >> (16u * (e & 1u))) & 0xFFFFu;
                const bool in_region = p >= B && p < E;
                const uint32_t v = in_region ? stored : 0u;
                if (in_region) in += stored; else out += stored;
                h = (h ^ (uint64_t)(v & 0xFFu)) * kFnvPrime;
                h = (h ^ (uint64_t)(v >> 8)) * kFnvPrime;
            }
        }
    }
    if (fnv) fnv[j] = h;
    if (inside) inside[j] = in;
    if (outside) outside[j] = out;
}

}  // namespace

void launch_pile_row_digests(const uint16_t* pile, const uint64_t* pile_off, const uint32_t* read_len, const uint32_t* begin,
                             const uint32_t* end, const uint8_t* alive, uint32_t n_rows, uint64_t* fnv, uint64_t* inside,
                             uint64_t* outside, hipStream_t stream) {
    if (n_rows == 0) return;
    hipLaunchKernelGGL(pile_row_digest_kernel, dim3((n_rows + 63u) / 64u), dim3(64), 0, stream, pile, pile_off, read_len, begin, end, alive,
                       n_rows, fnv, inside, outside);
}

}  // namespace rala_hip
