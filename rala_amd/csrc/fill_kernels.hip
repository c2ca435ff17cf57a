// Several buffers set to a byte value in ONE launch.
//
// `hipMemsetAsync` is a launch of the runtime's own fill kernel, and between two of them the queue
// stands still for 4 - 10 us (tools/trace_gaps.py: 45 fills per C3 step, most of them a few words of
// counters, a third of a millisecond in all).  The stages collect what they have to clear (FillList,
// kernels.h) and clear it together.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace rala_hip {

namespace {

constexpr int kFillBlock = 256;
constexpr uint32_t kFillChunk = kFillBlock * 16 * 4;     // bytes per workgroup: four 16-byte stores per thread

struct FillArgs {
    uint8_t* p[FillList::kMost];
    uint64_t bytes[FillList::kMost];
    uint32_t first_block[FillList::kMost + 1];
    uint32_t value[FillList::kMost];
    uint32_t n;
};

__global__ __launch_bounds__(kFillBlock) void fill_list_kernel(FillArgs a) {
    uint32_t seg = 0;
    while (seg + 1 < a.n && blockIdx.x >= a.first_block[seg + 1]) ++seg;
    uint8_t* p = a.p[seg];
    const uint64_t bytes = a.bytes[seg];
    const uint32_t v = a.value[seg];
    const uint64_t lo = (uint64_t)(blockIdx.x - a.first_block[seg]) * kFillChunk;
    const uint64_t hi = lo + kFillChunk < bytes ? lo + kFillChunk : bytes;
    // (chunks start at multiples of 16 behind an aligned base; a base that is not gets byte stores)
    if (((uintptr_t)p & 15u) == 0) {
        const uint4 v4 = make_uint4(v, v, v, v);
        uint64_t at = lo + (uint64_t)threadIdx.x * 16;
#pragma unroll
        for (int k = 0; k < 4; ++k, at += kFillBlock * 16) {
            if (at + 16 <= hi) *(uint4*)(p + at) = v4;
            else if (at < hi) for (uint64_t b = at; b < hi; ++b) p[b] = (uint8_t)v;
        }
    } else if (((uintptr_t)p & 3u) == 0) {
        for (uint64_t at = lo + (uint64_t)threadIdx.x * 4; at < hi; at += kFillBlock * 4) {
            if (at + 4 <= hi) *(uint32_t*)(p + at) = v;
            else for (uint64_t b = at; b < hi; ++b) p[b] = (uint8_t)v;
        }
    } else {
        for (uint64_t at = lo + threadIdx.x; at < hi; at += kFillBlock) p[at] = (uint8_t)v;
    }
}

}  // namespace

void FillList::add(void* p, int byte, size_t n_bytes) {
    if (n_bytes == 0) return;
    if (n == kMost) overflow = true;
    else { ptr[n] = p; value[n] = (uint8_t)byte; bytes[n] = n_bytes; ++n; }
}

hipError_t FillList::launch(hipStream_t s) {
    if (overflow) return hipErrorInvalidValue;
    if (n == 0) return hipSuccess;
    FillArgs a;
    uint32_t blocks = 0;
    for (uint32_t k = 0; k < n; ++k) {
        a.p[k] = (uint8_t*)ptr[k];
        a.bytes[k] = bytes[k];
        a.value[k] = 0x01010101u * value[k];
        a.first_block[k] = blocks;
        blocks += (uint32_t)((bytes[k] + kFillChunk - 1) / kFillChunk);
    }
    a.first_block[n] = blocks;
    a.n = n;
    n = 0;
    hipLaunchKernelGGL(fill_list_kernel, dim3(blocks), dim3(kFillBlock), 0, s, a);
    return hipGetLastError();
}

}  // namespace rala_hip
