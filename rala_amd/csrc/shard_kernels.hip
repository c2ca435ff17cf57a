// Small kernels of the sharded run: bound tuples partitioned by owner rank, per-read arrays moved
// between the global numbering (read r) and an owner's local numbering (read j * P + rank).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "device_utils.h"
#include "kernels.h"

namespace rala_hip {

namespace {

constexpr int kBlock = 256;
constexpr uint32_t kInf = 0xFFFFFFFFu;
constexpr uint32_t kPartChunk = 2048;

// Tuples {read, bound} grouped by the owner of the read (read % world; the stored read becomes
// the owner's local number read / world); tuples whose read is ~0u are left out.  A workgroup
// owns a chunk of 2048 tuples: bucket sizes in LDS; pass 0 adds them to the global per-owner
// counters, pass 1 reserves the chunk's range of every bucket with one atomic per owner and
// scatters.  The order inside a bucket is irrelevant (coverage is additive).
__global__ __launch_bounds__(kBlock) void partition_tuples_kernel(const uint2* __restrict__ in, uint64_t n, uint32_t world,
                                                                  uint32_t pass, uint32_t* counters,
                                                                  uint2* __restrict__ out) {
    __shared__ uint32_t s_cnt[64], s_base[64];
    constexpr uint32_t kPer = kPartChunk / kBlock;
    if (threadIdx.x < 64) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    uint2 t[kPer];
    uint32_t slot[kPer];
#pragma unroll
    for (uint32_t u = 0; u < kPer; ++u) {
        const uint64_t i = (uint64_t)blockIdx.x * kPartChunk + u * kBlock + threadIdx.x;
        t[u] = i < n ? in[i] : make_uint2(kInf, 0u);
        slot[u] = t[u].x != kInf ? atomicAdd(&s_cnt[t[u].x % world], 1u) : 0u;
    }
    __syncthreads();
    if (pass == 0) {
        if (threadIdx.x < world && s_cnt[threadIdx.x]) atomicAdd(&counters[threadIdx.x], s_cnt[threadIdx.x]);
        return;
    }
    if (threadIdx.x < world) s_base[threadIdx.x] = s_cnt[threadIdx.x] ? atomicAdd(&counters[threadIdx.x], s_cnt[threadIdx.x]) : 0u;
    __syncthreads();
#pragma unroll
    for (uint32_t u = 0; u < kPer; ++u) {
        if (t[u].x == kInf) continue;
        const uint32_t p = t[u].x % world;
        out[s_base[p] + slot[u]] = make_uint2(t[u].x / world, t[u].y);
    }
}

template <class T>
__global__ __launch_bounds__(kBlock) void localize_kernel(const T* __restrict__ global, uint64_t n_local, uint32_t world,
                                                          uint32_t rank, T* __restrict__ local) {
    const uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j < n_local) local[j] = global[j * world + rank];
}

// medians: one word per local read, p10 << 16 | median
__global__ __launch_bounds__(kBlock) void pack_median_kernel(const uint16_t* median, const uint16_t* p10, uint64_t n_local,
                                                             uint64_t nl_pad, uint32_t* out) {
    const uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j < nl_pad) out[j] = j < n_local ? ((uint32_t)p10[j] << 16) | median[j] : 0u;
}
__global__ __launch_bounds__(kBlock) void unpack_median_kernel(const uint32_t* all, uint32_t world, uint64_t nl_pad,
                                                               uint64_t n_reads, uint16_t* median, uint16_t* p10) {
    const uint64_t r = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (r >= n_reads) return;
    const uint32_t w = all[(r % world) * nl_pad + r / world];
    median[r] = (uint16_t)w;
    p10[r] = (uint16_t)(w >> 16);
}

// repeat hills: one 8-byte word per local read, slot << 32 | count
__global__ __launch_bounds__(kBlock) void pack_rep_kernel(const uint32_t* n_rep, const uint32_t* rep_slot, uint64_t n_local,
                                                          uint64_t nl_pad, uint64_t* out) {
    const uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j < nl_pad) out[j] = j < n_local && n_rep[j] ? ((uint64_t)rep_slot[j] << 32) | n_rep[j] : 0ull;
}
__global__ __launch_bounds__(kBlock) void unpack_rep_kernel(const uint64_t* all, uint32_t world, uint64_t nl_pad,
                                                            uint64_t n_reads, RankOffsets pool_base, uint32_t* n_rep,
                                                            uint32_t* rep_slot) {
    const uint64_t r = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (r >= n_reads) return;
    const uint32_t k = (uint32_t)(r % world);
    const uint64_t w = all[k * nl_pad + r / world];
    n_rep[r] = (uint32_t)w;
    rep_slot[r] = (uint32_t)(w >> 32) + pool_base.v[k];
}

// flags of the repeat hills (Interval::aux): mode 0 pool -> dense, mode 1 dense -> pool
__global__ __launch_bounds__(kBlock) void pool_aux_kernel(Interval* pool, uint32_t n, uint32_t* dense, uint32_t mode) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    if (mode == 0) dense[i] = pool[i].aux;
    else pool[i].aux = dense[i];
}

inline dim3 grid_for(uint64_t n) { return dim3((unsigned)((n + kBlock - 1) / kBlock)); }

// A rank's undecided killers as one block of 3 * each words (key, target, keeper; each = the longest list of any rank, which
// every rank knows from the status word): entries behind the rank's own are inert - the key "never" proposes nothing.  No
// exchange of the lists' lengths is needed to gather blocks of one size.
__global__ __launch_bounds__(kBlock) void pack_killers_kernel(const uint32_t* __restrict__ key, const uint32_t* __restrict__ target,
                                                              const uint32_t* __restrict__ keeper, const uint32_t* __restrict__ count,
                                                              uint32_t each, uint32_t* __restrict__ block) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= each) return;
    const bool mine = i < *count;
    block[i] = mine ? key[i] : kInf;
    block[each + i] = mine ? target[i] : 0u;
    block[2 * each + i] = mine ? keeper[i] : 0u;
}
__global__ __launch_bounds__(kBlock) void unpack_killers_kernel(const uint32_t* __restrict__ blocks, uint32_t world, uint32_t each,
                                                                uint32_t* __restrict__ key, uint32_t* __restrict__ target,
                                                                uint32_t* __restrict__ keeper, uint32_t* count) {
    const uint32_t j = blockIdx.x * kBlock + threadIdx.x;
    if (j == 0) *count = world * each;
    if (j >= world * each) return;
    const uint32_t* b = blocks + (size_t)(j / each) * 3 * each;
    const uint32_t i = j % each;
    key[j] = b[i]; target[j] = b[each + i]; keeper[j] = b[2 * each + i];
}

}  // namespace

void launch_pack_killers(const uint32_t* key, const uint32_t* target, const uint32_t* keeper, const uint32_t* count, uint32_t each,
                         uint32_t* block, hipStream_t s) {
    if (each) hipLaunchKernelGGL(pack_killers_kernel, grid_for(each), dim3(kBlock), 0, s, key, target, keeper, count, each, block);
}
void launch_unpack_killers(const uint32_t* blocks, uint32_t world, uint32_t each, uint32_t* key, uint32_t* target, uint32_t* keeper,
                           uint32_t* count, hipStream_t s) {
    hipLaunchKernelGGL(unpack_killers_kernel, grid_for(std::max<uint64_t>(1, (uint64_t)world * each)), dim3(kBlock), 0, s, blocks, world, each,
                       key, target, keeper, count);
}

void launch_partition_tuples(const uint2* in, uint64_t n, uint32_t world, uint32_t pass, uint32_t* counters, uint2* out,
                             hipStream_t s) {
    if (n) {
        hipLaunchKernelGGL(partition_tuples_kernel, dim3((unsigned)((n + kPartChunk - 1) / kPartChunk)), dim3(kBlock), 0, s,
                           in, n, world, pass, counters, out);
    }
}
void launch_localize_u32(const uint32_t* global, uint64_t n_local, uint32_t world, uint32_t rank, uint32_t* local, hipStream_t s) {
    if (n_local) hipLaunchKernelGGL(localize_kernel<uint32_t>, grid_for(n_local), dim3(kBlock), 0, s, global, n_local, world, rank, local);
}
void launch_localize_u16(const uint16_t* global, uint64_t n_local, uint32_t world, uint32_t rank, uint16_t* local, hipStream_t s) {
    if (n_local) hipLaunchKernelGGL(localize_kernel<uint16_t>, grid_for(n_local), dim3(kBlock), 0, s, global, n_local, world, rank, local);
}
void launch_localize_u8(const uint8_t* global, uint64_t n_local, uint32_t world, uint32_t rank, uint8_t* local, hipStream_t s) {
    if (n_local) hipLaunchKernelGGL(localize_kernel<uint8_t>, grid_for(n_local), dim3(kBlock), 0, s, global, n_local, world, rank, local);
}
void launch_pack_median(const uint16_t* median, const uint16_t* p10, uint64_t n_local, uint64_t nl_pad, uint32_t* out, hipStream_t s) {
    if (nl_pad) hipLaunchKernelGGL(pack_median_kernel, grid_for(nl_pad), dim3(kBlock), 0, s, median, p10, n_local, nl_pad, out);
}
void launch_unpack_median(const uint32_t* all, uint32_t world, uint64_t nl_pad, uint64_t n_reads, uint16_t* median, uint16_t* p10,
                          hipStream_t s) {
    if (n_reads) hipLaunchKernelGGL(unpack_median_kernel, grid_for(n_reads), dim3(kBlock), 0, s, all, world, nl_pad, n_reads, median, p10);
}
void launch_pack_rep(const uint32_t* n_rep, const uint32_t* rep_slot, uint64_t n_local, uint64_t nl_pad, uint64_t* out, hipStream_t s) {
    if (nl_pad) hipLaunchKernelGGL(pack_rep_kernel, grid_for(nl_pad), dim3(kBlock), 0, s, n_rep, rep_slot, n_local, nl_pad, out);
}
void launch_unpack_rep(const uint64_t* all, uint32_t world, uint64_t nl_pad, uint64_t n_reads, const RankOffsets& pool_base,
                       uint32_t* n_rep, uint32_t* rep_slot, hipStream_t s) {
    if (n_reads) {
        hipLaunchKernelGGL(unpack_rep_kernel, grid_for(n_reads), dim3(kBlock), 0, s, all, world, nl_pad, n_reads, pool_base, n_rep,
                           rep_slot);
    }
}
void launch_pool_aux(Interval* pool, uint32_t n, uint32_t* dense, uint32_t mode, hipStream_t s) {
    if (n) hipLaunchKernelGGL(pool_aux_kernel, grid_for(n), dim3(kBlock), 0, s, pool, n, dense, mode);
}

}  // namespace rala_hip
