// The end of an in-order containment removal (graph.cpp:464-483, :831-866) written as a fixed point.
//
// Given: base[r] - the death of read r as far as it is decided for good (all ones: "never") - and a list of
// CONDITIONAL killers {key, target, keeper}: killer i deletes its target at position key_i unless its
// keeper was deleted before that position.  Wanted: the one X with
//     X[t] = min( base[t], min { key_i : target_i = t, X[keeper_i] > key_i } ).
// The map is antitone, so X_r = F(X_(r-1)) from X_0 = base gives upper and lower bounds in turn and settles
// (by induction over the keys: what happens below a position is exact after as many rounds as the longest
// chain of killers below it).  Both callers - the second overlap pass after its first two rounds, the tail's
// two scans - arrive here with a few thousand conditional killers and need five to seven rounds: ONE
// workgroup does them, with workgroup barriers between the rounds, instead of four launches and (every few
// rounds) a look from the host per round.
//
// Up to kLdsEntries killers: X lives in LDS.  A target's slot is the smallest index among the entries that
// target it, found through map[] (one word per read, all ones between calls) by two small kernels in front of
// the rounds: fixed_point_assign_kernel (atomic minimum of the entry index per target) and
// fixed_point_prepare_kernel, which gathers for every entry what the rounds need of it - key, the slots of
// target and keeper, base[] of both - into packed arrays.  (The one workgroup doing those gathers itself
// took 130 us for 11 k entries: values written with atomics are read from the memory side, and one compute
// unit has only so many requests in flight.)  A keeper that no listed killer targets is a constant,
// base[keeper].  Longer lists (C5: 73 k undecided killers after the second round of the second pass, 17 k conditional
// ones in the tail): fixed_point_wide_kernel - X in four work arrays in global memory, a few workgroups that are
// resident together, a barrier between them per round.  What other wavefronts write with atomics is then read past
// the vector cache, on this device from the memory side: 20 - 30 us per round instead of 2, whatever the list's
// length (one workgroup alone took 160 us per round for 24 k entries).
#include <atomic>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "kernels.h"

namespace rala_hip {

namespace {

constexpr uint32_t kInf = 0xFFFFFFFFu;
constexpr int kFinishBlock = 1024;
constexpr uint32_t kFinishPer = 12;                                  // entries a thread keeps in registers
constexpr uint32_t kLdsEntries = kFinishPer * kFinishBlock;          // 12 288: three arrays of that many words = 144 KB

__device__ __forceinline__ uint32_t ld_past_l1(const uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_past_l1(uint32_t* p, uint32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

constexpr int kPrepBlock = 256;
constexpr uint32_t kPrepGroups = 48;

__global__ __launch_bounds__(kPrepBlock) void fixed_point_assign_kernel(FixedPointList list, uint32_t* map) {
    const uint32_t n = *list.count;
    if (n > list.lds_limit) return;
    for (uint32_t i = blockIdx.x * kPrepBlock + threadIdx.x; i < n; i += gridDim.x * kPrepBlock) atomicMin(map + list.target[i], i);
}

// pack: six arrays of kLdsEntries words: key, slot of the target, slot of the keeper (all ones: none), base[keeper],
// base[target], target
__global__ __launch_bounds__(kPrepBlock) void fixed_point_prepare_kernel(FixedPointList list, const uint32_t* __restrict__ map,
                                                                         const uint32_t* __restrict__ base, uint32_t* __restrict__ pack) {
    const uint32_t n = *list.count;
    if (n > list.lds_limit) return;
    for (uint32_t i = blockIdx.x * kPrepBlock + threadIdx.x; i < n; i += gridDim.x * kPrepBlock) {
        const uint32_t t = list.target[i], kp = list.keeper[i];
        pack[i] = list.key[i];
        pack[kLdsEntries + i] = map[t];
        pack[2 * kLdsEntries + i] = map[kp];
        pack[3 * kLdsEntries + i] = base[kp];
        pack[4 * kLdsEntries + i] = base[t];
        pack[5 * kLdsEntries + i] = t;
    }
}

// (one workgroup; lds: 3 * kLdsEntries words)
__device__ __forceinline__ void finish_in_lds(uint32_t* lds, const FixedPointList& list, uint32_t* base, uint32_t* map,
                                              const uint32_t* __restrict__ pack, uint32_t* error, uint32_t* rounds_out) {
    const uint32_t n = *list.count;
    const uint32_t tid = threadIdx.x;
    if (n == 0) {
        if (rounds_out && tid == 0) *rounds_out = 0;
        return;
    }
    uint32_t r = 1;
    {
        constexpr uint32_t kPer = kFinishPer;
        uint32_t* xb = lds;                         // base per slot
        uint32_t* x0 = lds + kLdsEntries;           // X of the even rounds; the odd ones' behind it
        for (uint32_t k = tid; k < n; k += kFinishBlock) { xb[k] = kInf; x0[k] = kInf; x0[kLdsEntries + k] = kInf; }
        __syncthreads();
        uint32_t ek[kPer], st[kPer], sp[kPer], bp[kPer];
#pragma unroll
        for (uint32_t u = 0; u < kPer; ++u) {
            const uint32_t i = tid + u * kFinishBlock;
            const uint32_t at = i < n ? i : 0u;                     // (an absent entry: a copy of entry 0 that never proposes)
            ek[u] = i < n ? pack[at] : kInf;
            st[u] = pack[kLdsEntries + at]; sp[u] = pack[2 * kLdsEntries + at]; bp[u] = pack[3 * kLdsEntries + at];
            const uint32_t bt = pack[4 * kLdsEntries + at];
            xb[st[u]] = bt; x0[st[u]] = bt;
        }
        __syncthreads();
        uint32_t cur = 1;
        for (;; ++r, cur ^= 1u) {
            uint32_t* xc = x0 + cur * kLdsEntries;
            const uint32_t* xp = x0 + (cur ^ 1u) * kLdsEntries;
            for (uint32_t k = tid; k < n; k += kFinishBlock) xc[k] = xb[k];
            __syncthreads();
#pragma unroll
            for (uint32_t u = 0; u < kPer; ++u) {
                const uint32_t vk = sp[u] == kInf ? bp[u] : xp[sp[u]];
                if (vk > ek[u] && ek[u] != kInf) atomicMin(&xc[st[u]], ek[u]);
            }
            __syncthreads();
            bool moved = false;
            for (uint32_t k = tid; k < n; k += kFinishBlock) moved = moved || xc[k] != xp[k];
            if (!__syncthreads_or(moved ? 1 : 0)) break;             // X_r = X_(r-1): settled
            if (r > n + 8u) {                                         // (every round settles at least one read)
                if (tid == 0) *error = 1u;
                break;
            }
        }
        // the deaths; map[] all ones again
        for (uint32_t i = tid; i < n; i += kFinishBlock) {
            const uint32_t t = pack[5 * kLdsEntries + i];
            base[t] = x0[cur * kLdsEntries + pack[kLdsEntries + i]];
            map[t] = kInf;
        }
        if (rounds_out && tid == 0) *rounds_out = r;
        return;
    }
}

// Barrier between the workgroups of a small grid that is USUALLY resident as a whole (nothing guarantees it: ranks that
// share a device, other tenants, a partitioned device).  sync[0] counts arrivals (zero at the launch), `phase` the barriers
// of this launch.  false: somebody gave up waiting (a workgroup that never got a compute unit) - everybody leaves, nobody
// passes this barrier, and the last workgroup to leave does the rounds alone (fixed_point_wide_kernel).
//
// ONE word decides (with a separate "give up" word one workgroup could see the count complete and pass the
// last barrier while another one's poll limit ran out a poll earlier - the first wrote its share of base[] and returned,
// never counted as having left, and nobody ran the rounds alone): giving up is a compare-and-swap that sets the top bit of
// the arrival count WHILE THE COUNT IS SHORT of this barrier's; a workgroup passes iff it reads a complete count without
// that bit.  The count only grows, so a barrier that anybody has passed cannot be given up on afterwards, and one that
// was given up on is passed by nobody: all or nothing.  (A later barrier's giving up can fail a slow workgroup at an
// earlier one; then the fast ones fail at the later one and everything is redone alone - the last barrier of a launch
// has no later one.)  give_up_now: tests - this workgroup does not wait at all.
constexpr uint32_t kWideGroups = 32;
constexpr uint32_t kGaveUp = 0x80000000u;
// (Everything the workgroups tell each other goes through memory-side atomics, stores and loads: what has to be
// complete before the arrival is this wavefront's own requests - a wait for its counters, not a fence, which on
// this device writes the L2 back and invalidates it: 40 us per round with fences.)
__device__ __forceinline__ bool grid_barrier(uint32_t* sync, uint32_t& phase, uint32_t groups, bool give_up_now = false) {
    __shared__ uint32_t s_passed;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    ++phase;
    if (threadIdx.x == 0) {
        const uint32_t target = phase * groups;
        uint32_t v = __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        uint32_t spins = give_up_now ? (1u << 19) : 0u;
        while (!(v & kGaveUp) && v < target) {
            if (++spins > (1u << 19)) {
                // give up - unless the count has become complete meanwhile (then v says so and the loop ends with a pass)
                if (__hip_atomic_compare_exchange_strong(sync, &v, v | kGaveUp, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    v |= kGaveUp;
                }
                continue;           // (a failed exchange left the current value in v)
            }
            __builtin_amdgcn_s_sleep(1);
            v = __hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        s_passed = (v & kGaveUp) ? 0u : 1u;
    }
    __syncthreads();
    return s_passed != 0;
}

// More killers than the LDS holds: X_r in work[r % 4]: round r compares X_(r-1) with X_(r-2), proposes into X_r and
// resets the targets in the array of X_(r+1) (last read a round ago) - one barrier per round.  sync: eight zeroed
// words ([0] arrivals | "given up" in the top bit, [2] workgroups that have left, [4 .. 6] "a value moved" per round % 3).
// kSolo: one workgroup alone (the barrier is the workgroup's own).  false: a barrier gave up, nothing has been written to base.
template <bool kSolo>
__device__ __forceinline__ bool finish_wide(const FixedPointList& list, uint32_t* base, uint32_t* w0, uint32_t* w1, uint32_t* w2,
                                            uint32_t* w3, uint32_t* sync, uint32_t* error, uint32_t* rounds_out) {
    const uint32_t n = *list.count;
    constexpr uint32_t kPer = 4;
    const uint32_t groups = kSolo ? 1u : gridDim.x;
    const uint32_t stride = groups * kFinishBlock;
    const uint32_t gtid = kSolo ? threadIdx.x : blockIdx.x * kFinishBlock + threadIdx.x;
    auto barrier = [&](uint32_t& phase) -> bool {
        if constexpr (kSolo) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __syncthreads();
            return true;
        } else {
            // (tests, debug_give_up: 1 - workgroup 1 does not wait at the first barrier, so nobody passes it; 2 - it does not
            // wait at any barrier from the second on: it passes those the others have reached and ends the others for everybody)
            const bool impatient = blockIdx.x == 1 && (list.debug_give_up == 1 ? phase == 0 : list.debug_give_up == 2 && phase >= 1);
            return grid_barrier(sync, phase, groups, impatient);
        }
    };
    uint32_t* const work[4] = {w0, w1, w2, w3};
    uint32_t* const flag = sync + 4;
    uint32_t phase = 0;
    uint32_t ek[kPer], et[kPer], ep[kPer], eb[kPer];
    auto load = [&](uint32_t i0) {                // entries i0 + u * stride; absent ones: copies of entry 0 with the key "never"
#pragma unroll
        for (uint32_t u = 0; u < kPer; ++u) {
            const uint32_t i = i0 + u * stride;
            const uint32_t at = i < n ? i : 0u;
            ek[u] = i < n ? list.key[at] : kInf;
            et[u] = list.target[at]; ep[u] = list.keeper[at];
        }
#pragma unroll
        for (uint32_t u = 0; u < kPer; ++u) eb[u] = base[et[u]];
    };
    for (uint32_t i0 = gtid; i0 < n; i0 += kPer * stride) {
        load(i0);
#pragma unroll
        for (uint32_t u = 0; u < kPer; ++u) {
            const uint32_t bp = base[ep[u]];
            for (int w = 0; w < 4; ++w) { st_past_l1(work[w] + et[u], eb[u]); st_past_l1(work[w] + ep[u], bp); }
        }
    }
    if (!barrier(phase)) return false;
    // (up to 131 072 entries stay in registers over the rounds: one trip to memory less per round)
    const bool resident = n <= kPer * stride;
    if (resident) load(gtid);
    uint32_t r = 1;
    for (;; ++r) {
        const uint32_t* prev2 = work[(r + 2) & 3];      // X_(r-2)
        const uint32_t* prev = work[(r + 3) & 3];       // X_(r-1)
        uint32_t* cur = work[r & 3];                    // X_r
        uint32_t* next = work[(r + 1) & 3];             // X_(r+1): reset here
        bool moved = false;
        for (uint32_t i0 = gtid; i0 < n; i0 += kPer * stride) {
            if (!resident) load(i0);
            uint32_t vk[kPer], v1[kPer], v2[kPer];
#pragma unroll
            for (uint32_t u = 0; u < kPer; ++u) { vk[u] = ld_past_l1(prev + ep[u]); v1[u] = ld_past_l1(prev + et[u]); v2[u] = ld_past_l1(prev2 + et[u]); }
#pragma unroll
            for (uint32_t u = 0; u < kPer; ++u) {
                moved = moved || (r >= 2 && v1[u] != v2[u]);
                if (vk[u] > ek[u] && ek[u] != kInf) atomicMin(cur + et[u], ek[u]);
                st_past_l1(next + et[u], eb[u]);
            }
        }
        if (r >= 2 && __syncthreads_or(moved ? 1 : 0) && threadIdx.x == 0) {
            __hip_atomic_store(flag + r % 3u, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (gtid == 0) __hip_atomic_store(flag + (r + 1u) % 3u, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!barrier(phase)) return false;
        if (r >= 2 && ld_past_l1(flag + r % 3u) == 0) break;     // X_(r-1) = X_(r-2): settled (and X_r is the same again)
        if (r > n + 8u) {                                         // (every round settles at least one read)
            if (gtid == 0) *error = 1u;
            return true;
        }
    }
    const uint32_t* settled = work[(r + 3) & 3];
    for (uint32_t i0 = gtid; i0 < n; i0 += kPer * stride) {
        if (!resident) load(i0);
#pragma unroll
        for (uint32_t u = 0; u < kPer; ++u) base[et[u]] = ld_past_l1(settled + et[u]);
    }
    if (rounds_out && gtid == 0) *rounds_out = r - 1;
    return true;
}

// The rounds: a list that fits the LDS is one workgroup's, a longer one the resident workgroups'.  Two kernels, both
// launched (the list's length is on the device; the one it is not meant for leaves at once): the second needs no LDS, and a
// kernel that asked for 144 KB of it for every workgroup would need a compute unit per workgroup - eight ranks sharing a GPU
// (the tests do that) would then need every compute unit of the device at the same time to get through their barriers.
__global__ __launch_bounds__(kFinishBlock) void fixed_point_finish_kernel(FixedPointList list, uint32_t* base, uint32_t* map,
                                                                          const uint32_t* __restrict__ pack, uint32_t* error,
                                                                          uint32_t* rounds_out) {
    extern __shared__ uint32_t lds[];
    if (*list.count <= list.lds_limit) finish_in_lds(lds, list, base, map, pack, error, rounds_out);
}

__global__ __launch_bounds__(kFinishBlock) void fixed_point_wide_kernel(FixedPointList list, uint32_t* base, uint32_t* w0, uint32_t* w1,
                                                                        uint32_t* w2, uint32_t* w3, uint32_t* sync, uint32_t* error,
                                                                        uint32_t* rounds_out) {
    if (*list.count <= list.lds_limit) return;
    if (finish_wide<false>(list, base, w0, w1, w2, w3, sync, error, rounds_out)) return;
    // The workgroups could not meet (one of them waited 2^19 polls for a compute unit): every workgroup comes by sooner or
    // later and leaves; the last one to leave does all the rounds alone - slower (C5: 1.1 ms instead of 0.36), never stuck.
    __shared__ uint32_t last;
    if (threadIdx.x == 0) last = __hip_atomic_fetch_add(sync + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (last) (void)finish_wide<true>(list, base, w0, w1, w2, w3, sync, error, rounds_out);
}

}  // namespace

size_t fixed_point_pack_words() { return 6 * (size_t)kLdsEntries; }

hipError_t launch_fixed_point_finish(const FixedPointList& list_in, uint32_t* base, uint32_t* map, uint32_t* pack, uint32_t* const work[4],
                                     uint32_t* sync8, uint32_t* error, uint32_t* rounds_out, hipStream_t s) {
    FixedPointList list = list_in;
    list.lds_limit = std::min<uint32_t>(list_in.lds_limit, kLdsEntries);
    if (const char* g = getenv("RALA_HIP_DEBUG_FP_GIVE_UP")) list.debug_give_up = atoi(g) == 2 ? 2u : 1u;     // tests: the long lists' workgroups do not meet
    constexpr size_t lds_bytes = 3 * (size_t)kLdsEntries * 4;
    // (once per DEVICE: the attribute belongs to the function on the current device, and the ranks of a sharded run are
    // threads of one process on different devices.  Every time, it was part of why the host fell behind the device in
    // this chain of 4-microsecond kernels: 8 - 19 us of queue idle time in front of each of them)
    {
        static std::atomic<uint64_t> set_on{0};
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        const uint64_t bit = 1ull << (dev & 63);
        if (!(set_on.load(std::memory_order_acquire) & bit)) {
            e = hipFuncSetAttribute((const void*)fixed_point_finish_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            if (e != hipSuccess) return e;
            set_on.fetch_or(bit, std::memory_order_release);
        }
    }
    // (the list's length is on the device: the kernels look at it)
    hipLaunchKernelGGL(fixed_point_assign_kernel, dim3(kPrepGroups), dim3(kPrepBlock), 0, s, list, map);
    hipLaunchKernelGGL(fixed_point_prepare_kernel, dim3(kPrepGroups), dim3(kPrepBlock), 0, s, list, (const uint32_t*)map,
                       (const uint32_t*)base, pack);
    hipLaunchKernelGGL(fixed_point_finish_kernel, dim3(1), dim3(kFinishBlock), lds_bytes, s, list, base, map, (const uint32_t*)pack,
                       error, rounds_out);
    hipLaunchKernelGGL(fixed_point_wide_kernel, dim3(kWideGroups), dim3(kFinishBlock), 0, s, list, base, work[0], work[1], work[2],
                       work[3], sync8, error, rounds_out);
    return hipGetLastError();
}

}  // namespace rala_hip
