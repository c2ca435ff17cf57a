// Sharded (multi-GPU) run of the hot path: C ABI rala_hip_mg_* (include/rala_hip.h).
//
// One rala_hip_mg object per rank (= per GPU).  Reads are the independent units: read r is owned
// by rank r % P and is local read r / P there.  The overlap file is cut into P contiguous slices
// on a_id-run boundaries (duplicate removal is per run, reference graph.cpp:346-350); rank k
// holds slice k and ALL read lengths.  One step (rala_hip_mg_run, collective):
//
//   slice   duplicate removal (beside:) the bounds of both sides of every overlap as 8-byte records
//           {local read & 4095, begin, end}, scattered ONCE by (owner, partition of 4096 of the owner's
//           reads) - the first level of the partitioned bucketing done where the overlaps are         O(N / P)
//   comm    ONE all-to-all(v): an owner's block = its groups' record counts + its records; a rank's own
//           block stays where it is
//   owner   second level of the bucketing over the blocks as they lie, rows, then build + annotate the
//           piles of the owned reads                                                  O(bases / P)
//           (reads of 2^25 bases and more, or option use_fused_emit = 0: bound records / tuples grouped by
//           owner, bucketed from the start by the owner - rounds 2 - 4)
//   comm    all-gather of the per-read state (19 B per read) and the interval pools
//   slice   classify (trim / type) against the gathered state                         O(N / P)
//   comm    containment fixed point: bounds all-reduced (min) per round; once few killers are
//           undecided they are gathered and the rest runs without collectives
//   slice   liveness, hill counters (all-reduce sum), survivors                       O(N / P)
//   comm    all-gather of the survivor lists (about 1 % of the overlaps)
//   all     preprocess tail, graph, transitive reduction on the survivors             replicated
//
// With sensitive overlaps (-s, graph.cpp:882-1054) the second add_layers and the repeat hills run
// on the owners as well (tuples by target owner, medians and hills gathered back).
//
// Two contexts per rank: `cs` (all reads, the slice's overlaps; ends up with the replicated
// result, so every getter of rala_hip.h works on it) and `cl` (the owned reads with their piles).
#include <hip/hip_runtime.h>
#include <string.h>
#include <sys/stat.h>

#include <algorithm>
#include <chrono>
#include <thread>

#include "comm.h"
#include "context.h"
#include "stages.h"

using namespace rala_hip;

struct rala_hip_mg {
    int device = 0;
    uint32_t rank = 0, world = 1;
    Comm* comm = nullptr;
    rala_hip_ctx* cs = nullptr;
    rala_hip_ctx* cl = nullptr;
    uint64_t n_reads = 0, n_local = 0, nl_pad = 0;
    bool have_reads = false, have_overlaps = false;
    bool verdict_shared = false;        // the failure this run returns is known to every rank (no abort needed)
    std::string err;
    DevBuf<uint2> d_send, d_recv;
    DevBuf<uint8_t> d_state_mine, d_state_all;
    DevBuf<uint8_t> d_bytes[2];
    rala_hip_mg_timings tm = {};
    hipEvent_t ev[8] = {};              // stage boundaries on cs->stream (timings without a wait)
};

namespace {

constexpr uint32_t kNoSlot = 0xFFFFFFFFu;
constexpr int kBlock = 256;

double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

uint64_t local_count(uint64_t n_reads, uint32_t rank, uint32_t world) {
    return n_reads > rank ? (n_reads - rank + world - 1) / world : 0;
}

// ---- packed per-read state of one rank: field arrays of nl entries back to back ------------
struct StateLayout {
    uint64_t nl;
    __host__ __device__ uint64_t begin() const { return 0; }
    __host__ __device__ uint64_t end() const { return 4 * nl; }
    __host__ __device__ uint64_t slot() const { return 8 * nl; }
    __host__ __device__ uint64_t n_pits() const { return 12 * nl; }
    __host__ __device__ uint64_t n_hills() const { return 16 * nl; }
    __host__ __device__ uint64_t median() const { return 20 * nl; }
    __host__ __device__ uint64_t p10() const { return 22 * nl; }
    __host__ __device__ uint64_t alive() const { return 24 * nl; }
    __host__ __device__ uint64_t bytes() const { return 25 * nl; }
};

struct ReadArrays {
    uint32_t *begin, *end, *slot;
    uint16_t *median, *p10;
    uint8_t* alive;
    uint32_t *n_pits, *n_hills;
};

ReadArrays read_arrays(rala_hip_ctx* c) {
    return ReadArrays{c->d_begin.p, c->d_end.p, c->d_iv_slot.p, c->d_median.p, c->d_p10.p,
                      c->d_alive.p, c->d_n_pits.p, c->d_n_hills.p};
}

__global__ __launch_bounds__(kBlock) void pack_state_kernel(ReadArrays a, uint64_t n_local, StateLayout L, uint8_t* out) {
    const uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= L.nl) return;
    const bool in = j < n_local;
    ((uint32_t*)(out + L.begin()))[j] = in ? a.begin[j] : 0u;
    ((uint32_t*)(out + L.end()))[j] = in ? a.end[j] : 0u;
    ((uint32_t*)(out + L.slot()))[j] = in ? a.slot[j] : kNoSlot;
    ((uint16_t*)(out + L.median()))[j] = in ? a.median[j] : (uint16_t)0;
    ((uint16_t*)(out + L.p10()))[j] = in ? a.p10[j] : (uint16_t)0;
    out[L.alive() + j] = in ? a.alive[j] : (uint8_t)0;
    ((uint32_t*)(out + L.n_pits()))[j] = in ? a.n_pits[j] : 0u;
    ((uint32_t*)(out + L.n_hills()))[j] = in ? a.n_hills[j] : 0u;
}

struct RankTable {
    uint32_t v[64];
};

// global read r = j * world + k  <-  entry j of rank k's block; pool slots rebased onto the
// concatenation of the ranks' interval pools
__global__ __launch_bounds__(kBlock) void unpack_state_kernel(const uint8_t* all, StateLayout L, uint32_t world,
                                                              uint64_t n_reads, RankTable pool_base, ReadArrays a) {
    const uint64_t r = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (r >= n_reads) return;
    const uint32_t k = (uint32_t)(r % world);
    const uint64_t j = r / world;
    const uint8_t* in = all + (uint64_t)k * L.bytes();
    a.begin[r] = ((const uint32_t*)(in + L.begin()))[j];
    a.end[r] = ((const uint32_t*)(in + L.end()))[j];
    const uint32_t s = ((const uint32_t*)(in + L.slot()))[j];
    a.slot[r] = s == kNoSlot ? kNoSlot : s + pool_base.v[k];
    a.median[r] = ((const uint16_t*)(in + L.median()))[j];
    a.p10[r] = ((const uint16_t*)(in + L.p10()))[j];
    a.alive[r] = in[L.alive() + j];
    a.n_pits[r] = ((const uint32_t*)(in + L.n_pits()))[j];
    a.n_hills[r] = ((const uint32_t*)(in + L.n_hills()))[j];
}


// ---- a rank's byte range of the PAF file, tokenised on its own GPU (rala_hip_mg_set_overlaps_from_paf) ------------------
// what the cuts between the ranks' rows need to know of a rank's rows: ends[0] = its first resolved row (query and target
// known; all ones: none), ends[1] = one behind its last, ends[2] = the first resolved row behind ends[0] whose query differs
// from that row's (n: none) - where the run that may have begun on the rank before ends
constexpr uint32_t kNoRow = 0xFFFFFFFFu;
__global__ __launch_bounds__(kBlock) void run_ends_kernel(const uint32_t* __restrict__ a_id, const uint32_t* __restrict__ b_id, uint32_t n,
                                                          uint32_t n_reads, uint32_t* ends) {
    uint32_t lo = kNoRow, hi = 0;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        if (a_id[i] < n_reads && b_id[i] < n_reads) { lo = lo < i ? lo : i; hi = i + 1; }
    }
    for (int d = 32; d; d >>= 1) {
        const uint32_t l2 = (uint32_t)__shfl_xor((int)lo, d, 64), h2 = (uint32_t)__shfl_xor((int)hi, d, 64);
        lo = lo < l2 ? lo : l2; hi = hi > h2 ? hi : h2;
    }
    if ((threadIdx.x & 63) == 0) {
        if (lo != kNoRow) atomicMin(&ends[0], lo);
        if (hi) atomicMax(&ends[1], hi);
    }
}
__global__ __launch_bounds__(kBlock) void head_run_kernel(const uint32_t* __restrict__ a_id, const uint32_t* __restrict__ b_id, uint32_t n,
                                                          uint32_t n_reads, uint32_t* ends) {
    const uint32_t first = ends[0];
    if (first == kNoRow) return;
    const uint32_t head = a_id[first];
    uint32_t lo = kNoRow;
    for (uint32_t i = first + 1 + blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        if (a_id[i] < n_reads && b_id[i] < n_reads && a_id[i] != head) { lo = i; break; }
    }
    for (int d = 32; d; d >>= 1) {
        const uint32_t l2 = (uint32_t)__shfl_xor((int)lo, d, 64);
        lo = lo < l2 ? lo : l2;
    }
    if ((threadIdx.x & 63) == 0 && lo != kNoRow) atomicMin(&ends[2], lo);
}
// ends[3], ends[4] = the queries of the first and the last resolved row
__global__ void run_ends_names_kernel(const uint32_t* __restrict__ a_id, uint32_t* ends) {
    ends[3] = ends[0] == kNoRow ? kNoRow : a_id[ends[0]];
    ends[4] = ends[1] == 0 ? kNoRow : a_id[ends[1] - 1];
}

// a column with room for `need` entries that keeps its first `keep`
template <class T>
hipError_t grow_keep(DevBuf<T>& b, size_t keep, size_t need, hipStream_t s) {
    if (need <= b.n && b.p) return hipSuccess;
    DevBuf<T> larger;
    hipError_t e = larger.ensure(need + need / 8);
    if (e != hipSuccess) return e;
    if (keep) e = hipMemcpyAsync(larger.p, b.p, keep * sizeof(T), hipMemcpyDeviceToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return e;
    std::swap(larger.p, b.p);
    std::swap(larger.n, b.n);
    return hipSuccess;
}

#define MGCHECK(call)                                                                    \
    do {                                                                                 \
        hipError_t e_ = (call);                                                          \
        if (e_ != hipSuccess) {                                                          \
            mg->err = std::string(#call) + ": " + hipGetErrorString(e_);                 \
            return e_ == hipErrorOutOfMemory ? RALA_HIP_ENOMEM : RALA_HIP_EDEVICE;       \
        }                                                                                \
    } while (0)

int mg_fail(rala_hip_mg* mg, int code, const std::string& msg) {
    mg->err = msg;
    return code;
}

int from_ctx(rala_hip_mg* mg, rala_hip_ctx* c, int rc, const char* where) {
    if (rc != RALA_HIP_OK) mg->err = std::string(where) + ": " + rala_hip_last_error(c);
    return rc;
}

int from_comm(rala_hip_mg* mg, int rc, const char* where) {
    if (rc == 0) return RALA_HIP_OK;
    mg->err = std::string(where) + ": " + mg->comm->error();
    return RALA_HIP_EDEVICE;
}

// Every rank learns whether any rank failed (otherwise the others would wait in the next collective for
// ever) - riding on a host exchange the step needs anyway: every rank contributes `n` words of its own
// behind a status word {code of its failure | flag << 32}; `flag` must be the same on every rank (with /
// without sensitive overlaps).  all = world * (n + 1) words.  Returns this rank's code, or
// RALA_HIP_EDEVICE if only another one failed.
int agree_with(rala_hip_mg* mg, int rc, const char* where, uint32_t flag, const uint64_t* mine, uint32_t n,
               std::vector<uint64_t>& all) {
    std::vector<uint64_t> out(n + 1);
    out[0] = (uint64_t)(uint32_t)(-rc) | (uint64_t)flag << 32;
    for (uint32_t k = 0; k < n; ++k) out[k + 1] = mine ? mine[k] : 0;
    all.assign((size_t)mg->world * (n + 1), 0);
    if (mg->comm->host_all_gather(out.data(), n + 1, all.data(), mg->cs->stream) != 0) {
        if (rc == RALA_HIP_OK) return from_comm(mg, -1, where);
        return rc;
    }
    mg->verdict_shared = true;          // whatever is returned below, every rank returns a failure too
    if (rc != RALA_HIP_OK) return rc;
    for (uint32_t p = 0; p < mg->world; ++p) {
        const uint64_t code = all[(size_t)p * (n + 1)] & 0xFFFFFFFFull;
        if (code != 0) {
            return mg_fail(mg, code == (uint64_t)(-RALA_HIP_EFILTERED) ? RALA_HIP_EFILTERED : RALA_HIP_EDEVICE,
                           std::string(where) + ": rank " + std::to_string(p) + " failed with code -" + std::to_string(code));
        }
    }
    for (uint32_t p = 0; p < mg->world; ++p) {
        const uint64_t theirs = all[(size_t)p * (n + 1)] >> 32;
        if (theirs != flag) {
            // bit 0: a sensitive pass; bit 1: bounds as 8-byte records (option use_bound_records)
            const char* what = ((theirs ^ flag) & 1) ? " disagrees on whether there are sensitive overlaps"
                                                     : " disagrees on the format of the bounds (option use_bound_records)";
            return mg_fail(mg, RALA_HIP_EINVAL, std::string(where) + ": rank " + std::to_string(p) + what);
        }
    }
    mg->verdict_shared = false;
    return RALA_HIP_OK;
}

int run_primary(rala_hip_mg* mg, bool with_sens) {
    rala_hip_ctx* cs = mg->cs;
    rala_hip_ctx* cl = mg->cl;
    Comm* comm = mg->comm;
    const uint32_t P = mg->world;
    hipStream_t s = cs->stream;
    // Stage boundaries are events on the slice context's stream, read at the end of the run: nobody waits for a stage
    // to end just to look at the clock.  (Rounds 1 - 4 synchronised both streams after every stage.)
    int n_marks = 0;
    auto mark = [&]() { if (n_marks < 8) (void)hipEventRecord(mg->ev[n_marks++], s); };
    mark();
    std::vector<uint64_t> all;

    // 1. slice: duplicates, owner-grouped bounds.  Scattered once by (owner, partition) where the reads suit the
    // partitioned bucketing (shard_path_fits: the same answer on every rank, checked with the status word); otherwise as
    // bound records {local read, begin, end} (8 bytes per overlap side) where the reads are short and few enough for that
    // format (rala_hip_bound_records_fit), as tuples {local read, bound} (two per side) in the last resort
    const bool fused = cs->use_fused_emit && cs->use_bound_records && shard_path_fits(mg->n_reads, cs->max_read_len, P);
    const bool records = !fused && cs->use_bound_records && rala_hip_bound_records_fit(cs, P) != 0;
    const ShardGeometry geom = shard_geometry(mg->n_reads, P);
    int rc = RALA_HIP_OK;
    std::vector<uint64_t> send_counts(P, 0);
    if (fused) {
        if (mg->d_send.ensure(shard_send_words(geom, cs->n_ovl)) != hipSuccess) rc = mg_fail(mg, RALA_HIP_ENOMEM, "send buffer");
        else rc = from_ctx(mg, cs, shard_emit(cs, geom, (uint64_t*)mg->d_send.p, send_counts.data()), "emit blocks");
    } else {
        rc = from_ctx(mg, cs, rala_hip_dedupe(cs), "dedupe");
        if (rc == RALA_HIP_OK) {
            if (mg->d_send.ensure(4 * std::max<uint64_t>(cs->n_ovl, 1) + 8) != hipSuccess) rc = mg_fail(mg, RALA_HIP_ENOMEM, "tuple buffer");
            else if (records) rc = from_ctx(mg, cs, rala_hip_emit_bound_records_bucketed(cs, P, (uint64_t*)mg->d_send.p, send_counts.data()), "emit records");
            else rc = from_ctx(mg, cs, rala_hip_emit_bound_tuples_bucketed(cs, P, (uint64_t*)mg->d_send.p, send_counts.data()), "emit tuples");
        }
    }
    // (the bucket sizes travel with the status: one host exchange)
    rc = agree_with(mg, rc, "emit", (with_sens ? 1u : 0u) | (records ? 2u : 0u) | (fused ? 4u : 0u), send_counts.data(), P, all);
    if (rc != RALA_HIP_OK) return rc;
    mark();

    // 2. ONE all-to-all(v) of 8-byte elements (the own part of the blocks is not moved)
    std::vector<uint64_t> recv_counts(P);
    uint64_t n_recv = 0;
    for (uint32_t p = 0; p < P; ++p) {
        recv_counts[p] = all[(size_t)p * (P + 1) + 1 + mg->rank];
        if (!(fused && p == mg->rank)) n_recv += recv_counts[p];
    }
    MGCHECK(mg->d_recv.ensure(n_recv + 8));
    rc = from_comm(mg, comm->all_to_all_v(mg->d_send.p, send_counts.data(), mg->d_recv.p, recv_counts.data(), sizeof(uint2), s, !fused),
                   "all-to-all of the bounds");
    if (rc != RALA_HIP_OK) return rc;
    mark();
    mg->tm.tuples_sent = 0;
    for (uint32_t p = 0; p < P; ++p) if (p != mg->rank) mg->tm.tuples_sent += send_counts[p];

    // 3. owner: piles of the reads this rank owns (its stream starts behind the exchange)
    MGCHECK(hipEventRecord(mg->ev[7], s));
    MGCHECK(hipStreamWaitEvent(cl->stream, mg->ev[7], 0));
    if (fused) {
        ShardBlocks blocks;
        uint64_t at = 0, mine_at = 0, n_records = 0, words = 0;
        for (uint32_t p = 0; p < P; ++p) {
            if (p < mg->rank) mine_at += send_counts[p];
            blocks.off[p] = (uint32_t)at;
            if (recv_counts[p] < geom.header) {           // (every block starts with its header)
                // (seen by this rank alone - the others' blocks may be whole: the verdict is NOT shared, the caller ends the group
                // instead of leaving the others waiting for this rank at their next agreement; advisor round 5)
                mg->verdict_shared = false;
                return mg_fail(mg, RALA_HIP_EDEVICE, "a block shorter than its header");
            }
            n_records += recv_counts[p] - geom.header;
            if (p != mg->rank) at += recv_counts[p];
            words += recv_counts[p];
        }
        for (uint32_t p = P; p < 64; ++p) blocks.off[p] = 0;
        blocks.off[mg->rank] = (uint32_t)mine_at;
        blocks.self = mg->rank;
        if (std::max<uint64_t>(at, mine_at + recv_counts[mg->rank]) >= 0xFFFFFFF0ull) {
            return mg_fail(mg, RALA_HIP_EINVAL, "positions of the bound records must fit 32 bits");
        }
        rc = from_ctx(mg, cl, set_bound_blocks(cl, (const uint64_t*)mg->d_recv.p, (const uint64_t*)mg->d_send.p, blocks, geom, n_records), "set blocks");
    } else {
        rc = from_ctx(mg, cl, records ? rala_hip_set_bound_records(cl, (const uint64_t*)mg->d_recv.p, n_recv, RALA_HIP_MEM_DEVICE)
                                      : rala_hip_set_bound_tuples(cl, (const uint64_t*)mg->d_recv.p, n_recv, RALA_HIP_MEM_DEVICE),
                      "set tuples");
    }
    if (rc == RALA_HIP_OK) {
        rc = rala_hip_initialize(cl);
        if (rc == RALA_HIP_EFILTERED) rc = RALA_HIP_OK;      // all of ONE rank's reads filtered is not the job's verdict
        rc = from_ctx(mg, cl, rc, "owner initialize");
    }
    // (the interval pools' sizes travel with the status)
    const uint64_t my_pool = rc == RALA_HIP_OK ? cl->pool_used : 0;
    rc = agree_with(mg, rc, "owner initialize", 0, &my_pool, 1, all);
    if (rc != RALA_HIP_OK) return rc;
    mark();

    // 4. all-gather of the per-read state and the interval pools; install in the slice context
    StateLayout L{mg->nl_pad};
    MGCHECK(mg->d_state_mine.ensure(L.bytes() + 16));
    MGCHECK(mg->d_state_all.ensure(L.bytes() * P + 16));
    hipLaunchKernelGGL(pack_state_kernel, dim3((unsigned)((L.nl + kBlock - 1) / kBlock)), dim3(kBlock), 0, s,
                       read_arrays(cl), mg->n_local, L, mg->d_state_mine.p);
    rc = from_comm(mg, comm->all_gather(mg->d_state_mine.p, mg->d_state_all.p, L.bytes(), s), "all-gather of the read state");
    if (rc != RALA_HIP_OK) return rc;
    std::vector<uint64_t> pool_counts(P);
    RankTable base;
    uint64_t pool_total = 0;
    for (uint32_t p = 0; p < P; ++p) {
        pool_counts[p] = all[(size_t)p * 2 + 1];
        base.v[p] = (uint32_t)pool_total;
        pool_total += pool_counts[p];
    }
    if (pool_total >= 0xFFFFFFF0ull) {          // (the same sum on every rank)
        mg->verdict_shared = true;
        return mg_fail(mg, RALA_HIP_ECAPACITY, "interval pools of all ranks exceed 32-bit slots");
    }
    if (pool_total > cs->pool_cap) {
        cs->pool_cap = (uint32_t)pool_total + 1024;
        MGCHECK(cs->d_pool.ensure(cs->pool_cap));
    }
    rc = from_comm(mg, comm->all_gather_v(cl->d_pool.p, cs->d_pool.p, pool_counts.data(), sizeof(Interval), s),
                   "all-gather of the interval pools");
    if (rc != RALA_HIP_OK) return rc;
    hipLaunchKernelGGL(unpack_state_kernel, dim3((unsigned)((mg->n_reads + kBlock - 1) / kBlock)), dim3(kBlock), 0, s,
                       (const uint8_t*)mg->d_state_all.p, L, P, mg->n_reads, base, read_arrays(cs));
    rc = install_read_state(cs, pool_total);
    if (rc != RALA_HIP_OK && rc != RALA_HIP_EFILTERED) return from_ctx(mg, cs, rc, "install state");
    if (rc == RALA_HIP_EFILTERED) {                 // the same verdict on every rank
        mg->verdict_shared = true;
        return from_ctx(mg, cs, rc, "initialize");
    }
    mark();

    // 5. second pass .. preprocess tail .. graph, sharded by slice where it is per overlap.  (No status
    // exchange behind it: a rank that fails in there aborts the group, the caller sees to that.)
    rc = from_ctx(mg, cs, construct_stages(cs, comm, with_sens), "construct");
    if (rc != RALA_HIP_OK) return rc;
    mark();
    return RALA_HIP_OK;
}

// the stages' times from the events run_primary recorded (after the run's last wait for the stream)
void read_marks(rala_hip_mg* mg) {
    float* const slot[5] = {&mg->tm.emit_ms, &mg->tm.exchange_ms, &mg->tm.owner_ms, &mg->tm.gather_ms, &mg->tm.construct_ms};
    for (int k = 0; k < 5; ++k) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, mg->ev[k], mg->ev[k + 1]) == hipSuccess) *slot[k] = ms;
    }
}

}  // namespace

extern "C" {

int rala_hip_mg_unique_id(void* id) {
    if (!id) return RALA_HIP_EINVAL;
    std::string err;
    return rccl_unique_id(id, &err) == 0 ? RALA_HIP_OK : RALA_HIP_EDEVICE;
}

int rala_hip_mg_local_group_create(uint32_t world, void** group) {
    if (!group) return RALA_HIP_EINVAL;
    *group = create_local_group(world);
    return *group ? RALA_HIP_OK : RALA_HIP_EINVAL;
}

void rala_hip_mg_local_group_destroy(void* group) { destroy_local_group((LocalGroup*)group); }

// Two steps, so that callers can make sure every rank has its device before anybody enters the collective part: a rank
// whose contexts cannot be created never reaches ncclCommInitRank, and the others would wait there for it.
int rala_hip_mg_create_contexts(int device, uint32_t rank, uint32_t world, rala_hip_mg** out) {
    if (!out) return RALA_HIP_EINVAL;
    *out = nullptr;
    if (world == 0 || world > 64 || rank >= world) return RALA_HIP_EINVAL;
    if (getenv("RALA_HIP_DEBUG_FAIL_RANK") && (uint32_t)atoi(getenv("RALA_HIP_DEBUG_FAIL_RANK")) == rank) return RALA_HIP_EDEVICE;    // tests
    rala_hip_mg* mg = new rala_hip_mg;
    mg->device = device; mg->rank = rank; mg->world = world;
    int rc = rala_hip_create(device, &mg->cs);
    if (rc == RALA_HIP_OK) rc = rala_hip_create(device, &mg->cl);
    for (auto& e : mg->ev) if (rc == RALA_HIP_OK && hipEventCreate(&e) != hipSuccess) rc = RALA_HIP_EDEVICE;
    if (rc != RALA_HIP_OK) { rala_hip_mg_destroy(mg); return rc; }
    *out = mg;
    return RALA_HIP_OK;
}

int rala_hip_mg_join(rala_hip_mg* mg, int transport, const void* token) {
    if (!mg || !token) return RALA_HIP_EINVAL;
    if (mg->comm) return mg_fail(mg, RALA_HIP_EINVAL, "the rank has joined a group already");
    (void)hipSetDevice(mg->device);
    std::string err;
    if (transport == RALA_HIP_COMM_RCCL) mg->comm = create_rccl_comm(mg->rank, mg->world, token, &err);
    else if (transport == RALA_HIP_COMM_LOCAL) mg->comm = create_local_comm((LocalGroup*)token, mg->rank, mg->device, &err);
    if (!mg->comm) {
        fprintf(stderr, "[rala_hip_mg_join] %s\n", err.empty() ? "unknown transport" : err.c_str());
        return mg_fail(mg, RALA_HIP_EDEVICE, err.empty() ? "unknown transport" : err);
    }
    return RALA_HIP_OK;
}

int rala_hip_mg_create(int device, uint32_t rank, uint32_t world, int transport, const void* token, rala_hip_mg** out) {
    if (!out) return RALA_HIP_EINVAL;
    *out = nullptr;
    if (!token) return RALA_HIP_EINVAL;
    rala_hip_mg* mg = nullptr;
    int rc = rala_hip_mg_create_contexts(device, rank, world, &mg);
    if (rc != RALA_HIP_OK) return rc;
    rc = rala_hip_mg_join(mg, transport, token);
    if (rc != RALA_HIP_OK) { rala_hip_mg_destroy(mg); return rc; }
    *out = mg;
    return RALA_HIP_OK;
}

void rala_hip_mg_destroy(rala_hip_mg* mg) {
    if (!mg) return;
    (void)hipSetDevice(mg->device);
    if (mg->cs) (void)hipStreamSynchronize(mg->cs->stream);
    if (mg->cl) (void)hipStreamSynchronize(mg->cl->stream);
    for (auto& e : mg->ev) if (e) (void)hipEventDestroy(e);
    delete mg->comm;
    mg->d_send.release(); mg->d_recv.release(); mg->d_state_mine.release(); mg->d_state_all.release();
    if (mg->cl) rala_hip_destroy(mg->cl);
    if (mg->cs) rala_hip_destroy(mg->cs);
    delete mg;
}

const char* rala_hip_mg_last_error(const rala_hip_mg* mg) { return mg ? mg->err.c_str() : "no object"; }

int rala_hip_mg_set_reads(rala_hip_mg* mg, const uint32_t* read_len, uint64_t n_reads) {
    if (!mg || (!read_len && n_reads)) return RALA_HIP_EINVAL;
    mg->n_reads = n_reads;
    mg->n_local = local_count(n_reads, mg->rank, mg->world);
    mg->nl_pad = (local_count(n_reads, 0, mg->world) + 15) / 16 * 16;        // the same on every rank
    int rc = from_ctx(mg, mg->cs, rala_hip_set_reads(mg->cs, read_len, n_reads), "set_reads");
    if (rc != RALA_HIP_OK) return rc;
    std::vector<uint32_t> local(mg->n_local);
    for (uint64_t j = 0; j < mg->n_local; ++j) local[j] = read_len[j * mg->world + mg->rank];
    rc = from_ctx(mg, mg->cl, rala_hip_set_reads(mg->cl, local.data(), mg->n_local), "set_reads (owner)");
    mg->have_reads = rc == RALA_HIP_OK;
    return rc;
}

int rala_hip_mg_set_overlaps(rala_hip_mg* mg, const rala_hip_overlaps* slice, uint64_t n, uint64_t first, int mem) {
    if (!mg) return RALA_HIP_EINVAL;
    if (first + n >= 0xFFFFFFF0ull) return mg_fail(mg, RALA_HIP_EINVAL, "file positions must fit 32 bits");
    const int rc = from_ctx(mg, mg->cs, rala_hip_set_overlaps(mg->cs, slice, n, mem), "set_overlaps");
    if (rc != RALA_HIP_OK) return rc;
    mg->cs->ovl.base = first;
    mg->have_overlaps = true;
    return RALA_HIP_OK;
}

// Cut points of the overlap file: world + 1 positions, every cut at the start of a run of equal
// a_id.  Records that do not resolve - query OR target unknown (RALA_HIP_NO_READ) - neither start
// nor end a run (graph.cpp:338-350: a failed transmute skips the record before the run logic
// sees it), so a cut never falls between X, <unresolved>, X.  b_id may be NULL (targets all known).
int rala_hip_mg_slice_cuts(const uint32_t* a_id, const uint32_t* b_id, uint64_t n, uint32_t world, uint64_t* cuts) {
    if (!cuts || world == 0 || (!a_id && n)) return RALA_HIP_EINVAL;
    auto skipped = [&](uint64_t i) { return a_id[i] == RALA_HIP_NO_READ || (b_id && b_id[i] == RALA_HIP_NO_READ); };
    cuts[0] = 0;
    for (uint32_t k = 1; k < world; ++k) {
        uint64_t i = std::max(cuts[k - 1], n / world * k + std::min<uint64_t>(k, n % world));
        // the resolved record in front of position i
        while (i < n) {
            if (i == 0) break;
            if (skipped(i)) { ++i; continue; }
            uint64_t j = i;
            while (j > 0 && skipped(j - 1)) --j;
            if (j == 0 || a_id[j - 1] != a_id[i]) break;      // a new run starts at i
            ++i;
        }
        cuts[k] = std::min(i, n);
    }
    cuts[world] = n;
    return RALA_HIP_OK;
}


// The overlaps of a sharded run from PAF TEXT, every rank's share tokenised on its own GPU (collective; round 5 - before,
// `rala --gpus N` parsed the whole file on the host and handed every rank a slice from host memory).  Rank k ships bytes
// [n k / P, n (k + 1) / P) of the file to its device and tokenises the lines that START there (ingest.hip); the rows of all
// ranks, in rank order, are the file's records.  A slice must not cut a run of equal queries (duplicate removal is per
// run, graph.cpp:343-350; rala_hip_mg_slice_cuts): the ranks exchange what the cuts depend on - their row counts, the
// queries of their first and last resolved rows, where their first run ends - with the status word, every rank computes
// the same cuts, and the rows in front of a rank's cut (the tail of a run that began on a rank before it, unresolved
// records in front of the first resolved one) travel to the rank that holds the run's start, column by column.
// *length_error_read: the first record in file order whose length differs from its sequence's (check_lengths); -1: none.
// *irregular != 0 (the same on every rank): not a file of 12-column records - nothing was set, take the host reader.
namespace {
int ingest_paf(rala_hip_mg* mg, const char* path, int check_lengths, uint32_t threads, int64_t* length_error_read, int* irregular);
}
int rala_hip_mg_set_overlaps_from_paf(rala_hip_mg* mg, const char* path, int check_lengths, uint32_t threads, int64_t* length_error_read,
                                      int* irregular) {
    if (!mg || !path || !length_error_read || !irregular) return RALA_HIP_EINVAL;
    *length_error_read = -1;
    *irregular = 0;
    if (!mg->comm) return mg_fail(mg, RALA_HIP_EINVAL, "the rank has not joined its group (rala_hip_mg_join)");
    if (!mg->have_reads) return mg_fail(mg, RALA_HIP_EINVAL, "set the reads first");
    mg->verdict_shared = false;
    const int rc = ingest_paf(mg, path, check_lengths, threads, length_error_read, irregular);
    // (a failure the others do not know of: they must not wait for this rank in the next collective)
    if (rc != RALA_HIP_OK && !mg->verdict_shared) mg->comm->abort();
    return rc;
}
namespace {
int ingest_paf(rala_hip_mg* mg, const char* path, int check_lengths, uint32_t threads, int64_t* length_error_read, int* irregular) {
    rala_hip_ctx* cs = mg->cs;
    const uint32_t P = mg->world, me = mg->rank;
    constexpr uint64_t kNone = 0xFFFFFFFFull;
    int rc = RALA_HIP_OK;
    mg->have_overlaps = false;
    if (hipSetDevice(mg->device) != hipSuccess) rc = mg_fail(mg, RALA_HIP_EDEVICE, "hipSetDevice");
    hipStream_t s = cs->stream;
    PafTarget T;
    for (int k = 0; k < 7; ++k) T.col[k] = &cs->d_paf_col[k];
    T.strand = &cs->d_paf_strand;
    PafRange R;
    uint64_t file_n = 0;
    if (rc == RALA_HIP_OK) {
        struct stat st;
        if (stat(path, &st) != 0) rc = mg_fail(mg, RALA_HIP_EINVAL, std::string("cannot open ") + path);
        else if (!S_ISREG(st.st_mode)) rc = mg_fail(mg, RALA_HIP_ENOTAFILE, std::string("not a regular file: ") + path);
        else file_n = (uint64_t)st.st_size;
    }
    cs->inputs_set = false;
    cs->n_ovl = 0;
    if (rc == RALA_HIP_OK) {
        const uint64_t lo = file_n / P * me + std::min<uint64_t>(me, file_n % P), hi = file_n / P * (me + 1) + std::min<uint64_t>(me + 1, file_n % P);
        rc = from_ctx(mg, cs, paf_tokenise_range(cs, path, lo, hi, check_lengths != 0, threads, 1u << 16, T, &R), "tokenise");
    }
    // what the cuts need of this rank's rows
    uint32_t ends[5] = {kNoRow, 0, 0, kNoRow, kNoRow};
    const uint32_t n_mine = (uint32_t)R.n_lines;
    if (rc == RALA_HIP_OK && R.flags == 0) {
        DevBuf<uint32_t>& d_ends = cs->d_shard_words;             // (eight words of scratch)
        hipError_t e = d_ends.ensure(64);
        ends[2] = n_mine;
        if (e == hipSuccess) e = hipMemcpyAsync(d_ends.p, ends, sizeof(ends), hipMemcpyHostToDevice, s);
        if (e == hipSuccess && n_mine) {
            const uint32_t grid = std::min<uint32_t>((n_mine + kBlock - 1) / kBlock, 2048);
            hipLaunchKernelGGL(run_ends_kernel, dim3(grid), dim3(kBlock), 0, s, (const uint32_t*)cs->d_paf_col[0].p, (const uint32_t*)cs->d_paf_col[1].p,
                               n_mine, (uint32_t)mg->n_reads, d_ends.p);
            hipLaunchKernelGGL(head_run_kernel, dim3(grid), dim3(kBlock), 0, s, (const uint32_t*)cs->d_paf_col[0].p, (const uint32_t*)cs->d_paf_col[1].p,
                               n_mine, (uint32_t)mg->n_reads, d_ends.p);
            hipLaunchKernelGGL(run_ends_names_kernel, dim3(1), dim3(1), 0, s, (const uint32_t*)cs->d_paf_col[0].p, d_ends.p);
        }
        if (e == hipSuccess) e = hipMemcpyAsync(ends, d_ends.p, sizeof(ends), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) rc = mg_fail(mg, RALA_HIP_EDEVICE, std::string("run ends: ") + hipGetErrorString(e));
    }
    // one host exchange: {rows, first resolved row, end of the first run, query of the first / last resolved row, tokeniser's
    // flags, first length error}
    const uint64_t mine[7] = {n_mine, ends[0] == kNoRow ? n_mine : ends[0], ends[2], ends[3], ends[4], R.flags, R.first_bad};
    std::vector<uint64_t> all;
    rc = agree_with(mg, rc, "device ingest", 0, mine, 7, all);
    if (rc != RALA_HIP_OK) return rc;
    auto of = [&](uint32_t p, uint32_t f) { return all[(size_t)p * 8 + 1 + f]; };
    uint32_t flags = 0;
    for (uint32_t p = 0; p < P; ++p) flags |= (uint32_t)of(p, 5);
    if (flags) { *irregular = (int)flags; return RALA_HIP_OK; }
    for (uint32_t p = 0; p < P; ++p) {
        if (of(p, 6) != ~0ull) { *length_error_read = (int64_t)(of(p, 6) & 0xFFFFFFFFull); return RALA_HIP_OK; }   // (file order: the lowest rank's)
    }
    // the cuts: how many of its first rows every rank gives away, to whom, what every rank ends up with
    std::vector<uint64_t> n_rows(P), move(P, 0), final_rows(P, 0), first_pos(P, 0);
    std::vector<uint32_t> dest(P, 0);
    uint64_t X = kNone;
    uint32_t keeper = 0;                                // the last rank so far that keeps a row (0 while there is none)
    for (uint32_t p = 0; p < P; ++p) {
        n_rows[p] = of(p, 0);
        const uint64_t lead = of(p, 1), head_end = of(p, 2), head_a = of(p, 3), tail_a = of(p, 4);
        if (p == 0) move[p] = 0;
        else if (head_a == kNone) move[p] = n_rows[p];                 // nothing resolves: the rows stand behind the record in front of them
        else if (X != kNone && head_a == X) move[p] = head_end;        // the run goes on: up to its end
        else move[p] = lead;                                           // a cut never falls on a record that does not resolve
        dest[p] = keeper;
        if (n_rows[p] > move[p]) keeper = p;
        if (tail_a != kNone) X = tail_a;
    }
    for (uint32_t p = 0; p < P; ++p) {
        final_rows[p] += n_rows[p] - move[p];
        if (move[p]) final_rows[dest[p]] += move[p];
    }
    for (uint32_t p = 1; p < P; ++p) first_pos[p] = first_pos[p - 1] + final_rows[p - 1];
    if (first_pos[P - 1] + final_rows[P - 1] >= 0xFFFFFFF0ull) {
        // (computed from the exchanged counts: the same answer on every rank - a shared verdict, the group stays usable and the host
        // readers' way into rala_hip_mg_set_overlaps can follow; advisor round 5)
        mg->verdict_shared = true;
        return mg_fail(mg, RALA_HIP_ETOOLARGE, "file positions must fit 32 bits");
    }
    // the rows in front of the cuts travel, column by column
    std::vector<uint64_t> send_counts(P, 0), recv_counts(P, 0);
    if (move[me]) send_counts[dest[me]] = move[me];
    uint64_t incoming = 0;
    for (uint32_t p = 0; p < P; ++p) if (p != me && move[p] && dest[p] == me) { recv_counts[p] = move[p]; incoming += move[p]; }
    for (int k = 0; k < 7 && rc == RALA_HIP_OK; ++k) {
        if (grow_keep(cs->d_paf_col[k], n_mine, (size_t)n_mine + incoming + 1, s) != hipSuccess) rc = mg_fail(mg, RALA_HIP_ENOMEM, "overlap columns");
    }
    if (rc == RALA_HIP_OK && grow_keep(cs->d_paf_strand, n_mine, (size_t)n_mine + incoming + 1, s) != hipSuccess) rc = mg_fail(mg, RALA_HIP_ENOMEM, "overlap columns");
    if (rc != RALA_HIP_OK) return rc;                   // (this rank's alone: the caller ends the group)
    for (int k = 0; k < 7; ++k) {
        rc = from_comm(mg, mg->comm->all_to_all_v(cs->d_paf_col[k].p, send_counts.data(), cs->d_paf_col[k].p + n_mine, recv_counts.data(), 4, s),
                       "rows in front of the cuts");
        if (rc != RALA_HIP_OK) return rc;
    }
    rc = from_comm(mg, mg->comm->all_to_all_v(cs->d_paf_strand.p, send_counts.data(), cs->d_paf_strand.p + n_mine, recv_counts.data(), 1, s),
                   "rows in front of the cuts");
    if (rc != RALA_HIP_OK) return rc;
    MGCHECK(hipStreamSynchronize(s));
    const uint64_t off = move[me];
    rala_hip_overlaps dev;
    dev.a_id = cs->d_paf_col[0].p + off; dev.b_id = cs->d_paf_col[1].p + off; dev.a_begin = cs->d_paf_col[2].p + off;
    dev.a_end = cs->d_paf_col[3].p + off; dev.b_begin = cs->d_paf_col[4].p + off; dev.b_end = cs->d_paf_col[5].p + off;
    dev.length = cs->d_paf_col[6].p + off; dev.strand = cs->d_paf_strand.p + off;
    return rala_hip_mg_set_overlaps(mg, &dev, final_rows[me], first_pos[me], RALA_HIP_MEM_DEVICE);
}
}  // namespace

namespace {

int run_all(rala_hip_mg* mg, const rala_hip_overlaps* sens_slice, uint64_t n_sens, uint32_t* n_pairs) {
    MGCHECK(hipSetDevice(mg->device));
    mg->tm = rala_hip_mg_timings();
    const double t0 = now_ms();
    int rc = run_primary(mg, sens_slice != nullptr);
    if (rc != RALA_HIP_OK) return rc;
    if (sens_slice != nullptr) {
        // collective: a rank whose share is empty still takes part (every rank passes a non-null
        // pointer or none does; the first agree() of the run checked that)
        const double t2 = now_ms();
        rc = from_ctx(mg, mg->cs, repeats_stage(mg->cs, mg->cl, mg->comm, sens_slice, n_sens), "sensitive pass");
        if (rc != RALA_HIP_OK) return rc;
        mg->tm.repeats_ms = (float)(now_ms() - t2);
    }
    const double t1 = now_ms();
    rc = from_ctx(mg, mg->cs, transitive_stage(mg->cs, mg->comm, n_pairs), "remove_transitive_edges");
    if (rc != RALA_HIP_OK) return rc;
    mg->tm.tr_ms = (float)(now_ms() - t1);
    mg->tm.total_ms = (float)(now_ms() - t0);
    read_marks(mg);
    return RALA_HIP_OK;
}

}  // namespace

// A failure every rank knows of (agree(), the all-filtered verdict) just returns.  Any other one is
// this rank's alone - an allocation, a kernel, a copy - and the next collective of the others
// would wait for this rank for ever: the group is aborted instead (Comm::abort), every rank's run
// fails, the rank objects are of no use afterwards.
int rala_hip_mg_run(rala_hip_mg* mg, const rala_hip_overlaps* sens_slice, uint64_t n_sens, uint32_t* n_pairs) {
    if (!mg || !n_pairs) return RALA_HIP_EINVAL;
    if (!mg->comm) return mg_fail(mg, RALA_HIP_EINVAL, "the rank has not joined its group (rala_hip_mg_join)");
    if (!mg->have_reads || !mg->have_overlaps) return mg_fail(mg, RALA_HIP_EINVAL, "set reads and overlaps first");
    mg->verdict_shared = false;
    const int rc = run_all(mg, sens_slice, n_sens, n_pairs);
    if (rc != RALA_HIP_OK && !mg->verdict_shared) mg->comm->abort();
    return rc;
}

int rala_hip_mg_run_threads(rala_hip_mg** ranks, uint32_t n, const rala_hip_overlaps* sens_slices, const uint64_t* n_sens,
                            uint32_t* n_pairs) {
    if (!ranks || n == 0 || !n_pairs) return RALA_HIP_EINVAL;
    std::vector<int> rc(n, RALA_HIP_OK);
    std::vector<uint32_t> pairs(n, 0);
    std::vector<std::thread> th;
    for (uint32_t k = 0; k < n; ++k) {
        th.emplace_back([&, k]() {
            rc[k] = rala_hip_mg_run(ranks[k], sens_slices ? &sens_slices[k] : nullptr, n_sens ? n_sens[k] : 0, &pairs[k]);
        });
    }
    for (auto& t : th) t.join();
    for (uint32_t k = 0; k < n; ++k) if (rc[k] != RALA_HIP_OK) return rc[k];
    for (uint32_t k = 1; k < n; ++k) {
        if (pairs[k] != pairs[0]) { ranks[0]->err = "ranks disagree on the transitive reduction"; return RALA_HIP_EDEVICE; }
    }
    *n_pairs = pairs[0];
    return RALA_HIP_OK;
}

int rala_hip_mg_get_slice(rala_hip_mg* mg, uint64_t* first, uint64_t* n) {
    if (!mg || !first || !n) return RALA_HIP_EINVAL;
    if (!mg->have_overlaps) return mg_fail(mg, RALA_HIP_EINVAL, "no overlaps set");
    *first = mg->cs->ovl.base;
    *n = mg->cs->n_ovl;
    return RALA_HIP_OK;
}

rala_hip_ctx* rala_hip_mg_context(rala_hip_mg* mg) { return mg ? mg->cs : nullptr; }
rala_hip_ctx* rala_hip_mg_owner_context(rala_hip_mg* mg) { return mg ? mg->cl : nullptr; }

int rala_hip_mg_get_pile_data(rala_hip_mg* mg, uint64_t read, uint16_t* data) {
    if (!mg || !data) return RALA_HIP_EINVAL;
    if (read >= mg->n_reads || read % mg->world != mg->rank) return mg_fail(mg, RALA_HIP_EINVAL, "the read's pile lives on rank read % world");
    struct DeviceGuard {                // the caller's current device is the caller's business
        int before = -1;
        DeviceGuard() { if (hipGetDevice(&before) != hipSuccess) before = -1; }
        ~DeviceGuard() { if (before >= 0) (void)hipSetDevice(before); }
    } guard;
    MGCHECK(hipSetDevice(mg->device));
    // the owner's context keeps the coverage; the valid region that applies is the final one
    const uint64_t j = read / mg->world;
    rala_hip_ctx* cl = mg->cl;
    const uint32_t n = cl->h_read_len[j];
    MGCHECK(hipMemcpy(data, cl->d_pile.p + cl->h_pile_off[j], (size_t)n * 2, hipMemcpyDeviceToHost));
    uint32_t be[2] = {0, 0};
    uint8_t alive = 0;
    MGCHECK(hipMemcpy(&be[0], mg->cs->d_begin.p + read, 4, hipMemcpyDeviceToHost));
    MGCHECK(hipMemcpy(&be[1], mg->cs->d_end.p + read, 4, hipMemcpyDeviceToHost));
    MGCHECK(hipMemcpy(&alive, mg->cs->d_alive.p + read, 1, hipMemcpyDeviceToHost));
    if (alive) {
        for (uint32_t p = 0; p < be[0] && p < n; ++p) data[p] = 0;      // Pile::shrink zeroes outside (pile.cpp:311-318)
        for (uint32_t p = be[1]; p < n; ++p) data[p] = 0;
    }
    return RALA_HIP_OK;
}

int rala_hip_mg_get_pile_row_digests(rala_hip_mg* mg, uint64_t* fnv, uint64_t* inside, uint64_t* outside) {
    if (!mg) return RALA_HIP_EINVAL;
    struct DeviceGuard {
        int before = -1;
        DeviceGuard() { if (hipGetDevice(&before) != hipSuccess) before = -1; }
        ~DeviceGuard() { if (before >= 0) (void)hipSetDevice(before); }
    } guard;
    MGCHECK(hipSetDevice(mg->device));
    // the owner's context keeps the rows (local row j = read j * world + rank); the regions that apply are the final ones of the
    // replicated state
    rala_hip_ctx* cl = mg->cl;
    const uint64_t n_own = cl->n_reads, n = mg->n_reads;
    std::vector<uint32_t> b(n), e(n), ob(n_own), oe(n_own);
    std::vector<uint8_t> a(n), oa(n_own);
    if (n) {
        MGCHECK(hipMemcpy(b.data(), mg->cs->d_begin.p, n * 4, hipMemcpyDeviceToHost));
        MGCHECK(hipMemcpy(e.data(), mg->cs->d_end.p, n * 4, hipMemcpyDeviceToHost));
        MGCHECK(hipMemcpy(a.data(), mg->cs->d_alive.p, n, hipMemcpyDeviceToHost));
    }
    for (uint64_t j = 0; j < n_own; ++j) {
        const uint64_t r = j * mg->world + mg->rank;
        ob[j] = b[r]; oe[j] = e[r]; oa[j] = a[r];
    }
    const int rc = pile_row_digests(cl, ob.data(), oe.data(), oa.data(), fnv, inside, outside);
    return rc == RALA_HIP_OK ? rc : mg_fail(mg, rc, rala_hip_last_error(cl));
}

int rala_hip_mg_get_timings(rala_hip_mg* mg, rala_hip_mg_timings* out) {
    if (!mg || !out) return RALA_HIP_EINVAL;
    *out = mg->tm;
    return RALA_HIP_OK;
}

}  // extern "C"
