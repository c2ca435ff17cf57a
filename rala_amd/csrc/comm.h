// Collectives of the sharded (multi-GPU) run, one object per rank.
//
// Two transports behind one interface:
//   * RCCL over xGMI (librccl is opened at run time, so librala_hip has no link-time
//     dependency on it): one process per GPU (bench.py under torchrun) or one host thread per
//     GPU inside one process (rala::Graph with --gpus N);
//   * "local": ranks are threads of one process that exchange device pointers through a
//     shared rendezvous object and move the bytes with hipMemcpyAsync / a small reduce kernel.
//     It lets several ranks share ONE device, which RCCL refuses - that is how the sharded
//     algorithm is tested on a single-GPU box - and it is the transport of last resort on a
//     node without a usable RCCL.
// All calls are collective: every rank of the group makes the same calls in the same order.
// Buffers are device memory unless named host_*; work is enqueued on the given stream and the
// call returns when it is safe to reuse the send buffer on that stream (stream order).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

namespace rala_hip {

enum class ReduceOp { kSum, kMin, kMax };

class Comm {
public:
    virtual ~Comm() {}
    uint32_t rank() const { return rank_; }
    uint32_t world() const { return world_; }
    const std::string& error() const { return err_; }

    // small host-side exchange: every rank contributes n uint64 values, all[world * n] receives
    // rank 0's values, then rank 1's, ...  Blocks until done.
    virtual int host_all_gather(const uint64_t* mine, uint32_t n, uint64_t* all, hipStream_t s) = 0;
    // elements of elem_bytes; send holds the part for rank 0, then rank 1, ... (send_counts[world]);
    // recv receives the parts from rank 0, 1, ... (recv_counts[world], known from a host exchange).
    // own_part = false: the rank's part for itself stays where it is - it is neither copied nor given room in recv
    // (recv_counts[rank] is ignored)
    virtual int all_to_all_v(const void* send, const uint64_t* send_counts, void* recv, const uint64_t* recv_counts,
                             size_t elem_bytes, hipStream_t s, bool own_part = true) = 0;
    // every rank contributes `bytes` bytes; recv = world * bytes
    virtual int all_gather(const void* send, void* recv, size_t bytes, hipStream_t s) = 0;
    // rank k contributes counts[k] elements; recv holds them back to back in rank order
    virtual int all_gather_v(const void* send, void* recv, const uint64_t* counts, size_t elem_bytes, hipStream_t s) = 0;
    // in place over n uint32 values
    virtual int all_reduce_u32(uint32_t* buf, size_t n, ReduceOp op, hipStream_t s) = 0;
    virtual int barrier(hipStream_t s) = 0;
    // NOT collective: called by a rank that failed between two collectives and will not make the
    // next one.  Releases the ranks of this process that wait for it (their calls fail from here on);
    // the group is unusable afterwards.
    virtual void abort() = 0;

protected:
    uint32_t rank_ = 0, world_ = 1;
    std::string err_;
};

// ---- RCCL ----------------------------------------------------------------------------------
constexpr size_t kCommIdBytes = 128;                   // sizeof(ncclUniqueId)
// fills id[128]; rank 0 calls it and ships the bytes to the other ranks (any channel)
int rccl_unique_id(void* id, std::string* err);
// blocks until all `world` ranks have joined; the device must be current
Comm* create_rccl_comm(uint32_t rank, uint32_t world, const void* id, std::string* err);

// ---- ranks as threads of one process -------------------------------------------------------
struct LocalGroup;
LocalGroup* create_local_group(uint32_t world);
void destroy_local_group(LocalGroup* g);
Comm* create_local_comm(LocalGroup* g, uint32_t rank, int device, std::string* err);

}  // namespace rala_hip
