// Stage functions of librala_hip shared by the single-GPU entry points (pipeline.hip) and the
// sharded runner (sharded.hip).  A non-null Comm makes a stage collective: every rank of the
// group calls it with its own slice of the overlaps and the same per-read state.
#pragma once

#include "comm.h"
#include "context.h"

namespace rala_hip {

// second overlap pass (graph.cpp:443-518) .. Graph::preprocess for chimeras (:699-880) .. node and
// edge construction (:553-632) on ctx->ovl (file positions ctx->ovl.base + i):
//   classify (trim / type, killer list)                      per overlap of the slice
//   containment fixed point                                  comm: bounds all-reduced (min) per round
//   liveness, hill counters, survivors                       comm: counters all-reduced (sum),
//                                                                  survivor lists all-gathered
//   preprocess tail + graph on the survivors                 replicated
int construct_stages(rala_hip_ctx* ctx, Comm* comm, bool sensitive_pass_follows = false);

// Graph::preprocess with the sensitive overlaps (graph.cpp:882-1054) behind construct_stages:
// cs = the context construct_stages ran on, with this rank's share of the sensitive overlaps;
// cl = the context that holds the piles (the same one on a single GPU, comm = null)
int repeats_stage(rala_hip_ctx* cs, rala_hip_ctx* cl, Comm* comm, const rala_hip_overlaps* sens, uint64_t n_sens);

// Graph::remove_transitive_edges (graph.cpp:1281-1335) on the graph construct_stages left on the
// device; with a communicator every rank probes from its share of the edges and the marks are
// all-reduced
int transitive_stage(rala_hip_ctx* ctx, Comm* comm, uint32_t* n_pairs);

// install per-read state that was computed elsewhere and already sits in ctx's device arrays
// (d_begin .. d_iv_slot, d_pool[0 .. pool_count)); validity bits of ctx's own overlaps must be
// there too (rala_hip_dedupe).  Counts the filtered reads.
int install_read_state(rala_hip_ctx* ctx, uint64_t pool_count);

// sharded runs, the bounds scattered once on the sender (bucket_kernels.hip): the sender's and the owner's side
int shard_emit(rala_hip_ctx* ctx, const ShardGeometry& g, uint64_t* send, uint64_t* words);
int set_bound_blocks(rala_hip_ctx* ctx, const uint64_t* base, const uint64_t* base_self, const ShardBlocks& blocks, const ShardGeometry& g,
                     uint64_t n_records);

// the device tokeniser (ingest.hip): the lines that start in bytes [lo, hi) of a PAF file into the target's columns
struct PafTarget {
    DevBuf<uint32_t>* col[7];       // a_id, b_id, a_begin, a_end, b_begin, b_end, length
    DevBuf<uint8_t>* strand;
    bool mhap = false;              // the file is MHAP: twelve blank-separated numeric columns, no names
};
struct PafRange {
    uint64_t n_lines = 0;
    unsigned long long first_bad = ~0ull;      // row << 32 | read of the first record whose length differs from its sequence's
    uint32_t flags = 0;                        // != 0: not a file of 12-column records (kernels.h: launch_paf_parse)
    uint64_t file_bytes = 0;
};
int paf_tokenise_range(rala_hip_ctx* ctx, const char* path, uint64_t lo, uint64_t hi, bool check_lengths, uint32_t threads,
                       size_t extra_rows, const PafTarget& T, PafRange* out);

// checksums of the rows cl holds (verify_kernels.hip) under the regions / liveness given per row (host arrays, cl->n_reads entries)
int pile_row_digests(rala_hip_ctx* cl, const uint32_t* begin, const uint32_t* end, const uint8_t* alive, uint64_t* fnv, uint64_t* inside,
                     uint64_t* outside);

hipError_t stream_sync(rala_hip_ctx* ctx, hipStream_t s);
hipError_t d2h_small(rala_hip_ctx* ctx, void* dst, const void* src, size_t bytes, hipStream_t s);
int flush_upload(rala_hip_ctx* ctx);       // (pipeline.hip) RALA_HIP_MEM_HOST_ASYNC columns not uploaded yet: now

}  // namespace rala_hip
