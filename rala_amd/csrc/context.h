// librala_hip context: device arenas + host-side state of one data set.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include <memory>
#include <string>
#include <vector>

#include "geom.h"
#include "host_pool.h"
#include "kernels.h"
#include "rala_hip.h"

namespace rala_hip {

struct HostOvl {
    uint32_t src;        // index in the input overlap arrays
    uint32_t a, b;
    Coords c;
    uint8_t strand;
    uint8_t dead;
    uint8_t type;        // cached Overlap::type, 255 = stale (a pile or the coordinates changed)
};

struct LaunchClass {
    uint32_t lw;         // LDS elements per array
    uint32_t first;      // range in the order array
    uint32_t count;
    bool in_lds;
};

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    // (round 6: the buffer as physical chunks mapped side by side into one reserved range - ensure_chunked; the rows of the piles)
    std::vector<hipMemGenericAllocationHandle_t> chunks;
    std::vector<void*> mapped;          // where chunk i is mapped (null: created, not mapped - a call that failed half way)
    size_t reserved = 0, chunk_size = 0;
    ~DevBuf() { release(); }
    void release() {
        if (reserved) {
            (void)hipDeviceSynchronize();       // (hipFree waits for the device's work by itself; unmapping does not)
            for (size_t i = 0; i < chunks.size(); ++i) {
                if (mapped[i]) (void)hipMemUnmap(mapped[i], chunk_size);
                (void)hipMemRelease(chunks[i]);
            }
            // The RANGE is not given back (hipMemAddressFree): the next reservation would be handed the same addresses, and a
            // kernel of the next context then wrote through translations of the OLD mapping - rows read back as zeros or as another
            // data set's in 3 runs of 8 of tests/test_gpu_parity.py::test_rows_in_mapped_chunks, none in 8 once no address is ever
            // mapped twice.  Address space is what leaks (a context's rows, out of 128 TB); when a reservation fails the caller
            // falls back to hipMalloc.
            chunks.clear();
            mapped.clear();
            reserved = 0;
            chunk_size = 0;
        } else if (p) {
            (void)hipFree(p);
        }
        p = nullptr;
        n = 0;
    }
    // chunk_bytes per physical allocation; order 1: chunk i lies at slot (i * stride) mod n_chunks of the range (neighbours in
    // the range are not neighbours in the order of allocation)
    hipError_t ensure_chunked(size_t count, size_t chunk_bytes, int order, int device) {
        if (count <= n && p) return hipSuccess;
        release();
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = device;
        size_t gran = 0;
        hipError_t e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
        if (e != hipSuccess) return e;
        if (gran == 0) gran = 2u << 20;
        chunk_bytes = (chunk_bytes + gran - 1) / gran * gran;
        const size_t n_chunks = (count * sizeof(T) + chunk_bytes - 1) / chunk_bytes;
        const size_t bytes = n_chunks * chunk_bytes;
        void* base = nullptr;
        e = hipMemAddressReserve(&base, bytes, 0, nullptr, 0);       // (an alignment asked for here is not honoured: 2 MB whatever)
        if (e != hipSuccess) return e;
        p = (T*)base;
        reserved = bytes;
        chunk_size = chunk_bytes;
        size_t stride = 1;
        if (order == 1 && n_chunks > 2) {
            stride = (size_t)((double)n_chunks * 0.6180339887) | 1u;
            auto gcd = [](size_t a, size_t b) { while (b) { const size_t t = a % b; a = b; b = t; } return a; };
            while (gcd(stride, n_chunks) != 1) stride += 2;
        }
        for (size_t i = 0; i < n_chunks && e == hipSuccess; ++i) {
            hipMemGenericAllocationHandle_t h;
            if (const char* f = getenv("RALA_HIP_DEBUG_CHUNK_FAIL")) {          // tests: the mapping fails at chunk k (the caller falls back)
                if ((size_t)atoll(f) == i) { e = hipErrorOutOfMemory; break; }
            }
            e = hipMemCreate(&h, chunk_bytes, &prop, 0);
            if (e != hipSuccess) break;
            chunks.push_back(h);
            mapped.push_back(nullptr);
            void* at = (char*)base + ((i * stride) % n_chunks) * chunk_bytes;
            e = hipMemMap(at, chunk_bytes, 0, h, 0);
            if (e == hipSuccess) mapped.back() = at;
        }
        if (e == hipSuccess) {
            hipMemAccessDesc acc = {};
            acc.location = prop.location;
            acc.flags = hipMemAccessFlagsProtReadWrite;
            e = hipMemSetAccess(base, bytes, &acc, 1);
        }
        if (e != hipSuccess) { release(); return e; }
        n = bytes / sizeof(T);
        return hipSuccess;
    }
    hipError_t ensure(size_t count) {
        if (count <= n && p) return hipSuccess;
        release();
        if (count == 0) count = 1;
        hipError_t e = hipMalloc((void**)&p, count * sizeof(T));
        if (e == hipSuccess) n = count;
        return e;
    }
    // Elements [first, first + count) to the host, chunk by chunk (every copy inside ONE of the runtime's allocations).
    hipError_t copy_to_host(void* dst, size_t first, size_t count) const {
        const char* src = (const char*)(p + first);
        size_t left = count * sizeof(T);
        while (left) {
            size_t take = left;
            if (chunk_size) take = std::min(left, chunk_size - (size_t)(src - (const char*)p) % chunk_size);
            const hipError_t e = hipMemcpy(dst, src, take, hipMemcpyDeviceToHost);
            if (e != hipSuccess) return e;
            dst = (char*)dst + take; src += take; left -= take;
        }
        return hipSuccess;
    }
    // room for `count` elements, the first `keep` kept (a blocking copy when the buffer has to move)
    hipError_t grow(size_t keep, size_t count) {
        if (count <= n && p) return hipSuccess;
        T* q = nullptr;
        const size_t want = count + count / 4 + 1;
        hipError_t e = hipMalloc((void**)&q, want * sizeof(T));
        if (e != hipSuccess) return e;
        if (keep && p) e = hipMemcpy(q, p, keep * sizeof(T), hipMemcpyDeviceToDevice);
        if (e != hipSuccess) { (void)hipFree(q); return e; }
        if (p) (void)hipFree(p);
        p = q;
        n = want;
        return hipSuccess;
    }
};

// pinned host staging buffer (fast, truly asynchronous device <-> host copies)
template <class T>
struct PinnedBuf {
    T* p = nullptr;
    size_t n = 0;
    ~PinnedBuf() { release(); }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        n = 0;
    }
    hipError_t ensure(size_t count) {
        if (count <= n && p) return hipSuccess;
        release();
        if (count == 0) count = 1;
        count += count / 4;                 // grow with slack: survivor counts vary little
        hipError_t e = hipHostMalloc((void**)&p, count * sizeof(T), hipHostMallocDefault);
        if (e == hipSuccess) n = count;
        return e;
    }
};

}  // namespace rala_hip

struct rala_hip_ctx {
    int device = 0;
    uint32_t n_compute_units = 256;
    hipStream_t stream = nullptr;
    hipStream_t side = nullptr;         // duplicate removal runs here, beside the bucketing
    hipStream_t aux = nullptr;          // the pile chain's small kernels (long and event-dense reads), beside the first one
    bool use_side_stream = true;
    std::string err;
    hipEvent_t ev[12] = {};
    // RALA_HIP_MEM_HOST_ASYNC: the columns' host addresses until rala_hip_initialize has uploaded them (copy stream; events:
    // ids, b coordinates, a coordinates, lengths there)
    hipStream_t copy = nullptr;
    hipEvent_t ev_up[4] = {};
    bool upload_pending = false, upload_queued = false;      // queued: every copy of this call's upload is on the copy stream
    const uint32_t* up_src[7] = {};
    const uint8_t* up_strand = nullptr;

    // options
    int64_t pool_per_read_x1000 = 1000;
    int64_t max_lds_read_len = 22000;
    int64_t debug_pile_stop_after = 99;
    int64_t host_threads = 0;                       // 0 = min(hardware threads, 16)
    std::unique_ptr<rala_hip::HostPool> pool;
    bool use_run_kernel = true;
    bool debug_fail_construct = false;          // tests: pass 2 fails on this context
    uint32_t debug_pile_variant = 0;            // measurements: PileArgs::variant
    uint32_t debug_dedupe_list_cap = 0;         // tests: duplicate removal's mark list holds this many marks (0 = 2^20); more: the pass over all overlaps
    uint32_t debug_fp_lds_limit = 0xFFFFFFFFu;  // tests: containment fixed points with more killers than this take the long lists' kernel
    bool use_bound_records = true;              // sharded runs: 8-byte bound records instead of two tuples per overlap side where they fit
    bool use_round_batches = true;              // containment fixed point: several rounds per look at the counter
    rala_hip::DevBuf<uint32_t> d_round_log;     // list sizes after the rounds the host did not look at

    // reads
    uint64_t n_reads = 0;
    uint32_t max_read_len = 0;
    // reads by length class (kernels.h: kPileClassBases): each class has its own first kernel in the
    // pile chain.  d_class_order lists the reads class by class; empty when every read is in class 0.
    uint32_t n_class[rala_hip::kPileClasses] = {};
    rala_hip::DevBuf<uint32_t> d_class_order;
    rala_hip::DevBuf<uint32_t> d_seg_base;            // tail: where each segment of the final overlap list ends
    rala_hip::DevBuf<uint32_t> d_dense;               // reads with more events than the first kernels take (+ its counter behind)
    std::vector<uint32_t> h_read_len;
    std::vector<uint64_t> h_pile_off;
    uint64_t pile_elems = 0;
    rala_hip::DevBuf<uint32_t> d_read_len;
    rala_hip::DevBuf<uint64_t> d_pile_off;
    rala_hip::DevBuf<uint16_t> d_pile;
    rala_hip::DevBuf<uint32_t> d_order, d_overflow;
    // reads whose region / interval lists outgrow the position-space kernels' LDS lists: noted in d_big_list[0], run
    // again with the lists in d_big_space at growing sizes (those that still do not fit: the other list)
    rala_hip::DevBuf<uint32_t> d_big_list[2], d_big_space;
    int64_t debug_big_caps = 0;         // tests: first sizes of the lists in global memory (0: the defaults)
    bool debug_force_big = false;       // tests: every read of the position-space kernels runs with its lists in global memory
    rala_hip::DevBuf<uint16_t> d_slab;
    std::vector<rala_hip::LaunchClass> classes;

    // the device tokeniser (ingest.hip): the host's name table as it is, the file's text, the columns it leaves
    rala_hip::DevBuf<uint8_t> d_name_buckets, d_name_arena, d_paf_text, d_paf_strand;
    uint64_t n_name_buckets = 0;
    rala_hip::DevBuf<uint32_t> d_paf_col[7], d_paf_chunk[2], d_paf_win[7];
    rala_hip::DevBuf<uint8_t> d_paf_win_strand;
    int64_t ingest_window_bytes = 0;    // option: the tokeniser's window over the file's text (0: a quarter of the free device memory)
    rala_hip::DevBuf<unsigned long long> d_paf_bad;
    rala_hip_ingest_timings ingest_tm = {};

    // overlaps
    uint64_t n_ovl = 0;
    rala_hip::OvlSoA ovl = {};
    rala_hip::DevBuf<uint32_t> d_ovl_u32[7];
    rala_hip::DevBuf<uint8_t> d_ovl_strand;
    rala_hip::DevBuf<uint8_t> d_valid, d_suspect;
    rala_hip::DevBuf<uint32_t> d_dedupe_list;      // where the counting pass marked queries (positions, then queries)
    bool valid_ready = false;
    bool inputs_set = false;            // rala_hip_set_overlaps / _set_bound_tuples was called

    // bound tuples shipped in by the caller instead of overlaps (multi-GPU owners)
    bool tuple_mode = false;
    uint64_t n_tuples = 0;
    const uint2* tuples = nullptr;      // {x = local read, y = bound}
    const uint64_t* records = nullptr;  // bound records instead (rala_hip_set_bound_records): n_records of them
    uint64_t n_records = 0;
    rala_hip::DevBuf<uint64_t> d_record;
    rala_hip::DevBuf<uint2> d_tuple;
    rala_hip::DevBuf<uint32_t> d_owner_cnt;
    bool piles_resident = false;
    // sharded runs, the bounds scattered ONCE on the sender (bucket_kernels.hip, "sharded runs").  A sender: the groups'
    // counts, the partitions' cursors, the blocks' lengths; duplicate removal runs beside the emit on the side stream and is
    // joined by the second pass (dedupe_pending).  An owner: its input is the blocks of all senders as they lie in the
    // runner's buffers (blocks_mode, with tuple_mode).
    bool use_fused_emit = true;
    rala_hip::DevBuf<uint32_t> d_shard_group, d_shard_part, d_shard_words, d_shard_tiles;
    bool dedupe_pending = false;
    bool lists_pending = false;         // a sharded run's survivor lists are being gathered on the side stream (event ev[3])
    bool blocks_mode = false;
    const uint64_t* blocks_base = nullptr;
    const uint64_t* blocks_base_self = nullptr;
    rala_hip::ShardBlocks blocks = {};
    rala_hip::ShardGeometry shard_geom = {};
    uint64_t n_block_records = 0;

    // bound CSR
    rala_hip::DevBuf<uint32_t> d_ev_off, d_cursor, d_ev, d_slot_rank[2], d_ev_fixed;
    uint32_t pile_chunk_mb = 1024;      // the rows' buffer as physical chunks of this size (0: one hipMalloc) - pipeline.hip
    bool debug_ev_events = false;
    uint32_t ev_shift = 0;              // units of d_ev_off: 1 << ev_shift events (kernels.h: kBucketPairShift)
    rala_hip::DevBuf<unsigned char> d_scan_ws;

    // per-read annotation
    rala_hip::DevBuf<uint32_t> d_begin, d_end, d_iv_slot;
    rala_hip::DevBuf<uint16_t> d_median, d_p10;
    rala_hip::DevBuf<uint8_t> d_alive;
    rala_hip::DevBuf<uint32_t> d_n_pits, d_n_hills;
    rala_hip::DevBuf<rala_hip::Interval> d_pool;
    rala_hip::DevBuf<uint32_t> d_small;      // [0] pool_count [1] error [2] changed [3] tr pairs
    uint32_t pool_cap = 0;
    uint32_t pool_cap_first = 0;        // what set_reads derived from the option
    uint32_t rep_pool_cap = 0;          // repeat hills (sensitive pass); starts at pool_cap_first

    // pass 2
    rala_hip::DevBuf<uint8_t> d_cls;
    rala_hip::DevBuf<uint32_t> d_death[2];
    rala_hip::DevBuf<uint32_t> d_chunk[4];      // survivors per chunk (overlaps, internals) and their scans
    rala_hip::DevBuf<uint4> d_rec;              // packed per-read records of the second pass
    rala_hip::DevBuf<uint8_t> d_crec;           // compact ones (valid region + flags, 4 or 8 bytes per read)
    rala_hip::DevBuf<uint8_t> d_fate;           // per read after the containment scan: never dies | has hills << 1
    rala_hip::DevBuf<uint64_t> d_scan_state;    // tile states of the single-pass scans (scan_pass.h)
    rala_hip::DevBuf<uint32_t> d_counts;        // small device counters of the second pass and the tail
    rala_hip::DevBuf<uint32_t> d_surv_u32[8];
    rala_hip::DevBuf<uint8_t> d_surv_u8[2];
    rala_hip::DevBuf<uint8_t> d_list_block[2];  // sharded runs: this slice's packed survivors / all slices'
    rala_hip::PinnedBuf<uint32_t> p_surv_u32[8];
    rala_hip::PinnedBuf<uint8_t> p_surv_u8[2];

    // host mirror of the per-read state (valid after initialize / construct)
    bool initialized = false, constructed = false;
    uint64_t n_prefiltered = 0;
    std::vector<uint32_t> h_begin, h_end, h_slot;
    std::vector<uint16_t> h_median, h_p10;
    std::vector<uint8_t> h_alive;
    std::vector<uint32_t> h_n_pits, h_n_hills;
    std::vector<rala_hip::Interval> h_pool;
    rala_hip::DevBuf<uint32_t> d_overflow_mid, d_overflow_long, d_chain_cnt;      // more overflow lists / chain counters
    // how initialize left the primary bound events (the sensitive pass reads them again):
    // fixed slots (d_ev_fixed, counts in d_cursor) or the CSR (d_ev_off, d_ev)
    bool ev_ready = false, ev_fixed = false;
    // bounds of the sensitive overlaps by read (CSR), the list of reads of a sensitive-pass launch
    rala_hip::DevBuf<uint32_t> d_sens_off, d_sens_cur, d_sens_ev, d_sens_list, d_sens_split;
    rala_hip::DevBuf<double> d_layout[4];
    rala_hip::DevBuf<uint32_t> d_layout_adj[2];
    // pinned staging of small device -> host reads (pipeline.hip: d2h_small / stream_sync)
    struct StagedCopy { void* dst; size_t offset, bytes; };
    rala_hip::PinnedBuf<uint32_t> p_stage;
    std::vector<StagedCopy> stage_pending;
    size_t stage_used = 0;
    int64_t use_fixed_buckets = 1;        // single-pass bucketing into fixed slots when they fit
    bool use_partitioned_buckets = true;  // target side through partitioning passes (bucket_kernels.hip) where the input suits
    rala_hip::DevBuf<uint32_t> d_bk_u32[3], d_bk_part, d_bk_group, d_bk_tiles;     // per-read counts; partition / group bases; level-2 tiles
    rala_hip::DevBuf<uint64_t> d_bk_rec[2];                            // target records after level 1 / level 2
    bool host_state_fresh = false;      // host mirrors of the per-read state match the device
    uint32_t pool_used = 0;             // interval pool records in use

    // sensitive pass (repeat hills)
    rala_hip::DevBuf<uint16_t> d_dataset_median;
    rala_hip::DevBuf<uint32_t> d_n_rep;
    // sensitive overlaps on the device: columns, transmuted target side, trimmed coordinates, tuples
    rala_hip::DevBuf<uint32_t> d_sens_col[7], d_sens_tb[2], d_sens_c[5];
    rala_hip::DevBuf<uint2> d_sens_tuples, d_sens_part, d_sens_recv;
    rala_hip::DevBuf<uint64_t> d_sens_rec;     // one context: the target bounds as 8-byte records     // as emitted / grouped by owner / received
    rala_hip::DevBuf<uint8_t> d_gather[2];     // sharded runs: this rank's block / all ranks' blocks of a gather
    rala_hip::DevBuf<uint8_t> d_sens_strand, d_sens_state;
    bool sens_in_device = false;        // option "sensitive_in_device_memory"
    rala_hip::DevBuf<uint32_t> d_rep_slot;
    rala_hip::DevBuf<rala_hip::Interval> d_rep_pool;
    std::vector<uint32_t> h_n_rep;
    std::vector<uint32_t> h_rep_slot;
    std::vector<rala_hip::Interval> h_rep_pool;
    bool have_repeats = false;
    bool rep_host_stale = false;        // the hills are on the device, the host vectors above not yet fetched
    uint32_t n_rep_hills = 0;

    // device-resident tail (tail_kernels.hip)
    bool use_gpu_tail = true;
    bool tail_on_device = false;          // results of the last construct live on the device
    bool host_stale = false;              // host mirrors (lists, graph, read state) need a download
    bool marks_on_device = false;         // transitive marks of the device graph not fetched yet
    uint32_t t_n0 = 0, t_n1 = 0, t_rounds = 0, t_n_kept = 0, t_n_nodes = 0, t_n_edges = 0, t_n_alive = 0;
    size_t t_med_tmp = 0;                 // bytes of d_med_tmp
    rala_hip::DevBuf<uint8_t> d_t_state, d_t_round, d_dirty, d_touched;
    rala_hip::DevBuf<uint32_t> d_n_pits0;
    rala_hip::DevBuf<uint16_t> d_cmed;
    rala_hip::DevBuf<uint8_t> d_med_tmp;
    rala_hip::DevBuf<uint32_t> d_cc_flags;
    rala_hip::DevBuf<uint32_t> d_kill[3], d_kill2[3], d_kill_count, d_death_sure;
    rala_hip::DevBuf<uint32_t> d_rank, d_alive_reads, d_t_tmp[2], d_kept_item, d_dovetail, d_epos, d_node_rank,
        d_node_read, d_e[3], d_t_death[2], d_t_work[2], d_t_fin, d_fp_map, d_fp_pack;
    rala_hip::DevBuf<uint8_t> d_t_mark;
    rala_hip::PinnedBuf<uint32_t> p_alive_reads, p_sens_rest;
    rala_hip::PinnedBuf<uint8_t> p_touched;
    rala_hip::PinnedBuf<uint16_t> p_cmed;

    // host tail
    std::vector<uint8_t> dirty, ever_dirty;       // reads whose valid region changed (this round / ever)
    std::vector<uint32_t> dirty_list;
    rala_hip::DevBuf<uint32_t> d_cc_edges, d_cc_label;
    rala_hip::DevBuf<uint32_t> d_tr[7];
    rala_hip::DevBuf<uint8_t> d_tr_marks;
    std::vector<rala_hip::HostOvl> overlaps, internals, scratch_ovl;
    std::vector<rala_hip::EdgePair> scratch_ep;
    std::vector<uint8_t> scratch_has, scratch_touched;
    std::vector<uint32_t> h_n_pits0;
    std::vector<uint32_t> scratch_u32a, scratch_u32b, scratch_u32c, alive_rank, alive_reads;
    rala_hip::PinnedBuf<uint32_t> p_cc_edges, p_cc_label;
    std::vector<uint32_t> node_read;
    std::vector<uint32_t> e_src, e_dst, e_len;
    std::vector<uint8_t> e_mark;

    rala_hip_timings tm = {};
};
