/*
 * librala_hip — C ABI of the MI355X (gfx950) implementation of Rala's
 * data-parallel hot path: pile-o-gram construction / annotation from
 * PAF/MHAP overlaps, overlap filtering, assembly-graph edge construction and
 * transitive-edge reduction.
 *
 * The reference (rvaser/rala, C++11) has no FFI; these entry points are what
 * a binding of its hot path would call.  Each one names the reference
 * interface it replaces (paths relative to the reference root).  All calls
 * are blocking and are made from one host thread per context, like the
 * reference's public API (src/graph.hpp:37-117).  Return value: 0 on success,
 * a negative RALA_HIP_E* code otherwise; rala_hip_last_error() gives the text.
 * No CPU fallback exists: without a usable HIP device every call fails.
 *
 * Data model.  Reads are numbered 0..n_reads-1 in sequence-file order
 * (src/graph.cpp:255-259).  Overlaps are handed over as a structure of arrays
 * in file order; a_id/b_id are read numbers (a name that is not in the
 * sequence file is passed as RALA_HIP_NO_READ: Overlap::transmute returns
 * false for it, src/overlap.cpp:44-47).  `length` is PAF column 11, or the
 * larger span for MHAP (src/overlap.cpp:16,29); `strand` is 0 for '+' /
 * equal rc flags and 1 otherwise (src/overlap.cpp:19,30).
 */
#ifndef RALA_HIP_H_
#define RALA_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RALA_HIP_NO_READ 0xFFFFFFFFu

enum {
    RALA_HIP_OK = 0,
    RALA_HIP_EDEVICE = -1,    /* HIP runtime error / no device */
    RALA_HIP_EINVAL = -2,     /* bad argument or call order */
    RALA_HIP_ECAPACITY = -3,  /* a fixed-capacity device list overflowed (raise via rala_hip_set_option) */
    RALA_HIP_EFILTERED = -4,  /* "filtered all sequences" (src/graph.cpp:418-421) */
    RALA_HIP_ENOMEM = -5,
    RALA_HIP_ENOTAFILE = -6,  /* the device tokeniser wants a regular file (a FIFO, a process substitution: take the host reader) */
    RALA_HIP_ETOOLARGE = -7   /* beyond the device tokeniser's 32-bit chunk / row counts (take the host reader) */
};

/* rala::OverlapType (src/overlap.hpp:27-33) */
enum { RALA_HIP_TYPE_X = 0, RALA_HIP_TYPE_A = 1, RALA_HIP_TYPE_B = 2, RALA_HIP_TYPE_AB = 3, RALA_HIP_TYPE_BA = 4 };

/* RALA_HIP_MEM_HOST_ASYNC (rala_hip_set_overlaps only): host memory - page-locked, if the copies are to run beside anything -
 * that stays valid until the next rala_hip_initialize has returned: the columns are then uploaded by that call, each in front
 * of the first kernel that reads it, on a stream of their own (ids -> the counting pass; b coordinates -> the first scatter;
 * a coordinates -> the query side; lengths, strands beside the pile kernels). */
enum { RALA_HIP_MEM_HOST = 0, RALA_HIP_MEM_DEVICE = 1, RALA_HIP_MEM_HOST_ASYNC = 2 };

typedef struct rala_hip_ctx rala_hip_ctx;

typedef struct rala_hip_overlaps {
    const uint32_t* a_id;
    const uint32_t* b_id;
    const uint32_t* a_begin;
    const uint32_t* a_end;
    const uint32_t* b_begin;
    const uint32_t* b_end;
    const uint32_t* length;
    const uint8_t* strand;
} rala_hip_overlaps;

/* Per-stage device time of the last rala_hip_construct / stage call, in
 * milliseconds (HIP events on the context's stream), and host time of the
 * sequential tail.  Duplicate removal runs on a second stream beside the bucketing and the
 * pile kernels: dedupe_ms is what it adds beyond them (normally 0). */
typedef struct rala_hip_timings {
    float dedupe_ms, bucket_ms, pile_ms, classify_ms, death_ms, finish_ms, tail_host_ms, tr_ms, total_ms;
    uint32_t pile_launches, death_rounds, pile_overflow_reads, pile_position_reads;
    /* reads whose slope-region / interval lists outgrew the LDS and ran with lists in global memory (initialize; the
     * sensitive pass adds its own); times an interval pool was grown and the stage repeated */
    uint32_t pile_unbounded_reads, pool_regrown;
    float repeats_ms;      /* the sensitive pass (Graph::preprocess, repeats: graph.cpp:882-1054), part of tail_host_ms */
} rala_hip_timings;

/* ---- context -------------------------------------------------------------- */
/* Replaces rala::createGraph's resource set-up (src/graph.cpp:184-238: parsers,
 * thread pool, logger) — here: device selection, one HIP stream, device arenas. */
int rala_hip_create(int device, rala_hip_ctx** out);
void rala_hip_destroy(rala_hip_ctx* ctx);
const char* rala_hip_last_error(const rala_hip_ctx* ctx);
/* options: "interval_pool_per_read_x1000" (default 1000 = one pit/hill slot per
 * read on average; a hint - a pool that turns out too small is grown to the counted need and the stage runs again), "max_lds_read_len" (position-space kernel: reads longer than this use
 * the HBM slab path), "use_run_kernel" (default 1; 0 sends every read through the
 * position-space kernel), "use_gpu_tail" (default 1; 0 runs the chimera stage of
 * Graph::preprocess on the host), "use_fixed_buckets" (default 1; 0 always buckets the bounds
 * through the exact count / scan / scatter path), "use_side_stream" (default 1; 0 runs duplicate
 * removal on the main stream before the bucketing and the pile chain's small kernels - event-dense
 * and longer reads - before its first one instead of beside it), "sensitive_in_device_memory" (default 0; 1 = the
 * sensitive overlaps handed to rala_hip_construct are device pointers), "host_threads",
 * "use_round_batches" (default 1: the containment fixed point of the second pass is finished on the device after two
 * rounds; 0 makes the host look at the killer list after every round), "use_partitioned_buckets" (default 1; 0 buckets the
 * target side through fixed slots), "debug_fp_lds_limit" (tests: containment fixed points with more killers than this
 * take the kernel for lists that do not fit the LDS), "use_fused_emit" (default 1; sharded runs: the senders scatter the
 * bounds once, by (owner, partition of the owner's reads), and the owners start at the second level of the partitioned
 * bucketing; 0: bounds grouped by owner only, bucketed by the owner from the start), "use_bound_records" (default 1; sharded
 * runs without the former: 0 ships two bound tuples per overlap side instead of one bound record),
 * "debug_part_shift" (tests / measurements, process-wide: the partitioned bucketing's first-level partitions hold 1 << value reads,
 * 12 .. 14; 0 = by the rule - 4096 reads, more where that would make more than 256 partitions),
 * "pile_chunk_mb" (default 1024: the rows of all piles lie in physical chunks of this many MB mapped side by side into one range -
 * hipMemCreate / hipMemMap - which the first pile kernel's stores like better than where one hipMalloc puts them; 0 = one hipMalloc),
 * "debug_count_window" (tests, process-wide: the partitioned bucketing counts this many groups of 128 reads per pass over the ids;
 * 0 = what a workgroup's LDS holds, 38 400 - one pass up to 4.9 M reads),
 * "debug_ev_events" (tests / measurements: 1 = the partitioned bucketing's row offsets count bound events where 4 n fits 32 bits, as
 * before round 6; 0, the default: bound pairs - up to 2^31 overlaps per context),
 * "debug_dedupe_list_cap" (tests: the list of the runs duplicate removal's counting pass marks holds this many marks, 0 = the
 * default 2^20; a list that does not hold them all is given up and the pass over all overlaps does the work),
 * "debug_pile_stop_after" (diagnostics: leave the run-space pile kernel after phase k, 99 = all;
 * 100 * m + k: the same without the row stores (m = 1), tools/phase_probe.py) */
int rala_hip_set_option(rala_hip_ctx* ctx, const char* key, int64_t value);
/* the context's hipStream_t, for callers that enqueue their own copies/collectives */
void* rala_hip_stream(rala_hip_ctx* ctx);

/* ---- inputs ------------------------------------------------------------------ */
/* Replaces the createPile loop of Graph::initialize (src/graph.cpp:249-264). */
int rala_hip_set_reads(rala_hip_ctx* ctx, const uint32_t* read_len, uint64_t n_reads);
/* Replaces the overlap stream of both parser passes (src/graph.cpp:328-382, :443-518):
 * parsed once, kept as binary SoA in HBM.  mem = RALA_HIP_MEM_HOST copies from host
 * memory; RALA_HIP_MEM_DEVICE adopts device pointers, which must stay valid; RALA_HIP_MEM_HOST_ASYNC
 * leaves the copies to rala_hip_initialize (above). */
int rala_hip_set_overlaps(rala_hip_ctx* ctx, const rala_hip_overlaps* ovl, uint64_t n, int mem);

/* The same from PAF TEXT, tokenised on the device (uncompressed files).  Replaces bioparser's PAF parser and the two
 * hash look-ups of Overlap::transmute (src/graph.cpp:328-352, src/overlap.cpp:36-82): `threads` reader threads ship the
 * file to the device in blocks (pinned staging, copies overlapped with the reads), two kernels count the lines and parse
 * them - one thread per line: names cut at the first blank and looked up in the name table, numbers as their leading
 * digits, column 11 as the overlap's length, the strand, Overlap::transmute's length check (check_lengths: the first
 * record in file order whose length differs from its sequence's is returned in *length_error_read, -1 = none; the
 * caller prints the reference's message).  The columns stay on the device and are the context's overlaps afterwards.
 * *irregular != 0: the file is not a plain list of 12-column records (a line with fewer columns, a name of more than a
 * kilobyte ...) - nothing was set, take the host reader (rala_amd/host/io.cpp), which knows what to do with such files.
 * The name table: rala::io::NameTable as built on the host (rala_amd/csrc/name_table.h: 32-byte buckets {hash32, id + 1,
 * length, arena offset, first 16 bytes}, n_buckets a power of two, the names' bytes in `arena`). */
typedef struct rala_hip_ingest_timings {
    float ship_ms;          /* file -> device memory (reads and copies overlapped) */
    float tokenize_ms;      /* count + scan + parse on the device */
    uint64_t bytes, lines;
} rala_hip_ingest_timings;
int rala_hip_set_name_table(rala_hip_ctx* ctx, const void* buckets, uint64_t n_buckets, const char* arena, uint64_t arena_bytes);
int rala_hip_set_overlaps_from_paf(rala_hip_ctx* ctx, const char* path, int check_lengths, uint32_t threads,
                                   int64_t* length_error_read, int* irregular);
/* The same for an uncompressed MHAP file (round 6): bioparser's MhapParser and the MHAP constructor of Overlap
 * (src/overlap.cpp:12-20) - twelve blank-separated columns "a_id b_id error minmers a_rc a_begin a_end a_length b_rc b_begin
 * b_end b_length", all numbers: id = the column minus one (ids are counted from 1; one that names no read does not
 * resolve), length = the longer of the two spans, strand = a_rc != b_rc, the length check of Overlap::transmute on columns
 * 8 and 12.  No name table is needed.  *irregular as above (fewer than twelve columns, ...). */
int rala_hip_set_overlaps_from_mhap(rala_hip_ctx* ctx, const char* path, int check_lengths, uint32_t threads,
                                    int64_t* length_error_read, int* irregular);
int rala_hip_get_ingest_timings(rala_hip_ctx* ctx, rala_hip_ingest_timings* out);
/* The sensitive overlaps (rala -s; Graph::preprocess, src/graph.cpp:901-939) of an uncompressed PAF file the same way, without
 * the length check (Overlap::transmute_ has none, src/overlap.cpp:84-114): the lines that start in bytes [lo, hi) of the file
 * (hi = ~0: to its end; a rank of a sharded run takes a share - any split of the sensitive set will do).  *out receives DEVICE
 * pointers that stay the context's (valid until the next call): hand them to rala_hip_construct / rala_hip_mg_run with the
 * option "sensitive_in_device_memory" set.  *irregular != 0: nothing was set, take the host reader. */
int rala_hip_tokenise_sensitive_paf(rala_hip_ctx* ctx, const char* path, uint64_t lo, uint64_t hi, uint32_t threads,
                                    rala_hip_overlaps* out, uint64_t* n, int* irregular);
/* the context's overlap columns, wherever they came from, into host buffers (*n entries each; cols / strand may be NULL to
 * ask for the count alone): a_id, b_id, a_begin, a_end, b_begin, b_end, length */
int rala_hip_get_overlap_columns(rala_hip_ctx* ctx, uint64_t* n, uint32_t* const cols[7], uint8_t* strand);

/* ---- stages ---------------------------------------------------------------------- */
/* Graph::initialize (src/graph.cpp:244-425): duplicate removal (:273-307), bound
 * emission (:311-326), Pile::add_layers for every read (:367-377), then per read
 * find_valid_region / find_median / find_chimeric_hills / find_chimeric_pits (:387-407).
 * Returns RALA_HIP_EFILTERED if no read survives. */
int rala_hip_initialize(rala_hip_ctx* ctx);
/* Graph::construct after initialize (src/graph.cpp:437-640): second overlap pass with
 * the in-order containment removal (:443-518), Graph::preprocess for chimeras
 * (:699-880), optionally Graph::preprocess for repeats with a sensitive overlap set
 * (:882-1054; pass sens = NULL / n_sens = 0 for none), node and edge construction
 * (:553-632).  Sensitive overlaps: a = query in untrimmed coordinates, b = target in
 * TRIMMED coordinates (misc/raven.sh), host memory. */
int rala_hip_construct(rala_hip_ctx* ctx, const rala_hip_overlaps* sens, uint64_t n_sens);
/* Pile::find_repetitive_hills(dataset_median) of ONE read on its current coverage (src/pile.cpp:500-566)
 * with the Pile's members begin_, end_, median_, p10_ as given: slopes at 1.42, the 0.84 / 0.9 / 0.336
 * rules; the read's repeat hills (flags cleared) are then what rala_hip_get_intervals(kind 2)
 * reports for it.  The sensitive pass of rala_hip_construct does this for all reads; this entry
 * point serves a stand-alone rala::Pile. */
int rala_hip_find_repetitive_hills(rala_hip_ctx* ctx, uint64_t read, uint32_t begin, uint32_t end, uint16_t median,
                                   uint16_t p10, uint16_t dataset_median);
/* Graph::remove_transitive_edges on the graph built by rala_hip_construct
 * (src/graph.cpp:1281-1335).  *n_pairs = its return value. */
int rala_hip_remove_transitive_edges(rala_hip_ctx* ctx, uint32_t* n_pairs);
/* The same on a caller-supplied graph (host arrays): edge e and e^1 are
 * reverse-complement twins (Edge::pair_), out-lists are in edge-id order.
 * marks[e] = 1 for every edge the reference would mark. */
int rala_hip_tr_mark(rala_hip_ctx* ctx, uint32_t n_nodes, uint32_t n_edges, const uint32_t* src,
                     const uint32_t* dst, const uint32_t* len, uint8_t* marks, uint32_t* n_pairs);

/* ---- multi-GPU building blocks (one process per GPU; the collective itself is the caller's,
 * e.g. RCCL through torch.distributed).  Reads are partitioned over ranks; rank k holds a
 * slice of the overlap file cut on a_id-run boundaries. ---------------------------------- */
/* remove_duplicate_overlaps only (src/graph.cpp:273-307) on this context's overlaps */
int rala_hip_dedupe(rala_hip_ctx* ctx);
/* A bound tuple is 8 bytes: the read in the low 32 bits, the bound ((position << 1) | is_end,
 * src/graph.cpp:317-324) in the high 32 bits - one element of the ONE all-to-all that ships every
 * bound to the owner of its read.
 * store_overlap_bounds (src/graph.cpp:311-326) as tuples: for overlap i the entries 4i..4i+3 of
 * the DEVICE buffer tuples_dev (4 * n_overlaps tuples, 16-byte aligned) receive
 * (a, (a_begin+15)<<1), (a, (a_end-15)<<1|1), (b, ...), (b, ...); the read is RALA_HIP_NO_READ
 * for records that do not resolve.  The caller routes them to the read owners. */
int rala_hip_emit_bound_tuples(rala_hip_ctx* ctx, uint64_t* tuples_dev);
/* The same tuples grouped by owner rank (owner = read % world, stored read = read / world =
 * the owner's local read number): the device buffer receives the bucket of rank 0, then rank 1,
 * ...; counts[world] (host) receives the bucket sizes.  Records that do not resolve are left
 * out. */
int rala_hip_emit_bound_tuples_bucketed(rala_hip_ctx* ctx, uint32_t world, uint64_t* tuples_dev, uint64_t* counts);
/* Feed a context whose reads are the locally owned ones with the tuples it received (read =
 * LOCAL read number; other values are ignored).  rala_hip_initialize then skips duplicate
 * removal and builds / annotates the piles from these bounds. */
int rala_hip_set_bound_tuples(rala_hip_ctx* ctx, const uint64_t* tuples, uint64_t n, int mem);
/* Bound records: both bounds of one overlap side in ONE 8-byte element - local read (22 bits, as above) << 42 | begin (21
 * bits) << 21 | end (21 bits), the raw coordinates of src/graph.cpp:317-324 (the +-15 is applied by the owner);
 * coordinates of 2^21 - 1 and more are stored as 2^21 - 1 (outside every read the format is used for).  Half the bytes
 * of the all-to-all, and the owner buckets them through the partitioned path (csrc/bucket_kernels.hip).
 * rala_hip_bound_records_fit: 1 if every read is shorter than 2^21 - 32 bases and a rank owns fewer than 2^22 reads.
 * _emit_: as rala_hip_emit_bound_tuples_bucketed, one record per overlap side (2 * n_overlaps at most).
 * _set_: as rala_hip_set_bound_tuples. */
int rala_hip_bound_records_fit(const rala_hip_ctx* ctx, uint32_t world);
int rala_hip_emit_bound_records_bucketed(rala_hip_ctx* ctx, uint32_t world, uint64_t* records_dev, uint64_t* counts);
int rala_hip_set_bound_records(rala_hip_ctx* ctx, const uint64_t* records, uint64_t n, int mem);
/* Install the result of Graph::initialize computed elsewhere (gathered from the owners) into a
 * context that holds all reads and overlaps, so that rala_hip_construct can follow.  Host
 * arrays; interval CSR as returned by rala_hip_get_intervals (kinds 0 and 1). */
int rala_hip_import_state(rala_hip_ctx* ctx, const uint8_t* valid, const uint32_t* begin, const uint32_t* end,
                          const uint16_t* median, const uint16_t* p10, const uint8_t* alive,
                          const uint64_t* pits_off, const uint32_t* pits_pairs, const uint32_t* pits_aux,
                          const uint64_t* hills_off, const uint32_t* hills_pairs);

/* Device-resident view of the result of rala_hip_initialize, for device-to-device gathers:
 * per-read arrays of n_reads entries, the interval pool (12-byte {first, second, aux} records;
 * a read's pits, then its hills, start at slot[read], or slot == 0xFFFFFFFF) and the validity
 * bytes.  rala_hip_get_device_state fills the view from a context;
 * rala_hip_import_state_device installs a (gathered) view into a context that holds all
 * reads and overlaps, like rala_hip_import_state does from host arrays. */
typedef struct rala_hip_device_state {
    const uint32_t* begin;
    const uint32_t* end;
    const uint16_t* median;
    const uint16_t* p10;
    const uint8_t* alive;
    const uint32_t* n_pits;     /* 32-bit counts: the reference's lists are vectors (pile.hpp:164-169) */
    const uint32_t* n_hills;
    const uint32_t* slot;
    const void* pool;
    uint64_t pool_count;
    const uint8_t* valid;       /* may be NULL in a tuple-fed context */
} rala_hip_device_state;
int rala_hip_get_device_state(rala_hip_ctx* ctx, rala_hip_device_state* out);
/* Copy the context's arrays into the caller's device buffers (non-NULL members of dst;
 * dst->pool_count = capacity of dst->pool in records).  Lets a caller that owns its device
 * memory (a torch tensor handed to RCCL) avoid aliasing the context's buffers. */
int rala_hip_copy_device_state(rala_hip_ctx* ctx, const rala_hip_device_state* dst);
int rala_hip_import_state_device(rala_hip_ctx* ctx, const rala_hip_device_state* in);

/* ---- sharded run over the GPUs of one node --------------------------------------------------
 * The reference fans its per-pile work out over a thread pool (src/graph.cpp:235, :367-377,
 * :387-407, ...); here reads are partitioned over P GPUs (owner(read) = read % P), the overlap
 * file is cut into P slices on a_id-run boundaries (rala_hip_mg_slice_cuts), and ONE all-to-all
 * of 8-byte bound tuples over xGMI ships every bound to the owner of its read.  Everything that
 * is per overlap (duplicate removal, bound emission, trim / type, liveness, hill counters) runs on
 * the slice; everything that is per read (piles, annotation) on the owner; the in-order
 * containment scan is a fixed point whose bounds are all-reduced (min) per round; the survivors
 * (about 1 % of the overlaps) are all-gathered and the preprocess tail, the graph and the
 * transitive reduction run replicated on them.
 *
 * One rala_hip_mg object per rank.  Ranks are either processes (one per GPU, RCCL: rank 0 obtains
 * a 128-byte id with rala_hip_mg_unique_id and ships it to the others by any means) or host
 * threads of one process (RCCL with an id made in that process, or the in-process transport
 * RALA_HIP_COMM_LOCAL, which also lets several ranks share one device - how the decomposition
 * is tested on a single GPU).  rala_hip_mg_create and rala_hip_mg_run are collective: every rank
 * calls them.  After a run, rala_hip_mg_context(mg) holds the replicated result: all getters
 * of this header work on it, except rala_hip_get_pile_data (the coverage of read r lives on
 * rank r % P: rala_hip_mg_get_pile_data). */
enum { RALA_HIP_COMM_RCCL = 0, RALA_HIP_COMM_LOCAL = 1 };
typedef struct rala_hip_mg rala_hip_mg;
typedef struct rala_hip_mg_timings {
    /* wall-clock milliseconds of this rank's last run, by step */
    float emit_ms, exchange_ms, owner_ms, gather_ms, construct_ms, repeats_ms, tr_ms, total_ms;
    uint64_t tuples_sent;       /* bound tuples this rank shipped to other ranks */
} rala_hip_mg_timings;
int rala_hip_mg_unique_id(void* id128);
int rala_hip_mg_local_group_create(uint32_t world, void** group);
void rala_hip_mg_local_group_destroy(void* group);
/* token: the 128-byte id (RALA_HIP_COMM_RCCL) or the group (RALA_HIP_COMM_LOCAL) */
int rala_hip_mg_create(int device, uint32_t rank, uint32_t world, int transport, const void* token, rala_hip_mg** out);
/* The same in two steps: the rank's device contexts (NOT collective), then joining the group (collective: ncclCommInitRank).
 * A launcher creates all contexts first, makes sure every rank has its own, and only then lets them join - a rank that
 * failed before the collective part would leave the others waiting inside it. */
int rala_hip_mg_create_contexts(int device, uint32_t rank, uint32_t world, rala_hip_mg** out);
int rala_hip_mg_join(rala_hip_mg* mg, int transport, const void* token);
void rala_hip_mg_destroy(rala_hip_mg* mg);
const char* rala_hip_mg_last_error(const rala_hip_mg* mg);
/* all read lengths, on every rank (src/graph.cpp:249-264) */
int rala_hip_mg_set_reads(rala_hip_mg* mg, const uint32_t* read_len, uint64_t n_reads);
/* cuts[world + 1]: slice k = records cuts[k] .. cuts[k + 1] of the file; a cut never splits a run of
 * equal a_id, and records that do not resolve (query or target RALA_HIP_NO_READ) do not break a run
 * (src/graph.cpp:338-350).  b_id may be NULL when every target is known. */
int rala_hip_mg_slice_cuts(const uint32_t* a_id, const uint32_t* b_id, uint64_t n, uint32_t world, uint64_t* cuts);
/* this rank's slice; first = file position of its record 0 */
int rala_hip_mg_set_overlaps(rala_hip_mg* mg, const rala_hip_overlaps* slice, uint64_t n, uint64_t first, int mem);
/* The same from PAF TEXT (an uncompressed file), collective: rank k ships bytes [n k / P, n (k + 1) / P) of the file to its own
 * GPU and tokenises the lines that start there (rala_hip_set_overlaps_from_paf's kernels; the name table and the reads must be
 * set on rala_hip_mg_context(mg)); the ranks exchange their row counts and the queries at their ends, compute the same cuts
 * between runs of equal queries (rala_hip_mg_slice_cuts' rule) and move the rows in front of the cuts to the rank that holds
 * the run's start.  Replaces, for N GPUs, bioparser's parser in front of Graph::initialize (src/graph.cpp:328-382).
 * *length_error_read, *irregular: as rala_hip_set_overlaps_from_paf (the same values on every rank). */
int rala_hip_mg_set_overlaps_from_paf(rala_hip_mg* mg, const char* path, int check_lengths, uint32_t threads, int64_t* length_error_read,
                                      int* irregular);
/* the rank's slice as it was set: file position of its record 0, records (the columns: rala_hip_get_overlap_columns on
 * rala_hip_mg_context(mg)) */
int rala_hip_mg_get_slice(rala_hip_mg* mg, uint64_t* first, uint64_t* n);
/* Graph::construct + remove_transitive_edges (src/graph.cpp:427-640, :1281-1335); sens_slice = this
 * rank's share of the sensitive overlaps (any contiguous share, n_sens = 0 allowed; NULL on EVERY
 * rank for a run without them - ranks that disagree get RALA_HIP_EINVAL).  A failure of one rank
 * alone (out of memory, a device error) aborts the group: every rank's call fails and the rank
 * objects must be destroyed. */
int rala_hip_mg_run(rala_hip_mg* mg, const rala_hip_overlaps* sens_slice, uint64_t n_sens, uint32_t* n_pairs);
/* the same for n ranks of this process, one host thread per rank, joined before it returns */
int rala_hip_mg_run_threads(rala_hip_mg** ranks, uint32_t n, const rala_hip_overlaps* sens_slices, const uint64_t* n_sens,
                            uint32_t* n_pairs);
rala_hip_ctx* rala_hip_mg_context(rala_hip_mg* mg);
/* the context of the reads this rank owns (local read j = read j * P + rank): stage timings of the pile kernels */
rala_hip_ctx* rala_hip_mg_owner_context(rala_hip_mg* mg);
int rala_hip_mg_get_pile_data(rala_hip_mg* mg, uint64_t read, uint16_t* data);
/* rala_hip_get_pile_row_digests (below) of the rows this rank owns: n_owned = the reads r with r % P == rank, entry j = read
 * j * P + rank, under the final valid regions */
int rala_hip_mg_get_pile_row_digests(rala_hip_mg* mg, uint64_t* fnv, uint64_t* inside, uint64_t* outside);
int rala_hip_mg_get_timings(rala_hip_mg* mg, rala_hip_mg_timings* out);

/* Force-directed layout of one connected component, the O(n^2) part of Graph::postprocess
 * (src/graph.cpp:1132-1226): n points x, y (host, in / out); the attraction partners of point
 * i are adj[adj_off[i] .. adj_off[i + 1]) (point indices, n = a point fixed at the origin: a
 * neighbour outside the component); `iterations` steps with step length t, decreased by dt
 * after every step, spring constant k.  Same arithmetic, in the same order, as the reference's
 * per-point task. */
int rala_hip_layout(rala_hip_ctx* ctx, uint32_t n, double* x, double* y, const uint32_t* adj_off, const uint32_t* adj,
                    uint32_t iterations, double k, double t, double dt);

/* ---- results (host buffers owned by the caller) --------------------------------------- */
/* is_valid_overlap_ (src/graph.hpp:168), one byte per overlap */
int rala_hip_get_valid(rala_hip_ctx* ctx, uint8_t* valid);
/* Pile::begin/end/median/p10 and liveness (piles_[i] != nullptr) of every read, as they
 * stand after the last completed stage.  Any pointer may be NULL. */
int rala_hip_get_piles(rala_hip_ctx* ctx, uint32_t* begin, uint32_t* end, uint16_t* median, uint16_t* p10,
                       uint8_t* alive);
/* Pile::data() of one read (src/pile.hpp:53): read_len[read] values.  Contents are
 * defined for reads that passed find_valid_region. */
int rala_hip_get_pile_data(rala_hip_ctx* ctx, uint64_t read, uint16_t* data);
/* Checksums of EVERY pile row where it lies in device memory (Pile::data() of all reads is 20 GB at 1 M reads: nothing a
 * caller wants copied), n_reads entries each, any pointer may be NULL: fnv[r] = FNV-1a-64 over the bytes of Pile::data() of
 * read r (src/pile.hpp:53: uint16 values, low byte first, zero outside the valid region as Pile::shrink leaves them,
 * src/pile.cpp:311-318) - what rala_hip_get_pile_data would return, hashed on the device; inside[r] = the sum of the row over
 * [begin, end); outside[r] = the sum of the values STORED outside it (zero right after rala_hip_initialize; later stages narrow a
 * region without rewriting its row).  A filtered read answers 0 to all three. */
int rala_hip_get_pile_row_digests(rala_hip_ctx* ctx, uint64_t* fnv, uint64_t* inside, uint64_t* outside);
/* Pits / hills / repeat hills of all reads as CSR: offsets[n_reads + 1], then pairs
 * (first, second) and one aux word per interval (pit: min coverage inside; hill:
 * spanning-overlap count; repeat hill: bridged flag).  kind: 0 pits, 1 hills, 2 repeat
 * hills.  Call with pairs = NULL to obtain offsets only. */
int rala_hip_get_intervals(rala_hip_ctx* ctx, int kind, uint64_t* offsets, uint32_t* pairs, uint32_t* aux);
/* Overlaps kept for the graph (which = 0) or as internals (which = 1) after construct.
 * Returns the count through *n; arrays may be NULL to query the count. */
int rala_hip_get_overlaps(rala_hip_ctx* ctx, int which, uint64_t* n, uint32_t* src_index, uint32_t* a_begin,
                          uint32_t* a_end, uint32_t* b_begin, uint32_t* b_end, uint32_t* length, uint8_t* type);
/* Assembly graph: node k belongs to read node_read[k] (k odd = reverse complement);
 * edges in id order with twin e^1; marks as set by rala_hip_remove_transitive_edges. */
int rala_hip_get_graph_size(rala_hip_ctx* ctx, uint64_t* n_nodes, uint64_t* n_edges);
int rala_hip_get_graph(rala_hip_ctx* ctx, uint32_t* node_read, uint32_t* src, uint32_t* dst, uint32_t* len,
                       uint8_t* marks);
int rala_hip_get_timings(rala_hip_ctx* ctx, rala_hip_timings* out);
/* number of reads dropped by find_valid_region (src/graph.cpp:409-424) */
int rala_hip_get_num_prefiltered(rala_hip_ctx* ctx, uint64_t* n);

#ifdef __cplusplus
}
#endif

#endif /* RALA_HIP_H_ */
