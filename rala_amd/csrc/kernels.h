// Kernel argument blocks and launchers of librala_hip (internal header).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rala_hip {

// zero padding in front of a pile inside the LDS image (multiple of 8 so the
// pile itself starts 16-byte aligned); >= the 847-base slope window
constexpr uint32_t kPadL = 848;

enum : uint32_t {
    kErrRegionCapacity = 1u,   // more slope regions than the kernel's lists hold
    kErrPoolCapacity = 2u,     // interval pool exhausted (the pool's counter holds what is needed)
    kErrRawCapacity = 4u,      // more raw intervals / candidate pairs than the kernel's lists hold
};
// Neither is an error of the call: a read that outgrows the LDS lists of the position-space kernels is noted in
// big_list and runs again with its lists in global memory, grown until it fits; a pool that is too small is grown
// to the counted size and the stage runs again (the reference keeps all of these in vectors, pile.hpp:164-169).

// one pit / hill / repeat hill of a read
struct Interval {
    uint32_t first, second;
    uint32_t aux;   // pit: min coverage inside [first, second]; hill: spanning-overlap counter
};

struct PileArgs {
    // inputs
    const uint32_t* read_len;
    const uint64_t* pile_off;      // element offset of each pile row (rows padded to 8 elements)
    uint16_t* pile;                // all piles, row after row
    const uint32_t* ev_off;        // CSR of bound events per read (n_reads + 1), in units of 1 << ev_shift events
    const uint32_t* ev;            // pos << 1 | is_end
    const uint32_t* ev_cnt;        // non-null: fixed slots instead of the CSR - read r has ev_cnt[r] events
    uint32_t ev_stride;            //           at ev + r * ev_stride
    uint32_t rows_chunked = 0;     // the rows lie in mapped chunks (pipeline.hip): the first kernel stores them non-temporally
    uint32_t ev_shift = 0;         // 1: ev_off counts bound PAIRS (the partitioned bucketing, round 6: 2^31 overlaps instead of 2^30)
    const uint32_t* order;         // reads of this launch
    uint32_t n_items;
    const uint32_t* n_items_dev;   // when non-null the item count is read from device memory
    uint32_t lw;                   // uint16 elements per big array (>= kPadL + len + 848, multiple of 8)
    uint32_t add_to_existing;
    uint32_t stop_after;           // diagnostics: leave the kernel after phase k (99 = run everything)
    uint16_t* slab;                // HBM scratch, 3 * lw elements per workgroup (long reads only)
    // outputs, indexed by read
    uint32_t* begin;
    uint32_t* end;
    uint16_t* median;
    uint16_t* p10;
    uint8_t* alive;
    uint32_t* n_pits;              // (32-bit counts: the reference's lists are vectors, pile.hpp:164-169)
    uint32_t* n_hills;
    uint32_t* iv_slot;             // first pool entry of the read (pits, then hills), ~0u if none
    Interval* pool;
    uint32_t* pool_count;
    uint32_t pool_cap;
    uint32_t* error;
    // sensitive pass (launch_pile_sens): begin / end are inputs; mode 1 writes median / p10 and
    // the row, mode 2 reads them and writes the repeat hills
    const uint32_t* sens_off = nullptr;        // CSR of the sensitive bounds per read
    const uint32_t* sens_ev = nullptr;
    const uint16_t* dataset_median = nullptr;  // mode 2: component median per read
    uint32_t* n_rep = nullptr;
    uint32_t* rep_slot = nullptr;
    Interval* rep_pool = nullptr;
    uint32_t* rep_pool_count = nullptr;
    uint32_t rep_pool_cap = 0;
    // position-space kernel: reads whose lists outgrow the LDS are appended here (null: error bits instead) ...
    uint32_t* big_list = nullptr;
    uint32_t* big_count = nullptr;
    // ... and run again with the lists in global memory: ListSpace::words(..) words per workgroup
    uint32_t* big_space = nullptr;
    uint32_t big_cap_reg = 0, big_cap_list = 0, big_cap_raw = 0;
    uint32_t force_big = 0;        // tests: every read of the position-space kernel takes that way
    // first pass: reads with more than kRunEventCap events are listed beforehand
    // (launch_pile_dense_list) and start in the cap-1024 kernel; the cap-512 kernels pass them over
    uint32_t skip_dense = 0;
    uint32_t variant = 0;          // measurements (a library built with -DRALA_PILE_AB): which variant of the first kernel runs
};

uint32_t pile_lds_bytes(uint32_t lw);
uint32_t pile_lw_for(uint32_t read_len);
uint64_t pile_big_words(uint32_t cap_reg, uint32_t cap_list, uint32_t cap_raw);    // per workgroup
void launch_pile_build_annotate(const PileArgs& args, uint32_t grid, bool in_lds, hipStream_t stream);

// verify_kernels.hip: per row FNV-1a-64 of Pile::data() (zero outside [begin, end)), the row's sum inside and the stored values' sum
// outside the region; a dead read answers 0 to all three.  Any output may be null.
void launch_pile_row_digests(const uint16_t* pile, const uint64_t* pile_off, const uint32_t* read_len, const uint32_t* begin,
                             const uint32_t* end, const uint8_t* alive, uint32_t n_rows, uint64_t* fnv, uint64_t* inside,
                             uint64_t* outside, hipStream_t stream);

// run-space kernel (pile_runs_kernel.hip): one wavefront per read; reads with more bound
// events than the instantiation's cap (or, with args.skip_dense, only those whose region lists do
// not fit) are appended to overflow_list instead (args.order may be null = identity).  Event caps
// 512 / 1024 / 2048 (tiers 0, 1, 2): the larger ones keep longer lists in LDS and run at lower
// occupancy; cap 512 comes in three layouts by read length (tiers 0 / 3 / 4, below).
constexpr uint32_t kRunEventCap = 512;
constexpr uint32_t kRunEventCapMid = 1024;
constexpr uint32_t kRunEventCapBig = 2048;
// length classes of the pile chain: the first kernel's bitmap, the second's (tier 3); longer reads
// take the cap-512 kernel for any length.  (A third bitmap class, up to 65535 bases at two and a
// half wavefronts per SIMD, was 3 % faster than that kernel on 50 kb reads - not kept.)
constexpr uint32_t kPileClasses = 3;
constexpr uint32_t kPileClassBases[kPileClasses - 1] = {16384u, 32768u};
// tier 0: cap 512 (one workgroup per read where the grid allows: reads of up to 16384 bases only),
// 3: cap 512 for reads of up to 32768 bases, 4: cap 512 for any length, 1: cap 1024, 2: cap 2048
void launch_pile_dense_list(const PileArgs& args, uint32_t n_reads, uint32_t* list, uint32_t* count, hipStream_t stream);
void launch_pile_runs(const PileArgs& args, uint32_t grid, int tier, uint32_t* overflow_list, uint32_t* overflow_count,
                      hipStream_t stream);
// the sensitive pass in run space (tier 0: cap 512, tier 1: cap 1024; reads of up to 16384 bases); the others are appended to
// overflow_list for the position-space kernel below
void launch_pile_sens(const PileArgs& args, uint32_t grid, int tier, int mode, uint32_t* overflow_list,
                      uint32_t* overflow_count, hipStream_t stream);

// sensitive pass (pile_repeats_kernel.hip): mode 1 = add layers on top + median for the
// targets; mode 2 = repeat hills for the members of connected components
struct RepeatArgs {
    const uint32_t* read_len;
    const uint64_t* pile_off;
    uint16_t* pile;
    const uint32_t* ev_off;            // mode 1: CSR of the new bounds per read
    const uint32_t* ev;
    const uint32_t* order;
    uint32_t n_items;
    uint32_t lw;
    uint16_t* slab;
    const uint32_t* begin;             // current valid regions
    const uint32_t* end;
    uint16_t* median;                  // mode 1 writes, mode 2 reads
    uint16_t* p10;
    const uint16_t* dataset_median;    // mode 2: component median per read
    uint32_t* n_rep;                   // mode 2 outputs
    uint32_t* rep_slot;
    Interval* pool;
    uint32_t* pool_count;
    uint32_t pool_cap;
    uint32_t* error;
    // as in PileArgs (cap_raw also bounds the candidate pairs)
    uint32_t* big_list = nullptr;
    uint32_t* big_count = nullptr;
    uint32_t* big_space = nullptr;
    uint32_t big_cap_reg = 0, big_cap_list = 0, big_cap_raw = 0;
    uint32_t force_big = 0;
};
uint32_t repeats_lds_bytes(uint32_t lw);
uint64_t repeats_big_words(uint32_t cap_reg, uint32_t cap_list, uint32_t cap_raw);
void launch_pile_repeats(const RepeatArgs& args, uint32_t grid, bool in_lds, int mode, hipStream_t stream);

// ---- overlap-side kernels (overlap_kernels.hip) ------------------------------
struct OvlSoA {
    const uint32_t *a_id, *b_id, *a_begin, *a_end, *b_begin, *b_end, *length;
    const uint8_t* strand;
    uint64_t n;
    uint64_t base;      // file position of record 0 (a rank of a sharded run holds a slice of the file)
};

// trimmed coordinates of the sensitive overlaps (second pass, -s) + kept / dropped
struct SensCoords {
    uint32_t *a_begin, *a_end, *b_begin, *b_end, *length;
    uint8_t* state;
};

struct ReadState {
    const uint32_t* begin;
    const uint32_t* end;
    const uint8_t* alive;
    const uint32_t* n_pits;
    const uint32_t* n_hills;
    const uint32_t* iv_slot;
    Interval* pool;
};

// suspect: n_reads bytes of scratch (queries whose runs are not strictly ordered by target)
void launch_dedupe(const OvlSoA& o, uint32_t n_reads, uint8_t* suspect, uint8_t* valid, hipStream_t s);
// Single-pass bucketing into fixed slots of `stride` events per read: the slot index comes
// straight from the counting atomic, so there is no scan and no second pass.  *over is set
// when a read has more events than a slot holds (the caller then falls back to the CSR path).
void launch_bucket_fixed(const OvlSoA& o, uint32_t n_reads, uint32_t stride, uint32_t* counts, uint32_t* ev_fixed,
                         uint32_t* over, hipStream_t s);
void launch_bucket_fixed_tuples(const uint2* tuples, uint64_t n, uint32_t n_reads, uint32_t stride, uint32_t* counts,
                                uint32_t* ev_fixed, uint32_t* over, hipStream_t s);
// counts -> (exclusive scan) -> ev_off; rank_a / rank_b: n_overlaps each, slot of the overlap's
// bounds inside the bucket of read a / read b
void launch_count_bounds(const OvlSoA& o, uint32_t n_reads, uint32_t* counts, uint32_t* rank_a, uint32_t* rank_b,
                         hipStream_t s);
void launch_scatter_bounds(const OvlSoA& o, uint32_t n_reads, const uint32_t* ev_off, const uint32_t* rank_a,
                           const uint32_t* rank_b, uint32_t* ev, hipStream_t s);
// bound tuples {x = read, y = bound} (8 bytes) instead of overlaps: multi-GPU owners receive them
// by all-to-all
void launch_emit_tuples(const OvlSoA& o, uint32_t n_reads, uint2* tuples, hipStream_t s);
// records: one bound record {local read : 22 | begin : 21 | end : 21} per overlap side instead of two tuples
void launch_bucket_tuples(const OvlSoA& o, uint32_t n_reads, uint32_t world, uint32_t pass, uint32_t* counters,
                          uint2* tuples, hipStream_t s, bool records = false);
void launch_records_to_tuples(const uint64_t* records, uint64_t n, uint2* tuples, hipStream_t s);
void launch_owner_offsets(const uint32_t* count, uint32_t world, uint32_t* cursor, hipStream_t s);
constexpr uint32_t kBoundRecordCoordBits = 21, kBoundRecordReadBits = 22;
void launch_count_tuples(const uint2* tuples, uint64_t n, uint32_t n_reads, uint32_t* counts, hipStream_t s);
// sensitive overlaps (graph.cpp:882-1054): transmute_ + target bounds as tuples 2i, 2i + 1;
// first trim; dovetails mark the repeat hills they bridge
void launch_sens_tuples(const OvlSoA& o, uint32_t n_reads, const uint32_t* begin, const uint8_t* alive, uint32_t* tb_begin,
                        uint32_t* tb_end, uint2* tuples, uint32_t* error, hipStream_t s);
// the same with ONE 8-byte bound record {read : 22, begin : 21, end : 21} per overlap instead of two tuples (the format of the
// owners' bound records: launch_bucket_partitioned_records takes it); a record of an overlap that is an error names no read
void launch_sens_records(const OvlSoA& o, uint32_t n_reads, const uint32_t* begin, const uint8_t* alive, uint32_t* tb_begin,
                         uint32_t* tb_end, uint64_t* records, uint32_t* error, hipStream_t s);
void launch_sens_trim(const OvlSoA& o, const uint32_t* tb_begin, const uint32_t* tb_end, const uint32_t* begin,
                      const uint32_t* end, const uint8_t* alive, const SensCoords& out, hipStream_t s);
void launch_sens_bridge(const OvlSoA& o, const SensCoords& sc, const uint32_t* begin, const uint32_t* end,
                        const uint8_t* alive, const uint32_t* n_rep, const uint32_t* rep_slot, Interval* rep_pool,
                        hipStream_t s);
void launch_scatter_tuples(const uint2* tuples, uint64_t n, uint32_t n_reads, uint32_t* cursor, uint32_t* ev,
                           hipStream_t s);
// overlaps that would delete a read (contained read, container without pits / hills)
struct KillList {
    uint32_t* count;            // device counter, zeroed before classify
    uint32_t *ovl, *target, *keeper;
};
// per-read records of the second pass: rec = {begin, n_hills, n_pits, pool slot};
// crec = valid region + "chimeric" + "has hills" in 4 bytes (small_records: no read longer than 32767
// bases) or 8; sure[r] = 0 for the reads that are gone already (sure is all ones on entry)
size_t compact_record_bytes(bool small_records);
void launch_pack_reads(const ReadState& rs, uint32_t n_reads, uint4* rec, void* crec, bool small_records, uint32_t* sure,
                       hipStream_t s);
uint32_t pass2_chunks(uint64_t n_overlaps);     // workgroups (= chunk counters) of survivor_masks / gather
// lo[n_reads], all ones on entry: lo[r] = the first overlap that would delete read r (the fixed point's
// first lower bound).  File positions are 1-based in the killer list and in lo / up / sure.
void launch_classify(const OvlSoA& o, uint32_t n_reads, const uint8_t* valid, const void* crec, bool small_records,
                     const KillList& kl, uint32_t* lo, hipStream_t s);
// containment fixed point on the killer list (overlap_kernels.hip): lo[t] = min(lo[t], i) over a
// list; one decision round (sure killers -> sure[], undecided ones -> out)
// (at_most: a bound of the list's length known to the host, sizes the grid; the length itself is on the device)
void launch_death_lower(const KillList& kl, uint32_t* lo, hipStream_t s, uint64_t at_most = ~0ull);
void launch_death_decide(const KillList& in, const uint32_t* lo, const uint32_t* up, uint32_t* sure, const KillList& out,
                         hipStream_t s, uint64_t at_most = ~0ull);
// *changed = 1 if the two differ anywhere; `older` is then filled with 0xFFFFFFFF (the next round's output)
void launch_death_diff(uint32_t* older, const uint32_t* newer, uint32_t n, uint32_t* changed, hipStream_t s);
// (count, status: launch_death_status in the same launch)
void launch_death_tighten(const uint32_t* sure, uint32_t* up, uint32_t* lo, uint32_t n, hipStream_t s, const uint32_t* count = nullptr,
                          uint32_t* status = nullptr);
// liveness + hill span counters + who survives (one mask word per 64 overlaps and kind, counts per chunk)
void launch_survivor_masks(const OvlSoA& o, uint32_t n_reads, const uint8_t* valid, const uint8_t* fate, const uint32_t* death, const void* crec, bool small_records,
                           const uint4* rec, Interval* pool, uint64_t* mask_ov, uint64_t* mask_in, uint32_t* chunk_ov,
                           uint32_t* chunk_in, hipStream_t s);
// alive[r] = 0 where death[r] is set; fate[r] = "never dies" | "has hills" << 1; *n_alive += the reads that are left
void launch_apply_death(const uint32_t* death, uint8_t* alive, const uint32_t* n_hills, uint8_t* fate, uint32_t n_reads,
                        uint32_t* n_alive, hipStream_t s);

// survivors gathered into dense arrays (trim re-applied against the pass-1 piles)
struct Survivors {
    uint32_t *src, *a_id, *b_id, *a_begin, *a_end, *b_begin, *b_end, *length;
    uint8_t *strand, *type;
};
void launch_gather_survivors(const OvlSoA& o, const uint64_t* mask_ov, const uint64_t* mask_in, const void* crec,
                             bool small_records, const uint32_t* chunk_ov_off, const uint32_t* chunk_in_off, uint32_t n_ov_total,
                             const Survivors& out, hipStream_t s);
// sharded runs (one slice of the overlap file per rank)
struct ListBlocks {             // the ranks' packed survivor blocks inside one gathered buffer
    uint64_t block_off[64];
    uint32_t n0[64], n1[64];    // overlaps / internals of the rank
    uint32_t dst0[64], dst1[64];
    uint32_t world;
};
struct RankOffsets {
    uint32_t v[64];
};
// shard_kernels.hip
void launch_partition_tuples(const uint2* in, uint64_t n, uint32_t world, uint32_t pass, uint32_t* counters, uint2* out,
                             hipStream_t s);
void launch_localize_u32(const uint32_t* global, uint64_t n_local, uint32_t world, uint32_t rank, uint32_t* local, hipStream_t s);
void launch_localize_u16(const uint16_t* global, uint64_t n_local, uint32_t world, uint32_t rank, uint16_t* local, hipStream_t s);
void launch_localize_u8(const uint8_t* global, uint64_t n_local, uint32_t world, uint32_t rank, uint8_t* local, hipStream_t s);
void launch_pack_median(const uint16_t* median, const uint16_t* p10, uint64_t n_local, uint64_t nl_pad, uint32_t* out, hipStream_t s);
void launch_unpack_median(const uint32_t* all, uint32_t world, uint64_t nl_pad, uint64_t n_reads, uint16_t* median, uint16_t* p10,
                          hipStream_t s);
void launch_pack_rep(const uint32_t* n_rep, const uint32_t* rep_slot, uint64_t n_local, uint64_t nl_pad, uint64_t* out, hipStream_t s);
void launch_unpack_rep(const uint64_t* all, uint32_t world, uint64_t nl_pad, uint64_t n_reads, const RankOffsets& pool_base,
                       uint32_t* n_rep, uint32_t* rep_slot, hipStream_t s);
void launch_pool_aux(Interval* pool, uint32_t n, uint32_t* dense, uint32_t mode, hipStream_t s);
void launch_death_status(const uint32_t* count, uint32_t* status, hipStream_t s);
// a rank's undecided killers padded with inert entries to `each` (block: 3 * each words), and all ranks' blocks back into lists
void launch_pack_killers(const uint32_t* key, const uint32_t* target, const uint32_t* keeper, const uint32_t* count, uint32_t each,
                         uint32_t* block, hipStream_t s);
void launch_unpack_killers(const uint32_t* blocks, uint32_t world, uint32_t each, uint32_t* key, uint32_t* target, uint32_t* keeper,
                           uint32_t* count, hipStream_t s);
void launch_hill_counts(const ReadState& rs, uint32_t n_reads, uint32_t* dense, uint32_t mode, hipStream_t s);
void launch_unpack_lists(const uint8_t* blocks, const ListBlocks& lb, const Survivors& out, hipStream_t s);

// ---- preprocess tail on device-resident lists (tail_kernels.hip) -------------------
struct TailList {               // survivors of the second pass: overlaps first, then internals
    uint32_t n;
    uint32_t *src, *a, *b, *a_begin, *a_end, *b_begin, *b_end, *length;
    uint8_t *strand, *type;
    uint8_t* state;             // 0 dead, 1 overlap, 2 internal, 3 internal promoted to overlap
    uint8_t* round;             // round of the promotion
};
struct TailReads {
    uint32_t *begin, *end;
    uint8_t *alive, *dirty;
    uint32_t *n_pits, *n_hills;
    const uint32_t* n_pits0;     // pit count the pile kernel wrote (hills sit behind those pits)
    const uint32_t* iv_slot;
    Interval* pool;
};
// sensitive pass on the device list (sens_kernels.hip)
void launch_sens_filter(const TailList& L, const uint32_t* begin, const uint32_t* end, const uint32_t* n_rep,
                        const uint32_t* rep_slot, const Interval* rep_pool, hipStream_t s);
void launch_finalize_states(const TailList& L, const uint8_t* alive, hipStream_t s);
// the read lists of the sensitive pass, made where the data is (*count zeroed by the caller): reads that received sensitive
// bounds; alive reads with an overlap (of a sharded run: this rank's, as local ids)
void launch_list_targets(const uint32_t* off, uint32_t n, uint32_t* list, uint32_t* count, hipStream_t s);
void launch_list_members(const uint32_t* alive_reads, const uint8_t* touched, uint32_t n_alive, uint32_t world, uint32_t rank,
                         uint32_t* list, uint32_t* count, hipStream_t s);
// The reads of a sensitive-pass list by the kernel that takes them (known beforehand from length and event counts, as in
// the first pass): class 0 up to 16384 bases and kRunEventCap - 2 events (primary + sensitive), 1 up to 32768 bases and as
// many events, 2 up to 16384 bases and kRunEventCapMid - 2 events, 3 up to 16384 bases and kRunEventCapBig - 2 events (round 5:
// the event-dense targets, a third of the pass at C5 while they ran in position space), 4 the rest (position space).  out[c] receives the
// class' reads, counts[c] (zeroed by the caller) how many.  A hand-over through the kernels' overflow lists costs one add
// to ONE counter per read: at C5 300 000 of them, 3 ms per kernel.
struct SensSplitArgs {
    const uint32_t* read_len;
    const uint32_t* ev_off;        // primary events: CSR (units of 1 << ev_shift events), or
    const uint32_t* ev_cnt;        // fixed slots (non-null)
    uint32_t ev_stride;
    uint32_t ev_shift;
    const uint32_t* sens_off;
    const uint32_t* begin;
    const uint32_t* end;
    uint32_t* out[5];
    uint32_t* counts;              // five words
};
// (the list's length is on the device: *n_dev, at most `bound`)
void launch_sens_split(const uint32_t* list, uint32_t bound, const uint32_t* n_dev, const SensSplitArgs& args, hipStream_t s);
// *status &= ~bits; *zero = 0 (zero may be null)
void launch_status_clear(uint32_t* status, uint32_t bits, uint32_t* zero, hipStream_t s);
void launch_scatter_component_medians(const uint32_t* alive_reads, const uint8_t* touched, const uint16_t* cmed, uint32_t n_alive,
                                      uint16_t* out, hipStream_t s);
void launch_break_hills(const TailReads& R, uint32_t n_reads, hipStream_t s);
// gate: a device word; the kernel does nothing when it holds 0 (a round enqueued before the host knows whether
// the loop goes on); *dropped is set to 1 if an overlap died
void launch_break_pits(const TailReads& R, const uint32_t* alive_reads, const uint8_t* touched, const uint16_t* comp_median,
                       uint32_t n_alive, hipStream_t s, const uint32_t* gate = nullptr);
void launch_retrim(const TailList& L, const TailReads& R, uint32_t promote, uint32_t round, uint32_t* dropped,
                   hipStream_t s, const uint32_t* gate = nullptr);
// (+ label[v] = v for v < n_labels: the components' start)
void launch_cc_edges(const TailList& L, const uint32_t* rank, uint32_t* edges, uint8_t* touched, uint32_t* label, uint32_t n_labels,
                     hipStream_t s);
void launch_refresh_types(const TailList& L, const TailReads& R, hipStream_t s);
// both in-order containment scans of the tail (graph.cpp:831-877) without a look from the host (tail_kernels.hip), stale
// types refreshed on the way (launch_refresh_types' work).  lists:
// six arrays of L.n words (killers, conditional killers); zeroed21: 21 zeroed words ([0] is set when a fixed point failed); work: four arrays of n_reads words, uninitialised; base2: 2 * n_reads words, all ones, and mark2: 2 * n_reads
// bytes, zero (tail_init); map, pack: launch_fixed_point_finish's
hipError_t launch_tail_contain(const TailList& L, const TailReads& R, uint8_t* alive, uint32_t* const lists[6], uint32_t* zeroed21, uint32_t* const work[4],
                               uint32_t* base2, uint8_t* mark2, uint32_t* map, uint32_t* pack, uint32_t n_reads, uint32_t lds_limit, hipStream_t s);
void launch_count_zero_u8(const uint8_t* x, uint32_t n, uint32_t* out, hipStream_t s);    // *out += #zeros
// list states (the first n0 items are overlaps, the rest internals), dirty[] = 0, n_pits0[] = n_pits[], base2[0 .. 2 n_reads) =
// all ones, mark2[0 .. 2 n_reads) = 0, map[0 .. n_reads) = all ones, zero22[0 .. 21] = 0 in one launch
void launch_tail_init(const TailList& L, uint32_t n0, const TailReads& R, uint32_t* n_pits0, uint32_t n_reads, uint32_t* base2,
                      uint8_t* mark2, uint32_t* map, uint32_t* zero22, hipStream_t s);
// single-pass scans with producer and consumer inside (scan_pass.h); false = out of tile states
struct ScanSpace;
bool launch_rank_pass(const uint8_t* alive, uint32_t* rank, uint32_t* alive_reads, uint32_t n_reads, ScanSpace& space, hipStream_t s);
bool launch_node_pass(const uint8_t* alive, uint32_t* node_rank, uint32_t* node_read, uint32_t* n_final, uint32_t n_reads,
                      ScanSpace& space, hipStream_t s);
// one segment of the final overlap list + its edges; base[0 .. 1] = kept items / dovetails in front of
// the segment, next[0 .. 1] receives the same for the segment behind it
bool launch_segment_pass(const TailList& L, const TailReads& R, uint32_t want_state, uint32_t want_round, uint32_t* base,
                         uint32_t* next, uint32_t* kept_item, const uint32_t* node_rank, uint32_t* e_src, uint32_t* e_dst,
                         uint32_t* e_len, ScanSpace& space, hipStream_t s);
// out[i] = in[0] + .. + in[i - 1], out[n] = the sum; copy (may be null) receives out[0 .. n) as well
bool launch_offsets_pass(const uint32_t* in, uint32_t* out, uint32_t* copy, uint32_t n, ScanSpace& space, hipStream_t s);
// two of them side by side (sums below 2^31): out0[n], out1[n] and totals[0 .. 1] = the sums
bool launch_pair_offsets_pass(const uint32_t* in0, const uint32_t* in1, uint32_t* out0, uint32_t* out1, uint32_t* totals, uint32_t n,
                              ScanSpace& space, hipStream_t s);

// ---- the end of a containment fixed point (fixed_point_kernels.hip) ------------------------------------
// conditional killers {key, target, keeper} (*count of them, on the device); base: the deaths decided for good (all ones:
// never), receives the deaths; map: one word per read, all ones, left all ones; pack: fixed_point_pack_words() words of
// scratch; work: four arrays of as many words as there are reads, uninitialised; sync8: eight zeroed words (one set per
// call); *error is set when the rounds do not settle (1) - workgroups of a long list's kernel that cannot meet are no error,
// the last of them does the rounds alone;
// *rounds_out (may be null) receives the number of rounds
struct FixedPointList {
    const uint32_t *key, *target, *keeper;
    const uint32_t* count;
    uint32_t lds_limit = 0xFFFFFFFFu;       // tests: lists longer than this take the long lists' kernel (at most what the LDS holds)
    uint32_t debug_give_up = 0;             // tests: the long lists' workgroups do not meet (one of them finishes alone)
};
size_t fixed_point_pack_words();
hipError_t launch_fixed_point_finish(const FixedPointList& list, uint32_t* base, uint32_t* map, uint32_t* pack, uint32_t* const work[4],
                                     uint32_t* sync8, uint32_t* error, uint32_t* rounds_out, hipStream_t s);

// ---- fills (fill_kernels.hip) -------------------------------------------------------------------
// What a stage has to clear, cleared in one launch: add(pointer, byte value, bytes) ..., then launch(stream).
struct FillList {
    static constexpr uint32_t kMost = 8;
    void* ptr[kMost];
    size_t bytes[kMost];
    uint8_t value[kMost];
    uint32_t n = 0;
    bool overflow = false;
    void add(void* p, int byte, size_t n_bytes);
    hipError_t launch(hipStream_t s);       // (more than kMost entries: hipErrorInvalidValue)
};

// ---- partitioned bucketing (bucket_kernels.hip) ------------------------------------------------
size_t partition_records_needed(uint32_t n_reads, uint64_t n_overlaps);
uint32_t partition_count(uint32_t n_reads);
uint32_t partition_group_slots(uint32_t n_reads);
size_t partition_tile_slots(uint32_t n_reads, uint64_t n_overlaps);
bool partition_path_fits(uint32_t n_reads, uint32_t max_read_len, uint64_t n_overlaps);
// bound events of all reads as an exact CSR (ev_off[n_reads + 1], ev); buffer sizes: bucket_kernels.hip
// dedupe (may be null): the counting pass does duplicate removal's first pass on the way and writes the validity bytes that hold
// for the unmarked queries; launch_dedupe_fix behind it (`counted` is recorded behind the counting pass) finishes the marked ones
constexpr uint32_t kDedupeListGivenUp = 0x80000000u;   // (list_cap is at most 2^20)
struct BucketDedupe {
    uint8_t* suspect;       // n_reads bytes (cleared by the call)
    uint8_t* valid;         // one byte per overlap
    uint32_t* list_pos;     // where the counting pass marked a query (a run that needs the full comparison): positions ...
    uint32_t* list_query;   // ... and queries, list_cap of each
    uint32_t list_cap;
    uint32_t* list_count;   // a zeroed word: marks listed; beyond list_cap (a trip that could not stage its marks sets kDedupeListGivenUp): the list was given up
    hipEvent_t counted;     // may be null
    // (may be null) columns that are still on their way (RALA_HIP_MEM_HOST_ASYNC): the stream waits for ids in front of the
    // counting pass, for b_coords in front of the first scatter, for a_coords in front of the query side
    hipEvent_t ids = nullptr, b_coords = nullptr, a_coords = nullptr;
};
extern uint32_t g_count_window;      // (bucket_kernels.hip, option "debug_count_window": groups of 128 reads per counting pass; 0 = what the LDS holds)
extern uint32_t g_part_shift;        // (bucket_kernels.hip, option "debug_part_shift": reads per first-level partition = 1 << shift, 12 .. 14; 0 = by the rule)
bool bucket_count_can_dedupe(const OvlSoA& o, const uint8_t* valid);      // (the id columns on 16-byte boundaries)
// Units of ev_off (round 6): the partitioned bucketing counts a read's bound PAIRS (an overlap puts one begin and one end on each of
// its two reads), so 2 n < 2^32 bounds it - 2.1 G overlaps per context instead of the 1.07 G of offsets in events; the readers
// shift (PileArgs::ev_shift).  The exact CSR and the tuples (single events) keep events.
constexpr uint32_t kBucketPairShift = 1;
hipError_t launch_bucket_partitioned(const OvlSoA& o, uint32_t n_reads, uint32_t* acount, uint32_t* written,
                                     uint32_t* part_cursor, uint32_t* group, uint32_t* tiles, uint64_t* rec1, uint64_t* rec2,
                                     uint32_t* ev_off, uint32_t* ev, uint32_t workgroups, FillList& fills, hipStream_t s,
                                     const BucketDedupe* dedupe = nullptr, uint32_t ev_shift = kBucketPairShift);
// the runs around the listed marks through the full comparison - or, when the list was given up, every overlap of a marked
// query (two launches; the one that is not needed leaves at once)
void launch_dedupe_fix(const OvlSoA& o, uint32_t n_reads, const BucketDedupe& d, hipStream_t s);
// the same from an owner rank's bound records (launch_bucket_tuples(.., records = true)): zero_counts = n_reads + 2 words
bool partition_path_fits_records(uint32_t n_reads, uint32_t max_read_len, uint64_t n_records);
// (shrink: what the bounds are drawn in by - 15 for the primary overlaps, graph.cpp:317-324; 0 for the sensitive ones, :929-933)
hipError_t launch_bucket_partitioned_records(const uint64_t* records, uint64_t n, uint32_t n_reads, uint32_t* zero_counts,
                                             uint32_t* part_cursor, uint32_t* group, uint32_t* tiles, uint64_t* rec1, uint64_t* rec2,
                                             uint32_t* ev_off, uint32_t* ev, uint32_t workgroups, FillList& fills, hipStream_t s,
                                             uint32_t shrink = 15u, uint32_t ev_shift = kBucketPairShift);

// ---- sharded runs: the bounds scattered once, on the sender, by (owner rank, partition of the owner's reads) ----------
// (bucket_kernels.hip) read r = local read r / world of rank r % world; every owner's reads in partitions of 4096 and groups
// of 128 - as many of each on every owner (what rank 0 needs)
struct ShardGeometry {
    uint32_t world;
    uint32_t n_part;        // partitions per owner
    uint32_t groups;        // groups per owner = 32 * n_part
    uint32_t header;        // 8-byte words in front of an owner's records: its groups' record counts (groups / 2)
};
struct ShardBlocks {
    uint32_t off[64];       // where rank p's block lies: 8-byte words from the base the owner is given ...
    uint32_t self;          // ... except this rank's (the owner's own, never copied): from the second base; 64 = none
};
ShardGeometry shard_geometry(uint64_t n_reads, uint32_t world);
bool shard_path_fits(uint64_t n_reads, uint32_t max_read_len, uint32_t world);         // the same on every rank
size_t shard_send_words(const ShardGeometry& g, uint64_t n_overlaps);                 // 8-byte words of a sender's buffer
size_t shard_tile_slots(const ShardGeometry& g, uint64_t n_records);
size_t shard_group_words(const ShardGeometry& g);
// sender: block after block - [header][records {local read & 4095 : 12 | begin : 26 | end : 26}, partition by partition];
// send_words[p] (device) = 8-byte words of block p
hipError_t launch_shard_emit(const OvlSoA& o, uint32_t n_reads, const ShardGeometry& g, uint32_t* group_count, uint32_t* part_cursor,
                             uint64_t* send, uint32_t* send_words, uint32_t workgroups, FillList& fills, hipStream_t s,
                             const BucketDedupe* dedupe = nullptr);      // (dedupe: as launch_bucket_partitioned's)
// owner: the blocks of all senders -> ev_off[n_reads_local + 1], ev (bounds drawn in by 15, graph.cpp:317-324)
hipError_t launch_bucket_from_blocks(const uint64_t* base, const uint64_t* base_self, const ShardBlocks& blocks, const ShardGeometry& g, uint32_t n_reads_local,
                                     uint64_t n_records, uint32_t* group, uint32_t* tiles, uint64_t* rec2, uint32_t* ev_off, uint32_t* ev,
                                     FillList& fills, hipStream_t s, uint32_t ev_shift = kBucketPairShift);

// ---- PAF text -> columns (ingest_kernels.hip) ----------------------------------------------
struct PafColumns {
    uint32_t *a_id, *b_id, *a_begin, *a_end, *b_begin, *b_end, *length;
    uint8_t* strand;
};
uint32_t paf_chunk_bytes();
uint32_t paf_halo_bytes();      // bytes behind a range's last own byte that its last lines' first eleven columns may use
// text: the n bytes of the file whose line starts belong to this launch, then up to n_avail (>= n) bytes that may be read for
// the last lines' columns - the buffer itself readable up to the next multiple of paf_chunk_bytes() + 4 KB behind n;
// first_is_start: byte 0 starts a line (the file's first byte, or the byte behind a newline); chunk_lines: one count per chunk
void launch_paf_count(const uint8_t* text, uint64_t n, bool first_is_start, uint32_t* chunk_lines, hipStream_t s);
// chunk_row: the exclusive scan of the counts; flags (zeroed): 1 a line with fewer than 12 columns, 2 more lines in a chunk than
// records can make, 4 a line whose first eleven columns reach beyond the halo; first_bad (all ones): row << 32 | read of the
// first record whose length differs from its sequence's; mhap: twelve blank-separated numeric columns (reference overlap.cpp:12-20),
// no name table
void launch_paf_parse(const uint8_t* text, uint64_t n, uint64_t n_avail, bool first_is_start, const uint32_t* chunk_row, const void* buckets,
                      uint64_t n_buckets, const char* arena, const uint32_t* read_len, uint32_t n_reads, bool check_lengths,
                      const PafColumns& out, uint32_t* flags, unsigned long long* first_bad, hipStream_t s, bool mhap = false);

// ---- scans (scan_kernels.hip) --------------------------------------------------
// exclusive prefix sum of n uint32 values; out may alias in; out[n] receives the total
size_t scan_workspace_bytes(uint64_t n);
void launch_exclusive_scan(const uint32_t* in, uint32_t* out, uint64_t n, void* workspace, hipStream_t s);

// ---- transitive reduction (tr_kernels.hip) ---------------------------------------
// edges [first_edge, last_edge) act as a->b (all of them, or one rank's share)
void launch_tr_mark(const uint32_t* row_ptr, const uint32_t* adj_edge, const uint32_t* edge_src,
                    const uint32_t* edge_dst, const uint32_t* edge_len, uint32_t n_nodes, uint32_t first_edge,
                    uint32_t last_edge, uint8_t* marks, hipStream_t s);
// out-degree per node (also flags endpoints >= n_nodes) / adjacency fill through a cursor
void launch_tr_degree(const uint32_t* src, const uint32_t* dst, uint32_t n_nodes, uint32_t n_edges, uint32_t* deg,
                      uint32_t* bad, hipStream_t s);
// adj: 2 * n_edges words, {edge, target of the edge} per out-list entry
void launch_tr_fill(const uint32_t* src, const uint32_t* dst, uint32_t n_nodes, uint32_t n_edges, uint32_t* cursor, uint32_t* adj,
                    hipStream_t s);
// connected components by min-label hooking + pointer jumping (edges = pairs a, b)
void launch_cc_init(uint32_t* label, uint32_t n, hipStream_t s);
void launch_cc_hook(const uint32_t* edges, uint32_t n_edges, uint32_t sample, uint32_t* label, uint32_t* changed,
                    hipStream_t s);
void launch_cc_compress(uint32_t* label, uint32_t n, hipStream_t s);
// the last compression of a component search with the components' sizes (reads with an overlap) for the component medians
void launch_cc_compress_count(uint32_t* label, uint32_t n, const uint8_t* touched, uint32_t* size, hipStream_t s);
// component medians (median_kernels.hip): tmp = component_median_workspace(n) bytes; before a search component_median_clear
// puts what has to be zero into the caller's fill; launch_cc_compress_count(.., component_median_sizes(tmp, n), ..) ends the
// search; then launch_component_medians: cmed[q] for every read q with an overlap
size_t component_median_workspace(uint32_t n);
void component_median_clear(void* tmp, uint32_t n, FillList& fills);
uint32_t* component_median_sizes(void* tmp, uint32_t n);
hipError_t launch_component_medians(const uint32_t* label, const uint8_t* touched, const uint32_t* alive_reads,
                                    const uint16_t* median, uint32_t n_alive, void* tmp, uint16_t* cmed, hipStream_t s);
void launch_tr_count(uint8_t* marks, uint32_t n_edges, uint32_t* n_pairs, hipStream_t s);    // also normalises the marks to 0 / 1

// ---- force-directed layout step (layout_kernels.hip) -------------------------------------------
void launch_layout_step(uint32_t n, const double* x, const double* y, double* x_out, double* y_out,
                        const uint32_t* adj_off, const uint32_t* adj, double k, double t, hipStream_t s);

}  // namespace rala_hip
