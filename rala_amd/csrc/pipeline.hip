// librala_hip: C ABI (include/rala_hip.h) and host orchestration of the stages.
//
// Device work: duplicate removal, bound bucketing, pile build + annotation,
// overlap classification, the containment fixed point, hill counters,
// survivor gathering, transitive-edge marking.  Host work (this file): the
// short sequential tail of Graph::preprocess on the few per cent of overlaps
// that survive containment removal (connected components, component medians,
// iterate-until-stable), node/edge numbering, CSR assembly.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <numeric>

#include "context.h"
#include "scan_pass.h"
#include "stages.h"

using namespace rala_hip;


// Small device -> host reads (counters, flags) go through a pinned staging area: an async copy
// into pageable memory is staged by the runtime and costs tens of microseconds more, and the
// tail does a dozen of them per call.  d2h_small queues the copy, stream_sync waits for the
// stream and delivers what was queued.
hipError_t rala_hip::d2h_small(rala_hip_ctx* ctx, void* dst, const void* src, size_t bytes, hipStream_t s) {
    if (ctx->p_stage.ensure(256) != hipSuccess || ctx->stage_used + bytes > 256 * sizeof(uint32_t) ||
        ctx->stage_pending.size() >= 16) {
        return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s);
    }
    char* at = (char*)ctx->p_stage.p + ctx->stage_used;
    const hipError_t e = hipMemcpyAsync(at, src, bytes, hipMemcpyDeviceToHost, s);
    if (e != hipSuccess) return e;
    ctx->stage_pending.push_back({dst, ctx->stage_used, bytes});
    ctx->stage_used += (bytes + 7) & ~(size_t)7;
    return hipSuccess;
}

hipError_t rala_hip::stream_sync(rala_hip_ctx* ctx, hipStream_t s) {
    const hipError_t e = hipStreamSynchronize(s);
    if (e == hipSuccess) {      // a failed wait delivers nothing
        for (const auto& c : ctx->stage_pending) memcpy(c.dst, (const char*)ctx->p_stage.p + c.offset, c.bytes);
    }
    ctx->stage_pending.clear();
    ctx->stage_used = 0;
    return e;
}

namespace {

// Every error return drops the staged device -> host copies that were queued but not delivered
// yet (d2h_small below): their destinations are mostly locals of the function that is returning.
#define HIPCHECK(call)                                                                       \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            ctx->err = std::string(#call) + ": " + hipGetErrorString(e_);                    \
            ctx->stage_pending.clear();                                                      \
            ctx->stage_used = 0;                                                             \
            return e_ == hipErrorOutOfMemory ? RALA_HIP_ENOMEM : RALA_HIP_EDEVICE;           \
        }                                                                                    \
    } while (0)

int fail(rala_hip_ctx* ctx, int code, const char* msg) {
    ctx->err = msg;
    ctx->stage_pending.clear();
    ctx->stage_used = 0;
    return code;
}

ReadState read_state(rala_hip_ctx* ctx) {
    ReadState rs;
    rs.begin = ctx->d_begin.p; rs.end = ctx->d_end.p; rs.alive = ctx->d_alive.p;
    rs.n_pits = ctx->d_n_pits.p; rs.n_hills = ctx->d_n_hills.p; rs.iv_slot = ctx->d_iv_slot.p;
    rs.pool = ctx->d_pool.p;
    return rs;
}

double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ---- host mirror of the per-read state -------------------------------------------
int download_read_state(rala_hip_ctx* ctx) {
    const uint64_t n = ctx->n_reads;
    HIPCHECK(hipSetDevice(ctx->device));        // (getters may be called with another rank's device current)
    ctx->h_begin.resize(n); ctx->h_end.resize(n); ctx->h_median.resize(n); ctx->h_p10.resize(n);
    ctx->h_alive.resize(n); ctx->h_n_pits.resize(n); ctx->h_n_hills.resize(n); ctx->h_slot.resize(n);
    hipStream_t s = ctx->stream;
    HIPCHECK(hipMemcpyAsync(ctx->h_begin.data(), ctx->d_begin.p, n * 4, hipMemcpyDeviceToHost, s));
    HIPCHECK(hipMemcpyAsync(ctx->h_end.data(), ctx->d_end.p, n * 4, hipMemcpyDeviceToHost, s));
    HIPCHECK(hipMemcpyAsync(ctx->h_median.data(), ctx->d_median.p, n * 2, hipMemcpyDeviceToHost, s));
    HIPCHECK(hipMemcpyAsync(ctx->h_p10.data(), ctx->d_p10.p, n * 2, hipMemcpyDeviceToHost, s));
    HIPCHECK(hipMemcpyAsync(ctx->h_alive.data(), ctx->d_alive.p, n, hipMemcpyDeviceToHost, s));
    HIPCHECK(hipMemcpyAsync(ctx->h_n_pits.data(), ctx->d_n_pits.p, n * 4, hipMemcpyDeviceToHost, s));
    HIPCHECK(hipMemcpyAsync(ctx->h_n_hills.data(), ctx->d_n_hills.p, n * 4, hipMemcpyDeviceToHost, s));
    HIPCHECK(hipMemcpyAsync(ctx->h_slot.data(), ctx->d_iv_slot.p, n * 4, hipMemcpyDeviceToHost, s));
    uint32_t small[4];
    HIPCHECK(d2h_small(ctx, small, ctx->d_small.p, sizeof(small), s));
    HIPCHECK(stream_sync(ctx, s));
    const uint32_t used = std::min(small[0], ctx->pool_cap);
    ctx->pool_used = used;
    ctx->h_pool.resize(used);
    if (used) {
        HIPCHECK(hipMemcpy(ctx->h_pool.data(), ctx->d_pool.p, (size_t)used * sizeof(Interval), hipMemcpyDeviceToHost));
    }
    ctx->host_state_fresh = true;
    return RALA_HIP_OK;
}

// ---- launch classes of the position-space kernel: reads grouped by LDS image size ------
void build_classes(rala_hip_ctx* ctx, const std::vector<uint32_t>& reads, std::vector<uint32_t>& order) {
    static const uint32_t kLw[] = {8192, 10240, 12288, 14336, 16384, 20480, 24576};
    const int n_lds = (int)(sizeof(kLw) / sizeof(kLw[0]));
    std::vector<std::vector<uint32_t>> bins(n_lds + 1);
    uint32_t max_long = 0;
    // a handful of reads (what the run-space chain hands on: a dozen at C3): ONE launch at the size of the longest - three
    // launches of four workgroups each are three times the latency of one
    int only = -1;
    if (reads.size() <= 32) {
        uint32_t top = 0;
        for (uint32_t r : reads) top = std::max(top, ctx->h_read_len[r]);
        only = n_lds;
        if ((int64_t)top <= ctx->max_lds_read_len) {
            for (int c = 0; c < n_lds; ++c) {
                if (pile_lw_for(top) <= kLw[c]) { only = c; break; }
            }
        }
    }
    for (uint32_t r : reads) {
        const uint32_t len = ctx->h_read_len[r];
        const uint32_t lw = pile_lw_for(len);
        int k = n_lds;
        if ((int64_t)len <= ctx->max_lds_read_len) {
            for (int c = 0; c < n_lds; ++c) {
                if (lw <= kLw[c]) { k = c; break; }
            }
        }
        if (only >= 0) k = only;
        bins[k].push_back(r);
        if (k == n_lds) max_long = std::max(max_long, lw);
    }
    ctx->classes.clear();
    order.clear();
    // big classes first: their workgroups are the long poles
    for (int k = n_lds; k >= 0; --k) {
        if (bins[k].empty()) continue;
        LaunchClass c;
        c.in_lds = k < n_lds;
        c.lw = c.in_lds ? kLw[k] : max_long;
        c.first = (uint32_t)order.size();
        c.count = (uint32_t)bins[k].size();
        // longest first (only a scheduling matter): counting sort on length / 64
        {
            const std::vector<uint32_t>& v = bins[k];
            uint32_t top = 0;
            for (uint32_t r : v) top = std::max(top, ctx->h_read_len[r] >> 6);
            std::vector<uint32_t> at(top + 2, 0);
            for (uint32_t r : v) ++at[top - (ctx->h_read_len[r] >> 6) + 1];
            for (uint32_t b = 0; b <= top; ++b) at[b + 1] += at[b];
            const size_t base = order.size();
            order.resize(base + v.size());
            for (uint32_t r : v) order[base + at[top - (ctx->h_read_len[r] >> 6)]++] = r;
        }
        ctx->classes.push_back(c);
    }
}

// position-space kernel over `reads` (all of them, or the run kernel's overflow)
// a.big_cap_reg != 0: with the region / interval lists in global memory at those sizes (a.big_space is set here)
int run_position_kernel(rala_hip_ctx* ctx, PileArgs a, const std::vector<uint32_t>& reads) {
    if (reads.empty()) return RALA_HIP_OK;
    std::vector<uint32_t> order;
    build_classes(ctx, reads, order);
    HIPCHECK(ctx->d_order.ensure(order.size() + 1));
    HIPCHECK(hipMemcpyAsync(ctx->d_order.p, order.data(), order.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    const bool big = a.big_cap_reg != 0;
    const uint64_t big_words = big ? pile_big_words(a.big_cap_reg, a.big_cap_list, a.big_cap_raw) : 0;
    for (const LaunchClass& c : ctx->classes) {
        uint32_t grid = c.count;
        if (!c.in_lds) grid = std::min<uint32_t>(c.count, 512);
        if (big) {
            grid = std::min<uint32_t>(grid, 64);
            HIPCHECK(ctx->d_big_space.ensure((size_t)(big_words * grid)));
            a.big_space = ctx->d_big_space.p;
        }
        if (!c.in_lds) HIPCHECK(ctx->d_slab.ensure((size_t)grid * 3 * c.lw));
        a.order = ctx->d_order.p + c.first;
        a.n_items = c.count;
        a.lw = c.lw;
        a.slab = ctx->d_slab.p;
        launch_pile_build_annotate(a, grid, c.in_lds, ctx->stream);
        ++ctx->tm.pile_launches;
    }
    HIPCHECK(stream_sync(ctx, ctx->stream));      // `order` is pageable host memory
    return RALA_HIP_OK;
}

// first sizes of the lists in global memory (the position-space kernels' LDS lists hold 192 regions per flag list and 64 raw
// intervals); the option debug_big_caps = regions << 32 | raw intervals makes tests start lower
void first_big_caps(const rala_hip_ctx* ctx, uint32_t& cap_reg, uint32_t& cap_list, uint32_t& cap_raw) {
    cap_reg = 1024; cap_raw = 4096;
    if (ctx->debug_big_caps) {
        cap_reg = std::max<uint32_t>(4, (uint32_t)((uint64_t)ctx->debug_big_caps >> 32));
        cap_raw = std::max<uint32_t>(4, (uint32_t)((uint64_t)ctx->debug_big_caps & 0xFFFFFFFFu));
    }
    cap_list = 4 * cap_reg;
}

// The reads the position-space kernel noted in d_big_list[0] (*count_dev of them): once more, with the region lists and raw
// intervals in global memory; lists that still do not fit are doubled for the reads concerned until they do (the reference
// keeps them in vectors, pile.cpp:66, 359, 448).  a: the call's arguments (a.error keeps its pool-capacity bit).
int run_unbounded_piles(rala_hip_ctx* ctx, PileArgs a, uint32_t count, uint32_t* count_dev) {
    hipStream_t s = ctx->stream;
    uint32_t cap_reg, cap_list, cap_raw;
    first_big_caps(ctx, cap_reg, cap_list, cap_raw);
    ctx->tm.pile_unbounded_reads += count;
    int from = 0;
    while (count) {
        std::vector<uint32_t> reads(count);
        HIPCHECK(hipMemcpy(reads.data(), ctx->d_big_list[from].p, (size_t)count * 4, hipMemcpyDeviceToHost));
        std::sort(reads.begin(), reads.end());
        HIPCHECK(ctx->d_big_list[from ^ 1].ensure(ctx->n_reads + 1));
        uint32_t status = 0;
        HIPCHECK(hipMemcpy(&status, a.error, 4, hipMemcpyDeviceToHost));
        status &= kErrPoolCapacity;
        HIPCHECK(hipMemcpyAsync(a.error, &status, 4, hipMemcpyHostToDevice, s));
        HIPCHECK(hipMemsetAsync(count_dev, 0, 4, s));
        a.big_list = ctx->d_big_list[from ^ 1].p;
        a.big_count = count_dev;
        a.big_cap_reg = cap_reg; a.big_cap_list = cap_list; a.big_cap_raw = cap_raw;
        a.n_items_dev = nullptr;
        const int rc = run_position_kernel(ctx, a, reads);
        if (rc != RALA_HIP_OK) return rc;
        uint32_t left = 0;
        HIPCHECK(hipMemcpy(&left, count_dev, 4, hipMemcpyDeviceToHost));
        HIPCHECK(hipMemcpy(&status, a.error, 4, hipMemcpyDeviceToHost));
        HIPCHECK(hipGetLastError());
        if (getenv("RALA_HIP_TRACE")) {
            fprintf(stderr, "[trace] lists in global memory: %u reads at %u regions / %u raw intervals, %u left (status %u)\n", count, cap_reg,
                    cap_raw, left, status);
        }
        if (left && !(status & (kErrRegionCapacity | kErrRawCapacity))) {
            return fail(ctx, RALA_HIP_EDEVICE, "position-space kernel: reads left without a reason");
        }
        if (status & kErrRegionCapacity) {
            if (cap_reg > (1u << 26)) return fail(ctx, RALA_HIP_ENOMEM, "slope-region lists beyond 2^26 entries");
            cap_reg *= 2; cap_list *= 2;
        }
        if (status & kErrRawCapacity) {
            if (cap_raw > (1u << 28)) return fail(ctx, RALA_HIP_ENOMEM, "interval lists beyond 2^28 entries");
            cap_raw *= 2;
        }
        count = left;
        from ^= 1;
    }
    // (the lists' bits are no error of the call any more)
    uint32_t status = 0;
    HIPCHECK(hipMemcpy(&status, a.error, 4, hipMemcpyDeviceToHost));
    status &= kErrPoolCapacity;
    HIPCHECK(hipMemcpy(a.error, &status, 4, hipMemcpyHostToDevice));
    return RALA_HIP_OK;
}

// ---- host tail helpers (Graph::preprocess, graph.cpp:699-880) --------------------------
bool host_trim(rala_hip_ctx* ctx, HostOvl& o) {
    if (!ctx->h_alive[o.a] || !ctx->h_alive[o.b]) return false;
    return ovl_trim(o.c, o.strand, ctx->h_begin[o.a], ctx->h_end[o.a], ctx->h_begin[o.b], ctx->h_end[o.b]);
}

uint32_t host_type(rala_hip_ctx* ctx, const HostOvl& o) {
    return ovl_type(o.c, o.strand, ctx->h_begin[o.a], ctx->h_end[o.a], ctx->h_begin[o.b], ctx->h_end[o.b]);
}

// Overlap::trim is idempotent while both valid regions stand still, so only overlaps that
// touch a read whose region changed since the last pass ("dirty") are recomputed.  Dropped
// items stay in place with dead = 1 (lists are compacted once, at the end); the pass is
// data parallel over the host pool.  Returns the number of items dropped.
uint64_t retrim(rala_hip_ctx* ctx, std::vector<HostOvl>& v) {
    const uint8_t* dirty = ctx->dirty.data();
    std::vector<uint64_t> dropped(ctx->pool->size(), 0);
    ctx->pool->chunks(v.size(), [&](unsigned t, size_t b, size_t e) {
        uint64_t d = 0;
        for (size_t k = b; k < e; ++k) {
            HostOvl& o = v[k];
            if (o.dead || !(dirty[o.a] | dirty[o.b])) continue;
            if (!host_trim(ctx, o)) { o.dead = 1; ++d; }
            else o.type = 255;
        }
        dropped[t] = d;
    });
    uint64_t total = 0;
    for (uint64_t d : dropped) total += d;
    return total;
}

// flags only (callable from pool threads, one read per thread); collect_dirty() lists them
void mark_dirty(rala_hip_ctx* ctx, uint32_t r) {
    ctx->dirty[r] = 1;
    ctx->ever_dirty[r] = 1;
}

void collect_dirty(rala_hip_ctx* ctx, const std::vector<uint32_t>& among) {
    for (uint32_t r : among) if (ctx->dirty[r]) ctx->dirty_list.push_back(r);
}

void clear_dirty(rala_hip_ctx* ctx) {
    for (uint32_t r : ctx->dirty_list) ctx->dirty[r] = 0;
    ctx->dirty_list.clear();
}

bool host_shrink(rala_hip_ctx* ctx, uint32_t r, uint32_t b, uint32_t e) {
    if (b > e || e - b < kMinRegion) { mark_dirty(ctx, r); return false; }
    if (ctx->h_begin[r] != b || ctx->h_end[r] != e) mark_dirty(ctx, r);
    ctx->h_begin[r] = b;
    ctx->h_end[r] = e;
    return true;
}

// Pile::break_over_chimeric_hills (pile.cpp:471-498): geom.h longest_piece, as on the device
bool break_hills(rala_hip_ctx* ctx, uint32_t r, const Interval* hills, uint32_t n) {
    const Piece keep = longest_piece(ctx->h_begin[r], ctx->h_end[r], n,
                                     [&](uint32_t i, uint32_t& f, uint32_t& s) { f = hills[i].first; s = hills[i].second; },
                                     [&](uint32_t i) { return hills[i].aux <= 3; });
    return host_shrink(ctx, r, keep.begin, keep.end);
}

// Pile::break_over_chimeric_pits (pile.cpp:366-402); a pit is real when some
// coverage inside it satisfies data*1.84 <= median — monotone in data, so the
// minimum recorded by the pile kernel decides.  Unreal pits are kept (in place).
bool break_pits(rala_hip_ctx* ctx, uint32_t r, Interval* pits, uint32_t& n_pits, uint16_t dataset_median) {
    uint32_t w = 0;
    const Piece keep = longest_piece(ctx->h_begin[r], ctx->h_end[r], n_pits,
                                     [&](uint32_t k, uint32_t& f, uint32_t& s) { f = pits[k].first; s = pits[k].second; },
                                     [&](uint32_t k) {
                                         if ((double)pits[k].aux * 1.84 <= (double)dataset_median) return true;
                                         pits[w++] = pits[k];
                                         return false;
                                     });
    n_pits = w;
    return host_shrink(ctx, r, keep.begin, keep.end);
}

// per read: the median of the pile medians of its connected component over the current
// overlaps (graph.cpp:740-783); components by min-label hooking on the GPU (cc_kernels).
int component_medians(rala_hip_ctx* ctx, std::vector<uint32_t>& members, std::vector<uint16_t>& med_of_member) {
    const size_t m = ctx->overlaps.size();
    members.clear();
    med_of_member.clear();
    if (m == 0) return RALA_HIP_OK;
    hipStream_t s = ctx->stream;
    // work on the ranks of the reads that were alive after the second pass (about a tenth of all)
    const std::vector<uint32_t>& rank = ctx->alive_rank;
    const std::vector<uint32_t>& reads = ctx->alive_reads;
    const size_t na = reads.size();
    HIPCHECK(ctx->p_cc_edges.ensure(2 * m));
    HIPCHECK(ctx->p_cc_label.ensure(na));
    uint32_t* edges = ctx->p_cc_edges.p;
    std::vector<uint8_t>& touched = ctx->scratch_touched;
    touched.assign(na, 0);
    ctx->pool->chunks(m, [&](unsigned, size_t b0, size_t e0) {
        for (size_t k = b0; k < e0; ++k) {
            const HostOvl& o = ctx->overlaps[k];
            if (o.dead) { edges[2 * k] = 0; edges[2 * k + 1] = 0; continue; }      // harmless self loop
            const uint32_t ra = rank[o.a], rb = rank[o.b];
            edges[2 * k] = ra; edges[2 * k + 1] = rb;
            touched[ra] = 1; touched[rb] = 1;
        }
    });
    HIPCHECK(ctx->d_cc_edges.ensure(2 * m));
    HIPCHECK(ctx->d_cc_label.ensure(na));
    HIPCHECK(hipMemcpyAsync(ctx->d_cc_edges.p, edges, 2 * m * 4, hipMemcpyHostToDevice, s));
    launch_cc_init(ctx->d_cc_label.p, (uint32_t)na, s);
    for (int round = 0;; ++round) {
        HIPCHECK(hipMemsetAsync(ctx->d_small.p + 2, 0, 4, s));
        for (int it = 0; it < 4; ++it) {          // several hook + jump steps per host round trip
            launch_cc_hook(ctx->d_cc_edges.p, (uint32_t)m, 0, ctx->d_cc_label.p, ctx->d_small.p + 2, s);
            launch_cc_compress(ctx->d_cc_label.p, (uint32_t)na, s);
        }
        uint32_t changed = 0;
        HIPCHECK(d2h_small(ctx, &changed, ctx->d_small.p + 2, 4, s));
        HIPCHECK(stream_sync(ctx, s));
        if (!changed) break;
        if (round > 10000) return fail(ctx, RALA_HIP_EDEVICE, "connected components did not converge");
    }
    uint32_t* label = ctx->p_cc_label.p;
    HIPCHECK(hipMemcpyAsync(label, ctx->d_cc_label.p, na * 4, hipMemcpyDeviceToHost, s));
    HIPCHECK(stream_sync(ctx, s));
    // members = reads with at least one overlap, in read order; grouped by label (counting sort)
    std::vector<uint32_t>& mrank = ctx->scratch_u32a;
    mrank.clear();
    for (size_t q = 0; q < na; ++q) if (touched[q]) { members.push_back(reads[q]); mrank.push_back((uint32_t)q); }
    std::vector<uint32_t>& cnt = ctx->scratch_u32b;
    cnt.assign(na + 1, 0);
    for (uint32_t q : mrank) ++cnt[label[q] + 1];
    for (size_t q = 0; q < na; ++q) cnt[q + 1] += cnt[q];
    std::vector<uint32_t> idx(members.size());
    for (size_t k = 0; k < members.size(); ++k) idx[cnt[label[mrank[k]]]++] = (uint32_t)k;
    med_of_member.assign(members.size(), 0);
    std::vector<uint16_t> mm;
    for (size_t b0 = 0; b0 < idx.size();) {
        const uint32_t lab = label[mrank[idx[b0]]];
        size_t t = b0;
        mm.clear();
        while (t < idx.size() && label[mrank[idx[t]]] == lab) { mm.push_back(ctx->h_median[members[idx[t]]]); ++t; }
        std::nth_element(mm.begin(), mm.begin() + mm.size() / 2, mm.end());
        const uint16_t med = mm[mm.size() / 2];
        for (size_t k = b0; k < t; ++k) med_of_member[idx[k]] = med;
        b0 = t;
    }
    return RALA_HIP_OK;
}

struct Trace {
    bool on;
    double t;
    Trace() : on(getenv("RALA_HIP_TRACE") != nullptr), t(now_ms()) {}
    void operator()(const char* what, size_t k = 0) {
        if (!on) return;
        const double u = now_ms();
        fprintf(stderr, "[trace] %-28s %8.3f ms  (%zu)\n", what, u - t, k);
        t = u;
    }
};

int preprocess_chimeras(rala_hip_ctx* ctx) {
    Trace tr;
    const uint64_t n = ctx->n_reads;
    ctx->dirty.assign(n, 0);
    ctx->ever_dirty.assign(n, 0);
    ctx->dirty_list.clear();
    ctx->alive_rank.assign(n, 0xFFFFFFFFu);
    ctx->alive_reads.clear();
    for (uint64_t r = 0; r < n; ++r) {
        if (!ctx->h_alive[r]) continue;
        ctx->alive_rank[r] = (uint32_t)ctx->alive_reads.size();
        ctx->alive_reads.push_back((uint32_t)r);
    }
    const std::vector<uint32_t>& n_pits0 = ctx->h_n_pits0;     // hills sit behind the initial pits
    ctx->h_n_pits0 = ctx->h_n_pits;
    // break over chimeric hills (graph.cpp:704-720)
    for (uint32_t r : ctx->alive_reads) {
        if (ctx->h_n_hills[r] == 0) continue;
        const Interval* hills = ctx->h_pool.data() + ctx->h_slot[r] + n_pits0[r];
        if (!break_hills(ctx, (uint32_t)r, hills, ctx->h_n_hills[r])) ctx->h_alive[r] = 0;
        ctx->h_n_hills[r] = 0;
    }
    collect_dirty(ctx, ctx->alive_reads);
    tr("break hills", ctx->dirty_list.size());
    retrim(ctx, ctx->overlaps);      // :722-728
    retrim(ctx, ctx->internals);     // :730-736
    clear_dirty(ctx);
    tr("retrim ov+int", ctx->overlaps.size() + ctx->internals.size());

    std::vector<uint32_t> members;
    std::vector<uint16_t> med;
    for (;;) {                       // :738-829
        const int rc = component_medians(ctx, members, med);
        if (rc != RALA_HIP_OK) return rc;
        tr("component medians", members.size());
        ctx->pool->chunks(members.size(), [&](unsigned, size_t b0, size_t e0) {
            for (size_t k = b0; k < e0; ++k) {
                const uint32_t r = members[k];
                if (ctx->h_n_pits[r] == 0) continue;     // no pits: shrink(begin, end) is a no-op
                Interval* pits = ctx->h_pool.data() + ctx->h_slot[r];
                if (!break_pits(ctx, r, pits, ctx->h_n_pits[r], med[k])) ctx->h_alive[r] = 0;
            }
        });
        collect_dirty(ctx, members);
        const bool changed = retrim(ctx, ctx->overlaps) != 0;
        // internals whose reads moved: trim again; the ones that became dovetails join the
        // overlaps, in internals order (graph.cpp:809-824)
        {
            const uint8_t* dirty = ctx->dirty.data();
            std::vector<HostOvl>& in = ctx->internals;
            std::vector<std::vector<uint32_t>> promoted(ctx->pool->size());
            ctx->pool->chunks(in.size(), [&](unsigned t, size_t b, size_t e) {
                for (size_t k = b; k < e; ++k) {
                    HostOvl& o = in[k];
                    // untouched since it was classified: still trimmed and still kX
                    if (o.dead || !((dirty[o.a] | dirty[o.b]) || o.type == 255)) continue;
                    if (!host_trim(ctx, o)) { o.dead = 1; continue; }
                    o.type = (uint8_t)host_type(ctx, o);
                    if (o.type == kTypeAB || o.type == kTypeBA) promoted[t].push_back((uint32_t)k);
                }
            });
            for (const auto& list : promoted) {
                for (uint32_t k : list) {
                    ctx->overlaps.push_back(in[k]);
                    in[k].dead = 1;
                }
            }
        }
        tr("break pits + retrim", ctx->dirty_list.size());
        clear_dirty(ctx);
        if (!changed) break;
    }

    // in-order containment removal without the chimera guard (:831-877); stale types are
    // refreshed in parallel first, the scan itself is sequential and cheap
    auto refresh = [&](std::vector<HostOvl>& v) {
        ctx->pool->chunks(v.size(), [&](unsigned, size_t b, size_t e) {
            for (size_t k = b; k < e; ++k) {
                HostOvl& o = v[k];
                if (!o.dead && o.type == 255 && ctx->h_alive[o.a] && ctx->h_alive[o.b]) o.type = (uint8_t)host_type(ctx, o);
            }
        });
    };
    // only containment overlaps (kA / kB) can delete a read: collect them in order (parallel),
    // walk them sequentially, then drop the items that had lost a read by the time the
    // reference's loop reached them (parallel; kill_pos = position of the deleting item)
    std::vector<uint32_t> kill_pos;
    auto kill_scan = [&](std::vector<HostOvl>& v, bool timed) {
        std::vector<std::vector<uint32_t>> cand(ctx->pool->size());
        ctx->pool->chunks(v.size(), [&](unsigned t, size_t b0, size_t e0) {
            for (size_t k = b0; k < e0; ++k) {
                const HostOvl& o = v[k];
                if (!o.dead && (o.type == kTypeA || o.type == kTypeB)) cand[t].push_back((uint32_t)k);
            }
        });
        if (timed) kill_pos.assign(n, 0xFFFFFFFFu);
        for (const auto& list : cand) {
            for (uint32_t k : list) {
                HostOvl& o = v[k];
                o.dead = 1;
                if (!ctx->h_alive[o.a] || !ctx->h_alive[o.b]) continue;
                const uint32_t victim = o.type == kTypeA ? o.b : o.a;
                ctx->h_alive[victim] = 0;
                if (timed) kill_pos[victim] = k;
            }
        }
        if (!timed) return;       // overlaps: a final sweep drops everything that lost a read
        ctx->pool->chunks(v.size(), [&](unsigned, size_t b0, size_t e0) {
            for (size_t k = b0; k < e0; ++k) {
                HostOvl& o = v[k];
                if (o.dead) continue;
                const bool gone_a = !ctx->h_alive[o.a] && (kill_pos[o.a] == 0xFFFFFFFFu || kill_pos[o.a] < k);
                const bool gone_b = !ctx->h_alive[o.b] && (kill_pos[o.b] == 0xFFFFFFFFu || kill_pos[o.b] < k);
                if (gone_a || gone_b) o.dead = 1;
            }
        });
    };
    refresh(ctx->overlaps);
    refresh(ctx->internals);
    kill_scan(ctx->overlaps, false);
    kill_scan(ctx->internals, true);
    tr("kill scans");
    // one compaction at the end: what is dead or (overlaps only, :869-877) lost a read goes
    auto compact = [&](std::vector<HostOvl>& v, bool check_piles) {
        const unsigned T = ctx->pool->size();
        std::vector<size_t> cnt(T + 1, 0);
        auto keep = [&](const HostOvl& o) {
            return !o.dead && !(check_piles && (!ctx->h_alive[o.a] || !ctx->h_alive[o.b]));
        };
        const bool single = v.size() < 4096;
        ctx->pool->chunks(v.size(), [&](unsigned t, size_t b0, size_t e0) {
            size_t c = 0;
            for (size_t k = b0; k < e0; ++k) c += keep(v[k]);
            cnt[t + 1] = c;
        });
        for (unsigned t = 0; t < T; ++t) cnt[t + 1] += cnt[t];
        std::vector<HostOvl>& out = ctx->scratch_ovl;      // capacity survives across calls
        out.resize(single ? cnt[1] : cnt[T]);
        ctx->pool->chunks(v.size(), [&](unsigned t, size_t b0, size_t e0) {
            size_t w = single ? 0 : cnt[t];
            for (size_t k = b0; k < e0; ++k) if (keep(v[k])) out[w++] = v[k];
        });
        v.swap(out);
    };
    compact(ctx->internals, false);
    compact(ctx->overlaps, true);
    tr("compact", ctx->overlaps.size());
    return RALA_HIP_OK;
}

// position-space sensitive-pass kernel over `reads`, grouped by LDS image size, on stream st (the main stream, or the aux
// stream beside a mode's run-space kernels: nothing here waits for another stream).  Mode 2: a read whose region lists or
// raw hills outgrow the kernel's LDS lists is noted and runs again with the lists in global memory, doubled until the read
// fits (as run_unbounded_piles).
int run_repeats_classes(rala_hip_ctx* ctx, RepeatArgs a, const std::vector<uint32_t>& reads, int mode, hipStream_t st) {
    Trace trc;
    std::vector<uint32_t> order;
    build_classes(ctx, reads, order);
    trc("repeats: classes", reads.size());
    HIPCHECK(ctx->d_order.ensure(order.size() + 1));
    HIPCHECK(hipMemcpyAsync(ctx->d_order.p, order.data(), order.size() * 4, hipMemcpyHostToDevice, st));
    const bool big = mode == 2 && a.big_cap_reg != 0;
    const uint64_t big_words = big ? repeats_big_words(a.big_cap_reg, a.big_cap_list, a.big_cap_raw) : 0;
    for (const LaunchClass& c : ctx->classes) {
        uint32_t grid = c.count;
        if (!c.in_lds) grid = std::min<uint32_t>(c.count, 512);
        if (big) {
            grid = std::min<uint32_t>(grid, 64);
            HIPCHECK(ctx->d_big_space.ensure((size_t)(big_words * grid)));
            a.big_space = ctx->d_big_space.p;
        }
        if (!c.in_lds) HIPCHECK(ctx->d_slab.ensure((size_t)grid * 3 * c.lw));
        a.order = ctx->d_order.p + c.first;
        a.n_items = c.count;
        a.lw = c.lw;
        a.slab = ctx->d_slab.p;
        launch_pile_repeats(a, grid, c.in_lds, mode, st);
    }
    HIPCHECK(hipStreamSynchronize(st));              // (`order` is pageable host memory)
    HIPCHECK(hipGetLastError());
    trc("repeats: kernels", ctx->classes.size());
    return RALA_HIP_OK;
}

int run_repeats_kernel(rala_hip_ctx* ctx, RepeatArgs a, const std::vector<uint32_t>& reads, int mode, hipStream_t st = nullptr) {
    if (reads.empty()) return RALA_HIP_OK;
    if (st == nullptr) st = ctx->stream;
    if (mode != 2) return run_repeats_classes(ctx, a, reads, mode, st);
    uint32_t* const count_dev = ctx->d_small.p + 8;
    HIPCHECK(ctx->d_big_list[0].ensure(ctx->n_reads + 1));
    HIPCHECK(hipMemsetAsync(count_dev, 0, 4, st));
    a.big_list = ctx->d_big_list[0].p;
    a.big_count = count_dev;
    a.big_space = nullptr; a.big_cap_reg = a.big_cap_list = a.big_cap_raw = 0;
    a.force_big = ctx->debug_force_big ? 1u : 0u;
    int rc = run_repeats_classes(ctx, a, reads, 2, st);
    if (rc != RALA_HIP_OK) return rc;
    // one look: how many reads were noted, the status word (copies on THIS stream - a blocking copy would wait for the
    // run-space kernels beside)
    uint32_t look[2] = {0, 0};                  // [0] status (d_small[7]) [1] noted reads (d_small[8])
    auto fetch = [&]() -> int {
        HIPCHECK(hipMemcpyAsync(look, ctx->d_small.p + 7, 8, hipMemcpyDeviceToHost, st));
        HIPCHECK(hipStreamSynchronize(st));
        return (int)RALA_HIP_OK;
    };
    rc = fetch();
    if (rc != RALA_HIP_OK) return rc;
    uint32_t count = look[1];
    if (count == 0) return RALA_HIP_OK;
    uint32_t cap_reg, cap_list, cap_raw;
    first_big_caps(ctx, cap_reg, cap_list, cap_raw);
    cap_list = 2 * cap_reg;
    ctx->tm.pile_unbounded_reads += count;
    int from = 0;
    while (count) {
        std::vector<uint32_t> again(count);
        HIPCHECK(hipMemcpyAsync(again.data(), ctx->d_big_list[from].p, (size_t)count * 4, hipMemcpyDeviceToHost, st));
        HIPCHECK(hipStreamSynchronize(st));
        std::sort(again.begin(), again.end());
        HIPCHECK(ctx->d_big_list[from ^ 1].ensure(ctx->n_reads + 1));
        // (the lists' bits of the status word start every round at zero; other kernels may be setting the pool's bit meanwhile)
        launch_status_clear(a.error, kErrRegionCapacity | kErrRawCapacity, count_dev, st);
        a.big_list = ctx->d_big_list[from ^ 1].p;
        a.big_cap_reg = cap_reg; a.big_cap_list = cap_list; a.big_cap_raw = cap_raw;
        rc = run_repeats_classes(ctx, a, again, 2, st);
        if (rc != RALA_HIP_OK) return rc;
        rc = fetch();
        if (rc != RALA_HIP_OK) return rc;
        const uint32_t left = look[1], status = look[0];
        if (getenv("RALA_HIP_TRACE")) {
            fprintf(stderr, "[trace] repeat hills, lists in global memory: %u reads at %u regions / %u raw intervals, %u left (status %u)\n",
                    count, cap_reg, cap_raw, left, status);
        }
        if (left && !(status & (kErrRegionCapacity | kErrRawCapacity))) {
            return fail(ctx, RALA_HIP_EDEVICE, "repeat-hill kernel: reads left without a reason");
        }
        if (status & kErrRegionCapacity) {
            if (cap_reg > (1u << 26)) return fail(ctx, RALA_HIP_ENOMEM, "slope-region lists beyond 2^26 entries");
            cap_reg *= 2; cap_list *= 2;
        }
        if (status & kErrRawCapacity) {
            if (cap_raw > (1u << 28)) return fail(ctx, RALA_HIP_ENOMEM, "interval lists beyond 2^28 entries");
            cap_raw *= 2;
        }
        count = left;
        from ^= 1;
    }
    // (the lists' bits are no error of the call any more)
    launch_status_clear(a.error, kErrRegionCapacity | kErrRawCapacity, nullptr, st);
    HIPCHECK(hipStreamSynchronize(st));
    return RALA_HIP_OK;
}

// One mode of the sensitive pass over the reads of a device list (its length in *count_dev, at most `bound`): every read
// starts in the kernel that fits it, known beforehand from its length and its primary + sensitive event count
// (launch_sens_split) - cap 512 / 16384 bases and cap 512 / 32768 bases on the main stream; cap 1024 / 16384 bases and, behind
// it, the position-space kernel for what fits neither (the event-dense reads: 6 842 of C5's 281 k targets, a third of the
// pass when they ran behind the others) BESIDE them on the aux stream (round 5).  What a kernel still hands on (its region
// lists) goes down the chain behind the join: cap 512 -> cap 1024 -> position space; usually nothing.
// Two looks from the host: the classes' sizes (with whatever the caller has queued on this context: d2h_small), and the
// hand-over counts at the end - which come with the sixteen words of d_small (the repeat hills' pool counter and the status
// word among them) in `small16`.  *count_out: the list's length.
int run_sens_pass(rala_hip_ctx* ctx, PileArgs pa, const RepeatArgs& ra, int mode, const uint32_t* list_dev, uint32_t bound,
                  const uint32_t* count_dev, uint32_t* count_out, uint32_t* small16) {
    *count_out = 0;
    if (bound == 0) return RALA_HIP_OK;
    hipStream_t s = ctx->stream;
    Trace trc;
    if (!ctx->use_run_kernel || !ctx->ev_ready) {
        uint32_t count = 0;
        HIPCHECK(d2h_small(ctx, &count, count_dev, 4, s));
        HIPCHECK(stream_sync(ctx, s));
        *count_out = count;
        std::vector<uint32_t> host(count);
        if (count) HIPCHECK(hipMemcpy(host.data(), list_dev, (size_t)count * 4, hipMemcpyDeviceToHost));
        std::sort(host.begin(), host.end());
        const int rc = run_repeats_kernel(ctx, ra, host, mode);
        if (rc != RALA_HIP_OK) return rc;
        HIPCHECK(hipMemcpy(small16, ctx->d_small.p, 64, hipMemcpyDeviceToHost));
        return RALA_HIP_OK;
    }
    HIPCHECK(ctx->d_overflow.ensure(ctx->n_reads + 1));
    HIPCHECK(ctx->d_overflow_mid.ensure(ctx->n_reads + 1));
    HIPCHECK(ctx->d_chain_cnt.ensure(16));
    // [0] handed on by the cap-512 kernels, [1] by the cap-1024 and cap-2048 kernels, [8 .. 12] the classes' sizes
    {
        FillList fills;
        fills.add(ctx->d_chain_cnt.p, 0, 4 * 4);
        fills.add(ctx->d_chain_cnt.p + 8, 0, 5 * 4);
        HIPCHECK(fills.launch(s));
    }
    HIPCHECK(ctx->d_sens_split.ensure(5 * ((size_t)bound + 1)));
    SensSplitArgs sp;
    sp.read_len = pa.read_len; sp.ev_off = pa.ev_off; sp.ev_cnt = pa.ev_cnt; sp.ev_stride = pa.ev_stride; sp.ev_shift = pa.ev_shift;
    sp.sens_off = pa.sens_off; sp.begin = pa.begin; sp.end = pa.end;
    for (int c = 0; c < 5; ++c) sp.out[c] = ctx->d_sens_split.p + (size_t)c * ((size_t)bound + 1);
    sp.counts = ctx->d_chain_cnt.p + 8;
    launch_sens_split(list_dev, bound, count_dev, sp, s);
    // look 1: the classes' sizes, the list's length; the first reads of the position-space class ride along
    constexpr uint32_t kRestAhead = 16384;
    HIPCHECK(ctx->p_sens_rest.ensure(kRestAhead));
    uint32_t cls[5] = {0, 0, 0, 0, 0}, count = 0;
    HIPCHECK(d2h_small(ctx, cls, ctx->d_chain_cnt.p + 8, 20, s));
    HIPCHECK(d2h_small(ctx, &count, count_dev, 4, s));
    HIPCHECK(hipMemcpyAsync(ctx->p_sens_rest.p, sp.out[4], (size_t)std::min<uint32_t>(bound, kRestAhead) * 4, hipMemcpyDeviceToHost, s));
    HIPCHECK(stream_sync(ctx, s));
    HIPCHECK(hipGetLastError());
    *count_out = count;
    if (count == 0) {
        HIPCHECK(hipMemcpy(small16, ctx->d_small.p, 64, hipMemcpyDeviceToHost));
        return RALA_HIP_OK;
    }
    uint32_t* const handed_512 = ctx->d_overflow.p;            // -> cap 1024
    uint32_t* const handed_1024 = ctx->d_overflow_mid.p;       // -> position space
    // (three streams: the two cap-512 kernels on the main one, the cap-1024 kernel on aux, the position-space kernels - whose
    // host side waits for its own stream between its steps - on side.  First version of this round: both of the latter on aux,
    // and the wait for the rest of a long class-3 list was a wait for the cap-1024 kernel: C5 12.4 ms, no gain)
    hipStream_t aux = ctx->use_side_stream ? ctx->aux : s;
    hipStream_t pos = ctx->use_side_stream ? ctx->side : s;
    if (aux != s) {
        HIPCHECK(hipEventRecord(ctx->ev[8], s));
        HIPCHECK(hipStreamWaitEvent(aux, ctx->ev[8], 0));
        HIPCHECK(hipStreamWaitEvent(pos, ctx->ev[8], 0));
    }
    pa.n_items_dev = nullptr;
    pa.order = sp.out[0]; pa.n_items = cls[0];
    launch_pile_sens(pa, cls[0], 0, mode, handed_512, ctx->d_chain_cnt.p, s);
    pa.order = sp.out[1]; pa.n_items = cls[1];
    launch_pile_sens(pa, std::min<uint32_t>(cls[1], 16384), 3, mode, handed_512, ctx->d_chain_cnt.p, s);
    pa.order = sp.out[2]; pa.n_items = cls[2];
    launch_pile_sens(pa, std::min<uint32_t>(cls[2], 8192), 1, mode, handed_1024, ctx->d_chain_cnt.p + 1, aux);
    if (aux != s) HIPCHECK(hipEventRecord(ctx->ev[9], aux));
    // (the event-dense reads: run space at 2048 events, one wavefront per SIMD - 6 842 of C5's 281 k targets took as long in
    // position space as all the others in theirs; what this kernel hands on goes down the chain with the cap-1024 kernel's)
    pa.order = sp.out[3]; pa.n_items = cls[3];
    launch_pile_sens(pa, std::min<uint32_t>(cls[3], 4096), 2, mode, handed_1024, ctx->d_chain_cnt.p + 1, pos);
    if (cls[4]) {
        std::vector<uint32_t> rest(ctx->p_sens_rest.p, ctx->p_sens_rest.p + std::min<uint32_t>(cls[4], kRestAhead));
        if (cls[4] > kRestAhead) {
            rest.resize(cls[4]);
            HIPCHECK(hipMemcpyAsync(rest.data() + kRestAhead, sp.out[4] + kRestAhead, (size_t)(cls[4] - kRestAhead) * 4, hipMemcpyDeviceToHost, pos));
            HIPCHECK(hipStreamSynchronize(pos));
        }
        std::sort(rest.begin(), rest.end());
        const int rc = run_repeats_kernel(ctx, ra, rest, mode, pos);
        if (rc != RALA_HIP_OK) return rc;
        trc("sens pass: position space beside", rest.size());
    }
    if (aux != s) {
        HIPCHECK(hipStreamWaitEvent(s, ctx->ev[9], 0));
        if (cls[3] || cls[4]) {
            HIPCHECK(hipEventRecord(ctx->ev[8], pos));
            HIPCHECK(hipStreamWaitEvent(s, ctx->ev[8], 0));
        }
    }
    pa.order = handed_512; pa.n_items = count; pa.n_items_dev = ctx->d_chain_cnt.p;
    launch_pile_sens(pa, std::min<uint32_t>(count, 2048), 1, mode, handed_1024, ctx->d_chain_cnt.p + 1, s);
    // look 2: what was handed on, and the call's counters
    uint32_t cnt[2] = {0, 0};
    HIPCHECK(d2h_small(ctx, cnt, ctx->d_chain_cnt.p, 8, s));
    HIPCHECK(d2h_small(ctx, small16, ctx->d_small.p, 64, s));
    HIPCHECK(stream_sync(ctx, s));
    HIPCHECK(hipGetLastError());
    trc(mode == 1 ? "sens pass 1: run space" : "sens pass 2: run space", count);
    if (getenv("RALA_HIP_TRACE")) {
        fprintf(stderr, "[trace] sens pass %d: classes %u / %u / %u / %u / %u, handed on %u + %u\n", mode, cls[0], cls[1], cls[2], cls[3], cls[4], cnt[0], cnt[1]);
    }
    if (cnt[1] == 0) return RALA_HIP_OK;
    std::vector<uint32_t> rest((size_t)cnt[1]);
    HIPCHECK(hipMemcpy(rest.data(), handed_1024, (size_t)cnt[1] * 4, hipMemcpyDeviceToHost));
    std::sort(rest.begin(), rest.end());
    const int rc = run_repeats_kernel(ctx, ra, rest, mode);
    trc("sens pass: position space", rest.size());
    if (rc != RALA_HIP_OK) return rc;
    HIPCHECK(hipMemcpy(small16, ctx->d_small.p, 64, hipMemcpyDeviceToHost));
    return RALA_HIP_OK;
}

// Pile::is_valid_overlap (pile.cpp:605-630)
bool is_valid_overlap(rala_hip_ctx* ctx, uint32_t r, uint32_t x, uint32_t y) {
    const uint32_t B = ctx->h_begin[r], E = ctx->h_end[r];
    const Interval* h = ctx->h_rep_pool.data() + (ctx->h_n_rep[r] ? ctx->h_rep_slot[r] : 0);
    for (uint32_t i = 0; i < ctx->h_n_rep[r]; ++i) {
        if (!(x < h[i].second && h[i].first < y)) continue;
        if ((double)h[i].first < 0.1 * (double)(E - B) + (double)B) {
            if (y < h[i].second + kHillFuzz && h[i].aux) return false;
        } else if ((double)h[i].second > 0.9 * (double)(E - B) + (double)B) {
            if (x + kHillFuzz > h[i].first && h[i].aux) return false;
        }
    }
    return true;
}

int materialize_host(rala_hip_ctx* ctx);
// the repeat hills (sensitive pass) as the host getters want them
int download_repeat_hills(rala_hip_ctx* ctx) {
    if (!ctx->rep_host_stale) return RALA_HIP_OK;
    const uint64_t n = ctx->n_reads;
    HIPCHECK(hipSetDevice(ctx->device));
    ctx->h_n_rep.resize(n); ctx->h_rep_slot.resize(n);
    HIPCHECK(hipMemcpy(ctx->h_n_rep.data(), ctx->d_n_rep.p, n * 4, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(ctx->h_rep_slot.data(), ctx->d_rep_slot.p, n * 4, hipMemcpyDeviceToHost));
    ctx->h_rep_pool.resize(ctx->n_rep_hills);
    if (ctx->n_rep_hills) {
        HIPCHECK(hipMemcpy(ctx->h_rep_pool.data(), ctx->d_rep_pool.p, (size_t)ctx->n_rep_hills * sizeof(Interval), hipMemcpyDeviceToHost));
    }
    ctx->rep_host_stale = false;
    return RALA_HIP_OK;
}
void build_graph(rala_hip_ctx* ctx);
TailList tail_list(rala_hip_ctx* ctx);
int tail_components(rala_hip_ctx* ctx, const TailList& L, uint32_t n_alive, bool touched_cleared = false);
int gpu_tail_part_b(rala_hip_ctx* ctx);

// Graph::preprocess(overlaps, sensitive path) (graph.cpp:882-1054).  Sensitive records:
// a = query (original read, untrimmed coordinates), b = target (trimmed read of the -p run).
//
// cs holds all reads and the lists the chimera stage left (on the host); cl holds the piles.  On
// one GPU they are the same context and comm is null.  In a sharded run cs holds this rank's
// share of the sensitive overlaps and cl the reads this rank owns (local read j = read j * P +
// rank): the target bounds travel to the owners in ONE all-to-all, the owners add the layers,
// take the medians and - once the component medians are known everywhere - look for the repeat
// hills; medians and hills are all-gathered, the hills' bridged flags all-reduced (max).
int preprocess_repeats(rala_hip_ctx* cs, rala_hip_ctx* cl, Comm* comm, const rala_hip_overlaps* sens, uint64_t n_sens,
                       bool device_lists) {
    // device_lists: the chimera stage left its lists, valid regions and liveness on the device
    // (gpu_tail_part_a); components, the final filter and (afterwards) the graph stay there too
    rala_hip_ctx* ctx = cs;                               // error reporting (HIPCHECK / fail)
    const bool sharded = comm != nullptr;
    const uint32_t P = sharded ? comm->world() : 1u, me = sharded ? comm->rank() : 0u;
    const uint64_t n = cs->n_reads;                       // all reads
    const uint64_t nl = cl->n_reads;                      // the reads whose piles are here
    const uint64_t nl_pad = ((n + P - 1) / P + 15) / 16 * 16;
    hipStream_t s = cs->stream;
    hipStream_t sl = cl->stream;
    Trace trc;
    auto comm_fail = [&](const char* what) {
        cs->err = std::string(what) + ": " + comm->error();
        cs->stage_pending.clear();
        cs->stage_used = 0;
        return RALA_HIP_EDEVICE;
    };
    if (n_sens >= 0x7FFFFFF0ull / 2) return fail(ctx, RALA_HIP_EINVAL, "too many sensitive overlaps");
    // the sensitive overlaps stay on the device from here on: columns as given (uploaded, or
    // adopted when the caller says they are device memory)
    OvlSoA so;
    {
        const uint32_t* src[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        const uint8_t* dev_strand = nullptr;
        if (n_sens) {
            src[0] = sens->a_id; src[1] = sens->b_id; src[2] = sens->a_begin; src[3] = sens->a_end;
            src[4] = sens->b_begin; src[5] = sens->b_end; src[6] = sens->length;
            dev_strand = sens->strand;
        }
        const uint32_t* dev[7];
        for (int k = 0; k < 7; ++k) dev[k] = src[k];
        if (!cs->sens_in_device) {
            for (int k = 0; k < 7; ++k) {
                HIPCHECK(cs->d_sens_col[k].ensure(n_sens));
                if (n_sens) HIPCHECK(hipMemcpyAsync(cs->d_sens_col[k].p, src[k], n_sens * 4, hipMemcpyHostToDevice, s));
                dev[k] = cs->d_sens_col[k].p;
            }
            HIPCHECK(cs->d_sens_strand.ensure(n_sens));
            if (n_sens) HIPCHECK(hipMemcpyAsync(cs->d_sens_strand.p, sens->strand, n_sens, hipMemcpyHostToDevice, s));
            dev_strand = cs->d_sens_strand.p;
        }
        so.a_id = dev[0]; so.b_id = dev[1]; so.a_begin = dev[2]; so.a_end = dev[3];
        so.b_begin = dev[4]; so.b_end = dev[5]; so.length = dev[6]; so.strand = dev_strand; so.n = n_sens; so.base = 0;
    }
    if (!device_lists) {
        // current valid regions and liveness on the device (a host tail narrowed them on the host)
        HIPCHECK(hipMemcpyAsync(cs->d_begin.p, cs->h_begin.data(), n * 4, hipMemcpyHostToDevice, s));
        HIPCHECK(hipMemcpyAsync(cs->d_end.p, cs->h_end.data(), n * 4, hipMemcpyHostToDevice, s));
        HIPCHECK(hipMemcpyAsync(cs->d_alive.p, cs->h_alive.data(), n, hipMemcpyHostToDevice, s));
    }
    // Overlap::transmute_ (overlap.cpp:84-114) + bounds of the targets, no +-15 (graph.cpp:929-933)
    for (int k = 0; k < 2; ++k) HIPCHECK(cs->d_sens_tb[k].ensure(n_sens));
    HIPCHECK(cs->d_sens_tuples.ensure(2 * n_sens + 8));
    HIPCHECK(cs->d_dataset_median.ensure(n));
    HIPCHECK(cs->d_n_rep.ensure(n));
    HIPCHECK(cs->d_rep_slot.ensure(n));
    cs->rep_pool_cap = std::max(cs->rep_pool_cap, cs->pool_cap_first);
    HIPCHECK(cs->d_rep_pool.ensure(cs->rep_pool_cap));
    HIPCHECK(hipMemsetAsync(cs->d_n_rep.p, 0, n * 4, s));
    HIPCHECK(hipMemsetAsync(cs->d_small.p + 6, 0, 8, s));           // [6] rep pool count [7] error
    HIPCHECK(hipMemsetAsync(cs->d_small.p + 2, 0, 4, s));           // [2] bad sensitive record
    // One context: the target bounds as ONE 8-byte record per overlap, bucketed by the partitioned path the owners' records
    // of a sharded run take (bucket_kernels.hip; no +-15 here) - where the set suits that path; two tuples per overlap through
    // count / scan / scatter otherwise (and in a sharded run, where the tuples travel first)
    const bool sens_records = !sharded && cl->use_partitioned_buckets && n < (1u << kBoundRecordReadBits) - 1u &&
                              partition_path_fits_records((uint32_t)nl, cl->max_read_len, 2 * n_sens);
    if (sens_records) {
        HIPCHECK(cs->d_sens_rec.ensure(n_sens + 8));
        launch_sens_records(so, (uint32_t)n, cs->d_begin.p, cs->d_alive.p, cs->d_sens_tb[0].p, cs->d_sens_tb[1].p, cs->d_sens_rec.p,
                            cs->d_small.p + 2, s);
    } else {
        launch_sens_tuples(so, (uint32_t)n, cs->d_begin.p, cs->d_alive.p, cs->d_sens_tb[0].p, cs->d_sens_tb[1].p,
                           cs->d_sens_tuples.p, cs->d_small.p + 2, s);
    }
    const uint2* tuples = cs->d_sens_tuples.p;          // what the pile holder buckets: {its read, bound}
    uint64_t n_tuples = 2 * n_sens;
    if (sharded) {
        // by owner of the target, ONE all-to-all
        HIPCHECK(cs->d_owner_cnt.ensure(2 * 64));
        HIPCHECK(cs->d_sens_part.ensure(2 * n_sens + 8));
        uint32_t* cnt = cs->d_owner_cnt.p;
        uint32_t* cur = cnt + 64;
        HIPCHECK(hipMemsetAsync(cnt, 0, 2 * 64 * 4, s));
        launch_partition_tuples(cs->d_sens_tuples.p, 2 * n_sens, P, 0, cnt, cs->d_sens_part.p, s);
        uint32_t h[64];
        HIPCHECK(hipMemcpyAsync(h, cnt, P * 4, hipMemcpyDeviceToHost, s));
        HIPCHECK(stream_sync(cs, s));
        uint32_t off[64];
        std::vector<uint64_t> send_counts(P), matrix((size_t)P * P), recv_counts(P);
        uint32_t acc = 0;
        for (uint32_t p = 0; p < P; ++p) { off[p] = acc; acc += h[p]; send_counts[p] = h[p]; }
        HIPCHECK(hipMemcpyAsync(cur, off, P * 4, hipMemcpyHostToDevice, s));
        launch_partition_tuples(cs->d_sens_tuples.p, 2 * n_sens, P, 1, cur, cs->d_sens_part.p, s);
        // a bad record anywhere is everybody's error
        if (comm->all_reduce_u32(cs->d_small.p + 2, 1, ReduceOp::kMax, s) != 0) return comm_fail("all-reduce of the record check");
        if (comm->host_all_gather(send_counts.data(), P, matrix.data(), s) != 0) return comm_fail("sensitive tuple counts");
        n_tuples = 0;
        for (uint32_t p = 0; p < P; ++p) { recv_counts[p] = matrix[(size_t)p * P + me]; n_tuples += recv_counts[p]; }
        HIPCHECK(cs->d_sens_recv.ensure(n_tuples + 8));
        if (comm->all_to_all_v(cs->d_sens_part.p, send_counts.data(), cs->d_sens_recv.p, recv_counts.data(), sizeof(uint2), s) != 0) {
            return comm_fail("all-to-all of the sensitive bounds");
        }
        tuples = cs->d_sens_recv.p;
        // the owner's copy of the current valid regions (the chimera stage narrowed them everywhere)
        HIPCHECK(stream_sync(cs, s));
        launch_localize_u32(cs->d_begin.p, nl, P, me, cl->d_begin.p, sl);
        launch_localize_u32(cs->d_end.p, nl, P, me, cl->d_end.p, sl);
        launch_localize_u8(cs->d_alive.p, nl, P, me, cl->d_alive.p, sl);
    }
    // (one context: the record check comes back with the first look of the pass below - a record that is an error names no
    // read, nothing downstream trips over it; a sharded run has waited for its collectives anyway)
    uint32_t bad = 0;
    HIPCHECK(d2h_small(cs, &bad, cs->d_small.p + 2, 4, s));
    auto check_records = [&]() -> int {
        if (bad & 1u) return fail(ctx, RALA_HIP_EINVAL, "sensitive overlap names must resolve");
        if (bad & 2u) return fail(ctx, RALA_HIP_EINVAL, "sensitive overlap targets a read that did not survive");
        return (int)RALA_HIP_OK;
    };
    if (sharded) {
        HIPCHECK(stream_sync(cs, s));
        HIPCHECK(hipGetLastError());
        const int rcb = check_records();
        if (rcb != RALA_HIP_OK) return rcb;
    }
    // bucketed by read on the pile holder: count -> scan -> scatter
    // (buffers of their own: the primary bound events stay where initialize left them)
    HIPCHECK(cl->d_sens_ev.ensure(n_tuples + 8));
    HIPCHECK(cl->d_sens_off.ensure(nl + 2));
    HIPCHECK(cl->d_sens_cur.ensure(nl + 2));
    HIPCHECK(cl->d_scan_ws.ensure(scan_workspace_bytes(std::max<uint64_t>(n_tuples, nl) + 2)));
    HIPCHECK(cl->d_dataset_median.ensure(nl));
    HIPCHECK(cl->d_n_rep.ensure(nl));
    HIPCHECK(cl->d_rep_slot.ensure(nl));
    cl->rep_pool_cap = std::max(cl->rep_pool_cap, cl->pool_cap_first);
    HIPCHECK(cl->d_rep_pool.ensure(cl->rep_pool_cap));
    if (sharded) {
        HIPCHECK(hipMemsetAsync(cl->d_n_rep.p, 0, nl * 4, sl));
        HIPCHECK(hipMemsetAsync(cl->d_small.p + 6, 0, 8, sl));
    }
    if (sens_records) {
        // (the first pass' bucketing buffers have served: the bound events themselves stay where initialize left them)
        const uint32_t n_reads_l = (uint32_t)nl;
        HIPCHECK(cl->d_bk_u32[0].ensure(n_reads_l + 2));
        HIPCHECK(cl->d_bk_part.ensure(partition_count(n_reads_l) + 2));
        HIPCHECK(cl->d_bk_group.ensure(3 * (size_t)partition_group_slots(n_reads_l)));
        HIPCHECK(cl->d_bk_tiles.ensure(3 * partition_tile_slots(n_reads_l, n_sens) + 2));
        for (int k = 0; k < 2; ++k) HIPCHECK(cl->d_bk_rec[k].ensure(partition_records_needed(n_reads_l, n_sens)));
        FillList fills;
        HIPCHECK(launch_bucket_partitioned_records(cs->d_sens_rec.p, n_sens, n_reads_l, cl->d_bk_u32[0].p, cl->d_bk_part.p, cl->d_bk_group.p,
                                                   cl->d_bk_tiles.p, cl->d_bk_rec[0].p, cl->d_bk_rec[1].p, cl->d_sens_off.p, cl->d_sens_ev.p,
                                                   cl->n_compute_units, fills, sl, 0u, 0u));       // (the sensitive CSR: in events)
    } else {
        HIPCHECK(hipMemsetAsync(cl->d_sens_cur.p, 0, (nl + 1) * 4, sl));
        launch_count_tuples(tuples, n_tuples, (uint32_t)nl, cl->d_sens_cur.p, sl);
        launch_exclusive_scan(cl->d_sens_cur.p, cl->d_sens_off.p, nl, cl->d_scan_ws.p, sl);
        HIPCHECK(hipMemcpyAsync(cl->d_sens_cur.p, cl->d_sens_off.p, nl * 4, hipMemcpyDeviceToDevice, sl));
        launch_scatter_tuples(tuples, n_tuples, (uint32_t)nl, cl->d_sens_cur.p, cl->d_sens_ev.p, sl);
    }
    // the targets: reads that received bounds - listed where the offsets are, the host learns how many
    HIPCHECK(cl->d_sens_list.ensure(nl + 1));
    HIPCHECK(cl->d_chain_cnt.ensure(16));
    HIPCHECK(hipMemsetAsync(cl->d_chain_cnt.p + 4, 0, 4, sl));
    launch_list_targets(cl->d_sens_off.p, (uint32_t)nl, cl->d_sens_list.p, cl->d_chain_cnt.p + 4, sl);
    uint32_t n_targets = 0;
    uint32_t small[16] = {};

    RepeatArgs a;
    a.read_len = cl->d_read_len.p; a.pile_off = cl->d_pile_off.p; a.pile = cl->d_pile.p;
    a.ev_off = cl->d_sens_off.p; a.ev = cl->d_sens_ev.p;
    a.begin = cl->d_begin.p; a.end = cl->d_end.p; a.median = cl->d_median.p; a.p10 = cl->d_p10.p;
    a.dataset_median = cl->d_dataset_median.p; a.n_rep = cl->d_n_rep.p; a.rep_slot = cl->d_rep_slot.p;
    a.pool = cl->d_rep_pool.p; a.pool_count = cl->d_small.p + 6; a.pool_cap = cl->rep_pool_cap;
    a.error = cl->d_small.p + 7;
    a.order = nullptr; a.n_items = 0; a.lw = 0; a.slab = nullptr;
    // the same in run space: the primary events + the sensitive bounds
    PileArgs pa;
    memset(&pa, 0, sizeof(pa));
    pa.read_len = cl->d_read_len.p; pa.pile_off = cl->d_pile_off.p; pa.pile = cl->d_pile.p;
    pa.ev_off = cl->d_ev_off.p; pa.ev = cl->ev_fixed ? cl->d_ev_fixed.p : cl->d_ev.p;
    pa.ev_cnt = cl->ev_fixed ? cl->d_cursor.p : nullptr; pa.ev_stride = kRunEventCapBig; pa.ev_shift = cl->ev_shift;
    pa.stop_after = 99;
    pa.begin = cl->d_begin.p; pa.end = cl->d_end.p; pa.median = cl->d_median.p; pa.p10 = cl->d_p10.p;
    pa.error = cl->d_small.p + 7;
    pa.sens_off = cl->d_sens_off.p; pa.sens_ev = cl->d_sens_ev.p;
    pa.dataset_median = cl->d_dataset_median.p; pa.n_rep = cl->d_n_rep.p; pa.rep_slot = cl->d_rep_slot.p;
    pa.rep_pool = cl->d_rep_pool.p; pa.rep_pool_count = cl->d_small.p + 6; pa.rep_pool_cap = cl->rep_pool_cap;
    // add_layers on top of the coverage + find_median for the targets (graph.cpp:941-969)
    int rc = run_sens_pass(cl, pa, a, 1, cl->d_sens_list.p, (uint32_t)nl, cl->d_chain_cnt.p + 4, &n_targets, small);
    if (!sharded) {
        // (the record check came back with the pass' first look; when the pass itself failed, a record that is an error is the
        // failure to report - the pass ran over what such records left behind: advisor round 5)
        const bool looked = rc == RALA_HIP_OK || stream_sync(cs, s) == hipSuccess;
        const int rcb = looked ? check_records() : (int)RALA_HIP_OK;
        if (rcb != RALA_HIP_OK) return rcb;
    }
    if (rc != RALA_HIP_OK) { if (cl != cs) cs->err = cl->err; return rc; }
    if (sharded) {
        // the new medians, everywhere
        HIPCHECK(cs->d_gather[0].ensure(nl_pad * 4 + 16));
        HIPCHECK(cs->d_gather[1].ensure(nl_pad * 4 * P + 16));
        launch_pack_median(cl->d_median.p, cl->d_p10.p, nl, nl_pad, (uint32_t*)cs->d_gather[0].p, s);
        if (comm->all_gather(cs->d_gather[0].p, cs->d_gather[1].p, nl_pad * 4, s) != 0) return comm_fail("all-gather of the medians");
        launch_unpack_median((const uint32_t*)cs->d_gather[1].p, P, nl_pad, n, cs->d_median.p, cs->d_p10.p, s);
    }
    if (!device_lists) {
        HIPCHECK(hipMemcpyAsync(cs->h_median.data(), cs->d_median.p, n * 2, hipMemcpyDeviceToHost, s));
        HIPCHECK(hipMemcpyAsync(cs->h_p10.data(), cs->d_p10.p, n * 2, hipMemcpyDeviceToHost, s));
        HIPCHECK(stream_sync(cs, s));
    }
    trc("rep: add_layers + median", n_targets);
    // first trim of the sensitive overlaps (graph.cpp:935-939)
    SensCoords sc;
    for (int k = 0; k < 5; ++k) HIPCHECK(cs->d_sens_c[k].ensure(n_sens));
    HIPCHECK(cs->d_sens_state.ensure(n_sens));
    sc.a_begin = cs->d_sens_c[0].p; sc.a_end = cs->d_sens_c[1].p; sc.b_begin = cs->d_sens_c[2].p;
    sc.b_end = cs->d_sens_c[3].p; sc.length = cs->d_sens_c[4].p; sc.state = cs->d_sens_state.p;
    launch_sens_trim(so, cs->d_sens_tb[0].p, cs->d_sens_tb[1].p, cs->d_begin.p, cs->d_end.p, cs->d_alive.p, sc, s);
    trc("rep: first trim", n_sens);
    // component medians over the primary overlaps -> repeat hills (graph.cpp:971-1026)
    std::vector<uint32_t> members;
    bool members_on_device = false;
    uint32_t n_members = 0;
    if (!device_lists) {
        std::vector<uint16_t> med;
        rc = component_medians(cs, members, med);
        if (rc != RALA_HIP_OK) return rc;
        std::vector<uint16_t> dm(n, 0);
        for (size_t k = 0; k < members.size(); ++k) dm[members[k]] = med[k];
        HIPCHECK(hipMemcpy(cs->d_dataset_median.p, dm.data(), n * 2, hipMemcpyHostToDevice));
    } else {
        // on the device list, in the rank space of the chimera stage (the new medians are in d_median)
        const TailList L = tail_list(cs);
        const uint32_t n_alive = cs->t_n_alive;
        rc = tail_components(cs, L, n_alive);
        if (rc != RALA_HIP_OK) return rc;
        HIPCHECK(hipMemsetAsync(cs->d_dataset_median.p, 0, n * 2, s));
        launch_scatter_component_medians(cs->d_alive_reads.p, cs->d_touched.p, cs->d_cmed.p, n_alive, cs->d_dataset_median.p, s);
        // the members (reads with an overlap; of a sharded run this rank's, as the owner's local ids) are listed where the
        // data is; the host learns how many
        HIPCHECK(cl->d_sens_list.ensure(std::max<uint64_t>(nl, 1) + 1));
        HIPCHECK(cs->d_chain_cnt.ensure(16));
        HIPCHECK(hipMemsetAsync(cs->d_chain_cnt.p + 5, 0, 4, s));
        launch_list_members(cs->d_alive_reads.p, cs->d_touched.p, n_alive, P, me, cl->d_sens_list.p, cs->d_chain_cnt.p + 5, s);
        if (sharded) HIPCHECK(stream_sync(cs, s));              // (the owner context's stream goes on from here)
        members_on_device = true;
    }
    std::vector<uint32_t> mine;
    if (sharded) {
        launch_localize_u16(cs->d_dataset_median.p, nl, P, me, cl->d_dataset_median.p, sl);
        if (!members_on_device) for (uint32_t r : members) if (r % P == me) mine.push_back(r / P);
    }
    uint32_t member_bound = (uint32_t)nl;
    if (!members_on_device) {
        // (host tail: the members were listed on the host)
        const std::vector<uint32_t>& list = sharded ? mine : members;
        HIPCHECK(cl->d_sens_list.ensure(std::max<uint64_t>(list.size(), 1) + 1));
        HIPCHECK(cs->d_chain_cnt.ensure(16));
        const uint32_t n_list = (uint32_t)list.size();
        if (n_list) HIPCHECK(hipMemcpy(cl->d_sens_list.p, list.data(), (size_t)n_list * 4, hipMemcpyHostToDevice));
        HIPCHECK(hipMemcpy(cs->d_chain_cnt.p + 5, &n_list, 4, hipMemcpyHostToDevice));
        member_bound = n_list;
    }
    for (;;) {
        rc = run_sens_pass(cl, pa, a, 2, cl->d_sens_list.p, member_bound, cs->d_chain_cnt.p + 5, &n_members, small);
        if (rc != RALA_HIP_OK) { if (cl != cs) cs->err = cl->err; return rc; }
        const uint32_t st[2] = {small[6], small[7]};
        if (!(st[1] & kErrPoolCapacity)) break;
        // more repeat hills than the pool holds: its counter holds what is needed - the pass once more (mode 2 neither
        // reads nor writes the rows) with a pool of that size
        const uint64_t need = (uint64_t)st[0] + st[0] / 8 + 1024;
        if (st[0] <= cl->rep_pool_cap || need > 0xFFFFFFF0ull) return fail(ctx, RALA_HIP_ENOMEM, "repeat-hill pool beyond 2^32 entries");
        cl->rep_pool_cap = (uint32_t)need;
        HIPCHECK(cl->d_rep_pool.ensure(cl->rep_pool_cap));
        a.pool = cl->d_rep_pool.p; a.pool_cap = cl->rep_pool_cap;
        pa.rep_pool = cl->d_rep_pool.p; pa.rep_pool_cap = cl->rep_pool_cap;
        HIPCHECK(hipMemsetAsync(cl->d_n_rep.p, 0, nl * 4, sl));
        HIPCHECK(hipMemsetAsync(cl->d_small.p + 6, 0, 8, sl));
        ++cl->tm.pool_regrown;
    }
    trc("rep: component medians + repeat hills", n_members);
    uint32_t n_hills = small[6];
    if (sharded) {
        // capacity errors are everybody's; then the hills of all owners, slots rebased onto the
        // concatenation of their pools
        HIPCHECK(hipMemcpyAsync(cs->d_small.p + 7, &small[7], 4, hipMemcpyHostToDevice, s));
        if (comm->all_reduce_u32(cs->d_small.p + 7, 1, ReduceOp::kMax, s) != 0) return comm_fail("all-reduce of the kernel status");
        HIPCHECK(d2h_small(cs, &small[7], cs->d_small.p + 7, 4, s));
        HIPCHECK(stream_sync(cs, s));
    }
    if (small[7] & (kErrRegionCapacity | kErrRawCapacity)) return fail(ctx, RALA_HIP_EDEVICE, "slope-region list overflow (repeat hills)");
    if (small[7] & kErrPoolCapacity) return fail(ctx, RALA_HIP_EDEVICE, "repeat-hill pool exhausted");
    if (sharded) {
        const uint64_t mine = std::min<uint32_t>(small[6], cl->rep_pool_cap);
        std::vector<uint64_t> counts(P);
        if (comm->host_all_gather(&mine, 1, counts.data(), s) != 0) return comm_fail("repeat-hill counts");
        RankOffsets base;
        uint64_t total = 0;
        for (uint32_t p = 0; p < P; ++p) { base.v[p] = (uint32_t)total; total += counts[p]; }
        HIPCHECK(cs->d_rep_pool.ensure(total + 1));
        HIPCHECK(cs->d_gather[0].ensure(nl_pad * 8 + 16));
        HIPCHECK(cs->d_gather[1].ensure(nl_pad * 8 * P + 16));
        launch_pack_rep(cl->d_n_rep.p, cl->d_rep_slot.p, nl, nl_pad, (uint64_t*)cs->d_gather[0].p, s);
        if (comm->all_gather(cs->d_gather[0].p, cs->d_gather[1].p, nl_pad * 8, s) != 0) return comm_fail("all-gather of the repeat hills");
        launch_unpack_rep((const uint64_t*)cs->d_gather[1].p, P, nl_pad, n, base, cs->d_n_rep.p, cs->d_rep_slot.p, s);
        if (comm->all_gather_v(cl->d_rep_pool.p, cs->d_rep_pool.p, counts.data(), sizeof(Interval), s) != 0) {
            return comm_fail("all-gather of the repeat-hill pools");
        }
        n_hills = (uint32_t)total;
    }
    trc("rep: repeat hills kernel", n_hills);
    // sensitive dovetails mark the hills they bridge (graph.cpp:1028-1043)
    launch_sens_bridge(so, sc, cs->d_begin.p, cs->d_end.p, cs->d_alive.p, cs->d_n_rep.p, cs->d_rep_slot.p,
                       cs->d_rep_pool.p, s);
    if (sharded && n_hills) {
        HIPCHECK(cs->d_t_tmp[0].ensure(std::max<size_t>(n_hills, n) + 2));
        launch_pool_aux(cs->d_rep_pool.p, n_hills, cs->d_t_tmp[0].p, 0, s);
        if (comm->all_reduce_u32(cs->d_t_tmp[0].p, n_hills, ReduceOp::kMax, s) != 0) return comm_fail("all-reduce of the bridged flags");
        launch_pool_aux(cs->d_rep_pool.p, n_hills, cs->d_t_tmp[0].p, 1, s);
    }
    if (sharded || trc.on) {                    // (one context: the next look is the graph's, gpu_tail_part_b)
        HIPCHECK(stream_sync(cs, s));
        HIPCHECK(hipGetLastError());
    }
    trc("rep: bridged hills", n_sens);
    // the host's copy of the hills: at once where the host filters the overlaps below, otherwise with the first getter
    cs->n_rep_hills = n_hills;
    cs->rep_host_stale = true;
    if (!device_lists) {
        const int rcd = download_repeat_hills(cs);
        if (rcd != RALA_HIP_OK) return rcd;
    }
    trc("rep: download hills", n_hills);
    // overlaps that end inside a bridged edge hill are dropped (graph.cpp:1045-1051)
    if (device_lists) {
        launch_sens_filter(tail_list(cs), cs->d_begin.p, cs->d_end.p, cs->d_n_rep.p, cs->d_rep_slot.p, cs->d_rep_pool.p, s);
    } else {
        size_t w = 0;
        for (size_t k = 0; k < cs->overlaps.size(); ++k) {
            const HostOvl& o = cs->overlaps[k];
            if (!is_valid_overlap(cs, o.a, o.c.a_begin, o.c.a_end) ||
                !is_valid_overlap(cs, o.b, o.c.b_begin, o.c.b_end)) {
                continue;
            }
            if (w != k) cs->overlaps[w] = cs->overlaps[k];
            ++w;
        }
        cs->overlaps.resize(w);
    }
    trc("rep: filter overlaps", cs->overlaps.size());
    cs->have_repeats = true;
    return RALA_HIP_OK;
}

// the sensitive pass behind the device-resident chimera stage (gpu_tail_part_a): repeats, final
// filter, then the final list and the graph (gpu_tail_part_b) - nothing comes to the host
int repeats_after_tail(rala_hip_ctx* cs, rala_hip_ctx* cl, Comm* comm, const rala_hip_overlaps* sens, uint64_t n_sens) {
    rala_hip_ctx* ctx = cs;
    hipStream_t s = cs->stream;
    Trace trs;
    launch_finalize_states(tail_list(cs), cs->d_alive.p, s);
    const int rc3 = preprocess_repeats(cs, cl, comm, sens, n_sens, true);
    if (rc3 != RALA_HIP_OK) return rc3;
    trs("sens: preprocess_repeats");
    const int rc4 = gpu_tail_part_b(cs);
    if (rc4 != RALA_HIP_OK) return rc4;
    trs("sens: final list + graph");
    HIPCHECK(stream_sync(cs, s));
    return RALA_HIP_OK;
}

// nodes for the surviving reads, two edges per dovetail overlap (graph.cpp:553-632)
void build_graph(rala_hip_ctx* ctx) {
    const uint64_t n = ctx->n_reads;
    std::vector<uint32_t> read_to_node(n, 0xFFFFFFFFu);
    ctx->node_read.clear();
    for (uint64_t r = 0; r < n; ++r) {
        if (!ctx->h_alive[r]) continue;
        read_to_node[r] = (uint32_t)ctx->node_read.size();
        ctx->node_read.push_back((uint32_t)r);
        ctx->node_read.push_back((uint32_t)r);
    }
    const std::vector<HostOvl>& ov = ctx->overlaps;
    const size_t m = ov.size();
    std::vector<EdgePair>& ep = ctx->scratch_ep;
    std::vector<uint8_t>& has = ctx->scratch_has;
    ep.resize(m);
    has.assign(m, 0);
    const unsigned T = ctx->pool->size();
    std::vector<uint64_t> cnt(T + 1, 0);
    ctx->pool->chunks(m, [&](unsigned t, size_t b, size_t e) {
        uint64_t c = 0;
        for (size_t k = b; k < e; ++k) {
            const HostOvl& o = ov[k];
            const uint32_t Ba = ctx->h_begin[o.a], Ea = ctx->h_end[o.a], Bb = ctx->h_begin[o.b], Eb = ctx->h_end[o.b];
            const uint32_t ty = ovl_type(o.c, o.strand, Ba, Ea, Bb, Eb);
            if (ovl_edges(o.c, o.strand, ty, read_to_node[o.a], read_to_node[o.b], Ba, Ea, Bb, Eb, ep[k])) {
                has[k] = 1;
                ++c;
            }
        }
        cnt[t + 1] = c;
    });
    for (unsigned t = 0; t < T; ++t) cnt[t + 1] += cnt[t];
    const uint64_t ne = 2 * cnt[T];
    ctx->e_src.resize(ne); ctx->e_dst.resize(ne); ctx->e_len.resize(ne);
    ctx->pool->chunks(m, [&](unsigned t, size_t b, size_t e) {
        uint64_t w = 2 * (m < 4096 ? 0 : cnt[t]);
        for (size_t k = b; k < e; ++k) {
            if (!has[k]) continue;
            const EdgePair& p = ep[k];
            ctx->e_src[w] = p.src0; ctx->e_dst[w] = p.dst0; ctx->e_len[w] = p.len0;
            ctx->e_src[w + 1] = p.src1; ctx->e_dst[w + 1] = p.dst1; ctx->e_len[w + 1] = p.len1;
            w += 2;
        }
    });
    ctx->e_mark.assign(ne, 0);
}

// transitive-edge marking on device edge arrays; marks stay in ctx->d_tr_marks
int scan_space(rala_hip_ctx* ctx, uint32_t scans, uint64_t items, ScanSpace& sp, FillList* fills = nullptr);

int tr_mark_device(rala_hip_ctx* ctx, uint32_t n_nodes, uint32_t n_edges, const uint32_t* d_src, const uint32_t* d_dst,
                   const uint32_t* d_len, uint32_t* n_pairs, Comm* comm = nullptr) {
    *n_pairs = 0;
    if (n_edges == 0) return RALA_HIP_OK;
    if (n_edges & 1) return fail(ctx, RALA_HIP_EINVAL, "edges must come in twin pairs (e, e^1)");
    hipStream_t s = ctx->stream;
    DevBuf<uint32_t>* B = ctx->d_tr;          // row, cursor, adj (persistent)
    HIPCHECK(B[0].ensure(n_nodes + 2)); HIPCHECK(B[1].ensure(n_nodes + 2)); HIPCHECK(B[2].ensure(2 * (size_t)n_edges));
    HIPCHECK(ctx->d_tr_marks.ensure((size_t)n_edges + 8));
    HIPCHECK(ctx->d_scan_ws.ensure(scan_workspace_bytes((uint64_t)n_nodes + 2)));
    HIPCHECK(hipEventRecord(ctx->ev[10], s));
    FillList fills;
    fills.add(ctx->d_tr_marks.p, 0, ((size_t)n_edges + 7) & ~(size_t)3);
    fills.add(ctx->d_small.p + 2, 0, 8);                             // [2] bad endpoint flag [3] pairs
    fills.add(B[1].p, 0, (size_t)(n_nodes + 1) * 4);
    // CSR on the device: out-degree count -> scan -> fill.  The fill order is arbitrary; the
    // candidate a->c is the edge with the highest id, which is the reference's "last one
    // in suffix_edges_" (graph.cpp:1291-1293, out-lists are in edge-id order there).
    ScanSpace sp;
    {
        const int rc = scan_space(ctx, 1, n_nodes, sp, &fills);
        if (rc != RALA_HIP_OK) return rc;
    }
    HIPCHECK(fills.launch(s));
    launch_tr_degree(d_src, d_dst, n_nodes, n_edges, B[1].p, ctx->d_small.p + 2, s);
    // row offsets (and the fill cursors, a second copy of them) from the out-degrees: one launch
    HIPCHECK(B[3 + 3].ensure(n_nodes + 2));
    if (!launch_offsets_pass(B[1].p, B[0].p, B[6].p, n_nodes, sp, s)) return fail(ctx, RALA_HIP_EDEVICE, "scan space");
    launch_tr_fill(d_src, d_dst, n_nodes, n_edges, B[6].p, B[2].p, s);
    if (!comm) {
        launch_tr_mark(B[0].p, B[2].p, d_src, d_dst, d_len, n_nodes, 0, n_edges, ctx->d_tr_marks.p, s);
    } else {
        // every rank probes from its share of the edges a->b; the marks (bytes 0 / 1) are summed
        const uint64_t P = comm->world(), k = comm->rank();
        launch_tr_mark(B[0].p, B[2].p, d_src, d_dst, d_len, n_nodes, (uint32_t)(n_edges * k / P), (uint32_t)(n_edges * (k + 1) / P),
                       ctx->d_tr_marks.p, s);
        if (comm->all_reduce_u32((uint32_t*)ctx->d_tr_marks.p, ((size_t)n_edges + 3) / 4, ReduceOp::kSum, s) != 0) {
            ctx->err = std::string("all-reduce of the transitive marks: ") + comm->error();
            return RALA_HIP_EDEVICE;
        }
    }
    launch_tr_count(ctx->d_tr_marks.p, n_edges, ctx->d_small.p + 3, s);
    HIPCHECK(hipEventRecord(ctx->ev[11], s));
    uint32_t res[2] = {0, 0};
    HIPCHECK(d2h_small(ctx, res, ctx->d_small.p + 2, 8, s));
    HIPCHECK(stream_sync(ctx, s));
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipEventElapsedTime(&ctx->tm.tr_ms, ctx->ev[10], ctx->ev[11]));
    if (res[0]) return fail(ctx, RALA_HIP_EINVAL, "edge endpoint out of range");
    *n_pairs = res[1];
    return RALA_HIP_OK;
}

int tr_mark_impl(rala_hip_ctx* ctx, uint32_t n_nodes, uint32_t n_edges, const uint32_t* src, const uint32_t* dst,
                 const uint32_t* len, uint8_t* marks, uint32_t* n_pairs) {
    *n_pairs = 0;
    if (n_edges == 0) return RALA_HIP_OK;
    hipStream_t s = ctx->stream;
    DevBuf<uint32_t>* B = ctx->d_tr;
    HIPCHECK(B[3].ensure(n_edges)); HIPCHECK(B[4].ensure(n_edges)); HIPCHECK(B[5].ensure(n_edges));
    HIPCHECK(hipMemcpyAsync(B[3].p, src, (size_t)n_edges * 4, hipMemcpyHostToDevice, s));
    HIPCHECK(hipMemcpyAsync(B[4].p, dst, (size_t)n_edges * 4, hipMemcpyHostToDevice, s));
    HIPCHECK(hipMemcpyAsync(B[5].p, len, (size_t)n_edges * 4, hipMemcpyHostToDevice, s));
    const int rc = tr_mark_device(ctx, n_nodes, n_edges, B[3].p, B[4].p, B[5].p, n_pairs);
    if (rc != RALA_HIP_OK) return rc;
    HIPCHECK(hipMemcpy(marks, ctx->d_tr_marks.p, n_edges, hipMemcpyDeviceToHost));
    return RALA_HIP_OK;
}

// ---- preprocess tail on the device (tail_kernels.hip) ---------------------------------------
TailList tail_list(rala_hip_ctx* ctx) {
    TailList L;
    L.n = ctx->t_n0 + ctx->t_n1;
    L.src = ctx->d_surv_u32[0].p; L.a = ctx->d_surv_u32[1].p; L.b = ctx->d_surv_u32[2].p;
    L.a_begin = ctx->d_surv_u32[3].p; L.a_end = ctx->d_surv_u32[4].p; L.b_begin = ctx->d_surv_u32[5].p;
    L.b_end = ctx->d_surv_u32[6].p; L.length = ctx->d_surv_u32[7].p;
    L.strand = ctx->d_surv_u8[0].p; L.type = ctx->d_surv_u8[1].p;
    L.state = ctx->d_t_state.p; L.round = ctx->d_t_round.p;
    return L;
}

TailReads tail_reads(rala_hip_ctx* ctx) {
    TailReads R;
    R.begin = ctx->d_begin.p; R.end = ctx->d_end.p; R.alive = ctx->d_alive.p; R.dirty = ctx->d_dirty.p;
    R.n_pits = ctx->d_n_pits.p; R.n_hills = ctx->d_n_hills.p; R.n_pits0 = ctx->d_n_pits0.p;
    R.iv_slot = ctx->d_iv_slot.p; R.pool = ctx->d_pool.p;
    return R;
}

// tile states for the single-pass scans of one stage (scan_pass.h): `scans` scans of up to `items` items;
// cleared (with the ticket counter behind them) by one fill - the caller's list of fills if it has one
int scan_space(rala_hip_ctx* ctx, uint32_t scans, uint64_t items, ScanSpace& sp, FillList* fills) {
    const size_t words = (size_t)scans * (scan_tiles_for(items) + 2) + 2;
    HIPCHECK(ctx->d_scan_state.ensure(words));
    sp.state = ctx->d_scan_state.p;
    sp.words = words - 2;
    sp.used = 0;
    sp.ticket = (uint32_t*)(ctx->d_scan_state.p + words - 2);
    if (fills) fills->add(ctx->d_scan_state.p, 0, words * 8);
    else HIPCHECK(hipMemsetAsync(ctx->d_scan_state.p, 0, words * 8, ctx->stream));
    return RALA_HIP_OK;
}

// connected components over the live overlaps of the device list (labels in the rank space of
// d_rank / d_alive_reads) and, per read with an overlap, the median of the pile medians of its
// component (graph.cpp:740-783): d_touched[rank], d_cmed[rank]
// (touched_cleared: the caller's fill has zeroed d_touched)
int tail_components(rala_hip_ctx* ctx, const TailList& L, uint32_t n_alive, bool touched_cleared) {
    hipStream_t s = ctx->stream;
    const uint32_t M = L.n;
    if (!touched_cleared) {
        FillList fills;
        fills.add(ctx->d_touched.p, 0, n_alive);
        component_median_clear(ctx->d_med_tmp.p, n_alive, fills);
        HIPCHECK(fills.launch(s));
    }
    launch_cc_edges(L, ctx->d_rank.p, ctx->d_cc_edges.p, ctx->d_touched.p, ctx->d_cc_label.p, n_alive, s);
    for (int k = 0; k < 2; ++k) {                               // sampled rounds, see cc_hook_kernel
        launch_cc_hook(ctx->d_cc_edges.p, M, 1, ctx->d_cc_label.p, ctx->d_cc_flags.p + 7, s);
        launch_cc_compress(ctx->d_cc_label.p, n_alive, s);
    }
    // one launch over all edges finishes the components (cc_hook_kernel unites to the end); the last compression
    // counts the components' reads with an overlap
    launch_cc_hook(ctx->d_cc_edges.p, M, 0, ctx->d_cc_label.p, ctx->d_cc_flags.p + 7, s);
    launch_cc_compress_count(ctx->d_cc_label.p, n_alive, ctx->d_touched.p, component_median_sizes(ctx->d_med_tmp.p, n_alive), s);
    // median of the pile medians per component (graph.cpp:777-783)
    HIPCHECK(launch_component_medians(ctx->d_cc_label.p, ctx->d_touched.p, ctx->d_alive_reads.p, ctx->d_median.p, n_alive,
                                      ctx->d_med_tmp.p, ctx->d_cmed.p, s));
    return RALA_HIP_OK;
}

// Graph::preprocess (chimeras) with the survivor lists resident on the device: everything up to
// and including the in-order containment removal (graph.cpp:699-877)
int gpu_tail_part_a(rala_hip_ctx* ctx) {
    hipStream_t s = ctx->stream;
    Trace trc;
    auto mark = [&](const char* what, size_t k = 0) {
        if (trc.on) { (void)stream_sync(ctx, s); trc(what, k); }
    };
    const uint32_t n_reads = (uint32_t)ctx->n_reads;
    const uint32_t M = ctx->t_n0 + ctx->t_n1;
    // (how many reads are alive came back with the survivor counts of the second pass: no look from the host here)
    const uint32_t n_alive = ctx->t_n_alive;
    const size_t big = (size_t)std::max<uint64_t>(n_reads, M) + 2;
    HIPCHECK(ctx->d_t_state.ensure(M)); HIPCHECK(ctx->d_t_round.ensure(M));
    HIPCHECK(ctx->d_dirty.ensure(n_reads)); HIPCHECK(ctx->d_n_pits0.ensure(n_reads));
    HIPCHECK(ctx->d_rank.ensure(n_reads)); HIPCHECK(ctx->d_alive_reads.ensure(n_reads));
    HIPCHECK(ctx->d_kept_item.ensure(M));
    HIPCHECK(ctx->d_node_rank.ensure(n_reads + 2)); HIPCHECK(ctx->d_t_death[0].ensure(n_reads));
    HIPCHECK(ctx->d_t_death[1].ensure(n_reads));
    HIPCHECK(ctx->d_touched.ensure(n_alive)); HIPCHECK(ctx->d_cmed.ensure(n_alive));
    HIPCHECK(ctx->d_cc_edges.ensure(2 * (size_t)M)); HIPCHECK(ctx->d_cc_label.ensure(n_alive));
    ctx->t_med_tmp = component_median_workspace(n_alive);
    HIPCHECK(ctx->d_med_tmp.ensure(ctx->t_med_tmp));
    HIPCHECK(ctx->d_cc_flags.ensure(8));
    (void)big;
    // (the containment scans order the items by key = items * (1 + round of promotion) + index, 32 bits: checked below with the
    // rounds this data set took - two to four - and not with the 255 a round counter could hold; before round 5 that was a
    // limit of 16.6 M surviving overlaps, 2.2 times C5's)
    if ((uint64_t)M >= 0x7FFFFFF0ull) return fail(ctx, RALA_HIP_ETOOLARGE, "too many surviving overlaps");
    const TailList L = tail_list(ctx);
    const TailReads R = tail_reads(ctx);
    ScanSpace sp;
    {
        FillList fills;
        const int rc = scan_space(ctx, 1, n_reads, sp, &fills);
        if (rc != RALA_HIP_OK) return rc;
        HIPCHECK(fills.launch(s));
    }
    // (d_counts: [4 .. 5] overlaps dropped per round of a batch, [8] reads left, [9] a containment scan that failed,
    // [10 .. 13] the scans' killer counts, [14 .. 29] their barriers - launch_tail_contain; [32 .. 39] the second pass's)
    HIPCHECK(ctx->d_counts.ensure(48));
    for (int k = 0; k < 2; ++k) HIPCHECK(ctx->d_t_work[k].ensure(n_reads));
    HIPCHECK(ctx->d_t_fin.ensure(2 * (size_t)n_reads));
    HIPCHECK(ctx->d_t_mark.ensure(2 * (size_t)n_reads));
    for (int k = 0; k < 3; ++k) { HIPCHECK(ctx->d_kill[k].ensure((size_t)M + 1)); HIPCHECK(ctx->d_kill2[k].ensure((size_t)M + 1)); }
    HIPCHECK(ctx->d_fp_map.ensure(n_reads));
    HIPCHECK(ctx->d_fp_pack.ensure(fixed_point_pack_words()));
    launch_tail_init(L, ctx->t_n0, R, ctx->d_n_pits0.p, n_reads, ctx->d_t_fin.p, ctx->d_t_mark.p, ctx->d_fp_map.p, ctx->d_counts.p + 9, s);
    // ranks of the reads that survived the second pass (the component graph lives on them)
    if (!launch_rank_pass(ctx->d_alive.p, ctx->d_rank.p, ctx->d_alive_reads.p, n_reads, sp, s)) return fail(ctx, RALA_HIP_EDEVICE, "scan space");
    mark("tail: ranks", n_alive);

    // break over chimeric hills, first re-trim (graph.cpp:704-736; no promotion here)
    HIPCHECK(ctx->d_counts.ensure(48));
    uint32_t* dropped = ctx->d_counts.p + 4;                        // one word per round of a batch
    launch_break_hills(R, n_reads, s);
    if (ctx->lists_pending) {           // (a sharded run's survivor lists, gathered beside the kernels above)
        HIPCHECK(hipStreamWaitEvent(s, ctx->ev[3], 0));
        ctx->lists_pending = false;
    }
    launch_retrim(L, R, 0, 0, dropped, s);
    mark("tail: hills + retrim", M);

    // graph.cpp:738-829: components, component medians, break over pits, re-trim - while an overlap died
    // in the round.  Hardly a data set is done after one round, and a look from the host costs a third of
    // one: two rounds are enqueued before the host looks, the second one gated on the device by the first
    // one's count (a round behind the reference's last would see the internals that round promoted as
    // overlaps - it must not act), then one at a time.
    uint32_t rounds = 0;
    for (;;) {
        const uint32_t batch = rounds == 0 ? 2u : 1u;
        for (uint32_t k = 0; k < batch; ++k) {
            if (rounds + k >= 255) return fail(ctx, RALA_HIP_EDEVICE, "chimera loop did not settle");
            const uint32_t* gate = k ? dropped + k - 1 : nullptr;
            // one fill: the dirt the last re-trim has dealt with, the components' marks, the batch's verdicts
            FillList fills;
            fills.add(ctx->d_dirty.p, 0, n_reads);
            fills.add(ctx->d_touched.p, 0, n_alive);
            component_median_clear(ctx->d_med_tmp.p, n_alive, fills);
            if (k == 0) fills.add(dropped, 0, 2 * 4);
            HIPCHECK(fills.launch(s));
            const int rcc = tail_components(ctx, L, n_alive, true);
            if (rcc != RALA_HIP_OK) return rcc;
            launch_break_pits(R, ctx->d_alive_reads.p, ctx->d_touched.p, ctx->d_cmed.p, n_alive, s, gate);
            launch_retrim(L, R, 1, rounds + k, dropped + k, s, gate);
        }
        uint32_t d[2] = {0, 0};
        HIPCHECK(d2h_small(ctx, d, dropped, sizeof(d), s));
        HIPCHECK(stream_sync(ctx, s));
        mark("tail: pits + retrim", d[0] + d[1]);
        if (!d[0]) { rounds += 1; break; }                     // (a second round of the batch did nothing)
        rounds += batch;
        if (!d[batch - 1]) break;
    }
    ctx->t_rounds = rounds;
    if ((uint64_t)M * ((uint64_t)rounds + 2ull) >= 0xFFFFFFF0ull) {
        return fail(ctx, RALA_HIP_ETOOLARGE, "too many surviving overlaps for 32-bit list positions");
    }

    // in-order containment removal (graph.cpp:831-877): overlaps (+ promoted), then internals
    {
        uint32_t* const work[4] = {ctx->d_t_death[0].p, ctx->d_t_death[1].p, ctx->d_t_work[0].p, ctx->d_t_work[1].p};
        uint32_t* const lists[6] = {ctx->d_kill[0].p, ctx->d_kill[1].p, ctx->d_kill[2].p, ctx->d_kill2[0].p, ctx->d_kill2[1].p,
                                    ctx->d_kill2[2].p};
        HIPCHECK(launch_tail_contain(L, R, ctx->d_alive.p, lists, ctx->d_counts.p + 9, work, ctx->d_t_fin.p, ctx->d_t_mark.p, ctx->d_fp_map.p,
                                     ctx->d_fp_pack.p, n_reads, ctx->debug_fp_lds_limit, s));
    }
    mark("tail: containment scans", M);
    return RALA_HIP_OK;
}

// ... the final overlap list, nodes and edges (graph.cpp:553-632, :869-877) from the device list
int gpu_tail_part_b(rala_hip_ctx* ctx) {
    hipStream_t s = ctx->stream;
    Trace trc;
    auto mark = [&](const char* what, size_t k = 0) {
        if (trc.on) { (void)stream_sync(ctx, s); trc(what, k); }
    };
    const uint32_t n_reads = (uint32_t)ctx->n_reads;
    const uint32_t M = ctx->t_n0 + ctx->t_n1;
    const TailList L = tail_list(ctx);
    const TailReads R = tail_reads(ctx);
    const uint32_t n_seg = ctx->t_rounds + 1;
    // sized by what the host knows: at most t_n_alive reads are left, at most M overlaps kept, all dovetails
    HIPCHECK(ctx->d_seg_base.ensure(2 * (size_t)n_seg + 4));
    HIPCHECK(ctx->d_node_read.ensure(2 * (size_t)ctx->t_n_alive + 2));
    for (int k = 0; k < 3; ++k) HIPCHECK(ctx->d_e[k].ensure(2 * (size_t)M + 2));
    HIPCHECK(ctx->d_counts.ensure(48));
    ScanSpace sp;
    {
        FillList fills;
        const int rc = scan_space(ctx, n_seg + 1, std::max<uint64_t>(n_reads, M), sp, &fills);
        if (rc != RALA_HIP_OK) return rc;
        fills.add(ctx->d_seg_base.p, 0, (2 * (size_t)n_seg + 4) * 4);
        fills.add(ctx->d_counts.p + 8, 0, 4);
        HIPCHECK(fills.launch(s));
    }
    // nodes: two per surviving read (graph.cpp:553-574)
    if (!launch_node_pass(ctx->d_alive.p, ctx->d_node_rank.p, ctx->d_node_read.p, ctx->d_counts.p + 8, n_reads, sp, s)) {
        return fail(ctx, RALA_HIP_EDEVICE, "scan space");
    }
    // the final overlap list: originals in order, then the promoted ones round by round; the two edges of
    // every dovetail (graph.cpp:576-632) with it
    for (uint32_t seg = 0; seg < n_seg; ++seg) {
        // (the last segment's sums - the totals - go next to the other counts the host fetches: d_counts[6 .. 7])
        if (!launch_segment_pass(L, R, seg == 0 ? 1u : 3u, seg == 0 ? 0u : seg - 1, ctx->d_seg_base.p + 2 * seg,
                                 seg + 1 == n_seg ? ctx->d_counts.p + 6 : ctx->d_seg_base.p + 2 * (seg + 1), ctx->d_kept_item.p,
                                 ctx->d_node_rank.p, ctx->d_e[0].p,
                                 ctx->d_e[1].p, ctx->d_e[2].p, sp, s)) {
            return fail(ctx, RALA_HIP_EDEVICE, "scan space");
        }
    }
    uint32_t four[4] = {0, 0, 0, 0};         // kept items, dovetails, reads left, a containment scan that failed
    HIPCHECK(d2h_small(ctx, four, ctx->d_counts.p + 6, 16, s));
    HIPCHECK(stream_sync(ctx, s));
    const uint32_t totals[2] = {four[0], four[1]}, left[2] = {four[2], four[3]};
    HIPCHECK(hipGetLastError());
    if (getenv("RALA_HIP_TRACE")) {
        uint32_t kc[4] = {0, 0, 0, 0};
        HIPCHECK(hipMemcpy(kc, ctx->d_counts.p + 10, 16, hipMemcpyDeviceToHost));
        fprintf(stderr, "[trace] tail containment killers: %u (%u conditional) among the overlaps, %u (%u) among the internals, of %u items\n",
                kc[0], kc[2], kc[1], kc[3], M);
    }
    if (left[1]) {
        return fail(ctx, RALA_HIP_EDEVICE, "containment fixed point did not converge");
    }
    const uint32_t n_final = left[0];
    ctx->t_n_kept = totals[0];
    ctx->t_n_nodes = 2 * n_final;
    ctx->t_n_edges = 2 * totals[1];
    mark("tail: final list, nodes, edges", ctx->t_n_edges);
    ctx->tail_on_device = true;
    ctx->host_stale = true;
    return RALA_HIP_OK;
}

int gpu_tail_run(rala_hip_ctx* ctx) {
    const int rc = gpu_tail_part_a(ctx);
    return rc != RALA_HIP_OK ? rc : gpu_tail_part_b(ctx);
}

// host mirrors of a device-resident result (lists in the reference's order, graph, read state)
int materialize_host(rala_hip_ctx* ctx) {
    if (ctx->have_repeats && ctx->rep_host_stale) {
        const int rcr = download_repeat_hills(ctx);
        if (rcr != RALA_HIP_OK) return rcr;
    }
    if (!ctx->host_stale) return ctx->host_state_fresh ? RALA_HIP_OK : download_read_state(ctx);
    hipStream_t s = ctx->stream;
    int rc = download_read_state(ctx);
    if (rc != RALA_HIP_OK) return rc;
    const uint32_t M = ctx->t_n0 + ctx->t_n1;
    std::vector<uint32_t> h[8];
    std::vector<uint8_t> strand(M), type(M), state(M);
    for (int f = 0; f < 8; ++f) {
        h[f].resize(M);
        if (M) HIPCHECK(hipMemcpyAsync(h[f].data(), ctx->d_surv_u32[f].p, (size_t)M * 4, hipMemcpyDeviceToHost, s));
    }
    std::vector<uint32_t> kept(ctx->t_n_kept);
    if (M) {
        HIPCHECK(hipMemcpyAsync(strand.data(), ctx->d_surv_u8[0].p, M, hipMemcpyDeviceToHost, s));
        HIPCHECK(hipMemcpyAsync(type.data(), ctx->d_surv_u8[1].p, M, hipMemcpyDeviceToHost, s));
        HIPCHECK(hipMemcpyAsync(state.data(), ctx->d_t_state.p, M, hipMemcpyDeviceToHost, s));
    }
    if (ctx->t_n_kept) {
        HIPCHECK(hipMemcpyAsync(kept.data(), ctx->d_kept_item.p, (size_t)ctx->t_n_kept * 4, hipMemcpyDeviceToHost, s));
    }
    ctx->node_read.resize(ctx->t_n_nodes);
    ctx->e_src.resize(ctx->t_n_edges); ctx->e_dst.resize(ctx->t_n_edges); ctx->e_len.resize(ctx->t_n_edges);
    if (ctx->t_n_nodes) {
        HIPCHECK(hipMemcpyAsync(ctx->node_read.data(), ctx->d_node_read.p, (size_t)ctx->t_n_nodes * 4, hipMemcpyDeviceToHost, s));
    }
    if (ctx->t_n_edges) {
        HIPCHECK(hipMemcpyAsync(ctx->e_src.data(), ctx->d_e[0].p, (size_t)ctx->t_n_edges * 4, hipMemcpyDeviceToHost, s));
        HIPCHECK(hipMemcpyAsync(ctx->e_dst.data(), ctx->d_e[1].p, (size_t)ctx->t_n_edges * 4, hipMemcpyDeviceToHost, s));
        HIPCHECK(hipMemcpyAsync(ctx->e_len.data(), ctx->d_e[2].p, (size_t)ctx->t_n_edges * 4, hipMemcpyDeviceToHost, s));
    }
    HIPCHECK(stream_sync(ctx, s));
    ctx->e_mark.assign(ctx->t_n_edges, 0);
    auto item = [&](uint32_t k) {
        HostOvl o;
        o.src = h[0][k]; o.a = h[1][k]; o.b = h[2][k];
        o.c.a_begin = h[3][k]; o.c.a_end = h[4][k]; o.c.b_begin = h[5][k]; o.c.b_end = h[6][k]; o.c.length = h[7][k];
        o.strand = strand[k]; o.dead = 0; o.type = type[k];
        return o;
    };
    ctx->overlaps.clear(); ctx->internals.clear();
    for (uint32_t k : kept) ctx->overlaps.push_back(item(k));
    for (uint32_t k = ctx->t_n0; k < M; ++k) if (state[k] == 2) ctx->internals.push_back(item(k));
    ctx->host_stale = false;
    return RALA_HIP_OK;
}


// ---- sharded survivor lists: a rank's survivors travel as one packed block -------------------
// block of m items: eight uint32 columns (src, a, b, a_begin, a_end, b_begin, b_end, length) of m
// entries each, then the strand and type bytes; overlaps first, internals behind them
size_t packed_list_bytes(uint64_t m) { return ((size_t)m * 34 + 15) & ~(size_t)15; }

Survivors packed_list_view(uint8_t* block, uint64_t m) {
    Survivors sv;
    uint32_t* c = (uint32_t*)block;
    sv.src = c; sv.a_id = c + m; sv.b_id = c + 2 * m; sv.a_begin = c + 3 * m; sv.a_end = c + 4 * m;
    sv.b_begin = c + 5 * m; sv.b_end = c + 6 * m; sv.length = c + 7 * m;
    sv.strand = block + 32 * m; sv.type = block + 33 * m;
    return sv;
}

// Second overlap pass on ctx->ovl (graph.cpp:443-518): static classification, the in-order
// containment removal as a fixed point, liveness + hill counters, survivors into the tail lists
// (ctx->d_surv_*, t_n0 overlaps then t_n1 internals).  With a communicator the overlaps are one
// slice of the file: the bounds of the fixed point are all-reduced (min) per round, the hill
// counters all-reduced (sum), the survivor lists all-gathered in slice (= file) order.
int pass2(rala_hip_ctx* ctx, Comm* comm) {
    hipStream_t s = ctx->stream;
    const uint32_t n_reads = (uint32_t)ctx->n_reads;
    const uint64_t N = ctx->n_ovl;
    const ReadState rs = read_state(ctx);
    auto comm_fail = [&](const char* what) {
        ctx->err = std::string(what) + ": " + comm->error();
        ctx->stage_pending.clear();
        ctx->stage_used = 0;
        return RALA_HIP_EDEVICE;
    };

    if (ctx->dedupe_pending) {          // (a sharded run's duplicate removal ran beside the emit, on the side stream)
        HIPCHECK(hipStreamWaitEvent(s, ctx->ev[1], 0));
        ctx->dedupe_pending = false;
    }
    // ---- static part ----
    // (a sharded run gathers up to 65 536 undecided killers of ALL ranks on every rank: room for them whatever the slice's size)
    const uint64_t kill_room = comm ? std::max<uint64_t>(N + 1, (1u << 16) + 1) : N + 1;
    for (int k = 0; k < 3; ++k) HIPCHECK(ctx->d_kill[k].ensure(kill_room));
    // the killer lists' counters: a fresh, pre-zeroed one per round of the fixed point
    constexpr uint32_t kCountRing = 128;
    HIPCHECK(ctx->d_kill_count.ensure(kCountRing + 4));
    KillList kl;
    kl.count = ctx->d_kill_count.p; kl.ovl = ctx->d_kill[0].p; kl.target = ctx->d_kill[1].p; kl.keeper = ctx->d_kill[2].p;
    HIPCHECK(hipEventRecord(ctx->ev[4], s));
    HIPCHECK(ctx->d_rec.ensure(n_reads));
    // compact per-read records (valid region + two flags): 4 bytes when no read is longer than 32767 bases
    const bool small_rec = ctx->max_read_len <= 32767u;
    HIPCHECK(ctx->d_crec.ensure((size_t)n_reads * compact_record_bytes(small_rec) + 16));
    HIPCHECK(ctx->d_counts.ensure(48));
    // one buffer: sure[n_reads] (min over sure killers = upper bound), lo[n_reads] (lower bound),
    // one status word - so that a sharded run needs ONE all-reduce (min) per round; up[] apart
    HIPCHECK(ctx->d_death_sure.ensure(2 * (size_t)n_reads + 8));
    HIPCHECK(ctx->d_fp_map.ensure(n_reads));
    HIPCHECK(ctx->d_fp_pack.ensure(fixed_point_pack_words()));
    uint32_t* sure = ctx->d_death_sure.p;
    uint32_t* lo = sure + n_reads;
    const size_t dbytes = (size_t)n_reads * 4;
    ScanSpace chunk_scan;
    {
        FillList fills;
        const int rc = scan_space(ctx, 1, pass2_chunks(N), chunk_scan, &fills);
        if (rc != RALA_HIP_OK) return rc;
        fills.add(ctx->d_kill_count.p, 0, (kCountRing + 4) * 4);
        fills.add(ctx->d_counts.p, 0, 48 * 4);
        fills.add(sure, 0xFF, 2 * dbytes + 4);
        fills.add(ctx->d_fp_map.p, 0xFF, dbytes);               // (launch_fixed_point_finish leaves it that way; here for good measure)
        HIPCHECK(fills.launch(s));
    }
    // (tests: a failure that only this rank sees, between two collectives of a sharded run)
    if (ctx->debug_fail_construct) return fail(ctx, RALA_HIP_EDEVICE, "debug_fail_construct");
    launch_pack_reads(rs, n_reads, ctx->d_rec.p, ctx->d_crec.p, small_rec, sure, s);
    launch_classify(ctx->ovl, n_reads, ctx->d_valid.p, ctx->d_crec.p, small_rec, kl, lo, s);
    HIPCHECK(hipEventRecord(ctx->ev[5], s));

    // ---- in-order containment removal as a fixed point (death_decide_kernel) ----
    for (int k = 0; k < 3; ++k) HIPCHECK(ctx->d_kill2[k].ensure(kill_room));
    KillList klist[2] = {kl, kl};
    uint32_t next_count = 1;
    klist[1].ovl = ctx->d_kill2[0].p; klist[1].target = ctx->d_kill2[1].p; klist[1].keeper = ctx->d_kill2[2].p;
    uint32_t* status = sure + 2 * (size_t)n_reads;          // 0xFFFFFFFF - undecided killers (min = most)
    uint32_t* up = ctx->d_death[1].p;
    // (the first lower bound - everybody's first killer - was taken by the classify kernel; the
    // upper bounds are all "never" in the first round, which is told so instead of reading them, and
    // written in full before the second)
    bool first_round = true;
    if (comm && comm->all_reduce_u32(lo, n_reads, ReduceOp::kMin, s) != 0) return comm_fail("all-reduce of the containment bounds");
    ctx->tm.death_rounds = 0;
    int cur = 0;
    bool gathered = comm == nullptr;                             // the killer lists are complete on this rank
    // Once few killers are left a round costs what the host's look at the counter costs.  A round
    // over an empty list changes nothing, so the host then enqueues several rounds at a time and
    // looks once; the list sizes of the rounds it did not look at are logged on the device (for
    // the round count) and read with the next results.
    constexpr uint32_t kFewKillers = 1u << 16, kRoundsPerLook = 5, kLogged = 64;
    constexpr uint32_t kNotSeen = 0xFFFFFFFFu;
    HIPCHECK(ctx->d_round_log.ensure(kLogged));
    std::vector<uint32_t> list_size;                             // after every round; kNotSeen = in the log
    // (one GPU: two rounds without a look from the host - C3: 1.09 M undecided after the first, 11 k after the
    // second - and what is left is finished on the device (fixed_point_kernels.hip: one workgroup with the bounds
    // in LDS, or a few resident workgroups for a long list): no more launches per round, no looks.  Before: six
    // more rounds of four launches, three looks.)
    const bool blind = comm == nullptr && ctx->use_round_batches;
    uint32_t unseen = blind ? 1u : 0u, n_logged = 0;
    constexpr uint32_t kFinishAtMost = 1u << 20;
    bool finished_on_device = false;
    uint64_t at_most = ~0ull;                                    // what the host knows of the current list's length
    for (;;) {
        if (next_count < kCountRing) {
            klist[cur ^ 1].count = ctx->d_kill_count.p + next_count++;
        } else {
            klist[cur ^ 1].count = ctx->d_kill_count.p + kCountRing + (cur ^ 1);
            HIPCHECK(hipMemsetAsync(klist[cur ^ 1].count, 0, 4, s));
        }
        launch_death_decide(klist[cur], lo, first_round ? nullptr : up, sure, klist[cur ^ 1], s, at_most);
        first_round = false;
        cur ^= 1;
        auto finish_on_device = [&]() -> int {
            // what is still undecided against sure[] as the deaths decided for good; up[], lo[] are free now
            HIPCHECK(ctx->d_death[0].ensure(n_reads));
            HIPCHECK(ctx->d_t_work[0].ensure(n_reads));
            uint32_t* const work[4] = {up, lo, ctx->d_death[0].p, ctx->d_t_work[0].p};
            const FixedPointList rest = {klist[cur].ovl, klist[cur].target, klist[cur].keeper, klist[cur].count, ctx->debug_fp_lds_limit};
            HIPCHECK(launch_fixed_point_finish(rest, sure, ctx->d_fp_map.p, ctx->d_fp_pack.p, work, ctx->d_counts.p + 32,
                                               ctx->d_counts.p + 1, ctx->d_counts.p + 2, s));
            finished_on_device = true;
            return (int)RALA_HIP_OK;
        };
        if (blind && list_size.size() == 1) {                   // the second round is done
            list_size.push_back(kNotSeen);
            const int rc = finish_on_device();
            if (rc != RALA_HIP_OK) return rc;
            break;
        }
        // tighter bounds for the next round: up = sure, lo = min(sure, undecided killers)
        uint32_t undecided = 0;
        if (gathered && unseen && n_logged < kLogged) {
            --unseen;
            list_size.push_back(kNotSeen);
            launch_death_tighten(sure, up, lo, n_reads, s, klist[cur].count, ctx->d_round_log.p + n_logged++);
            launch_death_lower(klist[cur], lo, s, at_most);
            continue;
        }
        if (gathered) {
            HIPCHECK(d2h_small(ctx, &undecided, klist[cur].count, 4, s));
            HIPCHECK(stream_sync(ctx, s));
        } else {
            HIPCHECK(hipMemcpyAsync(lo, sure, dbytes, hipMemcpyDeviceToDevice, s));
            launch_death_lower(klist[cur], lo, s);
            launch_death_status(klist[cur].count, status, s);
            if (comm->all_reduce_u32(sure, 2 * (size_t)n_reads + 1, ReduceOp::kMin, s) != 0) return comm_fail("all-reduce of the containment bounds");
            uint32_t st = 0;
            HIPCHECK(d2h_small(ctx, &st, status, 4, s));
            HIPCHECK(stream_sync(ctx, s));
            undecided = 0xFFFFFFFFu - st;                        // the largest list of any rank
        }
        list_size.push_back(undecided);
        if (getenv("RALA_HIP_TRACE")) fprintf(stderr, "[trace] containment round %d: %u undecided\n", (int)list_size.size(), undecided);
        if (undecided == 0) break;
        if (list_size.size() > 100000) return fail(ctx, RALA_HIP_EDEVICE, "containment fixed point did not converge");
        if (!gathered) HIPCHECK(hipMemcpyAsync(up, sure, dbytes, hipMemcpyDeviceToDevice, s));
        if (gathered && ctx->use_round_batches && undecided <= kFinishAtMost) {
            // (sharded run: the ranks hold the same gathered list and the same bounds)
            const int rc = finish_on_device();
            if (rc != RALA_HIP_OK) return rc;
            break;
        }
        if (gathered) {
            at_most = undecided;                                // the lists only shrink from here
            launch_death_tighten(sure, up, lo, n_reads, s);
            launch_death_lower(klist[cur], lo, s, at_most);
            if (undecided <= kFewKillers && ctx->use_round_batches) unseen = kRoundsPerLook;
        } else if ((uint64_t)undecided * comm->world() <= (1u << 16)) {
            // few killers left: every rank takes all of them and the rest needs no collective.  `undecided` is the longest
            // list of any rank, known everywhere: every rank pads its list to that length with inert entries and blocks of one
            // size are gathered - no exchange of the lists' lengths, no look from the host (rounds 2 - 4: a count to the host,
            // a host exchange, three gathers of different lengths, a count back to the device and a wait for it)
            const uint32_t each = undecided;
            const uint64_t total = (uint64_t)each * comm->world();
            if (total > kill_room) return fail(ctx, RALA_HIP_EDEVICE, "more undecided killers than there is room for");   // (cannot happen: <= 65536)
            HIPCHECK(ctx->d_gather[0].ensure((size_t)each * 12 + 16));
            HIPCHECK(ctx->d_gather[1].ensure((size_t)total * 12 + 16));
            launch_pack_killers(klist[cur].ovl, klist[cur].target, klist[cur].keeper, klist[cur].count, each, (uint32_t*)ctx->d_gather[0].p, s);
            if (comm->all_gather(ctx->d_gather[0].p, ctx->d_gather[1].p, (size_t)each * 12, s) != 0) return comm_fail("all-gather of the undecided killers");
            launch_unpack_killers((const uint32_t*)ctx->d_gather[1].p, comm->world(), each, klist[cur ^ 1].ovl, klist[cur ^ 1].target,
                                  klist[cur ^ 1].keeper, klist[cur ^ 1].count, s);
            cur ^= 1;
            gathered = true;
            at_most = total;
        }
    }
    uint32_t* death = sure;
    HIPCHECK(hipEventRecord(ctx->ev[6], s));

    // ---- liveness, hill counters, survivors ----
    const uint32_t n_chunks = pass2_chunks(N);
    const size_t mask_words = (size_t)((N + 63) / 64);
    HIPCHECK(ctx->d_cls.ensure(2 * 8 * mask_words + 16));
    uint64_t* mask_ov = (uint64_t*)ctx->d_cls.p;
    uint64_t* mask_in = mask_ov + mask_words;
    HIPCHECK(ctx->d_fate.ensure(n_reads));
    launch_apply_death(death, ctx->d_alive.p, ctx->d_n_hills.p, ctx->d_fate.p, n_reads, ctx->d_counts.p + 0, s);
    launch_survivor_masks(ctx->ovl, n_reads, ctx->d_valid.p, ctx->d_fate.p, death, ctx->d_crec.p, small_rec, ctx->d_rec.p,
                          ctx->d_pool.p, mask_ov, mask_in, ctx->d_chunk[0].p, ctx->d_chunk[1].p, s);
    if (comm && ctx->pool_used) {
        // Pile::check_chimeric_hills counters (pile.cpp:457-469): every slice counted its own overlaps
        HIPCHECK(ctx->d_t_tmp[0].ensure(std::max<size_t>(ctx->pool_used, n_reads) + 2));
        HIPCHECK(hipMemsetAsync(ctx->d_t_tmp[0].p, 0, (size_t)ctx->pool_used * 4, s));
        launch_hill_counts(rs, n_reads, ctx->d_t_tmp[0].p, 0, s);
        if (comm->all_reduce_u32(ctx->d_t_tmp[0].p, ctx->pool_used, ReduceOp::kSum, s) != 0) return comm_fail("all-reduce of the hill counters");
        launch_hill_counts(rs, n_reads, ctx->d_t_tmp[0].p, 1, s);
    }
    // the chunks' places in the two survivor lists: one scan; its sums come back with the other counts
    if (!launch_pair_offsets_pass(ctx->d_chunk[0].p, ctx->d_chunk[1].p, ctx->d_chunk[2].p, ctx->d_chunk[3].p, ctx->d_counts.p + 6,
                                  n_chunks, chunk_scan, s)) {
        return fail(ctx, RALA_HIP_EDEVICE, "scan space");
    }
    uint32_t round_log[64];
    if (n_logged && !finished_on_device) HIPCHECK(d2h_small(ctx, round_log, ctx->d_round_log.p, n_logged * 4, s));
    uint32_t counts8[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // [0] reads alive, [1] the finishing kernel's verdict, [2] its rounds, [6 .. 7] survivors
    HIPCHECK(d2h_small(ctx, counts8, ctx->d_counts.p + 0, 32, s));
    HIPCHECK(stream_sync(ctx, s));
    const uint32_t n_surv[2] = {counts8[6], counts8[7]};
    const uint32_t* counts3 = counts8;
    if (counts3[1]) {
        return fail(ctx, RALA_HIP_EDEVICE, "containment fixed point did not converge");
    }
    ctx->t_n_alive = counts3[0];            // the reads that survived the second pass (the tail's rank space)
    if (finished_on_device) {
        ctx->tm.death_rounds = (uint32_t)list_size.size() + counts3[2];
    } else {
        // rounds until the list was empty
        uint32_t at = 0;
        ctx->tm.death_rounds = (uint32_t)list_size.size();
        for (size_t r = 0; r < list_size.size(); ++r) {
            const uint32_t sz = list_size[r] == kNotSeen ? 0xFFFFFFFFu - round_log[at++] : list_size[r];
            if (sz == 0) { ctx->tm.death_rounds = (uint32_t)r + 1; break; }
        }
    }
    auto tail_view = [&]() {
        Survivors sv;
        sv.src = ctx->d_surv_u32[0].p; sv.a_id = ctx->d_surv_u32[1].p; sv.b_id = ctx->d_surv_u32[2].p;
        sv.a_begin = ctx->d_surv_u32[3].p; sv.a_end = ctx->d_surv_u32[4].p;
        sv.b_begin = ctx->d_surv_u32[5].p; sv.b_end = ctx->d_surv_u32[6].p;
        sv.length = ctx->d_surv_u32[7].p;
        sv.strand = ctx->d_surv_u8[0].p; sv.type = ctx->d_surv_u8[1].p;
        return sv;
    };
    if (!comm) {
        // both survivor lists side by side in one device list: overlaps, then internals
        const uint32_t M = n_surv[0] + n_surv[1];
        ctx->t_n0 = n_surv[0]; ctx->t_n1 = n_surv[1];
        for (int f = 0; f < 8; ++f) HIPCHECK(ctx->d_surv_u32[f].ensure(M));
        for (int f = 0; f < 2; ++f) HIPCHECK(ctx->d_surv_u8[f].ensure(M));
        // trim in the gather re-derives the coordinates against the pass-1 piles
        if (M) launch_gather_survivors(ctx->ovl, mask_ov, mask_in, ctx->d_crec.p, small_rec, ctx->d_chunk[2].p, ctx->d_chunk[3].p, n_surv[0], tail_view(), s);
    } else {
        // this slice's survivors as one packed block, all blocks gathered, unpacked in rank order
        const uint32_t P = comm->world();
        const uint64_t m = (uint64_t)n_surv[0] + n_surv[1];
        HIPCHECK(ctx->d_list_block[0].ensure(packed_list_bytes(m) + 16));
        if (m) launch_gather_survivors(ctx->ovl, mask_ov, mask_in, ctx->d_crec.p, small_rec, ctx->d_chunk[2].p, ctx->d_chunk[3].p, n_surv[0],
                                       packed_list_view(ctx->d_list_block[0].p, m), s);
        const uint64_t mine[2] = {n_surv[0], n_surv[1]};
        std::vector<uint64_t> all(2 * (size_t)P), bytes(P);
        if (comm->host_all_gather(mine, 2, all.data(), s) != 0) return comm_fail("survivor counts");
        ListBlocks lb;
        uint64_t tot0 = 0, tot1 = 0, off = 0;
        for (uint32_t p = 0; p < P; ++p) { tot0 += all[2 * p]; tot1 += all[2 * p + 1]; }
        if (tot0 + tot1 >= 0x7FFFFFF0ull) return fail(ctx, RALA_HIP_ETOOLARGE, "too many surviving overlaps");
        uint64_t at0 = 0, at1 = tot0;
        for (uint32_t p = 0; p < P; ++p) {
            const uint64_t mp = all[2 * p] + all[2 * p + 1];
            bytes[p] = packed_list_bytes(mp);
            lb.block_off[p] = off; lb.n0[p] = (uint32_t)all[2 * p]; lb.n1[p] = (uint32_t)all[2 * p + 1];
            lb.dst0[p] = (uint32_t)at0; lb.dst1[p] = (uint32_t)at1;
            off += bytes[p]; at0 += all[2 * p]; at1 += all[2 * p + 1];
        }
        lb.world = P;
        HIPCHECK(ctx->d_list_block[1].ensure(off + 16));
        const uint32_t M = (uint32_t)(tot0 + tot1);
        ctx->t_n0 = (uint32_t)tot0; ctx->t_n1 = (uint32_t)tot1;
        for (int f = 0; f < 8; ++f) HIPCHECK(ctx->d_surv_u32[f].ensure(M));
        for (int f = 0; f < 2; ++f) HIPCHECK(ctx->d_surv_u8[f].ensure(M));
        // The blocks travel on the side stream, beside the first kernels of the tail (list states, ranks of the reads that
        // are left, the breaks over hills: nothing of that looks at an item); gpu_tail_part_a joins in front of its first
        // kernel over the items.
        hipStream_t gs = ctx->use_side_stream ? ctx->side : s;
        if (gs != s) {
            HIPCHECK(hipEventRecord(ctx->ev[2], s));
            HIPCHECK(hipStreamWaitEvent(gs, ctx->ev[2], 0));
        }
        if (comm->all_gather_v(ctx->d_list_block[0].p, ctx->d_list_block[1].p, bytes.data(), 1, gs) != 0) return comm_fail("all-gather of the survivors");
        if (M) launch_unpack_lists(ctx->d_list_block[1].p, lb, tail_view(), gs);
        if (gs != s) {
            HIPCHECK(hipEventRecord(ctx->ev[3], gs));
            ctx->lists_pending = true;
        }
    }
    HIPCHECK(hipEventRecord(ctx->ev[7], s));
    HIPCHECK(hipGetLastError());
    return RALA_HIP_OK;
}

}  // namespace

// every row of cl where it lies, under the regions that apply (the getters' view: rala_hip_get_pile_data zeroes outside them)
int rala_hip::pile_row_digests(rala_hip_ctx* ctx, const uint32_t* begin, const uint32_t* end, const uint8_t* alive, uint64_t* fnv,
                               uint64_t* inside, uint64_t* outside) {
    const uint64_t n = ctx->n_reads;
    if (n == 0) return RALA_HIP_OK;
    HIPCHECK(hipSetDevice(ctx->device));
    DevBuf<uint32_t> d_be;
    DevBuf<uint8_t> d_al;
    DevBuf<uint64_t> d_out;
    HIPCHECK(d_be.ensure(2 * n)); HIPCHECK(d_al.ensure(n)); HIPCHECK(d_out.ensure(3 * n));
    HIPCHECK(hipMemcpy(d_be.p, begin, n * 4, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(d_be.p + n, end, n * 4, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(d_al.p, alive, n, hipMemcpyHostToDevice));
    hipStream_t s = ctx->stream;
    launch_pile_row_digests(ctx->d_pile.p, ctx->d_pile_off.p, ctx->d_read_len.p, d_be.p, d_be.p + n, d_al.p, (uint32_t)n,
                            fnv ? d_out.p : nullptr, inside ? d_out.p + n : nullptr, outside ? d_out.p + 2 * n : nullptr, s);
    HIPCHECK(hipGetLastError());
    HIPCHECK(stream_sync(ctx, s));
    if (fnv) HIPCHECK(hipMemcpy(fnv, d_out.p, n * 8, hipMemcpyDeviceToHost));
    if (inside) HIPCHECK(hipMemcpy(inside, d_out.p + n, n * 8, hipMemcpyDeviceToHost));
    if (outside) HIPCHECK(hipMemcpy(outside, d_out.p + 2 * n, n * 8, hipMemcpyDeviceToHost));
    return RALA_HIP_OK;
}

// ---- stage entry points shared with the sharded runner (stages.h) ---------------------------------
int rala_hip::construct_stages(rala_hip_ctx* ctx, Comm* comm, bool sensitive_pass_follows) {
    if (!ctx) return RALA_HIP_EINVAL;
    if (!ctx->initialized) return fail(ctx, RALA_HIP_EINVAL, "rala_hip_initialize must succeed first");
    if (ctx->tuple_mode || !ctx->inputs_set) return fail(ctx, RALA_HIP_EINVAL, "construct needs the overlaps (rala_hip_set_overlaps)");
    if (ctx->constructed) return fail(ctx, RALA_HIP_EINVAL, "object already constructed");
    ctx->have_repeats = false;
    HIPCHECK(hipSetDevice(ctx->device));
    int rc = pass2(ctx, comm);
    if (rc != RALA_HIP_OK) return rc;
    ctx->tail_on_device = false;
    ctx->host_stale = false;
    const double t0 = now_ms();
    rc = sensitive_pass_follows ? gpu_tail_part_a(ctx) : gpu_tail_run(ctx);      // (repeats_stage builds the graph)
    if (rc != RALA_HIP_OK) return rc;
    HIPCHECK(hipEventElapsedTime(&ctx->tm.classify_ms, ctx->ev[4], ctx->ev[5]));
    HIPCHECK(hipEventElapsedTime(&ctx->tm.death_ms, ctx->ev[5], ctx->ev[6]));
    HIPCHECK(hipEventElapsedTime(&ctx->tm.finish_ms, ctx->ev[6], ctx->ev[7]));
    ctx->tm.tail_host_ms = (float)(now_ms() - t0);
    ctx->constructed = true;
    return RALA_HIP_OK;
}

int rala_hip::repeats_stage(rala_hip_ctx* cs, rala_hip_ctx* cl, Comm* comm, const rala_hip_overlaps* sens, uint64_t n_sens) {
    if (!cs || !cl) return RALA_HIP_EINVAL;
    if (!cs->constructed) return fail(cs, RALA_HIP_EINVAL, "construct_stages must have run");
    if (!cl->piles_resident) return fail(cs, RALA_HIP_EINVAL, "the sensitive pass needs the piles on the owner context");
    const double t0 = now_ms();
    const int rc = repeats_after_tail(cs, cl, comm, sens, n_sens);
    cs->tm.repeats_ms = (float)(now_ms() - t0);
    cs->tm.tail_host_ms += cs->tm.repeats_ms;
    return rc;
}

int rala_hip::transitive_stage(rala_hip_ctx* ctx, Comm* comm, uint32_t* n_pairs) {
    if (!ctx || !n_pairs) return RALA_HIP_EINVAL;
    if (!ctx->constructed) return fail(ctx, RALA_HIP_EINVAL, "construct must succeed first");
    HIPCHECK(hipSetDevice(ctx->device));
    if (ctx->tail_on_device) {
        const int rc = tr_mark_device(ctx, ctx->t_n_nodes, ctx->t_n_edges, ctx->d_e[0].p, ctx->d_e[1].p, ctx->d_e[2].p, n_pairs, comm);
        ctx->marks_on_device = rc == RALA_HIP_OK;
        return rc;
    }
    return rala_hip_remove_transitive_edges(ctx, n_pairs);      // (after the sensitive pass the graph is a host graph)
}

// A sender of a sharded run: duplicate removal on the side stream (joined by the second pass), the bounds of the slice
// scattered once by (owner, partition) into `send` (shard_send_words() words); words[p] = 8-byte words of owner p's block.
// RALA_HIP_MEM_HOST_ASYNC columns that rala_hip_initialize has not uploaded yet: now, for a caller that reads them first
int rala_hip::flush_upload(rala_hip_ctx* ctx) {
    if (!ctx->upload_pending) return RALA_HIP_OK;
    for (int k = 0; k < 7; ++k) HIPCHECK(hipMemcpy(ctx->d_ovl_u32[k].p, ctx->up_src[k], (size_t)ctx->n_ovl * 4, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(ctx->d_ovl_strand.p, ctx->up_strand, ctx->n_ovl, hipMemcpyHostToDevice));
    ctx->upload_pending = false;
    return RALA_HIP_OK;
}

int rala_hip::shard_emit(rala_hip_ctx* ctx, const ShardGeometry& g, uint64_t* send, uint64_t* words) {
    if (!ctx || !send || !words) return RALA_HIP_EINVAL;
    if (ctx->n_reads == 0) return fail(ctx, RALA_HIP_EINVAL, "no reads set");
    if (!ctx->inputs_set || ctx->tuple_mode) return fail(ctx, RALA_HIP_EINVAL, "no overlaps set");
    HIPCHECK(hipSetDevice(ctx->device));
    { const int rcu = flush_upload(ctx); if (rcu != RALA_HIP_OK) return rcu; }
    hipStream_t s = ctx->stream;
    const uint32_t n_reads = (uint32_t)ctx->n_reads;
    // duplicate removal: its first pass inside the count where that kernel can take it (the id columns on 16-byte boundaries),
    // the marked runs' fix on the side stream beside the scatter - as on one GPU (rala_hip_initialize); otherwise the two
    // kernels of its own beside the whole emit
    const bool counted = ctx->use_side_stream && bucket_count_can_dedupe(ctx->ovl, ctx->d_valid.p);
    constexpr uint32_t kMarks = 1u << 20;
    BucketDedupe bd = {};
    if (counted) {
        HIPCHECK(ctx->d_dedupe_list.ensure(2 * (size_t)kMarks + 4));
        HIPCHECK(hipMemsetAsync(ctx->d_dedupe_list.p + 2 * (size_t)kMarks, 0, 4, s));
        bd.suspect = ctx->d_suspect.p; bd.valid = ctx->d_valid.p; bd.list_pos = ctx->d_dedupe_list.p; bd.list_query = ctx->d_dedupe_list.p + kMarks;
        bd.list_cap = ctx->debug_dedupe_list_cap ? std::min(kMarks, ctx->debug_dedupe_list_cap) : kMarks; bd.list_count = ctx->d_dedupe_list.p + 2 * (size_t)kMarks; bd.counted = ctx->ev[0];
    } else if (ctx->use_side_stream) {
        HIPCHECK(hipEventRecord(ctx->ev[0], s));
        HIPCHECK(hipStreamWaitEvent(ctx->side, ctx->ev[0], 0));
        launch_dedupe(ctx->ovl, n_reads, ctx->d_suspect.p, ctx->d_valid.p, ctx->side);
        HIPCHECK(hipEventRecord(ctx->ev[1], ctx->side));
        ctx->dedupe_pending = true;
    } else {
        launch_dedupe(ctx->ovl, n_reads, ctx->d_suspect.p, ctx->d_valid.p, s);
    }
    ctx->valid_ready = true;
    HIPCHECK(ctx->d_shard_group.ensure((size_t)g.world * g.groups + 2));
    HIPCHECK(ctx->d_shard_part.ensure((size_t)g.world * g.n_part + 2));
    HIPCHECK(ctx->d_shard_words.ensure(64));
    FillList fills;
    HIPCHECK(launch_shard_emit(ctx->ovl, n_reads, g, ctx->d_shard_group.p, ctx->d_shard_part.p, send, ctx->d_shard_words.p,
                               ctx->n_compute_units, fills, s, counted ? &bd : nullptr));
    if (counted && ctx->n_ovl) {
        HIPCHECK(hipStreamWaitEvent(ctx->side, ctx->ev[0], 0));
        launch_dedupe_fix(ctx->ovl, n_reads, bd, ctx->side);
        HIPCHECK(hipEventRecord(ctx->ev[1], ctx->side));
        ctx->dedupe_pending = true;
    }
    uint32_t h[64];
    HIPCHECK(d2h_small(ctx, h, ctx->d_shard_words.p, g.world * 4, s));
    HIPCHECK(stream_sync(ctx, s));
    HIPCHECK(hipGetLastError());
    for (uint32_t p = 0; p < g.world; ++p) words[p] = h[p];
    return RALA_HIP_OK;
}

// An owner of a sharded run: its input is the blocks of all senders where they lie (base: what was received; base_self:
// the send buffer, for the rank's own block); n_records of them in all.  rala_hip_initialize takes it from there.
int rala_hip::set_bound_blocks(rala_hip_ctx* ctx, const uint64_t* base, const uint64_t* base_self, const ShardBlocks& blocks,
                               const ShardGeometry& g, uint64_t n_records) {
    if (!ctx) return RALA_HIP_EINVAL;
    if (ctx->n_reads == 0) return fail(ctx, RALA_HIP_EINVAL, "no reads set");
    if (n_records >= 0x7FFFFFF0ull) return fail(ctx, RALA_HIP_EINVAL, "too many records");
    HIPCHECK(hipSetDevice(ctx->device));
    ctx->blocks_base = base; ctx->blocks_base_self = base_self; ctx->blocks = blocks; ctx->shard_geom = g;
    ctx->n_block_records = n_records;
    ctx->blocks_mode = true;
    ctx->records = nullptr; ctx->n_records = 0;
    ctx->tuples = nullptr; ctx->n_tuples = 2 * n_records;
    ctx->tuple_mode = true;
    ctx->inputs_set = true;
    ctx->n_ovl = 0;
    ctx->ovl = OvlSoA();
    HIPCHECK(ctx->d_ev.ensure(2 * n_records + 8));
    HIPCHECK(ctx->d_scan_ws.ensure(scan_workspace_bytes(std::max<uint64_t>(2 * n_records, ctx->n_reads) + 2)));
    ctx->initialized = ctx->constructed = ctx->ev_ready = false;
    return RALA_HIP_OK;
}

int rala_hip::install_read_state(rala_hip_ctx* ctx, uint64_t pool_count) {
    if (!ctx) return RALA_HIP_EINVAL;
    if (ctx->n_reads == 0) return fail(ctx, RALA_HIP_EINVAL, "no reads set");
    if (ctx->n_ovl && !ctx->valid_ready) return fail(ctx, RALA_HIP_EINVAL, "validity bits required (rala_hip_dedupe)");
    if (pool_count > ctx->pool_cap) return fail(ctx, RALA_HIP_ECAPACITY, "interval pool smaller than the installed state");
    HIPCHECK(hipSetDevice(ctx->device));
    { const int rcu = flush_upload(ctx); if (rcu != RALA_HIP_OK) return rcu; }
    hipStream_t s = ctx->stream;
    ctx->tm = rala_hip_timings();
    ctx->overlaps.clear(); ctx->internals.clear();
    const uint32_t small[8] = {(uint32_t)pool_count, 0, 0, 0, 0, 0, 0, 0};
    HIPCHECK(hipMemcpyAsync(ctx->d_small.p, small, sizeof(small), hipMemcpyHostToDevice, s));
    HIPCHECK(ctx->d_cc_flags.ensure(8));
    HIPCHECK(hipMemsetAsync(ctx->d_cc_flags.p + 6, 0, 4, s));
    launch_count_zero_u8(ctx->d_alive.p, (uint32_t)ctx->n_reads, ctx->d_cc_flags.p + 6, s);
    uint32_t n_dead = 0;
    HIPCHECK(d2h_small(ctx, &n_dead, ctx->d_cc_flags.p + 6, 4, s));
    HIPCHECK(stream_sync(ctx, s));                  // `small` is a local
    ctx->host_state_fresh = false;
    ctx->pool_used = (uint32_t)pool_count;
    ctx->n_prefiltered = n_dead;
    ctx->initialized = true;
    ctx->constructed = false;
    ctx->piles_resident = ctx->ev_ready = false;
    ctx->tail_on_device = ctx->host_stale = ctx->marks_on_device = false;
    if (ctx->n_prefiltered == ctx->n_reads) return fail(ctx, RALA_HIP_EFILTERED, "filtered all sequences");
    return RALA_HIP_OK;
}

extern "C" {

int rala_hip_create(int device, rala_hip_ctx** out) {
    if (!out) return RALA_HIP_EINVAL;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return RALA_HIP_EDEVICE;
    if (hipSetDevice(device) != hipSuccess) return RALA_HIP_EDEVICE;
    rala_hip_ctx* ctx = new rala_hip_ctx;
    ctx->device = device;
    if (const char* mb = getenv("RALA_HIP_PILE_CHUNK_MB")) ctx->pile_chunk_mb = (uint32_t)std::max(0, atoi(mb));      // (measurements: the option's default)
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) {
            ctx->n_compute_units = (uint32_t)prop.multiProcessorCount;
        }
    }
    // (the side stream at the highest priority the device offers: what runs there beside a kernel that fills the chip - the
    // position-space kernels of the sensitive pass, whose workgroups want most of a compute unit's LDS - gets the compute
    // units as they come free instead of when the other kernel has drained)
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    bool ok = hipStreamCreate(&ctx->stream) == hipSuccess &&
              hipStreamCreateWithPriority(&ctx->side, hipStreamDefault, prio_greatest) == hipSuccess &&
              hipStreamCreate(&ctx->aux) == hipSuccess;
    for (auto& e : ctx->ev) ok = ok && hipEventCreate(&e) == hipSuccess;
    ok = ok && ctx->d_small.ensure(16) == hipSuccess;
    if (!ok) {                                  // (rala_hip_destroy releases whatever was created)
        rala_hip_destroy(ctx);
        return RALA_HIP_EDEVICE;
    }
    ctx->pool.reset(new HostPool(std::min(16u, std::max(1u, std::thread::hardware_concurrency()))));
    *out = ctx;
    return RALA_HIP_OK;
}

void rala_hip_destroy(rala_hip_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    ctx->stage_pending.clear();             // nobody is waiting for staged copies any more
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->side) (void)hipStreamSynchronize(ctx->side);
    if (ctx->aux) (void)hipStreamSynchronize(ctx->aux);
    for (auto& e : ctx->ev) if (e) (void)hipEventDestroy(e);
    for (auto& e : ctx->ev_up) if (e) (void)hipEventDestroy(e);
    if (ctx->copy) (void)hipStreamDestroy(ctx->copy);
    if (ctx->side) (void)hipStreamDestroy(ctx->side);
    if (ctx->aux) (void)hipStreamDestroy(ctx->aux);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char* rala_hip_last_error(const rala_hip_ctx* ctx) { return ctx ? ctx->err.c_str() : "no context"; }

int rala_hip_set_option(rala_hip_ctx* ctx, const char* key, int64_t value) {
    if (!ctx || !key) return RALA_HIP_EINVAL;
    if (!strcmp(key, "interval_pool_per_read_x1000")) { ctx->pool_per_read_x1000 = value; return RALA_HIP_OK; }
    if (!strcmp(key, "debug_big_caps")) { ctx->debug_big_caps = value; return RALA_HIP_OK; }
    if (!strcmp(key, "debug_force_big")) { ctx->debug_force_big = value != 0; return RALA_HIP_OK; }
    if (!strcmp(key, "max_lds_read_len")) { ctx->max_lds_read_len = value; return RALA_HIP_OK; }
    if (!strcmp(key, "debug_pile_stop_after")) { ctx->debug_pile_stop_after = value; return RALA_HIP_OK; }
    if (!strcmp(key, "use_run_kernel")) { ctx->use_run_kernel = value != 0; return RALA_HIP_OK; }
    if (!strcmp(key, "use_round_batches")) { ctx->use_round_batches = value != 0; return RALA_HIP_OK; }
    if (!strcmp(key, "use_bound_records")) { ctx->use_bound_records = value != 0; return RALA_HIP_OK; }
    if (!strcmp(key, "use_fused_emit")) { ctx->use_fused_emit = value != 0; return RALA_HIP_OK; }
    if (!strcmp(key, "ingest_window_bytes")) { ctx->ingest_window_bytes = std::max<int64_t>(0, value); return RALA_HIP_OK; }
    if (!strcmp(key, "pile_chunk_mb")) { ctx->pile_chunk_mb = (uint32_t)std::max<int64_t>(0, std::min<int64_t>(value, 1 << 20)); return RALA_HIP_OK; }
    if (!strcmp(key, "debug_ev_events")) { ctx->debug_ev_events = value != 0; return RALA_HIP_OK; }
    if (!strcmp(key, "debug_count_window")) { g_count_window = (uint32_t)std::max<int64_t>(0, value); return RALA_HIP_OK; }
    if (!strcmp(key, "debug_part_shift")) { g_part_shift = (uint32_t)std::max<int64_t>(0, value); return RALA_HIP_OK; }
    if (!strcmp(key, "debug_pile_variant")) { ctx->debug_pile_variant = (uint32_t)std::max<int64_t>(0, value); return RALA_HIP_OK; }
    if (!strcmp(key, "debug_dedupe_list_cap")) { ctx->debug_dedupe_list_cap = (uint32_t)std::max<int64_t>(0, value); return RALA_HIP_OK; }
    if (!strcmp(key, "debug_fp_lds_limit")) { ctx->debug_fp_lds_limit = (uint32_t)std::max<int64_t>(0, value); return RALA_HIP_OK; }
    if (!strcmp(key, "debug_fail_construct")) { ctx->debug_fail_construct = value != 0; return RALA_HIP_OK; }
    if (!strcmp(key, "use_gpu_tail")) { ctx->use_gpu_tail = value != 0; return RALA_HIP_OK; }
    if (!strcmp(key, "use_fixed_buckets")) { ctx->use_fixed_buckets = value != 0; return RALA_HIP_OK; }
    if (!strcmp(key, "use_partitioned_buckets")) { ctx->use_partitioned_buckets = value != 0; return RALA_HIP_OK; }
    if (!strcmp(key, "use_side_stream")) { ctx->use_side_stream = value != 0; return RALA_HIP_OK; }
    if (!strcmp(key, "sensitive_in_device_memory")) { ctx->sens_in_device = value != 0; return RALA_HIP_OK; }
    if (!strcmp(key, "host_threads")) {
        ctx->host_threads = value;
        ctx->pool.reset(new HostPool((unsigned)std::max<int64_t>(1, std::min<int64_t>(value, 256))));
        return RALA_HIP_OK;
    }
    return fail(ctx, RALA_HIP_EINVAL, "unknown option");
}

void* rala_hip_stream(rala_hip_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int rala_hip_set_reads(rala_hip_ctx* ctx, const uint32_t* read_len, uint64_t n_reads) {
    if (!ctx || (!read_len && n_reads)) return RALA_HIP_EINVAL;
    if (n_reads >= 0x7FFFFFFFull) return fail(ctx, RALA_HIP_EINVAL, "too many reads");
    HIPCHECK(hipSetDevice(ctx->device));
    HIPCHECK(hipStreamSynchronize(ctx->side));
    HIPCHECK(hipStreamSynchronize(ctx->aux));
    ctx->n_reads = n_reads;
    ctx->h_read_len.assign(read_len, read_len + n_reads);
    ctx->max_read_len = 0;
    for (uint64_t r = 0; r < n_reads; ++r) ctx->max_read_len = std::max(ctx->max_read_len, read_len[r]);
    {
        // length classes of the pile chain (kernels.h: kPileClassBases)
        uint32_t cnt[kPileClasses] = {};
        auto cls = [](uint32_t n) {
            uint32_t c = 0;
            while (c + 1 < kPileClasses && n > kPileClassBases[c]) ++c;
            return c;
        };
        for (uint64_t r = 0; r < n_reads; ++r) ++cnt[cls(read_len[r])];
        for (uint32_t c = 0; c < kPileClasses; ++c) ctx->n_class[c] = cnt[c];
        if (cnt[0] != n_reads) {
            std::vector<uint32_t> order(n_reads);
            uint32_t at[kPileClasses] = {};
            for (uint32_t c = 1; c < kPileClasses; ++c) at[c] = at[c - 1] + cnt[c - 1];
            for (uint64_t r = 0; r < n_reads; ++r) order[at[cls(read_len[r])]++] = (uint32_t)r;
            HIPCHECK(ctx->d_class_order.ensure(n_reads));
            HIPCHECK(hipMemcpy(ctx->d_class_order.p, order.data(), n_reads * 4, hipMemcpyHostToDevice));
        }
    }
    ctx->h_pile_off.resize(n_reads + 1);
    // Rows start on 128-byte boundaries (64 elements): a row's first and last cache line are then its own, not shared
    // with the neighbouring rows, which other wavefronts write at other times.  Round 4, one box, pile kernel at C3:
    // 16-byte boundaries (rounds 1 - 3: what the 16-byte stores need) 4.26 ms, 128 bytes 4.17, 1 KB 4.18, 4 KB 4.21
    // (docs/history/gpurun/r4_rowalign.sh; RALA_PILE_ROW_ALIGN=<elements, a power of two >= 8> for the experiment).
    static const uint64_t row_align = [] {
        uint64_t want = getenv("RALA_PILE_ROW_ALIGN") ? (uint64_t)atoll(getenv("RALA_PILE_ROW_ALIGN")) : 64ull, a = 8;
        while (a * 2 <= want && a < (1ull << 20)) a *= 2;           // a power of two, 8 elements at least
        return a;
    }();
    uint64_t off = 0;
    for (uint64_t r = 0; r < n_reads; ++r) {
        ctx->h_pile_off[r] = off;
        off += ((uint64_t)read_len[r] + row_align - 1) & ~(row_align - 1);
    }
    ctx->h_pile_off[n_reads] = off;
    ctx->pile_elems = off;
    HIPCHECK(ctx->d_read_len.ensure(n_reads + 1));
    HIPCHECK(ctx->d_pile_off.ensure(n_reads + 1));
    ctx->d_pile.release();                       // allocated by rala_hip_initialize (owners only)
    HIPCHECK(hipMemcpy(ctx->d_read_len.p, read_len, n_reads * 4, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(ctx->d_pile_off.p, ctx->h_pile_off.data(), (n_reads + 1) * 8, hipMemcpyHostToDevice));
    HIPCHECK(ctx->d_order.ensure(n_reads + 1));
    HIPCHECK(ctx->d_overflow.ensure(n_reads + 1));
    HIPCHECK(ctx->d_ev_off.ensure(n_reads + 2)); HIPCHECK(ctx->d_cursor.ensure(n_reads + 2));
    HIPCHECK(ctx->d_suspect.ensure(n_reads + 1));
    HIPCHECK(ctx->d_begin.ensure(n_reads)); HIPCHECK(ctx->d_end.ensure(n_reads));
    HIPCHECK(ctx->d_median.ensure(n_reads)); HIPCHECK(ctx->d_p10.ensure(n_reads));
    HIPCHECK(ctx->d_alive.ensure(n_reads)); HIPCHECK(ctx->d_n_pits.ensure(n_reads));
    HIPCHECK(ctx->d_n_hills.ensure(n_reads)); HIPCHECK(ctx->d_iv_slot.ensure(n_reads));
    HIPCHECK(ctx->d_death[0].ensure(n_reads)); HIPCHECK(ctx->d_death[1].ensure(n_reads));
    // scans run over reads as well as over overlaps / tuples, whichever call comes first
    HIPCHECK(ctx->d_scan_ws.ensure(scan_workspace_bytes(std::max<uint64_t>(std::max(ctx->n_ovl, ctx->n_tuples), n_reads) + 2)));
    // host mirrors of the per-read state.  They are NOT registered with the runtime: registration
    // pins whole pages, small vectors of different contexts share pages of the heap, and
    // unregistering one context's vectors then unmapped pages another context's copies still went
    // to ("Memory access fault by GPU" in a later, unrelated call).
    ctx->h_begin.resize(n_reads); ctx->h_end.resize(n_reads); ctx->h_median.resize(n_reads);
    ctx->h_p10.resize(n_reads); ctx->h_alive.resize(n_reads); ctx->h_n_pits.resize(n_reads);
    ctx->h_n_hills.resize(n_reads); ctx->h_slot.resize(n_reads);
    // (a hint: a pool that turns out too small grows to the counted need; 0 = start at 16 slots, for the tests of that)
    ctx->pool_cap = (uint32_t)std::max<int64_t>(ctx->pool_per_read_x1000 == 0 ? 16 : 1024, (int64_t)n_reads * ctx->pool_per_read_x1000 / 1000);
    ctx->pool_cap_first = ctx->pool_cap;        // (the repeat hills' pool starts there too, and grows on its own)
    ctx->rep_pool_cap = 0;
    HIPCHECK(ctx->d_pool.ensure(ctx->pool_cap));
    ctx->initialized = ctx->constructed = ctx->ev_ready = false;
    return RALA_HIP_OK;
}

int rala_hip_set_overlaps(rala_hip_ctx* ctx, const rala_hip_overlaps* o, uint64_t n, int mem) {
    if (!ctx || (!o && n)) return RALA_HIP_EINVAL;
    // (2 n bound pairs in 32 bits - the partitioned bucketing's offsets, kernels.h: kBucketPairShift; the other bucketing paths count
    // 4 n events and say so in rala_hip_initialize when they meet more)
    if (n >= 0xFFFFFFF0ull / 2) return fail(ctx, RALA_HIP_EINVAL, "too many overlaps for 32-bit bound offsets");
    HIPCHECK(hipSetDevice(ctx->device));
    HIPCHECK(hipStreamSynchronize(ctx->side));          // a failed call may have left work there
    HIPCHECK(hipStreamSynchronize(ctx->aux));
    if (ctx->copy) HIPCHECK(hipStreamSynchronize(ctx->copy));
    ctx->upload_pending = false;
    if (mem == RALA_HIP_MEM_HOST_ASYNC && !ctx->copy) {
        // (made when first asked for: a process has few hardware queues - four by default - and its streams share them; a
        // stream that is not used must not push a context's main and aux streams onto one queue.  It did, in the sharded
        // runner's owner contexts: their small pile kernels ran in front of the big one instead of beside it, +0.13 ms at C3)
        // (all or nothing: a stream whose events could not all be made would leave later calls with null events - advisor round 5)
        hipStream_t made = nullptr;
        HIPCHECK(hipStreamCreate(&made));
        hipError_t bad = hipSuccess;
        for (auto& e : ctx->ev_up) {
            e = nullptr;
            if (bad == hipSuccess) bad = hipEventCreateWithFlags(&e, hipEventDisableTiming);
        }
        if (bad != hipSuccess) {
            for (auto& e : ctx->ev_up) { if (e) (void)hipEventDestroy(e); e = nullptr; }
            (void)hipStreamDestroy(made);
            HIPCHECK(bad);
        }
        ctx->copy = made;
    }
    ctx->n_ovl = n;
    const uint32_t* src[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (n) {
        src[0] = o->a_id; src[1] = o->b_id; src[2] = o->a_begin; src[3] = o->a_end;
        src[4] = o->b_begin; src[5] = o->b_end; src[6] = o->length;
    }
    const uint32_t* dev[7];
    const uint8_t* dev_strand;
    if (mem == RALA_HIP_MEM_DEVICE) {
        for (int k = 0; k < 7; ++k) dev[k] = src[k];
        dev_strand = n ? o->strand : nullptr;
    } else {
        // (RALA_HIP_MEM_HOST_ASYNC: room now, the copies in rala_hip_initialize - upload_columns)
        const bool later = mem == RALA_HIP_MEM_HOST_ASYNC && n != 0;
        for (int k = 0; k < 7; ++k) {
            HIPCHECK(ctx->d_ovl_u32[k].ensure(n));
            if (n && !later) HIPCHECK(hipMemcpy(ctx->d_ovl_u32[k].p, src[k], n * 4, hipMemcpyHostToDevice));
            dev[k] = ctx->d_ovl_u32[k].p;
            ctx->up_src[k] = src[k];
        }
        HIPCHECK(ctx->d_ovl_strand.ensure(n));
        if (n && !later) HIPCHECK(hipMemcpy(ctx->d_ovl_strand.p, o->strand, n, hipMemcpyHostToDevice));
        dev_strand = ctx->d_ovl_strand.p;
        ctx->up_strand = n ? o->strand : nullptr;
        ctx->upload_pending = later;
    }
    ctx->ovl.a_id = dev[0]; ctx->ovl.b_id = dev[1]; ctx->ovl.a_begin = dev[2]; ctx->ovl.a_end = dev[3];
    ctx->ovl.b_begin = dev[4]; ctx->ovl.b_end = dev[5]; ctx->ovl.length = dev[6]; ctx->ovl.strand = dev_strand;
    ctx->ovl.n = n;
    ctx->ovl.base = 0;
    ctx->tuple_mode = false;
    ctx->blocks_mode = false;
    ctx->dedupe_pending = false;
    ctx->inputs_set = true;
    ctx->valid_ready = false;
    HIPCHECK(ctx->d_valid.ensure(n));
    HIPCHECK(ctx->d_ev.ensure(4 * n + 8));
    HIPCHECK(ctx->d_cls.ensure(n / 4 + 64));        // survivor masks of the second pass: two bits per overlap
    for (int k = 0; k < 4; ++k) HIPCHECK(ctx->d_chunk[k].ensure(pass2_chunks(n) + 2));
    HIPCHECK(ctx->d_scan_ws.ensure(scan_workspace_bytes(std::max<uint64_t>(n, ctx->n_reads) + 2)));
    ctx->initialized = ctx->constructed = ctx->ev_ready = false;
    return RALA_HIP_OK;
}

static int initialize_stages(rala_hip_ctx* ctx);
int rala_hip_initialize(rala_hip_ctx* ctx) {
    if (!ctx) return RALA_HIP_EINVAL;
    const int rc = initialize_stages(ctx);
    // RALA_HIP_MEM_HOST_ASYNC: the host's columns are the caller's again when this call returns - however it returns
    if (ctx->upload_pending && ctx->copy) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->copy);
        // (a call that left before it had queued the copies has uploaded nothing: the next reader does, flush_upload)
        if (ctx->upload_queued) ctx->upload_pending = false;
    }
    ctx->upload_queued = false;
    return rc;
}
static int initialize_stages(rala_hip_ctx* ctx) {
    if (ctx->n_reads == 0) return fail(ctx, RALA_HIP_EINVAL, "no reads set");
    if (!ctx->inputs_set) return fail(ctx, RALA_HIP_EINVAL, "no overlaps or bound tuples set");
    HIPCHECK(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const uint32_t n_reads = (uint32_t)ctx->n_reads;
    ctx->tm = rala_hip_timings();
    ctx->overlaps.clear(); ctx->internals.clear();
    ctx->initialized = ctx->constructed = ctx->ev_ready = false;
    ctx->tail_on_device = ctx->host_stale = ctx->marks_on_device = false;

    {
        // The rows as physical chunks of 1 GB mapped side by side into one range (round 6).  Where the rows lie physically decides
        // whether the first pile kernel takes 3.8 or 4.3 ms at C3 (five placements inside one process: 3.84 / 4.20 / 4.12 / 4.27 /
        // 3.81) - its stores, a wavefront per row, 7 000 rows open at a time, are the only access pattern of the path that feels it.
        // hipMalloc's single block landed anywhere in that range; chunks of 1 GB mapped in order: 3.73 - 3.86 in 25 placements of 25,
        // the bench line 7.19 - 7.22 ms in five processes against 7.36 - 7.55 alternating with them.
        // Option pile_chunk_mb (0 = one hipMalloc; RALA_HIP_PILE_CHUNK_ORDER=1, measurements: the chunks permuted).  Falls back
        // to hipMalloc where the driver refuses the mapping calls.
        static const char* chunk_order = getenv("RALA_HIP_PILE_CHUNK_ORDER");
        hipError_t ec = hipErrorNotSupported;
        if (ctx->pile_chunk_mb > 0 && (ctx->pile_elems + 8) * sizeof(uint16_t) >= ((size_t)ctx->pile_chunk_mb << 20)) {
            ec = ctx->d_pile.ensure_chunked(ctx->pile_elems + 8, (size_t)ctx->pile_chunk_mb << 20, chunk_order ? atoi(chunk_order) : 0, ctx->device);
            if (ec != hipSuccess) (void)hipGetLastError();
        }
        if (ec != hipSuccess) HIPCHECK(ctx->d_pile.ensure(ctx->pile_elems + 8));
    }
    // (whatever a failed call may have left on the aux stream ends before the counters are reset)
    HIPCHECK(hipEventRecord(ctx->ev[9], ctx->aux));
    HIPCHECK(hipStreamWaitEvent(s, ctx->ev[9], 0));
    HIPCHECK(hipEventRecord(ctx->ev[0], s));
    // RALA_HIP_MEM_HOST_ASYNC: the columns leave the host now, in the order the kernels want them; the kernels below wait for
    // the event of the column they read first (the copy stream is in order: a later event covers the earlier columns)
    const bool uploading = ctx->upload_pending;
    if (uploading) {
        const size_t n4 = (size_t)ctx->n_ovl * 4;
        auto up = [&](int k) { return hipMemcpyAsync(ctx->d_ovl_u32[k].p, ctx->up_src[k], n4, hipMemcpyHostToDevice, ctx->copy); };
        // (a failed call may have left kernels that still read the columns: the copies behind them)
        HIPCHECK(hipStreamWaitEvent(ctx->copy, ctx->ev[0], 0));
        HIPCHECK(up(0)); HIPCHECK(up(1));
        HIPCHECK(hipEventRecord(ctx->ev_up[0], ctx->copy));
        HIPCHECK(up(4)); HIPCHECK(up(5));
        HIPCHECK(hipEventRecord(ctx->ev_up[1], ctx->copy));
        HIPCHECK(up(2)); HIPCHECK(up(3));
        HIPCHECK(hipEventRecord(ctx->ev_up[2], ctx->copy));
        HIPCHECK(up(6));
        HIPCHECK(hipMemcpyAsync(ctx->d_ovl_strand.p, ctx->up_strand, ctx->n_ovl, hipMemcpyHostToDevice, ctx->copy));
        HIPCHECK(hipEventRecord(ctx->ev_up[3], ctx->copy));      // everything
        ctx->upload_queued = true;
    }
    auto wait_for_all_columns = [&]() -> int {
        HIPCHECK(hipStreamWaitEvent(s, ctx->ev_up[3], 0));
        HIPCHECK(hipStreamWaitEvent(ctx->side, ctx->ev_up[3], 0));
        return (int)RALA_HIP_OK;
    };
    // the call's counters and what the bucketing wants cleared: one fill (fill_kernels.hip)
    FillList fills;
    fills.add(ctx->d_small.p, 0, 16 * 4);
    // duplicate removal only feeds the second pass (every resolvable overlap adds its bounds,
    // valid or not): it runs on a second stream and the main stream joins it after the pile kernels.
    // Beside WHAT it runs: beside the single-pass bucketing, both reading the same columns while the
    // atomics queue, the two slowed each other (0.2 - 0.5 ms per C3 step), so it starts when that
    // bucketing is done, beside the pile kernels.  Those are bound by instruction issue, and what runs
    // beside them costs them its whole stand-alone time (0.26 ms at C3: pile kernel 4.63 ms with it,
    // 4.37 without); the partitioned bucketing streams and has issue slots to spare - beside it the
    // duplicate removal costs 0.14 ms (round 4, docs/history/gpurun/r4_dedupe_early.sh: step 8.08 -> 7.99 ms).
    // RALA_DEDUPE_LATE keeps it beside the pile kernels.
    const bool forked = !ctx->tuple_mode && ctx->use_side_stream;
    if (uploading && !forked) {
        const int rcw = wait_for_all_columns();
        if (rcw != RALA_HIP_OK) return rcw;
    }
    if (!forked) {
        if (!ctx->tuple_mode) launch_dedupe(ctx->ovl, n_reads, ctx->d_suspect.p, ctx->d_valid.p, s);
        HIPCHECK(hipEventRecord(ctx->ev[1], s));
    }
    // bucket bounds by read.  Fast path: one kernel into fixed slots of kRunEventCapBig events per
    // read (8 KB; 8 GB at a million reads - HBM is 288 GB); the position inside the slot is what the
    // counting atomic returns, so there is no scan and no second pass over the overlaps.  A read
    // with more events than a slot (only the position-space kernel could take it) sends the whole
    // data set through the exact CSR path below.
    const uint32_t slot = kRunEventCapBig;
    // Partitioned path (bucket_kernels.hip): the target side through two partitioning passes instead of one
    // memory-side atomic and one partial write per overlap; ends in the exact CSR.  Needs the overlaps
    // (not tuples), coordinates below 2^25, enough overlaps per partition for the passes to pay, few enough
    // reads for a histogram of their groups of 128 to fit the LDS (4.9 M).
    // (round 6: whichever pile kernel follows - the position-space kernel reads the rows' offsets in pairs as well)
    const bool partition_allowed = ctx->use_fixed_buckets && ctx->use_partitioned_buckets;
    // (an owner rank's bound records: the same path from level 1 on, both sides as records; input that does not suit it
    // is turned into tuples)
    const bool from_records = ctx->tuple_mode && ctx->records != nullptr && partition_allowed &&
                              partition_path_fits_records(n_reads, ctx->max_read_len, ctx->n_records);
    if (ctx->tuple_mode && ctx->records != nullptr && !from_records) {
        HIPCHECK(ctx->d_tuple.ensure(2 * ctx->n_records + 8));
        launch_records_to_tuples(ctx->records, ctx->n_records, ctx->d_tuple.p, s);
        ctx->tuples = ctx->d_tuple.p;
        ctx->n_tuples = 2 * ctx->n_records;
    }
    // (an owner rank's blocks, scattered by partition on the senders: the same path from level 2 on - there is no other
    // form of that input)
    const bool from_blocks = ctx->tuple_mode && ctx->blocks_mode;
    const bool partitioned = from_blocks || from_records || (!ctx->tuple_mode && partition_allowed &&
                                                             partition_path_fits(n_reads, ctx->max_read_len, ctx->n_ovl));
    if (!partitioned && !ctx->tuple_mode && ctx->n_ovl >= 0xFFFFFFF0ull / 4) {
        return fail(ctx, RALA_HIP_ETOOLARGE, "2^30 overlaps and more need the partitioned bucketing (its offsets count bound pairs); this input or these options rule it out");
    }
    // (the partitioned bucketing's offsets count bound pairs; option debug_ev_events = 1, tests and measurements: events where they fit)
    const uint64_t n_bounds = from_blocks ? ctx->n_block_records : from_records ? ctx->n_records : 2 * ctx->n_ovl;
    const uint32_t ev_shift = !partitioned || (ctx->debug_ev_events && 2 * n_bounds < 0xFFFFFFF0ull) ? 0u : kBucketPairShift;
    bool fixed = !partitioned && ctx->use_run_kernel && ctx->use_fixed_buckets && (uint64_t)n_reads * slot * 4ull <= (64ull << 30);
    static const bool dedupe_late = getenv("RALA_DEDUPE_LATE") != nullptr;
    const bool dedupe_early = forked && partitioned && !dedupe_late;
    // (round 5) ... and inside the bucketing's counting pass where that pass can take it: both read the two id columns of every
    // overlap, the second stream is left with the marked queries' overlaps - usually none
    const bool dedupe_counted = dedupe_early && !from_records && !from_blocks && bucket_count_can_dedupe(ctx->ovl, ctx->d_valid.p);
    // (columns on their way: only the path below that names what it reads when takes them as they come)
    const bool fine_upload = uploading && dedupe_counted && partitioned;
    if (uploading && forked && !fine_upload) {
        const int rcw = wait_for_all_columns();
        if (rcw != RALA_HIP_OK) return rcw;
    }
    if (dedupe_early && !dedupe_counted) {
        HIPCHECK(hipStreamWaitEvent(ctx->side, ctx->ev[0], 0));
        launch_dedupe(ctx->ovl, n_reads, ctx->d_suspect.p, ctx->d_valid.p, ctx->side);
        HIPCHECK(hipEventRecord(ctx->ev[1], ctx->side));
    }
    if (from_blocks) {
        const ShardGeometry& g = ctx->shard_geom;
        HIPCHECK(ctx->d_shard_group.ensure(shard_group_words(g)));
        HIPCHECK(ctx->d_shard_tiles.ensure(3 * shard_tile_slots(g, ctx->n_block_records) + 2));
        HIPCHECK(ctx->d_bk_rec[1].ensure((size_t)ctx->n_block_records + 64));
        HIPCHECK(launch_bucket_from_blocks(ctx->blocks_base, ctx->blocks_base_self, ctx->blocks, g, n_reads, ctx->n_block_records,
                                           ctx->d_shard_group.p, ctx->d_shard_tiles.p, ctx->d_bk_rec[1].p, ctx->d_ev_off.p, ctx->d_ev.p, fills, s, ev_shift));
    } else if (partitioned) {
        const uint64_t n_rec = from_records ? ctx->n_records : ctx->n_ovl;
        for (int k = 0; k < 3; ++k) HIPCHECK(ctx->d_bk_u32[k].ensure(n_reads + 2));
        HIPCHECK(ctx->d_bk_part.ensure(partition_count(n_reads) + 2));
        HIPCHECK(ctx->d_bk_group.ensure(3 * (size_t)partition_group_slots(n_reads)));
        HIPCHECK(ctx->d_bk_tiles.ensure(3 * partition_tile_slots(n_reads, n_rec) + 2));
        for (int k = 0; k < 2; ++k) HIPCHECK(ctx->d_bk_rec[k].ensure(partition_records_needed(n_reads, n_rec)));
        if (from_records) {
            HIPCHECK(launch_bucket_partitioned_records(ctx->records, ctx->n_records, n_reads, ctx->d_bk_u32[0].p, ctx->d_bk_part.p,
                                                       ctx->d_bk_group.p, ctx->d_bk_tiles.p, ctx->d_bk_rec[0].p, ctx->d_bk_rec[1].p,
                                                       ctx->d_ev_off.p, ctx->d_ev.p, ctx->n_compute_units, fills, s, 15u, ev_shift));
        } else {
            constexpr uint32_t kMarks = 1u << 20;           // (C3: 1 - 2 % of a million queries)
            if (dedupe_counted) HIPCHECK(ctx->d_dedupe_list.ensure(2 * (size_t)kMarks));
            BucketDedupe bd = {ctx->d_suspect.p, ctx->d_valid.p, ctx->d_dedupe_list.p, ctx->d_dedupe_list.p + kMarks,
                               ctx->debug_dedupe_list_cap ? std::min(kMarks, ctx->debug_dedupe_list_cap) : kMarks,
                               ctx->d_small.p + 9, ctx->ev[8]};     // ([9]: zeroed with d_small above)
            if (fine_upload) { bd.ids = ctx->ev_up[0]; bd.b_coords = ctx->ev_up[1]; bd.a_coords = ctx->ev_up[2]; }
            HIPCHECK(launch_bucket_partitioned(ctx->ovl, n_reads, ctx->d_bk_u32[0].p, ctx->d_bk_u32[2].p,
                                               ctx->d_bk_part.p, ctx->d_bk_group.p, ctx->d_bk_tiles.p, ctx->d_bk_rec[0].p,
                                               ctx->d_bk_rec[1].p, ctx->d_ev_off.p, ctx->d_ev.p, ctx->n_compute_units, fills, s,
                                               dedupe_counted ? &bd : nullptr, ev_shift));
            if (dedupe_counted) {
                HIPCHECK(hipStreamWaitEvent(ctx->side, ctx->ev[8], 0));
                if (fine_upload) HIPCHECK(hipStreamWaitEvent(ctx->side, ctx->ev_up[3], 0));      // (the lengths)
                launch_dedupe_fix(ctx->ovl, n_reads, bd, ctx->side);
                HIPCHECK(hipEventRecord(ctx->ev[1], ctx->side));
            }
        }
    }
    if (fixed) {
        HIPCHECK(ctx->d_ev_fixed.ensure((size_t)n_reads * slot + 8));
        HIPCHECK(ctx->d_cc_flags.ensure(8));
        fills.add(ctx->d_cursor.p, 0, (size_t)(n_reads + 1) * 4);
        HIPCHECK(fills.launch(s));
        // (the slot-overflow flag and the count of dead reads below live in d_small, zeroed above and
        // fetched in one copy with the other results)
        if (ctx->tuple_mode) {
            launch_bucket_fixed_tuples(ctx->tuples, ctx->n_tuples, n_reads, slot, ctx->d_cursor.p,
                                       ctx->d_ev_fixed.p, ctx->d_small.p + 6, s);
        } else {
            launch_bucket_fixed(ctx->ovl, n_reads, slot, ctx->d_cursor.p, ctx->d_ev_fixed.p, ctx->d_small.p + 6, s);
        }
        // the overflow flag is read together with the other results at the end of this call; a
        // set flag repeats the call on the exact path (no host round trip in the common case)
    }
    if (!fixed && !partitioned) {
    // count -> exclusive scan -> scatter
    fills.add(ctx->d_cursor.p, 0, (size_t)(n_reads + 1) * 4);
    HIPCHECK(fills.launch(s));
    if (ctx->tuple_mode) {
        launch_count_tuples(ctx->tuples, ctx->n_tuples, n_reads, ctx->d_cursor.p, s);
    } else {
        HIPCHECK(ctx->d_slot_rank[0].ensure(ctx->n_ovl + 1)); HIPCHECK(ctx->d_slot_rank[1].ensure(ctx->n_ovl + 1));
        launch_count_bounds(ctx->ovl, n_reads, ctx->d_cursor.p, ctx->d_slot_rank[0].p, ctx->d_slot_rank[1].p, s);
    }
    launch_exclusive_scan(ctx->d_cursor.p, ctx->d_ev_off.p, n_reads, ctx->d_scan_ws.p, s);
    if (ctx->tuple_mode) {
        HIPCHECK(hipMemcpyAsync(ctx->d_cursor.p, ctx->d_ev_off.p, (size_t)n_reads * 4, hipMemcpyDeviceToDevice, s));
        launch_scatter_tuples(ctx->tuples, ctx->n_tuples, n_reads, ctx->d_cursor.p, ctx->d_ev.p, s);
    } else {
        launch_scatter_bounds(ctx->ovl, n_reads, ctx->d_ev_off.p, ctx->d_slot_rank[0].p, ctx->d_slot_rank[1].p, ctx->d_ev.p, s);
    }
    }
    HIPCHECK(hipEventRecord(ctx->ev[2], s));
    if (forked && !dedupe_early) {
        HIPCHECK(hipStreamWaitEvent(ctx->side, ctx->ev[2], 0));
        launch_dedupe(ctx->ovl, n_reads, ctx->d_suspect.p, ctx->d_valid.p, ctx->side);
        HIPCHECK(hipEventRecord(ctx->ev[1], ctx->side));
    }

    PileArgs a;
    a.read_len = ctx->d_read_len.p; a.pile_off = ctx->d_pile_off.p; a.pile = ctx->d_pile.p;
    a.ev_off = ctx->d_ev_off.p; a.ev = fixed ? ctx->d_ev_fixed.p : ctx->d_ev.p;
    a.ev_cnt = fixed ? ctx->d_cursor.p : nullptr; a.ev_stride = slot;
    ctx->ev_shift = ev_shift;
    a.ev_shift = ev_shift;
    a.rows_chunked = ctx->d_pile.reserved ? 1u : 0u;
    a.add_to_existing = 0; a.slab = ctx->d_slab.p;
    a.stop_after = (uint32_t)ctx->debug_pile_stop_after;
    a.variant = ctx->debug_pile_variant;
    a.begin = ctx->d_begin.p; a.end = ctx->d_end.p; a.median = ctx->d_median.p; a.p10 = ctx->d_p10.p;
    a.alive = ctx->d_alive.p; a.n_pits = ctx->d_n_pits.p; a.n_hills = ctx->d_n_hills.p; a.iv_slot = ctx->d_iv_slot.p;
    a.pool = ctx->d_pool.p; a.pool_count = ctx->d_small.p; a.pool_cap = ctx->pool_cap; a.error = ctx->d_small.p + 1;
    a.n_items_dev = nullptr;
    // (what outgrows the position-space kernel's LDS lists: noted, and dealt with after this call's one look from the host)
    HIPCHECK(ctx->d_big_list[0].ensure(n_reads + 1));
    a.big_list = ctx->d_big_list[0].p; a.big_count = ctx->d_small.p + 8;
    a.force_big = ctx->debug_force_big ? 1u : 0u;
    if (ctx->use_run_kernel) {
        // Chain without host synchronisation.  Every read starts in the kernel that fits it, known
        // beforehand: by length class (set_reads sorted them) and by event count (listed from the
        // bucket counts).  A hand-over through a list behind the first kernel costs one atomic on
        // one counter per read (1 ms for 100 000 reads) and puts the small, latency-bound kernels
        // behind the big one (0.2 ms per C3 step); now they run beside it on a stream of their own:
        //   main  up to 16384 bases, up to 512 events: run-space kernel, seven wavefronts per SIMD,
        //         one workgroup per read
        //   aux   more than 512 events (dense list) -> cap 1024;  up to 32768 bases -> the short
        //         layout with a bitmap twice the size (four wavefronts per SIMD, 9 % faster on
        //         20 kb reads than what follows);  longer -> cap 512 for any length (sorted events)
        // What still does not fit its kernel (slope-region lists, event caps) goes to a list: of
        // the cap-512 kernels (list 1) -> cap 1024, of those (list 2) -> cap 2048 -> what is left
        // (list 3) to the position-space kernel, sized for the longest read.  These run on the
        // main stream behind the join, usually over nothing.
        HIPCHECK(ctx->d_overflow_mid.ensure(n_reads + 1));
        HIPCHECK(ctx->d_dense.ensure(n_reads + 1));
        uint32_t* list_dense = ctx->d_dense.p;
        uint32_t* list1 = ctx->d_overflow.p;
        uint32_t* list2 = ctx->d_overflow_mid.p;
        uint32_t* list3 = ctx->d_order.p;
        // (all four zeroed with d_small at the top of this call)
        uint32_t* cnt_dense = ctx->d_small.p + 3;
        uint32_t* cnt1 = ctx->d_small.p + 4;
        uint32_t* cnt2 = ctx->d_small.p + 2;
        uint32_t* cnt3 = ctx->d_small.p + 5;
        const uint32_t n_short = ctx->n_class[0], n_medium = ctx->n_class[1], n_long = ctx->n_class[2];
        const uint32_t* by_class = n_short != n_reads ? ctx->d_class_order.p : nullptr;
        // few longer reads: the first kernel goes over all reads without the indirection (2 % at C3)
        // and leaves at once for a longer one
        static const bool force_order = getenv("RALA_PILE_FORCE_ORDER") != nullptr;
        const bool short_by_list = by_class && (force_order || (uint64_t)(n_reads - n_short) * 8 > n_reads);
        static const bool no_aux = getenv("RALA_PILE_NO_AUX") != nullptr;       // measurements: everything on one stream
        hipStream_t aux = ctx->use_side_stream && !no_aux ? ctx->aux : s;
        if (aux != s) HIPCHECK(hipStreamWaitEvent(aux, ctx->ev[2], 0));
        a.lw = 0;
        a.skip_dense = 1;
        launch_pile_dense_list(a, n_reads, list_dense, cnt_dense, aux);
        a.order = list_dense;
        a.n_items = n_reads;
        a.n_items_dev = cnt_dense;
        launch_pile_runs(a, std::min<uint32_t>(n_reads, 8192), 1, list2, cnt2, aux);
        a.n_items_dev = nullptr;
        a.order = by_class ? by_class + n_short : nullptr;
        a.n_items = n_medium;
        // (measurements: RALA_PILE_ANYLEN_FROM=1 sends this class to the any-length kernel, too)
        static const int anylen_from = getenv("RALA_PILE_ANYLEN_FROM") ? atoi(getenv("RALA_PILE_ANYLEN_FROM")) : 3;
        launch_pile_runs(a, std::min<uint32_t>(n_medium, 16384), anylen_from <= 1 ? 4 : 3, list1, cnt1, aux);
        a.order = by_class ? by_class + n_short + n_medium : nullptr;
        a.n_items = n_long;
        launch_pile_runs(a, std::min<uint32_t>(n_long, 20480), 4, list1, cnt1, aux);
        if (aux != s) HIPCHECK(hipEventRecord(ctx->ev[8], aux));
        a.order = short_by_list ? by_class : nullptr;
        a.n_items = short_by_list ? n_short : n_reads;
        launch_pile_runs(a, a.n_items, 0, list1, cnt1, s);
        if (aux != s) HIPCHECK(hipStreamWaitEvent(s, ctx->ev[8], 0));
        a.n_items = n_reads;
        a.order = list1;
        a.n_items_dev = cnt1;
        launch_pile_runs(a, std::min<uint32_t>(n_reads, 8192), 1, list2, cnt2, s);
        a.order = list2;
        a.n_items_dev = cnt2;
        launch_pile_runs(a, std::min<uint32_t>(n_reads, 2048), 2, list3, cnt3, s);
        a.order = list3;
        a.n_items_dev = cnt3;
        const uint32_t max_len = ctx->max_read_len;
        const bool in_lds = (int64_t)max_len <= ctx->max_lds_read_len && pile_lw_for(max_len) <= 24576;
        a.lw = pile_lw_for(max_len);
        const uint32_t grid = std::min<uint32_t>(n_reads, 256);
        if (!in_lds) HIPCHECK(ctx->d_slab.ensure((size_t)grid * 3 * a.lw));
        a.slab = ctx->d_slab.p;
        launch_pile_build_annotate(a, grid, in_lds, s);
        ctx->tm.pile_launches = 5 + (n_medium ? 1 : 0) + (n_long ? 1 : 0);
    } else {
        std::vector<uint32_t> reads(n_reads);
        std::iota(reads.begin(), reads.end(), 0u);
        const int rc2 = run_position_kernel(ctx, a, reads);
        if (rc2 != RALA_HIP_OK) return rc2;
    }
    HIPCHECK(hipEventRecord(ctx->ev[3], s));
    HIPCHECK(hipGetLastError());
    if (forked) HIPCHECK(hipStreamWaitEvent(s, ctx->ev[1], 0));

    // the per-read results stay on the device; host mirrors are fetched by the first getter
    launch_count_zero_u8(ctx->d_alive.p, n_reads, ctx->d_small.p + 7, s);
    uint32_t small[16];
    HIPCHECK(d2h_small(ctx, small, ctx->d_small.p, sizeof(small), s));
    HIPCHECK(stream_sync(ctx, s));
    if (uploading) {
        // (the host's columns are free from here on)
        HIPCHECK(hipStreamSynchronize(ctx->copy));
        ctx->upload_pending = false;
    }
    const uint32_t slot_overflow = fixed ? small[6] : 0;
    if (small[8] && !slot_overflow) {
        // reads whose region / interval lists outgrew the position-space kernel's LDS lists
        const int rc_big = run_unbounded_piles(ctx, a, small[8], ctx->d_small.p + 8);
        if (rc_big != RALA_HIP_OK) return rc_big;
        HIPCHECK(hipMemsetAsync(ctx->d_small.p + 7, 0, 4, s));
        launch_count_zero_u8(ctx->d_alive.p, n_reads, ctx->d_small.p + 7, s);
        const uint32_t noted = small[8];
        HIPCHECK(d2h_small(ctx, small, ctx->d_small.p, sizeof(small), s));
        HIPCHECK(stream_sync(ctx, s));
        small[8] = noted;
    }
    const uint32_t n_dead = small[7];
    if (slot_overflow) {
        // some read has more events than a fixed slot holds: once more, through the exact path
        const int64_t keep = ctx->use_fixed_buckets;
        ctx->use_fixed_buckets = 0;
        const int rc_again = rala_hip_initialize(ctx);
        ctx->use_fixed_buckets = keep;
        return rc_again;
    }
    ctx->host_state_fresh = false;
    ctx->pool_used = std::min(small[0], ctx->pool_cap);
    ctx->tm.pile_overflow_reads = ctx->use_run_kernel ? small[3] + small[4] : 0;      // event-dense + handed on
    if (getenv("RALA_HIP_TRACE") || getenv("RALA_HIP_TRACE_BUFFERS")) {
        fprintf(stderr, "[trace] buffers: events %p (CSR %p) slots %p counts %p piles %p a_id %p b_id %p b_begin %p\n", (void*)ctx->d_ev.p, (void*)ctx->d_ev_off.p, (void*)ctx->d_ev_fixed.p,
                (void*)ctx->d_cursor.p, (void*)ctx->d_pile.p, (const void*)ctx->ovl.a_id, (const void*)ctx->ovl.b_id,
                (const void*)ctx->ovl.b_begin);
        if (getenv("RALA_HIP_TRACE")) fprintf(stderr, "[trace] pile chain: %u reads listed as event-dense, %u handed on by the cap-512 kernels, %u on to cap 2048, %u to position space\n",
                small[3], small[4], small[2], small[5]);
    }
    ctx->tm.pile_position_reads = ctx->use_run_kernel ? small[5] : n_reads;
    // dedupe_ms: what duplicate removal adds to the critical path (it runs beside the bucketing
    // and the pile kernels; the main stream joins it after them)
    {
        float dd = 0, bk = 0, all = 0;
        HIPCHECK(hipEventElapsedTime(&dd, ctx->ev[0], ctx->ev[1]));
        HIPCHECK(hipEventElapsedTime(&bk, forked ? ctx->ev[0] : ctx->ev[1], ctx->ev[2]));
        HIPCHECK(hipEventElapsedTime(&all, ctx->ev[0], ctx->ev[3]));
        ctx->tm.bucket_ms = bk;
        ctx->tm.dedupe_ms = forked ? std::max(0.0f, dd - all) : dd;
    }
    HIPCHECK(hipEventElapsedTime(&ctx->tm.pile_ms, ctx->ev[2], ctx->ev[3]));
    if (small[1] & (kErrRegionCapacity | kErrRawCapacity)) return fail(ctx, RALA_HIP_EDEVICE, "slope-region list overflow in the pile kernel");
    if (small[1] & kErrPoolCapacity) {
        // more pits and hills than the pool holds (interval_pool_per_read_x1000 is a hint): its counter holds what is needed -
        // once more with a pool of that size (and a little more: the slots of reads that ran twice are not handed back)
        const uint64_t need = (uint64_t)small[0] + small[0] / 8 + 1024;
        if (small[0] <= ctx->pool_cap || need > 0xFFFFFFF0ull) return fail(ctx, RALA_HIP_ENOMEM, "interval pool beyond 2^32 entries");
        ctx->pool_cap = (uint32_t)need;
        HIPCHECK(ctx->d_pool.ensure(ctx->pool_cap));
        const uint32_t regrown = ctx->tm.pool_regrown + 1;
        const int rc_again = rala_hip_initialize(ctx);
        ctx->tm.pool_regrown = regrown;
        return rc_again;
    }
    ctx->n_prefiltered = n_dead;
    ctx->ev_ready = true;
    ctx->ev_fixed = fixed;
    ctx->initialized = true;
    ctx->valid_ready = !ctx->tuple_mode;
    ctx->piles_resident = true;
    if (ctx->n_prefiltered == ctx->n_reads) return fail(ctx, RALA_HIP_EFILTERED, "filtered all sequences");
    return RALA_HIP_OK;
}

int rala_hip_dedupe(rala_hip_ctx* ctx) {
    if (!ctx) return RALA_HIP_EINVAL;
    if (ctx->n_reads == 0) return fail(ctx, RALA_HIP_EINVAL, "no reads set");
    if (!ctx->inputs_set || ctx->tuple_mode) return fail(ctx, RALA_HIP_EINVAL, "no overlaps set");
    HIPCHECK(hipSetDevice(ctx->device));
    { const int rcu = flush_upload(ctx); if (rcu != RALA_HIP_OK) return rcu; }
    launch_dedupe(ctx->ovl, (uint32_t)ctx->n_reads, ctx->d_suspect.p, ctx->d_valid.p, ctx->stream);
    HIPCHECK(stream_sync(ctx, ctx->stream));
    HIPCHECK(hipGetLastError());
    ctx->valid_ready = true;
    return RALA_HIP_OK;
}

int rala_hip_emit_bound_tuples(rala_hip_ctx* ctx, uint64_t* tuples_dev) {
    if (!ctx || !tuples_dev) return RALA_HIP_EINVAL;
    if (((uintptr_t)tuples_dev & 15u) != 0) return fail(ctx, RALA_HIP_EINVAL, "tuple buffer must be 16-byte aligned");
    HIPCHECK(hipSetDevice(ctx->device));
    { const int rcu = flush_upload(ctx); if (rcu != RALA_HIP_OK) return rcu; }
    launch_emit_tuples(ctx->ovl, (uint32_t)ctx->n_reads, (uint2*)tuples_dev, ctx->stream);
    HIPCHECK(stream_sync(ctx, ctx->stream));
    HIPCHECK(hipGetLastError());
    return RALA_HIP_OK;
}

namespace {
int emit_bucketed(rala_hip_ctx* ctx, uint32_t world, uint64_t* tuples_dev, uint64_t* counts, bool records);
}
int rala_hip_emit_bound_tuples_bucketed(rala_hip_ctx* ctx, uint32_t world, uint64_t* tuples_dev, uint64_t* counts) {
    return emit_bucketed(ctx, world, tuples_dev, counts, false);
}
int rala_hip_bound_records_fit(const rala_hip_ctx* ctx, uint32_t world) {
    if (!ctx || world == 0) return 0;
    const uint64_t local = (ctx->n_reads + world - 1) / world;
    return ctx->max_read_len < (1u << kBoundRecordCoordBits) - 32u && local < (1ull << kBoundRecordReadBits) ? 1 : 0;
}
int rala_hip_emit_bound_records_bucketed(rala_hip_ctx* ctx, uint32_t world, uint64_t* records_dev, uint64_t* counts) {
    if (!ctx) return RALA_HIP_EINVAL;
    if (!rala_hip_bound_records_fit(ctx, world)) return fail(ctx, RALA_HIP_EINVAL, "reads too long or too many for bound records");
    return emit_bucketed(ctx, world, records_dev, counts, true);
}
namespace {
int emit_bucketed(rala_hip_ctx* ctx, uint32_t world, uint64_t* tuples_dev, uint64_t* counts, bool records) {
    if (!ctx || !tuples_dev || !counts || world == 0 || world > 64) return RALA_HIP_EINVAL;
    if (((uintptr_t)tuples_dev & 15u) != 0) return fail(ctx, RALA_HIP_EINVAL, "tuple buffer must be 16-byte aligned");
    HIPCHECK(hipSetDevice(ctx->device));
    { const int rcu = flush_upload(ctx); if (rcu != RALA_HIP_OK) return rcu; }
    hipStream_t s = ctx->stream;
    HIPCHECK(ctx->d_owner_cnt.ensure(2 * 64));
    uint32_t* cnt = ctx->d_owner_cnt.p;
    uint32_t* cur = cnt + 64;
    HIPCHECK(hipMemsetAsync(cnt, 0, 2 * 64 * 4, s));
    launch_bucket_tuples(ctx->ovl, (uint32_t)ctx->n_reads, world, 0, cnt, (uint2*)tuples_dev, s, records);
    // (the buckets' places from the counts on the device: the host looks once, at the end)
    launch_owner_offsets(cnt, world, cur, s);
    launch_bucket_tuples(ctx->ovl, (uint32_t)ctx->n_reads, world, 1, cur, (uint2*)tuples_dev, s, records);
    uint32_t h[64];
    HIPCHECK(d2h_small(ctx, h, cnt, world * 4, s));
    HIPCHECK(stream_sync(ctx, s));
    HIPCHECK(hipGetLastError());
    for (uint32_t p = 0; p < world; ++p) counts[p] = h[p];
    return RALA_HIP_OK;
}
}  // namespace

int rala_hip_get_device_state(rala_hip_ctx* ctx, rala_hip_device_state* out) {
    if (!ctx || !out) return RALA_HIP_EINVAL;
    if (!ctx->initialized) return fail(ctx, RALA_HIP_EINVAL, "not initialized");
    out->begin = ctx->d_begin.p; out->end = ctx->d_end.p; out->median = ctx->d_median.p; out->p10 = ctx->d_p10.p;
    out->alive = ctx->d_alive.p; out->n_pits = ctx->d_n_pits.p; out->n_hills = ctx->d_n_hills.p;
    out->slot = ctx->d_iv_slot.p; out->pool = ctx->d_pool.p; out->pool_count = (size_t)ctx->pool_used;
    out->valid = ctx->valid_ready ? ctx->d_valid.p : nullptr;
    return RALA_HIP_OK;
}

int rala_hip_copy_device_state(rala_hip_ctx* ctx, const rala_hip_device_state* dst) {
    if (!ctx || !dst) return RALA_HIP_EINVAL;
    const bool per_read = dst->begin || dst->end || dst->median || dst->p10 || dst->alive || dst->n_pits ||
                          dst->n_hills || dst->slot || dst->pool;
    if (per_read && !ctx->initialized) return fail(ctx, RALA_HIP_EINVAL, "not initialized");
    HIPCHECK(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const uint64_t n = ctx->n_reads;
    auto cp = [&](const void* to, const void* from, size_t bytes) -> hipError_t {
        if (!to || bytes == 0) return hipSuccess;
        return hipMemcpyAsync(const_cast<void*>(to), from, bytes, hipMemcpyDeviceToDevice, s);
    };
    HIPCHECK(cp(dst->begin, ctx->d_begin.p, n * 4));
    HIPCHECK(cp(dst->end, ctx->d_end.p, n * 4));
    HIPCHECK(cp(dst->median, ctx->d_median.p, n * 2));
    HIPCHECK(cp(dst->p10, ctx->d_p10.p, n * 2));
    HIPCHECK(cp(dst->alive, ctx->d_alive.p, n));
    HIPCHECK(cp(dst->n_pits, ctx->d_n_pits.p, n * 4));
    HIPCHECK(cp(dst->n_hills, ctx->d_n_hills.p, n * 4));
    HIPCHECK(cp(dst->slot, ctx->d_iv_slot.p, n * 4));
    if (dst->pool) {
        if (dst->pool_count < (size_t)ctx->pool_used) return fail(ctx, RALA_HIP_ECAPACITY, "pool buffer too small");
        HIPCHECK(cp(dst->pool, ctx->d_pool.p, (size_t)ctx->pool_used * sizeof(Interval)));
    }
    if (dst->valid) {
        if (!ctx->valid_ready) return fail(ctx, RALA_HIP_EINVAL, "no validity bits on this context");
        HIPCHECK(cp(dst->valid, ctx->d_valid.p, ctx->n_ovl));
    }
    HIPCHECK(stream_sync(ctx, s));
    return RALA_HIP_OK;
}

int rala_hip_import_state_device(rala_hip_ctx* ctx, const rala_hip_device_state* in) {
    if (!ctx || !in || !in->begin || !in->end || !in->median || !in->p10 || !in->alive || !in->n_pits ||
        !in->n_hills || !in->slot) {
        return RALA_HIP_EINVAL;
    }
    if (ctx->n_reads == 0) return fail(ctx, RALA_HIP_EINVAL, "no reads set");
    if (ctx->n_ovl && !in->valid) return fail(ctx, RALA_HIP_EINVAL, "valid bits required");
    if (in->pool_count && !in->pool) return RALA_HIP_EINVAL;
    HIPCHECK(hipSetDevice(ctx->device));
    // (columns handed over with RALA_HIP_MEM_HOST_ASYNC leave inside rala_hip_initialize; a caller that installs the read state
    // instead must not find them still on the host - advisor round 5)
    { const int rcu = flush_upload(ctx); if (rcu != RALA_HIP_OK) return rcu; }
    hipStream_t s = ctx->stream;
    const uint64_t n = ctx->n_reads;
    ctx->tm = rala_hip_timings();
    ctx->overlaps.clear(); ctx->internals.clear();
    if (in->pool_count > ctx->pool_cap) {
        ctx->pool_cap = (uint32_t)in->pool_count + 1024;
        HIPCHECK(ctx->d_pool.ensure(ctx->pool_cap));
    }
    const uint32_t small[8] = {(uint32_t)in->pool_count, 0, 0, 0, 0, 0, 0, 0};
    HIPCHECK(hipMemcpyAsync(ctx->d_small.p, small, sizeof(small), hipMemcpyHostToDevice, s));
    HIPCHECK(hipMemcpyAsync(ctx->d_begin.p, in->begin, n * 4, hipMemcpyDeviceToDevice, s));
    HIPCHECK(hipMemcpyAsync(ctx->d_end.p, in->end, n * 4, hipMemcpyDeviceToDevice, s));
    HIPCHECK(hipMemcpyAsync(ctx->d_median.p, in->median, n * 2, hipMemcpyDeviceToDevice, s));
    HIPCHECK(hipMemcpyAsync(ctx->d_p10.p, in->p10, n * 2, hipMemcpyDeviceToDevice, s));
    HIPCHECK(hipMemcpyAsync(ctx->d_alive.p, in->alive, n, hipMemcpyDeviceToDevice, s));
    HIPCHECK(hipMemcpyAsync(ctx->d_n_pits.p, in->n_pits, n * 4, hipMemcpyDeviceToDevice, s));
    HIPCHECK(hipMemcpyAsync(ctx->d_n_hills.p, in->n_hills, n * 4, hipMemcpyDeviceToDevice, s));
    HIPCHECK(hipMemcpyAsync(ctx->d_iv_slot.p, in->slot, n * 4, hipMemcpyDeviceToDevice, s));
    if (in->pool_count) {
        HIPCHECK(hipMemcpyAsync(ctx->d_pool.p, in->pool, (size_t)in->pool_count * sizeof(Interval), hipMemcpyDeviceToDevice, s));
    }
    if (ctx->n_ovl) HIPCHECK(hipMemcpyAsync(ctx->d_valid.p, in->valid, ctx->n_ovl, hipMemcpyDeviceToDevice, s));
    // host mirrors are fetched by the first getter; only the number of filtered reads is needed now
    HIPCHECK(ctx->d_cc_flags.ensure(8));
    HIPCHECK(hipMemsetAsync(ctx->d_cc_flags.p + 6, 0, 4, s));
    launch_count_zero_u8(ctx->d_alive.p, (uint32_t)n, ctx->d_cc_flags.p + 6, s);
    uint32_t n_dead = 0;
    HIPCHECK(d2h_small(ctx, &n_dead, ctx->d_cc_flags.p + 6, 4, s));
    HIPCHECK(stream_sync(ctx, s));
    ctx->host_state_fresh = false;
    ctx->pool_used = (uint32_t)in->pool_count;
    ctx->n_prefiltered = n_dead;
    ctx->initialized = true;
    ctx->valid_ready = true;
    ctx->constructed = false;
    ctx->piles_resident = ctx->ev_ready = false;
    ctx->tail_on_device = ctx->host_stale = ctx->marks_on_device = false;
    if (ctx->n_prefiltered == n) return fail(ctx, RALA_HIP_EFILTERED, "filtered all sequences");
    return RALA_HIP_OK;
}

int rala_hip_set_bound_tuples(rala_hip_ctx* ctx, const uint64_t* tuples, uint64_t n, int mem) {
    if (!ctx || (n && !tuples)) return RALA_HIP_EINVAL;
    if (ctx->n_reads == 0) return fail(ctx, RALA_HIP_EINVAL, "no reads set");
    if (n >= 0xFFFFFFF0ull) return fail(ctx, RALA_HIP_EINVAL, "too many tuples");
    HIPCHECK(hipSetDevice(ctx->device));
    if (mem == RALA_HIP_MEM_DEVICE) {
        ctx->tuples = (const uint2*)tuples;
    } else {
        HIPCHECK(ctx->d_tuple.ensure(n));
        if (n) HIPCHECK(hipMemcpy(ctx->d_tuple.p, tuples, n * 8, hipMemcpyHostToDevice));
        ctx->tuples = ctx->d_tuple.p;
    }
    ctx->n_tuples = n;
    ctx->records = nullptr;
    ctx->n_records = 0;
    ctx->blocks_mode = false;
    ctx->tuple_mode = true;
    ctx->inputs_set = true;
    ctx->n_ovl = 0;
    ctx->ovl = OvlSoA();
    HIPCHECK(ctx->d_ev.ensure(n + 8));
    HIPCHECK(ctx->d_scan_ws.ensure(scan_workspace_bytes(std::max<uint64_t>(n, ctx->n_reads) + 2)));
    ctx->initialized = ctx->constructed = ctx->ev_ready = false;
    return RALA_HIP_OK;
}

int rala_hip_set_bound_records(rala_hip_ctx* ctx, const uint64_t* records, uint64_t n, int mem) {
    if (!ctx || (n && !records)) return RALA_HIP_EINVAL;
    if (ctx->n_reads == 0) return fail(ctx, RALA_HIP_EINVAL, "no reads set");
    if (n >= 0x7FFFFFF0ull) return fail(ctx, RALA_HIP_EINVAL, "too many records");
    if (!rala_hip_bound_records_fit(ctx, 1)) return fail(ctx, RALA_HIP_EINVAL, "reads too long or too many for bound records");
    HIPCHECK(hipSetDevice(ctx->device));
    if (mem == RALA_HIP_MEM_DEVICE) {
        ctx->records = records;
    } else {
        HIPCHECK(ctx->d_record.ensure(n + 8));
        if (n) HIPCHECK(hipMemcpy(ctx->d_record.p, records, n * 8, hipMemcpyHostToDevice));
        ctx->records = ctx->d_record.p;
    }
    if (!ctx->records) { HIPCHECK(ctx->d_record.ensure(8)); ctx->records = ctx->d_record.p; }   // (an empty share)
    ctx->n_records = n;
    ctx->tuples = nullptr;
    ctx->n_tuples = 2 * n;              // (what the tuple paths would see)
    ctx->blocks_mode = false;
    ctx->tuple_mode = true;
    ctx->inputs_set = true;
    ctx->n_ovl = 0;
    ctx->ovl = OvlSoA();
    HIPCHECK(ctx->d_ev.ensure(2 * n + 8));
    HIPCHECK(ctx->d_scan_ws.ensure(scan_workspace_bytes(std::max<uint64_t>(2 * n, ctx->n_reads) + 2)));
    ctx->initialized = ctx->constructed = ctx->ev_ready = false;
    return RALA_HIP_OK;
}

int rala_hip_import_state(rala_hip_ctx* ctx, const uint8_t* valid, const uint32_t* begin, const uint32_t* end,
                          const uint16_t* median, const uint16_t* p10, const uint8_t* alive,
                          const uint64_t* pits_off, const uint32_t* pits_pairs, const uint32_t* pits_aux,
                          const uint64_t* hills_off, const uint32_t* hills_pairs) {
    if (!ctx || !begin || !end || !median || !p10 || !alive || !pits_off || !hills_off) return RALA_HIP_EINVAL;
    if (ctx->n_reads == 0) return fail(ctx, RALA_HIP_EINVAL, "no reads set");
    if (ctx->n_ovl && !valid) return fail(ctx, RALA_HIP_EINVAL, "valid bits required");
    HIPCHECK(hipSetDevice(ctx->device));
    const uint64_t n = ctx->n_reads;
    ctx->tm = rala_hip_timings();
    ctx->overlaps.clear(); ctx->internals.clear();
    std::vector<uint32_t> np(n), nh(n);
    std::vector<uint32_t> slot(n, 0xFFFFFFFFu);
    std::vector<Interval> pool;
    for (uint64_t r = 0; r < n; ++r) {
        const uint64_t a = pits_off[r + 1] - pits_off[r], b = hills_off[r + 1] - hills_off[r];
        if (a > 0xFFFFFFFFull || b > 0xFFFFFFFFull) return fail(ctx, RALA_HIP_EINVAL, "too many intervals on a read");
        np[r] = (uint32_t)a; nh[r] = (uint32_t)b;
        if (a + b == 0) continue;
        slot[r] = (uint32_t)pool.size();
        for (uint64_t k = pits_off[r]; k < pits_off[r + 1]; ++k) {
            Interval iv; iv.first = pits_pairs[2 * k]; iv.second = pits_pairs[2 * k + 1]; iv.aux = pits_aux[k];
            pool.push_back(iv);
        }
        for (uint64_t k = hills_off[r]; k < hills_off[r + 1]; ++k) {
            Interval iv; iv.first = hills_pairs[2 * k]; iv.second = hills_pairs[2 * k + 1]; iv.aux = 0;
            pool.push_back(iv);
        }
    }
    if (pool.size() > ctx->pool_cap) {
        ctx->pool_cap = (uint32_t)pool.size() + 1024;
        HIPCHECK(ctx->d_pool.ensure(ctx->pool_cap));
    }
    const uint32_t used = (uint32_t)pool.size();
    uint32_t small[8] = {used, 0, 0, 0, 0, 0, 0, 0};
    HIPCHECK(hipMemcpy(ctx->d_small.p, small, sizeof(small), hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(ctx->d_begin.p, begin, n * 4, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(ctx->d_end.p, end, n * 4, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(ctx->d_median.p, median, n * 2, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(ctx->d_p10.p, p10, n * 2, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(ctx->d_alive.p, alive, n, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(ctx->d_n_pits.p, np.data(), n * 4, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(ctx->d_n_hills.p, nh.data(), n * 4, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(ctx->d_iv_slot.p, slot.data(), n * 4, hipMemcpyHostToDevice));
    if (used) HIPCHECK(hipMemcpy(ctx->d_pool.p, pool.data(), (size_t)used * sizeof(Interval), hipMemcpyHostToDevice));
    if (ctx->n_ovl) HIPCHECK(hipMemcpy(ctx->d_valid.p, valid, ctx->n_ovl, hipMemcpyHostToDevice));
    const int rc = download_read_state(ctx);
    if (rc != RALA_HIP_OK) return rc;
    ctx->n_prefiltered = 0;
    for (uint64_t r = 0; r < n; ++r) if (!ctx->h_alive[r]) ++ctx->n_prefiltered;
    ctx->initialized = true;
    ctx->valid_ready = true;
    ctx->constructed = false;
    ctx->piles_resident = ctx->ev_ready = false;
    ctx->tail_on_device = ctx->host_stale = ctx->marks_on_device = false;
    if (ctx->n_prefiltered == n) return fail(ctx, RALA_HIP_EFILTERED, "filtered all sequences");
    return RALA_HIP_OK;
}

int rala_hip_construct(rala_hip_ctx* ctx, const rala_hip_overlaps* sens, uint64_t n_sens) {
    if (!ctx) return RALA_HIP_EINVAL;
    if (!ctx->initialized) return fail(ctx, RALA_HIP_EINVAL, "rala_hip_initialize must succeed first");
    if (ctx->tuple_mode || !ctx->inputs_set) return fail(ctx, RALA_HIP_EINVAL, "construct needs the overlaps (rala_hip_set_overlaps)");
    if (ctx->constructed) return fail(ctx, RALA_HIP_EINVAL, "object already constructed");
    if (sens != nullptr && n_sens != 0 && !ctx->piles_resident) {
        return fail(ctx, RALA_HIP_EINVAL, "the sensitive pass needs the piles on this context");
    }
    ctx->have_repeats = false;
    HIPCHECK(hipSetDevice(ctx->device));
    { const int rcu = flush_upload(ctx); if (rcu != RALA_HIP_OK) return rcu; }
    hipStream_t s = ctx->stream;
    const uint32_t n_reads = (uint32_t)ctx->n_reads;
    const uint64_t N = ctx->n_ovl;
    const ReadState rs = read_state(ctx);

    {
        const int rc_p2 = pass2(ctx, nullptr);
        if (rc_p2 != RALA_HIP_OK) return rc_p2;
    }
    const uint32_t M = ctx->t_n0 + ctx->t_n1;
    const uint32_t n_surv[2] = {ctx->t_n0, ctx->t_n1};
    const bool with_sens = sens != nullptr && n_sens != 0;
    ctx->tail_on_device = false;
    ctx->host_stale = false;
    if (ctx->use_gpu_tail) {
        const double t0 = now_ms();
        const int rc5 = with_sens ? gpu_tail_part_a(ctx) : gpu_tail_run(ctx);
        if (rc5 != RALA_HIP_OK) return rc5;
        HIPCHECK(hipEventElapsedTime(&ctx->tm.classify_ms, ctx->ev[4], ctx->ev[5]));
        HIPCHECK(hipEventElapsedTime(&ctx->tm.death_ms, ctx->ev[5], ctx->ev[6]));
        HIPCHECK(hipEventElapsedTime(&ctx->tm.finish_ms, ctx->ev[6], ctx->ev[7]));
        if (with_sens) {
            // Graph::preprocess(sensitive overlaps) (graph.cpp:882-1054) works on the lists the
            // chimera stage leaves: bring them to the host, annotate repeats (kernels + host
            // orchestration), rebuild the graph from what is left
            const double t1 = now_ms();
            const int rc6 = repeats_after_tail(ctx, ctx, nullptr, sens, n_sens);
            if (rc6 != RALA_HIP_OK) return rc6;
            ctx->tm.repeats_ms = (float)(now_ms() - t1);
        }
        ctx->tm.tail_host_ms = (float)(now_ms() - t0);
        ctx->constructed = true;
        return RALA_HIP_OK;
    }
    // host tail: lists to the host
    std::vector<HostOvl>* lists[2] = {&ctx->overlaps, &ctx->internals};
    for (int f = 0; f < 8; ++f) {
        HIPCHECK(ctx->p_surv_u32[f].ensure(M));
        if (M) HIPCHECK(hipMemcpyAsync(ctx->p_surv_u32[f].p, ctx->d_surv_u32[f].p, (size_t)M * 4, hipMemcpyDeviceToHost, s));
    }
    for (int f = 0; f < 2; ++f) {
        HIPCHECK(ctx->p_surv_u8[f].ensure(M));
        if (M) HIPCHECK(hipMemcpyAsync(ctx->p_surv_u8[f].p, ctx->d_surv_u8[f].p, M, hipMemcpyDeviceToHost, s));
    }
    HIPCHECK(stream_sync(ctx, s));
    for (int k = 0; k < 2; ++k) {
        const uint32_t m = n_surv[k];
        const uint32_t off = k == 0 ? 0 : n_surv[0];
        lists[k]->clear();
        // promoted internals are appended to the overlaps later: room for them up front
        lists[k]->reserve(k == 0 ? (size_t)M : (size_t)m);
        lists[k]->resize(m);
        std::vector<HostOvl>& dst = *lists[k];
        uint32_t* hp[8];
        for (int f = 0; f < 8; ++f) hp[f] = ctx->p_surv_u32[f].p + off;
        const uint8_t* hs = ctx->p_surv_u8[0].p + off;
        const uint8_t* ht = ctx->p_surv_u8[1].p + off;
        ctx->pool->chunks(m, [&](unsigned, size_t b0, size_t e0) {
            for (size_t i = b0; i < e0; ++i) {
                HostOvl& o = dst[i];
                o.src = hp[0][i]; o.a = hp[1][i]; o.b = hp[2][i];
                o.c.a_begin = hp[3][i]; o.c.a_end = hp[4][i]; o.c.b_begin = hp[5][i]; o.c.b_end = hp[6][i];
                o.c.length = hp[7][i];
                o.strand = hs[i]; o.dead = 0; o.type = ht[i];
            }
        });
    }
    // refreshed read state: liveness after the scan, hill counters in the pool
    int rc = download_read_state(ctx);
    if (rc != RALA_HIP_OK) return rc;
    HIPCHECK(hipEventElapsedTime(&ctx->tm.classify_ms, ctx->ev[4], ctx->ev[5]));
    HIPCHECK(hipEventElapsedTime(&ctx->tm.death_ms, ctx->ev[5], ctx->ev[6]));
    HIPCHECK(hipEventElapsedTime(&ctx->tm.finish_ms, ctx->ev[6], ctx->ev[7]));

    // ---- tail on the host (cross-check path; also used with sensitive overlaps) ----
    const double t0 = now_ms();
    Trace trc;
    {
        const int rc4 = preprocess_chimeras(ctx);
        if (rc4 != RALA_HIP_OK) return rc4;
    }
    if (sens != nullptr && n_sens != 0) {
        const double t1 = now_ms();
        const int rc3 = preprocess_repeats(ctx, ctx, nullptr, sens, n_sens, false);
        if (rc3 != RALA_HIP_OK) return rc3;
        ctx->tm.repeats_ms = (float)(now_ms() - t1);
    }
    trc("preprocess total");
    build_graph(ctx);
    trc("build_graph", ctx->e_src.size());
    ctx->tm.tail_host_ms = (float)(now_ms() - t0);

    // push the final valid regions / liveness back, re-zero the piles outside them
    HIPCHECK(hipMemcpyAsync(ctx->d_begin.p, ctx->h_begin.data(), (size_t)n_reads * 4, hipMemcpyHostToDevice, s));
    HIPCHECK(hipMemcpyAsync(ctx->d_end.p, ctx->h_end.data(), (size_t)n_reads * 4, hipMemcpyHostToDevice, s));
    HIPCHECK(hipMemcpyAsync(ctx->d_alive.p, ctx->h_alive.data(), (size_t)n_reads, hipMemcpyHostToDevice, s));
    HIPCHECK(stream_sync(ctx, s));
    ctx->constructed = true;
    return RALA_HIP_OK;
}

int rala_hip_find_repetitive_hills(rala_hip_ctx* ctx, uint64_t read, uint32_t begin, uint32_t end, uint16_t median,
                                   uint16_t p10, uint16_t dataset_median) {
    if (!ctx) return RALA_HIP_EINVAL;
    if (!ctx->initialized || read >= ctx->n_reads) return fail(ctx, RALA_HIP_EINVAL, "bad read / not initialized");
    if (!ctx->piles_resident) return fail(ctx, RALA_HIP_EINVAL, "the piles live on another context");
    HIPCHECK(hipSetDevice(ctx->device));
    { const int rcm = materialize_host(ctx); if (rcm != RALA_HIP_OK) return rcm; }
    if (!ctx->h_alive[read]) return fail(ctx, RALA_HIP_EINVAL, "the read was filtered");
    if (begin > end || end > ctx->h_read_len[read]) return fail(ctx, RALA_HIP_EINVAL, "bad valid region");
    ctx->h_begin[read] = begin; ctx->h_end[read] = end; ctx->h_median[read] = median; ctx->h_p10[read] = p10;
    const uint64_t n = ctx->n_reads;
    hipStream_t s = ctx->stream;
    HIPCHECK(ctx->d_dataset_median.ensure(n));
    HIPCHECK(ctx->d_n_rep.ensure(n));
    HIPCHECK(ctx->d_rep_slot.ensure(n));
    ctx->rep_pool_cap = std::max(ctx->rep_pool_cap, ctx->pool_cap_first);
    HIPCHECK(ctx->d_rep_pool.ensure(ctx->rep_pool_cap));
    if (!ctx->have_repeats) {
        HIPCHECK(hipMemsetAsync(ctx->d_n_rep.p, 0, n * 4, s));
        HIPCHECK(hipMemsetAsync(ctx->d_small.p + 6, 0, 8, s));
        ctx->h_n_rep.assign(n, 0); ctx->h_rep_slot.assign(n, 0); ctx->h_rep_pool.clear();
    }
    // the valid region as it stands (the host mirrors are authoritative after a host tail)
    HIPCHECK(hipMemcpyAsync(ctx->d_begin.p + read, &ctx->h_begin[read], 4, hipMemcpyHostToDevice, s));
    HIPCHECK(hipMemcpyAsync(ctx->d_end.p + read, &ctx->h_end[read], 4, hipMemcpyHostToDevice, s));
    HIPCHECK(hipMemcpyAsync(ctx->d_median.p + read, &ctx->h_median[read], 2, hipMemcpyHostToDevice, s));
    HIPCHECK(hipMemcpyAsync(ctx->d_p10.p + read, &ctx->h_p10[read], 2, hipMemcpyHostToDevice, s));
    HIPCHECK(hipMemcpyAsync(ctx->d_dataset_median.p + read, &dataset_median, 2, hipMemcpyHostToDevice, s));
    HIPCHECK(hipMemsetAsync(ctx->d_small.p + 7, 0, 4, s));
    HIPCHECK(stream_sync(ctx, s));
    RepeatArgs a;
    a.read_len = ctx->d_read_len.p; a.pile_off = ctx->d_pile_off.p; a.pile = ctx->d_pile.p;
    a.ev_off = ctx->d_ev_off.p; a.ev = ctx->d_ev.p;
    a.begin = ctx->d_begin.p; a.end = ctx->d_end.p; a.median = ctx->d_median.p; a.p10 = ctx->d_p10.p;
    a.dataset_median = ctx->d_dataset_median.p; a.n_rep = ctx->d_n_rep.p; a.rep_slot = ctx->d_rep_slot.p;
    a.pool = ctx->d_rep_pool.p; a.pool_count = ctx->d_small.p + 6; a.pool_cap = ctx->rep_pool_cap;
    a.error = ctx->d_small.p + 7;
    a.order = nullptr; a.n_items = 0; a.lw = 0; a.slab = nullptr;
    const std::vector<uint32_t> one(1, (uint32_t)read);
    uint32_t small[8];
    for (;;) {
        uint32_t used_before = 0;
        HIPCHECK(hipMemcpy(&used_before, ctx->d_small.p + 6, 4, hipMemcpyDeviceToHost));
        const int rc = run_repeats_kernel(ctx, a, one, 2);
        if (rc != RALA_HIP_OK) return rc;
        HIPCHECK(hipMemcpy(small, ctx->d_small.p, sizeof(small), hipMemcpyDeviceToHost));
        if (!(small[7] & kErrPoolCapacity)) break;
        // the pool (it keeps the hills of earlier calls) is too small for this read's: a larger one, the call once more
        const uint64_t need = (uint64_t)small[6] + small[6] / 8 + 1024;
        if (small[6] <= ctx->rep_pool_cap || need > 0xFFFFFFF0ull) return fail(ctx, RALA_HIP_ENOMEM, "repeat-hill pool beyond 2^32 entries");
        rala_hip::DevBuf<Interval> larger;
        HIPCHECK(larger.ensure(need));
        if (used_before) HIPCHECK(hipMemcpy(larger.p, ctx->d_rep_pool.p, (size_t)used_before * sizeof(Interval), hipMemcpyDeviceToDevice));
        std::swap(larger.p, ctx->d_rep_pool.p);
        std::swap(larger.n, ctx->d_rep_pool.n);
        ctx->rep_pool_cap = (uint32_t)need;
        a.pool = ctx->d_rep_pool.p; a.pool_cap = ctx->rep_pool_cap;
        const uint32_t reset[2] = {used_before, 0};
        HIPCHECK(hipMemcpy(ctx->d_small.p + 6, reset, 8, hipMemcpyHostToDevice));
    }
    if (small[7] & (kErrRegionCapacity | kErrRawCapacity)) return fail(ctx, RALA_HIP_EDEVICE, "slope-region list overflow (repeat hills)");
    ctx->h_n_rep.resize(n); ctx->h_rep_slot.resize(n);
    HIPCHECK(hipMemcpy(&ctx->h_n_rep[read], ctx->d_n_rep.p + read, 4, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(&ctx->h_rep_slot[read], ctx->d_rep_slot.p + read, 4, hipMemcpyDeviceToHost));
    ctx->h_rep_pool.resize(small[6]);
    if (small[6]) HIPCHECK(hipMemcpy(ctx->h_rep_pool.data(), ctx->d_rep_pool.p, (size_t)small[6] * sizeof(Interval), hipMemcpyDeviceToHost));
    ctx->n_rep_hills = small[6];
    ctx->rep_host_stale = false;
    ctx->have_repeats = true;
    return RALA_HIP_OK;
}

int rala_hip_remove_transitive_edges(rala_hip_ctx* ctx, uint32_t* n_pairs) {
    if (!ctx || !n_pairs) return RALA_HIP_EINVAL;
    if (!ctx->constructed) return fail(ctx, RALA_HIP_EINVAL, "rala_hip_construct must succeed first");
    HIPCHECK(hipSetDevice(ctx->device));
    if (ctx->tail_on_device) {
        const int rc = tr_mark_device(ctx, ctx->t_n_nodes, ctx->t_n_edges, ctx->d_e[0].p, ctx->d_e[1].p, ctx->d_e[2].p,
                                      n_pairs);
        ctx->marks_on_device = rc == RALA_HIP_OK;
        return rc;
    }
    const uint32_t ne = (uint32_t)ctx->e_src.size();
    ctx->e_mark.assign(ne, 0);
    return tr_mark_impl(ctx, (uint32_t)ctx->node_read.size(), ne, ctx->e_src.data(), ctx->e_dst.data(),
                        ctx->e_len.data(), ctx->e_mark.data(), n_pairs);
}

int rala_hip_tr_mark(rala_hip_ctx* ctx, uint32_t n_nodes, uint32_t n_edges, const uint32_t* src,
                     const uint32_t* dst, const uint32_t* len, uint8_t* marks, uint32_t* n_pairs) {
    if (!ctx || !n_pairs || (n_edges && (!src || !dst || !len || !marks))) return RALA_HIP_EINVAL;
    HIPCHECK(hipSetDevice(ctx->device));
    return tr_mark_impl(ctx, n_nodes, n_edges, src, dst, len, marks, n_pairs);
}

int rala_hip_layout(rala_hip_ctx* ctx, uint32_t n, double* x, double* y, const uint32_t* adj_off, const uint32_t* adj,
                    uint32_t iterations, double k, double t, double dt) {
    if (!ctx || (n && (!x || !y || !adj_off))) return RALA_HIP_EINVAL;
    if (n == 0) return RALA_HIP_OK;
    const uint32_t n_adj = adj_off[n];
    if (n_adj && !adj) return RALA_HIP_EINVAL;
    HIPCHECK(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    for (int b = 0; b < 4; ++b) HIPCHECK(ctx->d_layout[b].ensure(n));
    HIPCHECK(ctx->d_layout_adj[0].ensure(n + 1)); HIPCHECK(ctx->d_layout_adj[1].ensure(n_adj + 1));
    HIPCHECK(hipMemcpyAsync(ctx->d_layout[0].p, x, (size_t)n * 8, hipMemcpyHostToDevice, s));
    HIPCHECK(hipMemcpyAsync(ctx->d_layout[1].p, y, (size_t)n * 8, hipMemcpyHostToDevice, s));
    HIPCHECK(hipMemcpyAsync(ctx->d_layout_adj[0].p, adj_off, (size_t)(n + 1) * 4, hipMemcpyHostToDevice, s));
    if (n_adj) HIPCHECK(hipMemcpyAsync(ctx->d_layout_adj[1].p, adj, (size_t)n_adj * 4, hipMemcpyHostToDevice, s));
    int cur = 0;
    for (uint32_t it = 0; it < iterations; ++it) {
        launch_layout_step(n, ctx->d_layout[2 * cur].p, ctx->d_layout[2 * cur + 1].p, ctx->d_layout[2 * (cur ^ 1)].p,
                           ctx->d_layout[2 * (cur ^ 1) + 1].p, ctx->d_layout_adj[0].p, ctx->d_layout_adj[1].p, k, t, s);
        cur ^= 1;
        t -= dt;
    }
    HIPCHECK(hipMemcpyAsync(x, ctx->d_layout[2 * cur].p, (size_t)n * 8, hipMemcpyDeviceToHost, s));
    HIPCHECK(hipMemcpyAsync(y, ctx->d_layout[2 * cur + 1].p, (size_t)n * 8, hipMemcpyDeviceToHost, s));
    HIPCHECK(stream_sync(ctx, s));
    HIPCHECK(hipGetLastError());
    return RALA_HIP_OK;
}

// ---- results ---------------------------------------------------------------------------
int rala_hip_get_valid(rala_hip_ctx* ctx, uint8_t* valid) {
    if (!ctx || !valid) return RALA_HIP_EINVAL;
    if (!ctx->valid_ready) return fail(ctx, RALA_HIP_EINVAL, "run rala_hip_dedupe or rala_hip_initialize first");
    HIPCHECK(hipSetDevice(ctx->device));
    if (ctx->n_ovl) HIPCHECK(hipMemcpy(valid, ctx->d_valid.p, ctx->n_ovl, hipMemcpyDeviceToHost));
    return RALA_HIP_OK;
}

int rala_hip_get_piles(rala_hip_ctx* ctx, uint32_t* begin, uint32_t* end, uint16_t* median, uint16_t* p10,
                       uint8_t* alive) {
    if (!ctx) return RALA_HIP_EINVAL;
    if (!ctx->initialized) return fail(ctx, RALA_HIP_EINVAL, "not initialized");
    { const int rcm = materialize_host(ctx); if (rcm != RALA_HIP_OK) return rcm; }
    const uint64_t n = ctx->n_reads;
    for (uint64_t r = 0; r < n; ++r) {
        const bool a = ctx->h_alive[r];
        if (begin) begin[r] = a ? ctx->h_begin[r] : 0;
        if (end) end[r] = a ? ctx->h_end[r] : 0;
        if (median) median[r] = a ? ctx->h_median[r] : 0;
        if (p10) p10[r] = a ? ctx->h_p10[r] : 0;
        if (alive) alive[r] = a;
    }
    return RALA_HIP_OK;
}

int rala_hip_get_pile_data(rala_hip_ctx* ctx, uint64_t read, uint16_t* data) {
    if (!ctx || !data) return RALA_HIP_EINVAL;
    if (!ctx->initialized || read >= ctx->n_reads) return fail(ctx, RALA_HIP_EINVAL, "bad read / not initialized");
    if (!ctx->piles_resident) return fail(ctx, RALA_HIP_EINVAL, "piles live on the owning rank's context");
    HIPCHECK(hipSetDevice(ctx->device));
    { const int rcm = materialize_host(ctx); if (rcm != RALA_HIP_OK) return rcm; }
    const uint32_t n = ctx->h_read_len[read];
    HIPCHECK(ctx->d_pile.copy_to_host(data, ctx->h_pile_off[read], n));
    // Pile::shrink zeroes outside the current valid region (pile.cpp:311-318); the
    // host tail narrows regions after the pile was written
    if (ctx->h_alive[read]) {
        const uint32_t B = ctx->h_begin[read], E = ctx->h_end[read];
        for (uint32_t j = 0; j < B && j < n; ++j) data[j] = 0;
        for (uint32_t j = E; j < n; ++j) data[j] = 0;
    }
    return RALA_HIP_OK;
}

int rala_hip_get_pile_row_digests(rala_hip_ctx* ctx, uint64_t* fnv, uint64_t* inside, uint64_t* outside) {
    if (!ctx) return RALA_HIP_EINVAL;
    if (!ctx->initialized) return fail(ctx, RALA_HIP_EINVAL, "not initialized");
    if (!ctx->piles_resident) return fail(ctx, RALA_HIP_EINVAL, "piles live on the owning rank's context");
    HIPCHECK(hipSetDevice(ctx->device));
    { const int rcm = materialize_host(ctx); if (rcm != RALA_HIP_OK) return rcm; }
    return pile_row_digests(ctx, ctx->h_begin.data(), ctx->h_end.data(), ctx->h_alive.data(), fnv, inside, outside);
}

int rala_hip_get_intervals(rala_hip_ctx* ctx, int kind, uint64_t* offsets, uint32_t* pairs, uint32_t* aux) {
    if (!ctx || !offsets) return RALA_HIP_EINVAL;
    if (!ctx->initialized) return fail(ctx, RALA_HIP_EINVAL, "not initialized");
    if (kind < 0 || kind > 2) return fail(ctx, RALA_HIP_EINVAL, "bad interval kind");
    { const int rcm = materialize_host(ctx); if (rcm != RALA_HIP_OK) return rcm; }
    const uint64_t n = ctx->n_reads;
    uint64_t off = 0;
    for (uint64_t r = 0; r < n; ++r) {
        offsets[r] = off;
        if (!ctx->h_alive[r]) continue;
        if (kind == 2) {
            if (!ctx->have_repeats || ctx->h_n_rep[r] == 0) continue;
            const Interval* iv = ctx->h_rep_pool.data() + ctx->h_rep_slot[r];
            const uint32_t c2 = ctx->h_n_rep[r];
            if (pairs) {
                for (uint32_t k = 0; k < c2; ++k) {
                    pairs[2 * (off + k)] = iv[k].first;
                    pairs[2 * (off + k) + 1] = iv[k].second;
                    if (aux) aux[off + k] = iv[k].aux;
                }
            }
            off += c2;
            continue;
        }
        const uint32_t cnt = kind == 0 ? ctx->h_n_pits[r] : ctx->h_n_hills[r];
        if (cnt == 0) continue;
        // hills follow the pits found by the pile kernel; the tail only ever drops pits
        // from the front segment in place, so the initial pit count is what the
        // device wrote into the pool layout: recover it from the device-side array
        const Interval* base = ctx->h_pool.data() + ctx->h_slot[r];
        const Interval* iv = base;
        if (kind == 1) iv = base + ctx->h_n_pits[r];
        if (pairs) {
            for (uint32_t k = 0; k < cnt; ++k) {
                pairs[2 * (off + k)] = iv[k].first;
                pairs[2 * (off + k) + 1] = iv[k].second;
                if (aux) aux[off + k] = iv[k].aux;
            }
        }
        off += cnt;
    }
    offsets[n] = off;
    return RALA_HIP_OK;
}

int rala_hip_get_overlaps(rala_hip_ctx* ctx, int which, uint64_t* n, uint32_t* src_index, uint32_t* a_begin,
                          uint32_t* a_end, uint32_t* b_begin, uint32_t* b_end, uint32_t* length, uint8_t* type) {
    if (!ctx || !n) return RALA_HIP_EINVAL;
    if (!ctx->constructed) return fail(ctx, RALA_HIP_EINVAL, "not constructed");
    { const int rcm = materialize_host(ctx); if (rcm != RALA_HIP_OK) return rcm; }
    const std::vector<HostOvl>& v = which == 0 ? ctx->overlaps : ctx->internals;
    *n = v.size();
    if (!src_index) return RALA_HIP_OK;
    for (size_t k = 0; k < v.size(); ++k) {
        src_index[k] = v[k].src;
        if (a_begin) a_begin[k] = v[k].c.a_begin;
        if (a_end) a_end[k] = v[k].c.a_end;
        if (b_begin) b_begin[k] = v[k].c.b_begin;
        if (b_end) b_end[k] = v[k].c.b_end;
        if (length) length[k] = v[k].c.length;
        if (type) {
            // internals can outlive their piles (reference graph.cpp:849-867 never re-checks them)
            const bool live = ctx->h_alive[v[k].a] && ctx->h_alive[v[k].b];
            type[k] = live ? (uint8_t)host_type(ctx, v[k]) : (uint8_t)255;
        }
    }
    return RALA_HIP_OK;
}

int rala_hip_get_graph_size(rala_hip_ctx* ctx, uint64_t* n_nodes, uint64_t* n_edges) {
    if (!ctx || !n_nodes || !n_edges) return RALA_HIP_EINVAL;
    if (!ctx->constructed) return fail(ctx, RALA_HIP_EINVAL, "not constructed");
    if (ctx->tail_on_device) {
        *n_nodes = ctx->t_n_nodes;
        *n_edges = ctx->t_n_edges;
        return RALA_HIP_OK;
    }
    *n_nodes = ctx->node_read.size();
    *n_edges = ctx->e_src.size();
    return RALA_HIP_OK;
}

int rala_hip_get_graph(rala_hip_ctx* ctx, uint32_t* node_read, uint32_t* src, uint32_t* dst, uint32_t* len,
                       uint8_t* marks) {
    if (!ctx) return RALA_HIP_EINVAL;
    if (!ctx->constructed) return fail(ctx, RALA_HIP_EINVAL, "not constructed");
    { const int rcm = materialize_host(ctx); if (rcm != RALA_HIP_OK) return rcm; }
    if (ctx->marks_on_device) {
        HIPCHECK(hipSetDevice(ctx->device));
        ctx->e_mark.resize(ctx->t_n_edges);
        if (ctx->t_n_edges) HIPCHECK(hipMemcpy(ctx->e_mark.data(), ctx->d_tr_marks.p, ctx->t_n_edges, hipMemcpyDeviceToHost));
        ctx->marks_on_device = false;
    }
    if (node_read) memcpy(node_read, ctx->node_read.data(), ctx->node_read.size() * 4);
    const size_t ne = ctx->e_src.size();
    if (src) memcpy(src, ctx->e_src.data(), ne * 4);
    if (dst) memcpy(dst, ctx->e_dst.data(), ne * 4);
    if (len) memcpy(len, ctx->e_len.data(), ne * 4);
    if (marks) memcpy(marks, ctx->e_mark.data(), ne);
    return RALA_HIP_OK;
}

int rala_hip_get_timings(rala_hip_ctx* ctx, rala_hip_timings* out) {
    if (!ctx || !out) return RALA_HIP_EINVAL;
    *out = ctx->tm;
    out->total_ms = ctx->tm.dedupe_ms + ctx->tm.bucket_ms + ctx->tm.pile_ms + ctx->tm.classify_ms + ctx->tm.death_ms +
                    ctx->tm.finish_ms + ctx->tm.tail_host_ms + ctx->tm.tr_ms;
    return RALA_HIP_OK;
}

int rala_hip_get_num_prefiltered(rala_hip_ctx* ctx, uint64_t* n) {
    if (!ctx || !n) return RALA_HIP_EINVAL;
    *n = ctx->n_prefiltered;
    return RALA_HIP_OK;
}

}  // extern "C"
