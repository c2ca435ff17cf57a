// Host side of the device tokeniser (ingest_kernels.hip): the file's text to device memory - reader threads with pinned
// staging blocks of their own, every block's copy queued behind its read, so that the disk / page cache and PCIe work at the
// same time - then count, scan, parse, and the columns become the context's overlaps (rala_hip_set_overlaps, device memory).
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <stdio.h>
#include <stdlib.h>

#include <atomic>
#include <chrono>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "kernels.h"
#include "name_table.h"
#include "stages.h"

using namespace rala_hip;

namespace {

constexpr size_t kBlockBytes = 32u << 20;       // one staging block
constexpr uint32_t kMaxReaders = 8;        // (measured at C3: 4 readers 107 ms, 8: 70 - 73 ms, 12 - 16: 85 - 130 ms on the 16 CPUs a box allows)

// Pinned staging blocks are expensive to make (the pages are locked one by one) and cheap to keep: a pool of the
// process, two blocks per reader.
struct StagingPool {
    std::mutex m;
    std::vector<void*> free_blocks;
    // (never freed: at process exit the runtime may be gone before this object is, and the memory goes with the process)
    void* take() {
        {
            std::lock_guard<std::mutex> hold(m);
            if (!free_blocks.empty()) { void* p = free_blocks.back(); free_blocks.pop_back(); return p; }
        }
        void* p = nullptr;
        return hipHostMalloc(&p, kBlockBytes, hipHostMallocDefault) == hipSuccess ? p : nullptr;
    }
    void give(void* p) {
        std::lock_guard<std::mutex> hold(m);
        free_blocks.push_back(p);
    }
};
StagingPool& staging() {
    static StagingPool pool;
    return pool;
}

int ingest_fail(rala_hip_ctx* ctx, int code, const std::string& msg) {
    ctx->err = msg;
    return code;
}

#define INGEST_CHECK(call)                                                                                  \
    do {                                                                                                    \
        const hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess) {                                                                             \
            return ingest_fail(ctx, e_ == hipErrorOutOfMemory ? RALA_HIP_ENOMEM : RALA_HIP_EDEVICE,         \
                               std::string(#call) + ": " + hipGetErrorString(e_));                          \
        }                                                                                                   \
    } while (0)

double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

// The lines that START in bytes [lo, hi) of the file (hi = ~0: to its end), tokenised on the device into `T`'s columns (room for
// extra_rows more rows than the range holds: a rank of a sharded run receives the heads of its neighbours' first runs).  The
// text is shipped by reader threads - two pinned staging blocks each, a block's copy queued behind its read - with what the last
// lines' first eleven columns may need behind hi.
int rala_hip::paf_tokenise_range(rala_hip_ctx* ctx, const char* path, uint64_t lo, uint64_t hi, bool check_lengths, uint32_t threads,
                                 size_t extra_rows, const PafTarget& T, PafRange* out) {
    if (!ctx || !path || !out) return RALA_HIP_EINVAL;
    *out = PafRange();
    if (ctx->n_reads == 0) return ingest_fail(ctx, RALA_HIP_EINVAL, "no reads set");
    if (ctx->n_name_buckets == 0 && !T.mhap) return ingest_fail(ctx, RALA_HIP_EINVAL, "no name table set (rala_hip_set_name_table)");
    INGEST_CHECK(hipSetDevice(ctx->device));
    struct Fd {                                     // (closed on every way out - ADVICE round 4)
        int fd = -1;
        ~Fd() { if (fd >= 0) close(fd); }
    } file;
    file.fd = open(path, O_RDONLY);
    const int fd = file.fd;
    if (fd < 0) return ingest_fail(ctx, RALA_HIP_EINVAL, std::string("cannot open ") + path);
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) return ingest_fail(ctx, RALA_HIP_ENOTAFILE, std::string("not a regular file: ") + path);
    const uint64_t file_n = (uint64_t)st.st_size;
    hi = std::min(hi, file_n);
    lo = std::min(lo, hi);
    const uint64_t n = hi - lo;                                             // bytes whose line starts are ours
    const uint64_t n_avail = std::min<uint64_t>(file_n - lo, n + paf_halo_bytes());
    bool first_is_start = lo == 0;
    if (lo) {
        char before = 0;
        if (pread(fd, &before, 1, (off_t)(lo - 1)) != 1) return ingest_fail(ctx, RALA_HIP_EDEVICE, std::string("reading ") + path + " failed");
        first_is_start = before == '\n';
    }
    const uint32_t chunk = paf_chunk_bytes();
    const uint64_t n_chunks = (n + chunk - 1) / chunk;
    if (n_chunks >= 0xFFFFFFF0ull) return ingest_fail(ctx, RALA_HIP_ETOOLARGE, "file too large for 32-bit chunk ids");
    const uint64_t cap = n_chunks * chunk + 4096 + 64;
    hipStream_t s = ctx->stream;
    const double t0 = now_ms();
    if (ctx->d_paf_text.ensure(cap) != hipSuccess) return ingest_fail(ctx, RALA_HIP_ENOMEM, "device memory for the file's text");
    // what lies behind the text reads as newlines
    INGEST_CHECK(hipMemsetAsync(ctx->d_paf_text.p + n_avail, '\n', cap - n_avail, s));

    // ---- ship: reader threads, two pinned blocks each, a block's copy queued behind its read ----
    const uint64_t n_blocks = (n_avail + kBlockBytes - 1) / kBlockBytes;
    const uint32_t n_readers = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint32_t>(threads ? threads : 1, kMaxReaders), n_blocks));
    std::atomic<uint64_t> next(0);
    std::atomic<int> failed(0);
    std::vector<std::thread> readers;
    uint8_t* const text = ctx->d_paf_text.p;
    const int device = ctx->device;
    for (uint32_t t = 0; t < n_readers && n_blocks; ++t) {
        readers.emplace_back([&]() {
            if (hipSetDevice(device) != hipSuccess) { failed = 1; return; }
            hipStream_t cs = nullptr;
            hipEvent_t ev[2] = {nullptr, nullptr};
            void* blk[2] = {staging().take(), staging().take()};
            bool ok = blk[0] && blk[1] && hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) == hipSuccess &&
                      hipEventCreateWithFlags(&ev[0], hipEventDisableTiming) == hipSuccess &&
                      hipEventCreateWithFlags(&ev[1], hipEventDisableTiming) == hipSuccess;
            bool busy[2] = {false, false};
            for (int k = 0; ok && !failed; k ^= 1) {
                const uint64_t b = next.fetch_add(1);
                if (b >= n_blocks) break;
                if (busy[k]) ok = hipEventSynchronize(ev[k]) == hipSuccess;       // the block's last copy has left it
                const uint64_t off = b * kBlockBytes;
                const size_t len = (size_t)std::min<uint64_t>(kBlockBytes, n_avail - off);
                size_t got = 0;
                while (ok && got < len) {
                    const ssize_t r = pread(fd, (char*)blk[k] + got, len - got, (off_t)(lo + off + got));
                    if (r <= 0) { ok = false; break; }
                    got += (size_t)r;
                }
                ok = ok && hipMemcpyAsync(text + off, blk[k], len, hipMemcpyHostToDevice, cs) == hipSuccess &&
                     hipEventRecord(ev[k], cs) == hipSuccess;
                busy[k] = ok;
            }
            if (cs) ok = (hipStreamSynchronize(cs) == hipSuccess) && ok;
            if (!ok) failed = 1;
            for (int k = 0; k < 2; ++k) {
                if (ev[k]) (void)hipEventDestroy(ev[k]);
                if (blk[k]) staging().give(blk[k]);
            }
            if (cs) (void)hipStreamDestroy(cs);
        });
    }
    // While the readers work: the device memory the tokeniser will want.  hipMalloc of a few hundred megabytes takes a
    // millisecond and more, eight columns of them 4 - 9 ms - behind the copies that is free.  The number of records is not
    // known yet: a record has at least 23 bytes, files in the wild 60 - 150 per line; room for one per 32 bytes is made now
    // (twice what a synthetic file needs) and the exact count decides later whether that was enough.
    {
        const size_t guess = (size_t)(n / 32) + 1024 + extra_rows;
        bool ok = true;
        for (int k = 0; k < 7 && ok; ++k) ok = T.col[k]->ensure(guess) == hipSuccess;
        ok = ok && T.strand->ensure(guess) == hipSuccess && ctx->d_paf_bad.ensure(2) == hipSuccess;
        if (n_chunks) {
            ok = ok && ctx->d_paf_chunk[0].ensure(n_chunks + 2) == hipSuccess && ctx->d_paf_chunk[1].ensure(n_chunks + 2) == hipSuccess &&
                 ctx->d_scan_ws.ensure(scan_workspace_bytes(std::max<uint64_t>(std::max<uint64_t>(n_chunks, ctx->n_reads), ctx->n_ovl) + 2)) == hipSuccess;
        }
        if (!ok) failed = 2;
    }
    for (auto& th : readers) th.join();
    if (failed == 2) return ingest_fail(ctx, RALA_HIP_ENOMEM, "device memory for the overlap columns");
    if (failed) return ingest_fail(ctx, RALA_HIP_EDEVICE, std::string("reading / copying ") + path + " failed");
    INGEST_CHECK(hipStreamSynchronize(s));
    const double t1 = now_ms();

    // ---- count, scan, parse ----
    const bool trace = getenv("RALA_HIP_TRACE") != nullptr;
    double tc = t1, tp0 = t1, tp1 = t1;
    uint32_t n_lines = 0;
    if (n_chunks) {
        launch_paf_count(text, n, first_is_start, ctx->d_paf_chunk[0].p, s);
        launch_exclusive_scan(ctx->d_paf_chunk[0].p, ctx->d_paf_chunk[1].p, n_chunks, ctx->d_scan_ws.p, s);
        INGEST_CHECK(hipMemcpyAsync(&n_lines, ctx->d_paf_chunk[1].p + n_chunks, 4, hipMemcpyDeviceToHost, s));
        INGEST_CHECK(hipStreamSynchronize(s));
    }
    tc = now_ms();
    if ((uint64_t)n_lines >= 0xFFFFFFF0ull / 2) return ingest_fail(ctx, RALA_HIP_ETOOLARGE, "too many overlaps for 32-bit bound offsets");
    for (int k = 0; k < 7; ++k) INGEST_CHECK(T.col[k]->ensure((size_t)n_lines + 1 + extra_rows));
    INGEST_CHECK(T.strand->ensure((size_t)n_lines + 1 + extra_rows));
    unsigned long long bad = ~0ull;
    uint32_t flags = 0;
    if (n_lines) {
        tp0 = now_ms();
        INGEST_CHECK(hipMemsetAsync(ctx->d_paf_bad.p, 0xFF, 8, s));
        INGEST_CHECK(hipMemsetAsync(ctx->d_paf_bad.p + 1, 0, 8, s));
        PafColumns cols;
        cols.a_id = T.col[0]->p; cols.b_id = T.col[1]->p; cols.a_begin = T.col[2]->p; cols.a_end = T.col[3]->p;
        cols.b_begin = T.col[4]->p; cols.b_end = T.col[5]->p; cols.length = T.col[6]->p; cols.strand = T.strand->p;
        launch_paf_parse(text, n, n_avail, first_is_start, ctx->d_paf_chunk[1].p, ctx->d_name_buckets.p, std::max<uint64_t>(ctx->n_name_buckets, 1),
                         (const char*)ctx->d_name_arena.p, ctx->d_read_len.p, (uint32_t)ctx->n_reads, check_lengths, cols,
                         (uint32_t*)(ctx->d_paf_bad.p + 1), ctx->d_paf_bad.p, s, T.mhap);
        unsigned long long back[2] = {0, 0};
        INGEST_CHECK(hipMemcpyAsync(back, ctx->d_paf_bad.p, 16, hipMemcpyDeviceToHost, s));
        INGEST_CHECK(hipStreamSynchronize(s));
        INGEST_CHECK(hipGetLastError());
        bad = back[0];
        flags = (uint32_t)back[1];
        tp1 = now_ms();
    }
    ctx->d_paf_text.release();                  // (the text is as large as the file: not kept)
    const double t2 = now_ms();
    ctx->ingest_tm.ship_ms = (float)(t1 - t0);
    ctx->ingest_tm.tokenize_ms = (float)(t2 - t1);
    ctx->ingest_tm.bytes = n;
    ctx->ingest_tm.lines = n_lines;
    if (trace) {
        fprintf(stderr, "[trace] device ingest: %.2f GB of text shipped in %.1f ms by %u readers, %u lines tokenised in %.2f ms (count + scan %.2f, "
                "columns' memory %.2f, parse %.2f, the text's memory back %.2f; flags %u)\n",
                n / 1e9, t1 - t0, n_readers, n_lines, t2 - t1, tc - t1, tp0 - tc, tp1 - tp0, t2 - tp1, flags);
    }
    out->n_lines = n_lines;
    out->first_bad = bad;
    out->flags = flags;
    out->file_bytes = file_n;
    return RALA_HIP_OK;
}

extern "C" {

int rala_hip_set_name_table(rala_hip_ctx* ctx, const void* buckets, uint64_t n_buckets, const char* arena, uint64_t arena_bytes) {
    if (!ctx || !buckets || n_buckets == 0 || (n_buckets & (n_buckets - 1)) != 0 || (!arena && arena_bytes)) return RALA_HIP_EINVAL;
    INGEST_CHECK(hipSetDevice(ctx->device));
    INGEST_CHECK(ctx->d_name_buckets.ensure(n_buckets * sizeof(NameBucket)));
    INGEST_CHECK(ctx->d_name_arena.ensure(arena_bytes + 16));
    INGEST_CHECK(hipMemcpy(ctx->d_name_buckets.p, buckets, n_buckets * sizeof(NameBucket), hipMemcpyHostToDevice));
    if (arena_bytes) INGEST_CHECK(hipMemcpy(ctx->d_name_arena.p, arena, arena_bytes, hipMemcpyHostToDevice));
    ctx->n_name_buckets = n_buckets;
    return RALA_HIP_OK;
}

static int set_overlaps_from_text(rala_hip_ctx* ctx, const char* path, bool mhap, int check_lengths, uint32_t threads,
                                  int64_t* length_error_read, int* irregular);

int rala_hip_set_overlaps_from_paf(rala_hip_ctx* ctx, const char* path, int check_lengths, uint32_t threads,
                                   int64_t* length_error_read, int* irregular) {
    return set_overlaps_from_text(ctx, path, false, check_lengths, threads, length_error_read, irregular);
}

int rala_hip_set_overlaps_from_mhap(rala_hip_ctx* ctx, const char* path, int check_lengths, uint32_t threads,
                                    int64_t* length_error_read, int* irregular) {
    return set_overlaps_from_text(ctx, path, true, check_lengths, threads, length_error_read, irregular);
}

static int set_overlaps_from_text(rala_hip_ctx* ctx, const char* path, bool mhap, int check_lengths, uint32_t threads,
                                  int64_t* length_error_read, int* irregular) {
    if (!ctx || !path || !length_error_read || !irregular) return RALA_HIP_EINVAL;
    *length_error_read = -1;
    *irregular = 0;
    // (the columns a successful call before this one left are about to be written over - whatever happens
    // here, the context has no overlaps until this call has set them)
    ctx->inputs_set = false;
    ctx->n_ovl = 0;
    ctx->ovl = OvlSoA();
    ctx->initialized = ctx->constructed = false;
    PafTarget T;
    for (int k = 0; k < 7; ++k) T.col[k] = &ctx->d_paf_col[k];
    T.strand = &ctx->d_paf_strand;
    T.mhap = mhap;
    PafRange R;
    // The file's text goes through device memory in WINDOWS (round 5; before, all of it had to fit at once): at most a
    // quarter of what is free (option ingest_window_bytes; the reference streams the file in chunks of 1 GiB,
    // graph.cpp:24, 329-365 - its run carry-over is not needed here, rows are tokenised independently of each other and
    // a window takes the lines that START in it).  One window - the usual case - tokenises straight into the columns.
    uint64_t window = (uint64_t)ctx->ingest_window_bytes;
    if (window == 0 && getenv("RALA_INGEST_WINDOW")) window = (uint64_t)atoll(getenv("RALA_INGEST_WINDOW"));      // (tests)
    if (window == 0) {
        size_t free_b = 0, total_b = 0;
        INGEST_CHECK(hipSetDevice(ctx->device));
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
        window = std::max<uint64_t>(256ull << 20, free_b / 4);
    }
    struct stat st_;
    const uint64_t file_n = stat(path, &st_) == 0 && S_ISREG(st_.st_mode) ? (uint64_t)st_.st_size : 0;
    if (file_n <= window) {
        const int rc = paf_tokenise_range(ctx, path, 0, ~0ull, check_lengths != 0, threads, 0, T, &R);
        if (rc != RALA_HIP_OK) return rc;
    } else {
        PafTarget W;
        for (int k = 0; k < 7; ++k) W.col[k] = &ctx->d_paf_win[k];
        W.strand = &ctx->d_paf_win_strand;
        W.mhap = mhap;
        uint64_t rows = 0;
        float ship = 0, tok = 0;
        for (uint64_t lo = 0; lo < file_n; lo += window) {
            PafRange part;
            const int rc = paf_tokenise_range(ctx, path, lo, lo + window, check_lengths != 0, threads, 0, W, &part);
            if (rc != RALA_HIP_OK) return rc;
            ship += ctx->ingest_tm.ship_ms; tok += ctx->ingest_tm.tokenize_ms;
            R.flags |= part.flags;
            if (part.flags) break;
            if (part.first_bad != ~0ull) {          // (windows come in file order: the first one with an offender holds the first offender)
                R.first_bad = (((part.first_bad >> 32) + rows) << 32) | (part.first_bad & 0xFFFFFFFFull);
                break;
            }
            if (rows + part.n_lines >= 0xFFFFFFF0ull / 2) return ingest_fail(ctx, RALA_HIP_ETOOLARGE, "too many overlaps for 32-bit bound offsets");
            for (int k = 0; k < 7; ++k) {
                if (ctx->d_paf_col[k].grow(rows, rows + part.n_lines + 1) != hipSuccess) return ingest_fail(ctx, RALA_HIP_ENOMEM, "device memory for the overlap columns");
                if (part.n_lines) INGEST_CHECK(hipMemcpy(ctx->d_paf_col[k].p + rows, ctx->d_paf_win[k].p, part.n_lines * 4, hipMemcpyDeviceToDevice));
            }
            if (ctx->d_paf_strand.grow(rows, rows + part.n_lines + 1) != hipSuccess) return ingest_fail(ctx, RALA_HIP_ENOMEM, "device memory for the overlap columns");
            if (part.n_lines) INGEST_CHECK(hipMemcpy(ctx->d_paf_strand.p + rows, ctx->d_paf_win_strand.p, part.n_lines, hipMemcpyDeviceToDevice));
            rows += part.n_lines;
        }
        for (int k = 0; k < 7; ++k) ctx->d_paf_win[k].release();
        ctx->d_paf_win_strand.release();
        R.n_lines = rows;
        ctx->ingest_tm.ship_ms = ship; ctx->ingest_tm.tokenize_ms = tok;
        ctx->ingest_tm.bytes = file_n; ctx->ingest_tm.lines = rows;
    }
    if (R.flags) {
        *irregular = (int)R.flags;
        return RALA_HIP_OK;
    }
    if (R.first_bad != ~0ull) {
        *length_error_read = (int64_t)(R.first_bad & 0xFFFFFFFFull);
        return RALA_HIP_OK;
    }
    rala_hip_overlaps dev;
    dev.a_id = ctx->d_paf_col[0].p; dev.b_id = ctx->d_paf_col[1].p; dev.a_begin = ctx->d_paf_col[2].p; dev.a_end = ctx->d_paf_col[3].p;
    dev.b_begin = ctx->d_paf_col[4].p; dev.b_end = ctx->d_paf_col[5].p; dev.length = ctx->d_paf_col[6].p; dev.strand = ctx->d_paf_strand.p;
    return rala_hip_set_overlaps(ctx, &dev, R.n_lines, RALA_HIP_MEM_DEVICE);
}

// The sensitive overlaps (-s; Graph::preprocess, graph.cpp:901-939) of an uncompressed PAF file tokenised on the device, no
// length check (Overlap::transmute_ has none, overlap.cpp:84-114): bytes [lo, hi) of the file's lines (hi = ~0: to its end; a
// rank of a sharded run takes a share - any split of the sensitive set will do).  out: device pointers that stay the
// context's (valid until the next call); hand them to rala_hip_construct / rala_hip_mg_run with the option
// "sensitive_in_device_memory" set.  *irregular != 0: not a file of 12-column records, nothing was set - take the host reader.
int rala_hip_tokenise_sensitive_paf(rala_hip_ctx* ctx, const char* path, uint64_t lo, uint64_t hi, uint32_t threads, rala_hip_overlaps* out,
                                    uint64_t* n, int* irregular) {
    if (!ctx || !path || !out || !n || !irregular) return RALA_HIP_EINVAL;
    *irregular = 0;
    *n = 0;
    PafTarget T;
    for (int k = 0; k < 7; ++k) T.col[k] = &ctx->d_sens_col[k];
    T.strand = &ctx->d_sens_strand;
    PafRange R;
    const int rc = paf_tokenise_range(ctx, path, lo, hi, false, threads, 0, T, &R);
    if (rc != RALA_HIP_OK) return rc;
    if (R.flags) { *irregular = (int)R.flags; return RALA_HIP_OK; }
    out->a_id = ctx->d_sens_col[0].p; out->b_id = ctx->d_sens_col[1].p; out->a_begin = ctx->d_sens_col[2].p; out->a_end = ctx->d_sens_col[3].p;
    out->b_begin = ctx->d_sens_col[4].p; out->b_end = ctx->d_sens_col[5].p; out->length = ctx->d_sens_col[6].p; out->strand = ctx->d_sens_strand.p;
    *n = R.n_lines;
    return RALA_HIP_OK;
}

int rala_hip_get_overlap_columns(rala_hip_ctx* ctx, uint64_t* n, uint32_t* const cols[7], uint8_t* strand) {
    if (!ctx || !n) return RALA_HIP_EINVAL;
    if (!ctx->inputs_set || ctx->tuple_mode) return ingest_fail(ctx, RALA_HIP_EINVAL, "no overlaps set");
    *n = ctx->n_ovl;
    if (!cols && !strand) return RALA_HIP_OK;
    INGEST_CHECK(hipSetDevice(ctx->device));
    { const int rcu = rala_hip::flush_upload(ctx); if (rcu != RALA_HIP_OK) return rcu; }
    const uint32_t* src[7] = {ctx->ovl.a_id, ctx->ovl.b_id, ctx->ovl.a_begin, ctx->ovl.a_end, ctx->ovl.b_begin, ctx->ovl.b_end, ctx->ovl.length};
    for (int k = 0; cols && k < 7; ++k) {
        if (cols[k] && ctx->n_ovl) INGEST_CHECK(hipMemcpy(cols[k], src[k], ctx->n_ovl * 4, hipMemcpyDeviceToHost));
    }
    if (strand && ctx->n_ovl) INGEST_CHECK(hipMemcpy(strand, ctx->ovl.strand, ctx->n_ovl, hipMemcpyDeviceToHost));
    return RALA_HIP_OK;
}

int rala_hip_get_ingest_timings(rala_hip_ctx* ctx, rala_hip_ingest_timings* out) {
    if (!ctx || !out) return RALA_HIP_EINVAL;
    *out = ctx->ingest_tm;
    return RALA_HIP_OK;
}

}  // extern "C"
