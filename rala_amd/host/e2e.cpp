// End-to-end figure of SURVEY.md section 8(d): overlaps/s from PAF text to the transitively
// reduced graph (ingest -> rala_hip_initialize / _construct / _remove_transitive_edges), sequence
// loading excluded like in the metric's definition.  Read names are "r<i>" (what rala_amd.synth
// writes).  Ingest: an uncompressed file's text goes to the device and is tokenised there
// (rala_hip_set_overlaps_from_paf; ms_parse = ship + tokenise, ms_upload = the name table); a
// compressed file, a file the device tokeniser calls irregular, or device_ingest = 0: the host
// readers (multi-threaded parse, then the columns' upload).
#include <stdint.h>
#include <stdio.h>

#include <chrono>
#include <string>
#include <vector>

#include "io.hpp"
#include "rala_hip.h"

extern "C" int rala_e2e_from_paf_with(const char* paf_path, const uint32_t* read_len, uint64_t n_reads, uint32_t num_threads,
                                      int device_ingest, double* ms_parse, double* ms_upload, double* ms_device, uint64_t* n_overlaps,
                                      uint32_t* n_transitive, int* used_device_ingest) {
    using clock = std::chrono::steady_clock;
    auto ms = [](clock::time_point a, clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    std::vector<std::string> names(n_reads);
    for (uint64_t i = 0; i < n_reads; ++i) names[i] = "r" + std::to_string(i);
    std::vector<uint32_t> len(read_len, read_len + n_reads);
    rala::io::NameTable table;
    table.build(names);
    rala_hip_ctx* ctx = nullptr;
    if (rala_hip_create(0, &ctx) != RALA_HIP_OK) return -1;
    int rc = rala_hip_set_reads(ctx, read_len, n_reads);
    if (rc != RALA_HIP_OK) { rala_hip_destroy(ctx); return rc; }

    auto t0 = clock::now();
    rala::io::OverlapColumns c;
    int64_t bad = -1;
    const std::string path(paf_path);
    uint64_t n_ovl = 0;
    bool on_device = false;
    clock::time_point t1, t2;
    if (device_ingest && !rala::io::has_suffix(path, ".gz")) {
        int irregular = 0;
        rc = rala_hip_set_name_table(ctx, table.buckets(), table.n_buckets(), table.arena().data(), table.arena().size());
        t1 = clock::now();
        if (rc == RALA_HIP_OK) rc = rala_hip_set_overlaps_from_paf(ctx, paf_path, 1, num_threads, &bad, &irregular);
        t2 = clock::now();
        if (rc != RALA_HIP_OK || bad >= 0) { rala_hip_destroy(ctx); return -2; }
        on_device = irregular == 0;
        if (on_device) {
            rala_hip_get_overlap_columns(ctx, &n_ovl, nullptr, nullptr);
            // (reported as: ms_upload = the name table, ms_parse = the text's way to the device + the tokeniser)
            const auto ship = t2 - t1, table_up = t1 - t0;
            t1 = t0 + ship;
            t2 = t1 + table_up;
        } else {
            t0 = clock::now();          // the host reader starts over
        }
    }
    if (!on_device) {
        // (a gzip-compressed file: one thread inflates, the others parse)
        const bool ok = rala::io::has_suffix(path, ".gz")
                            ? rala::io::read_overlaps_streamed(path, false, table, len, true, num_threads, c, &bad)
                            : rala::io::read_paf_parallel(path, table, len, true, num_threads, c, &bad);
        if (!ok || bad >= 0) {
            rala_hip_destroy(ctx);
            return -2;
        }
        t1 = clock::now();
        rala_hip_overlaps soa = {c.a_id.data(), c.b_id.data(), c.a_begin.data(), c.a_end.data(), c.b_begin.data(),
                                 c.b_end.data(), c.length.data(), c.strand.data()};
        rc = rala_hip_set_overlaps(ctx, &soa, c.size(), RALA_HIP_MEM_HOST);
        t2 = clock::now();
        n_ovl = c.size();
    }
    if (used_device_ingest) *used_device_ingest = on_device ? 1 : 0;
    if (rc == RALA_HIP_OK) rc = rala_hip_initialize(ctx);
    if (rc == RALA_HIP_OK) rc = rala_hip_construct(ctx, nullptr, 0);
    uint32_t n_tr = 0;
    if (rc == RALA_HIP_OK) rc = rala_hip_remove_transitive_edges(ctx, &n_tr);
    const auto t3 = clock::now();
    if (rc != RALA_HIP_OK) fprintf(stderr, "[rala_e2e_from_paf] error: %s\n", rala_hip_last_error(ctx));
    *ms_parse = ms(t0, t1); *ms_upload = ms(t1, t2); *ms_device = ms(t2, t3);
    *n_overlaps = n_ovl; *n_transitive = n_tr;
    rala_hip_destroy(ctx);
    return rc;
}

extern "C" int rala_e2e_from_paf(const char* paf_path, const uint32_t* read_len, uint64_t n_reads, uint32_t num_threads,
                                 double* ms_parse, double* ms_upload, double* ms_device, uint64_t* n_overlaps,
                                 uint32_t* n_transitive) {
    return rala_e2e_from_paf_with(paf_path, read_len, n_reads, num_threads, 1, ms_parse, ms_upload, ms_device, n_overlaps,
                                  n_transitive, nullptr);
}
