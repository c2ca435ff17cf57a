// End-to-end figure of SURVEY.md section 8(d): overlaps/s from PAF text to the transitively
// reduced graph (ingest -> rala_hip_initialize / _construct / _remove_transitive_edges), sequence
// loading excluded like in the metric's definition.  Read names are "r<i>" (what rala_amd.synth
// writes).  Ingest: an uncompressed file's text goes to the device and is tokenised there
// (rala_hip_set_overlaps_from_paf; ms_parse = ship + tokenise, ms_upload = the name table); a
// compressed file, a file the device tokeniser calls irregular, or device_ingest = 0: the host
// readers (multi-threaded parse, then the columns' upload).
#include <stdint.h>
#include <stdio.h>

#include <algorithm>
#include <chrono>
#include <string>
#include <thread>
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

// The same over several ranks (threads of this process, one per device ordinal in `devices`; transport: 0 RCCL, 1 the
// in-process one, which also lets ranks share a device): every rank ships and tokenises its own byte range of the file on
// its own GPU (rala_hip_mg_set_overlaps_from_paf), then the sharded step.  ms_ingest: name tables + ship + tokenise + the
// cuts' exchange (the slowest rank); ms_device: the step (first call).
extern "C" int rala_e2e_from_paf_ranks(const char* paf_path, const uint32_t* read_len, uint64_t n_reads, uint32_t num_threads, uint32_t world,
                                       const int* devices, int transport, double* ms_ingest, double* ms_device, uint64_t* n_overlaps,
                                       uint32_t* n_transitive) {
    using clock = std::chrono::steady_clock;
    auto ms = [](clock::time_point a, clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    if (world == 0 || world > 64) return RALA_HIP_EINVAL;
    std::vector<std::string> names(n_reads);
    for (uint64_t i = 0; i < n_reads; ++i) names[i] = "r" + std::to_string(i);
    rala::io::NameTable table;
    table.build(names);
    void* group = nullptr;
    unsigned char id[128] = {0};
    if (transport == RALA_HIP_COMM_LOCAL) {
        if (rala_hip_mg_local_group_create(world, &group) != RALA_HIP_OK) return -1;
    } else if (rala_hip_mg_unique_id(id) != RALA_HIP_OK) {
        return -1;
    }
    std::vector<rala_hip_mg*> ranks(world, nullptr);
    std::vector<int> rc(world, RALA_HIP_OK);
    for (uint32_t k = 0; k < world; ++k) rc[k] = rala_hip_mg_create_contexts(devices[k], k, world, &ranks[k]);
    int bad_rc = RALA_HIP_OK;
    for (uint32_t k = 0; k < world; ++k) if (rc[k] != RALA_HIP_OK) bad_rc = rc[k];
    std::vector<uint64_t> rows(world, 0);
    std::vector<int> irregular(world, 0);
    std::vector<int64_t> bad(world, -1);
    auto on_every_rank = [&](auto f) {
        std::vector<std::thread> th;
        for (uint32_t k = 0; k < world; ++k) th.emplace_back([&, k]() { rc[k] = f(k); });
        for (auto& t : th) t.join();
        for (uint32_t k = 0; k < world; ++k) if (rc[k] != RALA_HIP_OK && bad_rc == RALA_HIP_OK) {
            bad_rc = rc[k];
            fprintf(stderr, "[rala_e2e_from_paf_ranks] rank %u: %s\n", k, rala_hip_mg_last_error(ranks[k]));
        }
    };
    if (bad_rc == RALA_HIP_OK) {
        on_every_rank([&](uint32_t k) {
            int r = rala_hip_mg_join(ranks[k], transport, transport == RALA_HIP_COMM_LOCAL ? group : (void*)id);
            if (r == RALA_HIP_OK) r = rala_hip_mg_set_reads(ranks[k], read_len, n_reads);
            return r;
        });
    }
    const auto t0 = clock::now();
    if (bad_rc == RALA_HIP_OK) {
        on_every_rank([&](uint32_t k) {
            int r = rala_hip_set_name_table(rala_hip_mg_context(ranks[k]), table.buckets(), table.n_buckets(), table.arena().data(), table.arena().size());
            const int r2 = rala_hip_mg_set_overlaps_from_paf(ranks[k], r == RALA_HIP_OK ? paf_path : "", 1, std::max(1u, num_threads / world), &bad[k],
                                                             &irregular[k]);
            if (r == RALA_HIP_OK) r = r2;
            uint64_t first = 0;
            if (r == RALA_HIP_OK && !irregular[k] && bad[k] < 0) r = rala_hip_mg_get_slice(ranks[k], &first, &rows[k]);
            return r;
        });
        for (uint32_t k = 0; k < world; ++k) if (bad_rc == RALA_HIP_OK && (irregular[k] || bad[k] >= 0)) bad_rc = -2;
    }
    const auto t1 = clock::now();
    uint32_t pairs = 0;
    if (bad_rc == RALA_HIP_OK) {
        bad_rc = rala_hip_mg_run_threads(ranks.data(), world, nullptr, nullptr, &pairs);
        if (bad_rc != RALA_HIP_OK) {
            for (uint32_t k = 0; k < world; ++k) fprintf(stderr, "[rala_e2e_from_paf_ranks] rank %u: %s\n", k, rala_hip_mg_last_error(ranks[k]));
        }
    }
    const auto t2 = clock::now();
    *ms_ingest = ms(t0, t1); *ms_device = ms(t1, t2);
    *n_overlaps = 0;
    for (uint64_t r : rows) *n_overlaps += r;
    *n_transitive = pairs;
    for (rala_hip_mg* r : ranks) if (r) rala_hip_mg_destroy(r);
    if (group) rala_hip_mg_local_group_destroy(group);
    return bad_rc;
}
