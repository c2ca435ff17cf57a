/*!
 * @file host_api_capi.cpp
 *
 * @brief C entry points over stand-alone rala::Pile / rala::Overlap objects, for the tests that
 * drive the class interface the way the reference's Graph::initialize does (createPile,
 * add_layers, find_valid_region, find_median, find_chimeric_hills, find_chimeric_pits,
 * break_over_*, Overlap::transmute / trim / type; reference src/pile.hpp:19-170,
 * src/overlap.hpp:27-117).  Built into rala_amd/host/librala_api.so; not part of librala.so.
 */

#include <stdio.h>
#include <string.h>
#include <stdint.h>

#include <memory>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "io.hpp"
#include "overlap.hpp"
#include "pile.hpp"
#include "sequence.hpp"
#include "rala_hip.h"

namespace {

struct Handle {
    std::vector<std::unique_ptr<rala::Pile>> piles;
    std::unordered_map<std::string, uint64_t> name_to_id;
    std::vector<uint32_t> length;
};

std::string read_name(uint64_t r) { return "read_" + std::to_string(r); }

}  // namespace

extern "C" {

void* hp_create(const uint32_t* read_len, uint64_t n) {
    Handle* h = new Handle();
    for (uint64_t r = 0; r < n; ++r) {
        h->piles.emplace_back(rala::createPile(r, read_len[r]));
        h->name_to_id[read_name(r)] = r;
        h->length.push_back(read_len[r]);
    }
    return h;
}

void hp_destroy(void* p) { delete (Handle*)p; }

void hp_add_layers(void* p, uint64_t r, const uint32_t* bounds, uint64_t n) {
    std::vector<uint32_t> b(bounds, bounds + n);
    ((Handle*)p)->piles[r]->add_layers(b);
}

int hp_find_valid_region(void* p, uint64_t r) { return ((Handle*)p)->piles[r]->find_valid_region() ? 1 : 0; }
void hp_find_median(void* p, uint64_t r) { ((Handle*)p)->piles[r]->find_median(); }
void hp_find_chimeric_hills(void* p, uint64_t r) { ((Handle*)p)->piles[r]->find_chimeric_hills(); }
void hp_find_chimeric_pits(void* p, uint64_t r) { ((Handle*)p)->piles[r]->find_chimeric_pits(); }
int hp_break_over_chimeric_pits(void* p, uint64_t r, uint16_t median) {
    return ((Handle*)p)->piles[r]->break_over_chimeric_pits(median) ? 1 : 0;
}
int hp_break_over_chimeric_hills(void* p, uint64_t r) {
    return ((Handle*)p)->piles[r]->break_over_chimeric_hills() ? 1 : 0;
}
int hp_shrink(void* p, uint64_t r, uint32_t begin, uint32_t end) {
    return ((Handle*)p)->piles[r]->shrink(begin, end) ? 1 : 0;
}
void hp_find_repetitive_hills(void* p, uint64_t r, uint16_t median) { ((Handle*)p)->piles[r]->find_repetitive_hills(median); }
int hp_has_repetitive_hills(void* p, uint64_t r) { return ((Handle*)p)->piles[r]->has_repetitive_hills() ? 1 : 0; }
/*! @brief Pile::to_json; returns the length, copies at most cap bytes */
uint64_t hp_to_json(void* p, uint64_t r, char* out, uint64_t cap) {
    const auto& pile = ((Handle*)p)->piles[r];
    if (pile == nullptr) return 0;
    const std::string s = pile->to_json();
    if (out) memcpy(out, s.data(), s.size() < cap ? s.size() : cap);
    return s.size();
}
/*! @brief Graph::initialize drops a pile whose valid region is too short (graph.cpp:396-399) */
void hp_reset(void* p, uint64_t r) { ((Handle*)p)->piles[r].reset(); }

/*! @brief out = {alive, begin, end, median, p10, has_chimeric_pit, has_chimeric_hill} */
void hp_get(void* p, uint64_t r, uint32_t* out) {
    const auto& pile = ((Handle*)p)->piles[r];
    for (int i = 0; i < 7; ++i) out[i] = 0;
    if (pile == nullptr) return;
    out[0] = 1; out[1] = pile->begin(); out[2] = pile->end(); out[3] = pile->median(); out[4] = pile->p10();
    out[5] = pile->has_chimeric_pit(); out[6] = pile->has_chimeric_hill();
}

uint64_t hp_data(void* p, uint64_t r, uint16_t* out) {
    const auto& pile = ((Handle*)p)->piles[r];
    if (pile == nullptr) return 0;
    const std::vector<uint16_t>& d = pile->data();
    for (size_t i = 0; i < d.size(); ++i) out[i] = d[i];
    return d.size();
}

/*!
 * @brief one PAF record against the current piles: createOverlap, transmute, trim, type.
 * coords = {a_begin, a_end, b_begin, b_end, length} in / out.  Returns 0 when transmute or
 * trim drops the overlap.
 */
int hp_overlap_trim_type(void* p, uint32_t a, uint32_t b, uint32_t strand, uint32_t* coords, int* type_out) {
    Handle* h = (Handle*)p;
    auto o = rala::createOverlap(read_name(a), h->length[a], coords[0], coords[1], strand ? '-' : '+',
        read_name(b), h->length[b], coords[2], coords[3], coords[4]);
    if (!o->transmute(h->piles, h->name_to_id)) return 0;
    if (o->a_id() != a || o->b_id() != b || o->orientation() != strand) return -1;
    if (!o->trim(h->piles)) return 0;
    coords[0] = o->a_begin(); coords[1] = o->a_end(); coords[2] = o->b_begin(); coords[3] = o->b_end();
    coords[4] = o->length();
    *type_out = (int)o->type(h->piles);
    return 1;
}

/*! @brief an MHAP record (1-based ids): the fields createOverlap derives */
int hp_overlap_from_mhap(uint64_t a_id, uint64_t b_id, uint32_t a_rc, uint32_t a_begin, uint32_t a_end,
    uint32_t a_length, uint32_t b_rc, uint32_t b_begin, uint32_t b_end, uint32_t b_length, uint32_t* out) {
    auto o = rala::createOverlap(a_id, b_id, a_rc, a_begin, a_end, a_length, b_rc, b_begin, b_end, b_length);
    out[0] = o->a_id(); out[1] = o->b_id(); out[2] = o->length(); out[3] = o->orientation();
    return 0;
}

// ---- the device tokeniser (rala_hip_set_overlaps_from_paf) for tests/test_gpu_ingest.py: the columns it leaves, to be
// compared with the host readers' (libassembly_graph.so: io_paf_parse).  names: n_reads names separated by '\n'.
struct PafOnDevice {
    int rc = 0, irregular = 0;
    int64_t bad = -1;
    uint64_t n = 0;
    std::vector<uint32_t> col[7];
    std::vector<uint8_t> strand;
    rala_hip_ingest_timings tm = {};
};

static void* text_on_device(const char* path, const char* names, const uint32_t* read_len, uint64_t n_reads, int check_lengths, uint32_t threads, bool mhap);
void* hp_paf_device(const char* path, const char* names, const uint32_t* read_len, uint64_t n_reads, int check_lengths, uint32_t threads) {
    return text_on_device(path, names, read_len, n_reads, check_lengths, threads, false);
}
// (an MHAP file names its reads by number: no name table)
void* hp_mhap_device(const char* path, const uint32_t* read_len, uint64_t n_reads, int check_lengths, uint32_t threads) {
    return text_on_device(path, "", read_len, n_reads, check_lengths, threads, true);
}
static void* text_on_device(const char* path, const char* names, const uint32_t* read_len, uint64_t n_reads, int check_lengths, uint32_t threads, bool mhap) {
    std::vector<std::string> nm;
    const char* p = names;
    for (uint64_t i = 0; i < n_reads && !mhap; ++i) {
        const char* e = strchr(p, '\n');
        nm.emplace_back(p, e ? (size_t)(e - p) : strlen(p));
        p = e ? e + 1 : p + strlen(p);
    }
    rala::io::NameTable table;
    if (!mhap) table.build(nm);
    auto* out = new PafOnDevice();
    rala_hip_ctx* ctx = nullptr;
    out->rc = rala_hip_create(0, &ctx);
    if (out->rc != RALA_HIP_OK) return out;
    out->rc = rala_hip_set_reads(ctx, read_len, n_reads);
    if (out->rc == RALA_HIP_OK && !mhap) out->rc = rala_hip_set_name_table(ctx, table.buckets(), table.n_buckets(), table.arena().data(), table.arena().size());
    if (out->rc == RALA_HIP_OK) {
        out->rc = mhap ? rala_hip_set_overlaps_from_mhap(ctx, path, check_lengths, threads, &out->bad, &out->irregular)
                       : rala_hip_set_overlaps_from_paf(ctx, path, check_lengths, threads, &out->bad, &out->irregular);
    }
    if (out->rc == RALA_HIP_OK) rala_hip_get_ingest_timings(ctx, &out->tm);
    if (out->rc == RALA_HIP_OK && !out->irregular && out->bad < 0) {
        out->rc = rala_hip_get_overlap_columns(ctx, &out->n, nullptr, nullptr);
        uint32_t* cols[7];
        for (int k = 0; k < 7; ++k) { out->col[k].resize(out->n); cols[k] = out->col[k].data(); }
        out->strand.resize(out->n);
        if (out->rc == RALA_HIP_OK) out->rc = rala_hip_get_overlap_columns(ctx, &out->n, cols, out->strand.data());
    }
    rala_hip_destroy(ctx);
    return out;
}
// The same over `world` ranks that share device 0 (in-process transport), every rank on a thread of its own: rank k ships and
// tokenises its byte range of the file, the ranks settle the cuts between runs of equal queries and move the rows in front
// of them (rala_hip_mg_set_overlaps_from_paf).  The handle holds the slices' columns back to back (which must be the
// file's records in order); slices[2 k], [2 k + 1] = file position of rank k's first record and its record count.
// sensitive != 0: the ranks' shares of a sensitive file instead (rala_hip_tokenise_sensitive_paf, no collective, no cuts).
void* hp_paf_device_ranks(const char* path, const char* names, const uint32_t* read_len, uint64_t n_reads, int check_lengths, uint32_t threads,
                          uint32_t world, int sensitive, uint64_t* slices) {
    std::vector<std::string> nm;
    const char* p = names;
    for (uint64_t i = 0; i < n_reads; ++i) {
        const char* e = strchr(p, '\n');
        nm.emplace_back(p, e ? (size_t)(e - p) : strlen(p));
        p = e ? e + 1 : p + strlen(p);
    }
    rala::io::NameTable table;
    table.build(nm);
    auto* out = new PafOnDevice();
    void* group = nullptr;
    out->rc = rala_hip_mg_local_group_create(world, &group);
    if (out->rc != RALA_HIP_OK) return out;
    std::vector<rala_hip_mg*> ranks(world, nullptr);
    for (uint32_t k = 0; k < world && out->rc == RALA_HIP_OK; ++k) out->rc = rala_hip_mg_create_contexts(0, k, world, &ranks[k]);
    std::vector<int> rc(world, RALA_HIP_OK), irregular(world, 0);
    std::vector<int64_t> bad(world, -1);
    std::vector<std::vector<uint32_t>> col[7];
    for (auto& c : col) c.resize(world);
    std::vector<std::vector<uint8_t>> strand(world);
    uint64_t file_bytes = 0;
    if (sensitive) {
        FILE* f = fopen(path, "rb");
        if (f) { fseek(f, 0, SEEK_END); file_bytes = (uint64_t)ftell(f); fclose(f); }
    }
    if (out->rc == RALA_HIP_OK) {
        std::vector<std::thread> th;
        for (uint32_t k = 0; k < world; ++k) {
            th.emplace_back([&, k]() {
                rala_hip_mg* mg = ranks[k];
                int r = rala_hip_mg_join(mg, RALA_HIP_COMM_LOCAL, group);
                if (r == RALA_HIP_OK) r = rala_hip_mg_set_reads(mg, read_len, n_reads);
                rala_hip_ctx* cs = rala_hip_mg_context(mg);
                if (r == RALA_HIP_OK) r = rala_hip_set_name_table(cs, table.buckets(), table.n_buckets(), table.arena().data(), table.arena().size());
                uint64_t first = 0, n = 0;
                std::vector<uint32_t> got[7];
                std::vector<uint8_t> got_strand;
                if (sensitive) {
                    rala_hip_overlaps dev = {};
                    if (r == RALA_HIP_OK) r = rala_hip_tokenise_sensitive_paf(cs, path, file_bytes * k / world, file_bytes * (k + 1) / world, threads, &dev, &n, &irregular[k]);
                    if (r == RALA_HIP_OK && !irregular[k]) {
                        // (read back through the context: the columns are plain device memory, adopted as its overlaps)
                        r = rala_hip_set_overlaps(cs, &dev, n, RALA_HIP_MEM_DEVICE);
                        uint32_t* dst[7];
                        for (int c = 0; c < 7; ++c) { got[c].resize(n); dst[c] = got[c].data(); }
                        got_strand.resize(n);
                        uint64_t n2 = 0;
                        if (r == RALA_HIP_OK) r = rala_hip_get_overlap_columns(cs, &n2, dst, got_strand.data());
                    }
                } else {
                    if (r == RALA_HIP_OK) r = rala_hip_mg_set_overlaps_from_paf(mg, path, check_lengths, threads, &bad[k], &irregular[k]);
                    if (r == RALA_HIP_OK && !irregular[k] && bad[k] < 0) {
                        r = rala_hip_mg_get_slice(mg, &first, &n);
                        uint32_t* dst[7];
                        for (int c = 0; c < 7; ++c) { got[c].resize(n); dst[c] = got[c].data(); }
                        got_strand.resize(n);
                        uint64_t n2 = 0;
                        if (r == RALA_HIP_OK) r = rala_hip_get_overlap_columns(cs, &n2, dst, got_strand.data());
                        if (r == RALA_HIP_OK && n2 != n) r = RALA_HIP_EDEVICE;
                    }
                }
                slices[2 * k] = first; slices[2 * k + 1] = n;
                for (int c = 0; c < 7; ++c) col[c][k] = std::move(got[c]);
                strand[k] = std::move(got_strand);
                rc[k] = r;
            });
        }
        for (auto& t : th) t.join();
    }
    for (uint32_t k = 0; k < world; ++k) {
        if (rc[k] != RALA_HIP_OK && out->rc == RALA_HIP_OK) { out->rc = rc[k]; fprintf(stderr, "[hp_paf_device_ranks] rank %u: %s\n", k, rala_hip_mg_last_error(ranks[k])); }
        out->irregular |= irregular[k];
        if (bad[k] >= 0 && out->bad < 0) out->bad = bad[k];
        for (int c = 0; c < 7; ++c) out->col[c].insert(out->col[c].end(), col[c][k].begin(), col[c][k].end());
        out->strand.insert(out->strand.end(), strand[k].begin(), strand[k].end());
    }
    out->n = out->strand.size();
    for (rala_hip_mg* r : ranks) if (r) rala_hip_mg_destroy(r);
    rala_hip_mg_local_group_destroy(group);
    return out;
}
// info[0 .. 5] = return code, irregular flags, first read with a length mismatch (-1 none), records, ship us, tokenise us
void hp_paf_device_info(void* h, int64_t* info) {
    const auto* o = (const PafOnDevice*)h;
    info[0] = o->rc; info[1] = o->irregular; info[2] = o->bad; info[3] = (int64_t)o->n;
    info[4] = (int64_t)(o->tm.ship_ms * 1000.0f); info[5] = (int64_t)(o->tm.tokenize_ms * 1000.0f);
}
void hp_paf_device_copy(void* h, uint32_t* a_id, uint32_t* b_id, uint32_t* a_begin, uint32_t* a_end, uint32_t* b_begin, uint32_t* b_end,
                        uint32_t* length, uint8_t* strand) {
    const auto* o = (const PafOnDevice*)h;
    uint32_t* dst[7] = {a_id, b_id, a_begin, a_end, b_begin, b_end, length};
    for (int k = 0; k < 7; ++k) memcpy(dst[k], o->col[k].data(), o->n * 4);
    memcpy(strand, o->strand.data(), o->n);
}
void hp_paf_device_free(void* h) { delete (PafOnDevice*)h; }

// rala::createSequence (reference src/sequence.cpp:12-25): the length of what it made - it leaves the process on an empty name / data
uint64_t hp_sequence_length(const char* name, const char* data) { return rala::createSequence(name, data)->data().size(); }

}  // extern "C"
