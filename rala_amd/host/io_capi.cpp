// C surface of the overlap readers for the CPU test-suite (tests/test_ingest_cpu.py): the
// multi-threaded PAF reader against the line-by-line one.  Not part of the product boundary.
#include <stdint.h>
#include <string.h>

#include <algorithm>

#include <string>
#include <unordered_map>
#include <vector>

#include "io.hpp"

namespace {

struct Parsed {
    rala::io::OverlapColumns cols;
    int64_t length_error = -1;
    bool ok = false;
};

}  // namespace

extern "C" {

// names: n_reads names separated by '\n'.  parallel = 0: io::read_paf + std::unordered_map (the line-by-line
// reader), 1: io::read_paf_parallel, 2: io::read_overlaps_streamed (PAF, plain or gzip), 3: the same for
// MHAP, 4: io::read_mhap line by line
void* io_paf_parse(const char* path, const char* names, const uint32_t* read_len, uint64_t n_reads, int check_lengths,
                   uint32_t threads, int parallel) {
    std::vector<std::string> nm;
    const char* p = names;
    for (uint64_t i = 0; i < n_reads; ++i) {
        const char* e = strchr(p, '\n');
        nm.emplace_back(p, e ? (size_t)(e - p) : strlen(p));
        p = e ? e + 1 : p + strlen(p);
    }
    std::vector<uint32_t> len(read_len, read_len + n_reads);
    auto* out = new Parsed();
    if (parallel == 1 || parallel == 2 || parallel == 3) {
        rala::io::NameTable table;
        table.build(nm);
        if (parallel == 1) out->ok = rala::io::read_paf_parallel(path, table, len, check_lengths != 0, threads, out->cols, &out->length_error);
        else out->ok = rala::io::read_overlaps_streamed(path, parallel == 3, table, len, check_lengths != 0, threads, out->cols, &out->length_error);
        return out;
    }
    if (parallel == 4) {
        auto& c = out->cols;
        out->ok = rala::io::read_mhap(path, [&](const rala::io::MhapRecord& r) {
            const uint64_t a = r.a_id - 1, b = r.b_id - 1;
            const uint32_t ia = a < len.size() ? (uint32_t)a : 0xFFFFFFFFu, ib = b < len.size() ? (uint32_t)b : 0xFFFFFFFFu;
            if (out->length_error < 0) {
                if (check_lengths && ia != 0xFFFFFFFFu && r.a_length != len[ia]) out->length_error = ia;
                else if (check_lengths && ia != 0xFFFFFFFFu && ib != 0xFFFFFFFFu && r.b_length != len[ib]) out->length_error = ib;
            }
            c.a_id.push_back(ia); c.b_id.push_back(ib);
            c.a_begin.push_back(r.a_begin); c.a_end.push_back(r.a_end);
            c.b_begin.push_back(r.b_begin); c.b_end.push_back(r.b_end);
            c.length.push_back(std::max(r.a_end - r.a_begin, r.b_end - r.b_begin)); c.strand.push_back(r.a_rc == r.b_rc ? 0 : 1);
        });
        return out;
    }
    std::unordered_map<std::string, uint64_t> map;
    for (uint64_t i = 0; i < n_reads; ++i) map[nm[i]] = i;
    auto& c = out->cols;
    out->ok = rala::io::read_paf(path, [&](const rala::io::PafRecord& r) {
        auto a = map.find(r.q_name), b = map.find(r.t_name);
        const uint32_t ia = a == map.end() ? 0xFFFFFFFFu : (uint32_t)a->second;
        const uint32_t ib = b == map.end() ? 0xFFFFFFFFu : (uint32_t)b->second;
        if (out->length_error < 0) {
            if (check_lengths && ia != 0xFFFFFFFFu && r.q_length != len[ia]) out->length_error = ia;
            else if (check_lengths && ia != 0xFFFFFFFFu && ib != 0xFFFFFFFFu && r.t_length != len[ib]) out->length_error = ib;
        }
        c.a_id.push_back(ia); c.b_id.push_back(ib);
        c.a_begin.push_back(r.q_begin); c.a_end.push_back(r.q_end);
        c.b_begin.push_back(r.t_begin); c.b_end.push_back(r.t_end);
        c.length.push_back(r.overlap_length); c.strand.push_back(r.orientation == '+' ? 0 : 1);
    });
    return out;
}

int io_paf_ok(void* h) { return ((Parsed*)h)->ok; }
uint64_t io_paf_size(void* h) { return ((Parsed*)h)->cols.size(); }
int64_t io_paf_length_error(void* h) { return ((Parsed*)h)->length_error; }
void io_paf_copy(void* h, uint32_t* a_id, uint32_t* b_id, uint32_t* a_begin, uint32_t* a_end, uint32_t* b_begin,
                 uint32_t* b_end, uint32_t* length, uint8_t* strand) {
    const auto& c = ((Parsed*)h)->cols;
    const size_t n = c.size();
    memcpy(a_id, c.a_id.data(), n * 4); memcpy(b_id, c.b_id.data(), n * 4);
    memcpy(a_begin, c.a_begin.data(), n * 4); memcpy(a_end, c.a_end.data(), n * 4);
    memcpy(b_begin, c.b_begin.data(), n * 4); memcpy(b_end, c.b_end.data(), n * 4);
    memcpy(length, c.length.data(), n * 4); memcpy(strand, c.strand.data(), n);
}
void io_paf_free(void* h) { delete (Parsed*)h; }

}  // extern "C"
