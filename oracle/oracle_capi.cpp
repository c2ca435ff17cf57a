// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_core.hpp).
//
// C interface (for ctypes) over oracle_graph.hpp's Driver.  Compiled twice by
// oracle/Makefile:
//   liboracle.so      -> flat restatement            (travels with the repo)
//   _ref/liboracle_ref.so (-DORA_USE_REF) -> real rala::Pile / rala::Overlap
//                        objects from /root/reference (built only where the
//                        reference exists)
#include <string.h>

#ifdef ORA_USE_REF
#include "ref_backend.hpp"
typedef ora::RefBackend Backend;
#else
#include "flat_backend.hpp"
typedef ora::FlatBackend Backend;
#endif
#include "oracle_graph.hpp"

typedef ora::Driver<Backend> Drv;

namespace {
struct Handle {
    Drv d;
    // owned copies of the inputs
    std::vector<uint32_t> a_id, b_id, a_begin, a_end, b_begin, b_end, length;
    std::vector<uint8_t> strand;
};

void fill_input(ora::OvlInput& in, uint64_t n, std::vector<uint32_t>& a_id, std::vector<uint32_t>& b_id,
                std::vector<uint32_t>& ab, std::vector<uint32_t>& ae, std::vector<uint32_t>& bb,
                std::vector<uint32_t>& be, std::vector<uint32_t>& len, std::vector<uint8_t>& st) {
    in.n = n;
    in.a_id = a_id.data(); in.b_id = b_id.data();
    in.a_begin = ab.data(); in.a_end = ae.data();
    in.b_begin = bb.data(); in.b_end = be.data();
    in.length = len.data(); in.strand = st.data();
}

uint64_t copy_intervals(const std::vector<ora::Iv>& v, uint32_t* out) {
    if (out) {
        for (size_t i = 0; i < v.size(); ++i) { out[2 * i] = v[i].first; out[2 * i + 1] = v[i].second; }
    }
    return v.size();
}
}  // namespace

extern "C" {

const char* ora_backend_name() { return Backend::name(); }

void* ora_create(uint32_t n_threads) {
    Handle* h = new Handle;
    h->d.n_threads = n_threads ? n_threads : 1;
    return h;
}

// one task per pile through a pool of n_threads workers (the reference's thread_pool fan-out, graph.cpp:367-377) instead of
// the chunked loop
void ora_use_task_pool(void* p) {
    Handle* h = (Handle*)p;
    h->d.task_pool = thread_pool::createThreadPool(h->d.n_threads);
}

void ora_destroy(void* p) { delete (Handle*)p; }

void ora_set_reads(void* p, const uint32_t* len, uint64_t n) { ((Handle*)p)->d.set_reads(len, n); }

void ora_set_overlaps(void* p, uint64_t n, const uint32_t* a_id, const uint32_t* b_id, const uint32_t* a_begin,
                      const uint32_t* a_end, const uint32_t* b_begin, const uint32_t* b_end,
                      const uint32_t* length, const uint8_t* strand) {
    Handle* h = (Handle*)p;
    h->a_id.assign(a_id, a_id + n); h->b_id.assign(b_id, b_id + n);
    h->a_begin.assign(a_begin, a_begin + n); h->a_end.assign(a_end, a_end + n);
    h->b_begin.assign(b_begin, b_begin + n); h->b_end.assign(b_end, b_end + n);
    h->length.assign(length, length + n); h->strand.assign(strand, strand + n);
    fill_input(h->d.in, n, h->a_id, h->b_id, h->a_begin, h->a_end, h->b_begin, h->b_end, h->length, h->strand);
}

// ---- stages --------------------------------------------------------------
void ora_pass1(void* p) { ((Handle*)p)->d.pass1_dedupe_and_layers(); }
int ora_annotate(void* p) { return ((Handle*)p)->d.annotate() ? 0 : -1; }
int ora_initialize(void* p) { return ((Handle*)p)->d.initialize() ? 0 : -1; }
void ora_pass2(void* p) { ((Handle*)p)->d.pass2(); }
void ora_preprocess_chimeras(void* p) { ((Handle*)p)->d.preprocess_chimeras(); }
void ora_preprocess_repeats(void* p, uint64_t n, const uint32_t* a_id, const uint32_t* b_id,
                            const uint32_t* a_begin, const uint32_t* a_end, const uint32_t* b_begin,
                            const uint32_t* b_end, const uint32_t* length, const uint8_t* strand) {
    ora::OvlInput s;
    s.n = n; s.a_id = a_id; s.b_id = b_id; s.a_begin = a_begin; s.a_end = a_end;
    s.b_begin = b_begin; s.b_end = b_end; s.length = length; s.strand = strand;
    ((Handle*)p)->d.preprocess_repeats(s);
}
void ora_build_graph(void* p) { ((Handle*)p)->d.build_graph(); }
uint32_t ora_remove_transitive_edges(void* p) { return ((Handle*)p)->d.remove_transitive_edges(); }

// ---- per-read state --------------------------------------------------------
uint64_t ora_n_reads(void* p) { return ((Handle*)p)->d.n_reads; }
uint64_t ora_n_prefiltered(void* p) { return ((Handle*)p)->d.n_prefiltered; }

void ora_get_valid(void* p, uint8_t* out) {
    Handle* h = (Handle*)p;
    memcpy(out, h->d.valid.data(), h->d.valid.size());
}

void ora_get_piles(void* p, uint32_t* begin, uint32_t* end, uint16_t* median, uint16_t* p10, uint8_t* alive) {
    Drv& d = ((Handle*)p)->d;
    for (uint64_t r = 0; r < d.n_reads; ++r) {
        const bool a = d.bk.alive(r);
        alive[r] = a;
        begin[r] = a ? d.bk.begin(r) : 0;
        end[r] = a ? d.bk.end(r) : 0;
        median[r] = a ? d.bk.median(r) : 0;
        p10[r] = a ? d.bk.p10(r) : 0;
    }
}

// copies pile r into out (may be null); returns its length, 0 if dead
uint64_t ora_pile_data(void* p, uint64_t r, uint16_t* out) {
    Drv& d = ((Handle*)p)->d;
    if (!d.bk.alive(r)) return 0;
    const std::vector<uint16_t>& v = d.bk.data(r);
    if (out) memcpy(out, v.data(), v.size() * sizeof(uint16_t));
    return v.size();
}

// Checksums of every pile's data_ as the backend's objects hold it (reference src/pile.hpp:53, a std::vector<uint16_t>): per read the
// FNV-1a-64 of its bytes and the sum of its values; 0 for a read without a pile.  Either output may be null.
void ora_pile_row_digests(void* p, uint64_t* fnv, uint64_t* sum) {
    Drv& d = ((Handle*)p)->d;
    for (uint64_t r = 0; r < d.n_reads; ++r) {
        uint64_t h = 0, s = 0;
        if (d.bk.alive(r)) {
            const std::vector<uint16_t>& v = d.bk.data(r);
            const uint8_t* b = (const uint8_t*)v.data();
            h = 1469598103934665603ull;
            for (size_t i = 0; i < v.size() * sizeof(uint16_t); ++i) h = (h ^ b[i]) * 1099511628211ull;
            for (uint16_t x : v) s += x;
        }
        if (fnv) fnv[r] = h;
        if (sum) sum[r] = s;
    }
}

// kind: 0 pits, 1 hills, 2 repeat hills.  out may be null (count only).
uint64_t ora_pile_intervals(void* p, uint64_t r, int kind, uint32_t* out) {
    Drv& d = ((Handle*)p)->d;
    if (!d.bk.alive(r)) return 0;
    if (kind == 0) return copy_intervals(d.bk.pits(r), out);
    if (kind == 1) return copy_intervals(d.bk.hills(r), out);
    return copy_intervals(d.bk.rep_hills(r), out);
}

uint64_t ora_pile_hill_counts(void* p, uint64_t r, uint32_t* out) {
    Drv& d = ((Handle*)p)->d;
    if (!d.bk.alive(r)) return 0;
    const std::vector<uint32_t>& v = d.bk.hill_cnt(r);
    if (out) memcpy(out, v.data(), v.size() * sizeof(uint32_t));
    return v.size();
}

uint64_t ora_pile_repeat_flags(void* p, uint64_t r, uint8_t* out) {
    Drv& d = ((Handle*)p)->d;
    if (!d.bk.alive(r)) return 0;
    std::vector<uint8_t> v = d.bk.rep_flag(r);
    if (out && !v.empty()) memcpy(out, v.data(), v.size());
    return v.size();
}

// ---- unit-level entry points (crafted pile contents) -----------------------
void ora_pile_set_state(void* p, uint64_t r, const uint16_t* data, uint32_t n, uint32_t begin, uint32_t end) {
    ((Handle*)p)->d.bk.set_pile_state(r, data, n, begin, end);
}
void ora_pile_add_layers(void* p, uint64_t r, const uint32_t* bounds, uint64_t n) {
    std::vector<uint32_t> b(bounds, bounds + n);
    ((Handle*)p)->d.bk.add_layers(r, b);
}
int ora_pile_find_valid_region(void* p, uint64_t r) { return ((Handle*)p)->d.bk.find_valid_region(r) ? 1 : 0; }
void ora_pile_find_median(void* p, uint64_t r) { ((Handle*)p)->d.bk.find_median(r); }
void ora_pile_find_chimeric_hills(void* p, uint64_t r) { ((Handle*)p)->d.bk.find_chimeric_hills(r); }
void ora_pile_find_chimeric_pits(void* p, uint64_t r) { ((Handle*)p)->d.bk.find_chimeric_pits(r); }
// Pile::to_json; returns the length, copies at most cap bytes
uint64_t ora_pile_to_json(void* p, uint64_t r, char* out, uint64_t cap) {
    Handle* h = (Handle*)p;
    if (!h->d.bk.alive(r)) return 0;
    const std::string s = h->d.bk.to_json(r);
    if (out) memcpy(out, s.data(), s.size() < cap ? s.size() : cap);
    return s.size();
}

void ora_pile_find_repetitive_hills(void* p, uint64_t r, uint16_t med) {
    ((Handle*)p)->d.bk.find_repetitive_hills(r, med);
}
int ora_pile_break_over_chimeric_pits(void* p, uint64_t r, uint16_t med) {
    return ((Handle*)p)->d.bk.break_over_chimeric_pits(r, med) ? 1 : 0;
}
int ora_pile_break_over_chimeric_hills(void* p, uint64_t r) {
    return ((Handle*)p)->d.bk.break_over_chimeric_hills(r) ? 1 : 0;
}
// slope regions as (key, last) pairs, key = first<<1 | is_up
uint64_t ora_pile_find_slopes(void* p, uint64_t r, double q, uint32_t* out, uint64_t cap) {
    std::vector<ora::Iv> R = ((Handle*)p)->d.bk.find_slopes(r, q);
    for (size_t i = 0; i < R.size() && i < cap; ++i) { out[2 * i] = R[i].first; out[2 * i + 1] = R[i].second; }
    return R.size();
}
// in/out: iv holds n (first, second) pairs; returns the merged count
uint64_t ora_interval_merge(uint32_t* iv, uint64_t n) {
    std::vector<ora::Iv> v;
    for (uint64_t i = 0; i < n; ++i) v.push_back(ora::Iv(iv[2 * i], iv[2 * i + 1]));
    ora::interval_merge(v);
    for (size_t i = 0; i < v.size(); ++i) { iv[2 * i] = v[i].first; iv[2 * i + 1] = v[i].second; }
    return v.size();
}

// One overlap against the current piles: trim (returns 0 if dropped) then
// type.  coords: {a_begin, a_end, b_begin, b_end, length} in/out.
int ora_overlap_trim_type(void* p, uint32_t a, uint32_t b, uint32_t strand, uint32_t* coords, int* type_out) {
    Drv& d = ((Handle*)p)->d;
    if (!d.bk.alive(a) || !d.bk.alive(b)) return 0;
    Backend::OvlH h = d.bk.make_ovl(a, b, coords[0], coords[1], d.read_len[a], coords[2], coords[3],
                                    d.read_len[b], coords[4], strand);
    if (h == nullptr) return 0;
    int ok = d.bk.trim(h) ? 1 : 0;
    if (ok) {
        coords[0] = d.bk.a_begin(h); coords[1] = d.bk.a_end(h);
        coords[2] = d.bk.b_begin(h); coords[3] = d.bk.b_end(h);
        coords[4] = d.bk.length(h);
        *type_out = d.bk.type(h);
    }
    d.bk.free_ovl(h);
    return ok;
}

// ---- overlap lists -----------------------------------------------------------
// which: 0 = overlaps, 1 = internals.  Arrays may be null (count only).
uint64_t ora_get_overlaps(void* p, int which, uint64_t* src, uint32_t* a_begin, uint32_t* a_end,
                          uint32_t* b_begin, uint32_t* b_end, uint32_t* length, uint8_t* type) {
    Drv& d = ((Handle*)p)->d;
    std::vector<Drv::Item>& v = which == 0 ? d.overlaps : d.internals;
    if (src) {
        for (size_t k = 0; k < v.size(); ++k) {
            src[k] = v[k].src;
            a_begin[k] = d.bk.a_begin(v[k].h); a_end[k] = d.bk.a_end(v[k].h);
            b_begin[k] = d.bk.b_begin(v[k].h); b_end[k] = d.bk.b_end(v[k].h);
            length[k] = d.bk.length(v[k].h);
            // internals can outlive their piles (graph.cpp:849-867 never re-checks them)
            const bool live = d.bk.alive(d.bk.a_id(v[k].h)) && d.bk.alive(d.bk.b_id(v[k].h));
            type[k] = live ? (uint8_t)d.bk.type(v[k].h) : (uint8_t)255;
        }
    }
    return v.size();
}

// ---- graph ---------------------------------------------------------------------
uint64_t ora_n_nodes(void* p) { return ((Handle*)p)->d.node_read.size(); }
uint64_t ora_n_edges(void* p) { return ((Handle*)p)->d.edges.size(); }
void ora_get_nodes(void* p, uint32_t* node_read) {
    Drv& d = ((Handle*)p)->d;
    memcpy(node_read, d.node_read.data(), d.node_read.size() * sizeof(uint32_t));
}
void ora_get_edges(void* p, uint32_t* src, uint32_t* dst, uint32_t* len, uint8_t* marked) {
    Drv& d = ((Handle*)p)->d;
    for (size_t e = 0; e < d.edges.size(); ++e) {
        src[e] = d.edges[e].src; dst[e] = d.edges[e].dst; len[e] = d.edges[e].len;
        marked[e] = d.edge_marked[e];
    }
}
// load an arbitrary graph for unit tests of the transitive reduction
void ora_set_graph(void* p, uint64_t n_nodes, uint64_t n_edges, const uint32_t* src, const uint32_t* dst,
                   const uint32_t* len) {
    Drv& d = ((Handle*)p)->d;
    d.node_read.assign(n_nodes, 0);
    d.edges.resize(n_edges);
    for (uint64_t e = 0; e < n_edges; ++e) { d.edges[e].src = src[e]; d.edges[e].dst = dst[e]; d.edges[e].len = len[e]; }
    d.edge_marked.assign(n_edges, 0);
}

}  // extern "C"
