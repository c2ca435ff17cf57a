// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_core.hpp).
// Backend for oracle_graph.hpp's Driver that runs on the flat restatement.
#pragma once

#include <string>

#include "oracle_core.hpp"

namespace ora {

struct FlatBackend {
    typedef Ovl* OvlH;
    std::vector<Pile> piles;

    static const char* name() { return "flat-restatement"; }

    void create_piles(const uint32_t* len, uint64_t n) {
        piles.resize(n);
        for (uint64_t i = 0; i < n; ++i) piles[i].init(i, len[i]);
    }
    bool alive(uint64_t r) const { return piles[r].alive; }
    void kill(uint64_t r) {
        piles[r].alive = false;
        std::vector<uint16_t>().swap(piles[r].data);
    }
    uint32_t begin(uint64_t r) const { return piles[r].begin; }
    uint32_t end(uint64_t r) const { return piles[r].end; }
    uint16_t median(uint64_t r) const { return piles[r].median; }
    uint16_t p10(uint64_t r) const { return piles[r].p10; }
    const std::vector<uint16_t>& data(uint64_t r) const { return piles[r].data; }
    const std::vector<Iv>& pits(uint64_t r) const { return piles[r].pits; }
    const std::vector<Iv>& hills(uint64_t r) const { return piles[r].hills; }
    const std::vector<uint32_t>& hill_cnt(uint64_t r) const { return piles[r].hill_cnt; }
    const std::vector<Iv>& rep_hills(uint64_t r) const { return piles[r].rep_hills; }
    std::vector<uint8_t> rep_flag(uint64_t r) const { return piles[r].rep_flag; }

    void set_pile_state(uint64_t r, const uint16_t* d, uint32_t n, uint32_t b, uint32_t e) {
        piles[r].data.assign(d, d + n);
        piles[r].begin = b;
        piles[r].end = e;
    }

    void add_layers(uint64_t r, std::vector<uint32_t>& b) { ora::add_layers(piles[r], b); }
    bool find_valid_region(uint64_t r) { return ora::find_valid_region(piles[r]); }
    void find_median(uint64_t r) { ora::find_median(piles[r]); }
    void find_chimeric_hills(uint64_t r) { ora::find_chimeric_hills(piles[r]); }
    void find_chimeric_pits(uint64_t r) { ora::find_chimeric_pits(piles[r]); }
    std::vector<Iv> find_slopes(uint64_t r, double q) { return ora::find_slopes(piles[r], q); }
    bool has_hill(uint64_t r) const { return piles[r].has_hill(); }
    bool has_chimeric_region(uint64_t r) const { return piles[r].has_chimeric_region(); }
    // restatement of Pile::to_json (pile.cpp:632-663): "<id>":{"y":[...],"b":..,"e":..,"h":[f,s,...],"m":..,"p10":..}
    std::string to_json(uint64_t r) const {
        std::string out = "\"" + std::to_string(r) + "\":{\"y\":[";
        const std::vector<uint16_t>& d = data(r);
        for (size_t i = 0; i < d.size(); ++i) { out += std::to_string(d[i]); if (i + 1 < d.size()) out += ","; }
        out += "],\"b\":" + std::to_string(begin(r)) + ",\"e\":" + std::to_string(end(r)) + ",\"h\":[";
        const std::vector<Iv>& h = rep_hills(r);
        for (size_t i = 0; i < h.size(); ++i) {
            out += std::to_string(h[i].first) + "," + std::to_string(h[i].second);
            if (i + 1 < h.size()) out += ",";
        }
        out += "],\"m\":" + std::to_string(median(r)) + ",\"p10\":" + std::to_string(p10(r)) + "}";
        return out;
    }
    bool has_rep_hills(uint64_t r) const { return piles[r].has_rep_hills(); }
    void check_chimeric_hills(uint64_t r, OvlH h) { ora::check_chimeric_hills(piles[r], *h); }
    bool break_over_chimeric_hills(uint64_t r) { return ora::break_over_chimeric_hills(piles[r]); }
    bool break_over_chimeric_pits(uint64_t r, uint16_t m) { return ora::break_over_chimeric_pits(piles[r], m); }
    void find_repetitive_hills(uint64_t r, uint16_t m) { ora::find_repetitive_hills(piles[r], m); }
    void check_repetitive_hills(uint64_t r, OvlH h) { ora::check_repetitive_hills(piles[r], *h); }
    bool is_valid_overlap(uint64_t r, uint32_t x, uint32_t y) const { return ora::is_valid_overlap(piles[r], x, y); }

    // Overlap::Overlap (PAF) + transmute (overlap.cpp:22-31,36-82)
    OvlH make_ovl(uint32_t a, uint32_t b, uint32_t ab, uint32_t ae, uint32_t al, uint32_t bb, uint32_t be,
                  uint32_t bl, uint32_t length, uint32_t strand) {
        Ovl* o = new Ovl;
        o->a_id = a; o->b_id = b;
        o->a_begin = ab; o->a_end = ae; o->a_len = al;
        o->b_begin = bb; o->b_end = be; o->b_len = bl;
        o->length = length; o->strand = strand ? 1 : 0;
        return o;
    }
    // Overlap::transmute_ (overlap.cpp:84-114): target coordinates shifted
    // by the target pile's begin
    OvlH make_sensitive(uint32_t a, uint32_t b, uint32_t ab, uint32_t ae, uint32_t al, uint32_t bb, uint32_t be,
                        uint32_t length, uint32_t strand) {
        Ovl* o = make_ovl(a, b, ab, ae, al, bb, be, 0, length, strand);
        o->b_begin += piles[b].begin;
        o->b_end += piles[b].begin;
        o->b_len = (uint32_t)piles[b].data.size();
        return o;
    }
    void free_ovl(OvlH h) { delete h; }
    bool trim(OvlH h) {
        return ora::ovl_trim(*h, &piles[h->a_id], &piles[h->b_id]);
    }
    int type(OvlH h) const { return (int)ora::ovl_type(*h, piles[h->a_id], piles[h->b_id]); }
    uint32_t a_id(OvlH h) const { return h->a_id; }
    uint32_t b_id(OvlH h) const { return h->b_id; }
    uint32_t a_begin(OvlH h) const { return h->a_begin; }
    uint32_t a_end(OvlH h) const { return h->a_end; }
    uint32_t b_begin(OvlH h) const { return h->b_begin; }
    uint32_t b_end(OvlH h) const { return h->b_end; }
    uint32_t length(OvlH h) const { return h->length; }
    uint32_t strand(OvlH h) const { return h->strand; }
};

}  // namespace ora
