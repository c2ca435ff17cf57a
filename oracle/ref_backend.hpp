// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_core.hpp).
//
// Backend for oracle_graph.hpp's Driver that runs on the REAL reference
// objects: rala::Pile and rala::Overlap, compiled unmodified from
// /root/reference/src/pile.cpp and overlap.cpp (oracle/Makefile target
// `ref`; outputs only under oracle/_ref/).  No reference source is copied
// into this repository; this header only includes the reference's own
// headers from where they lie.
//
// Private members (Overlap's parser-only constructor, Pile::find_slopes,
// pile state for crafted-data unit tests) are reached by re-declaring the
// access keyword for this translation unit only; class layout is unchanged.
#pragma once

#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#define private public
#include "pile.hpp"      // -I/root/reference/src
#include "overlap.hpp"
#undef private

#include "oracle_core.hpp"   // Iv, enum values only

namespace ora {

struct RefBackend {
    typedef rala::Overlap* OvlH;
    std::vector<std::unique_ptr<rala::Pile>> piles;
    std::unordered_map<std::string, uint64_t> name_to_id;
    std::vector<std::string> names;

    static const char* name() { return "reference-objects"; }

    void create_piles(const uint32_t* len, uint64_t n) {
        piles.clear();
        names.resize(n);
        for (uint64_t i = 0; i < n; ++i) {
            piles.emplace_back(rala::createPile(i, len[i]));       // graph.cpp:257
            names[i] = "r" + std::to_string(i);
            name_to_id[names[i]] = i;                              // graph.cpp:256
        }
    }
    bool alive(uint64_t r) const { return piles[r] != nullptr; }
    void kill(uint64_t r) { piles[r].reset(); }
    uint32_t begin(uint64_t r) const { return piles[r]->begin(); }
    uint32_t end(uint64_t r) const { return piles[r]->end(); }
    uint16_t median(uint64_t r) const { return piles[r]->median(); }
    uint16_t p10(uint64_t r) const { return piles[r]->p10(); }
    const std::vector<uint16_t>& data(uint64_t r) const { return piles[r]->data(); }
    const std::vector<Iv>& pits(uint64_t r) const { return piles[r]->chimeric_pits_; }
    const std::vector<Iv>& hills(uint64_t r) const { return piles[r]->chimeric_hills_; }
    const std::vector<uint32_t>& hill_cnt(uint64_t r) const { return piles[r]->chimeric_hill_coverage_; }
    const std::vector<Iv>& rep_hills(uint64_t r) const { return piles[r]->repeat_hills_; }
    std::vector<uint8_t> rep_flag(uint64_t r) const {
        std::vector<uint8_t> f;
        for (bool b : piles[r]->repeat_hill_coverage_) f.push_back(b ? 1 : 0);
        return f;
    }

    void set_pile_state(uint64_t r, const uint16_t* d, uint32_t n, uint32_t b, uint32_t e) {
        piles[r]->data_.assign(d, d + n);
        piles[r]->begin_ = b;
        piles[r]->end_ = e;
    }

    void add_layers(uint64_t r, std::vector<uint32_t>& b) { piles[r]->add_layers(b); }
    bool find_valid_region(uint64_t r) { return piles[r]->find_valid_region(); }
    void find_median(uint64_t r) { piles[r]->find_median(); }
    void find_chimeric_hills(uint64_t r) { piles[r]->find_chimeric_hills(); }
    void find_chimeric_pits(uint64_t r) { piles[r]->find_chimeric_pits(); }
    std::vector<Iv> find_slopes(uint64_t r, double q) { return piles[r]->find_slopes(q); }
    bool has_hill(uint64_t r) const { return piles[r]->has_chimeric_hill(); }
    bool has_chimeric_region(uint64_t r) const { return piles[r]->has_chimeric_region(); }
    bool has_rep_hills(uint64_t r) const { return piles[r]->has_repetitive_hills(); }
    void check_chimeric_hills(uint64_t r, OvlH h) {
        std::unique_ptr<rala::Overlap> tmp(h);
        piles[r]->check_chimeric_hills(tmp);
        tmp.release();
    }
    bool break_over_chimeric_hills(uint64_t r) { return piles[r]->break_over_chimeric_hills(); }
    bool break_over_chimeric_pits(uint64_t r, uint16_t m) { return piles[r]->break_over_chimeric_pits(m); }
    void find_repetitive_hills(uint64_t r, uint16_t m) { piles[r]->find_repetitive_hills(m); }
    void check_repetitive_hills(uint64_t r, OvlH h) {
        std::unique_ptr<rala::Overlap> tmp(h);
        piles[r]->check_repetitive_hills(tmp);
        tmp.release();
    }
    bool is_valid_overlap(uint64_t r, uint32_t x, uint32_t y) const { return piles[r]->is_valid_overlap(x, y); }
    std::string to_json(uint64_t r) const { return piles[r]->to_json(); }          // pile.cpp:632-663

    // PAF constructor (overlap.cpp:22-31) followed by the reference's own
    // Overlap::transmute (overlap.cpp:36-82)
    OvlH make_ovl(uint32_t a, uint32_t b, uint32_t ab, uint32_t ae, uint32_t al, uint32_t bb, uint32_t be,
                  uint32_t bl, uint32_t length, uint32_t strand) {
        rala::Overlap* o = new rala::Overlap(names[a].c_str(), (uint32_t)names[a].size(), al, ab, ae,
                                             strand ? '-' : '+', names[b].c_str(), (uint32_t)names[b].size(), bl,
                                             bb, be, 0, length, 255);
        if (!o->transmute(piles, name_to_id)) {
            delete o;
            return nullptr;
        }
        return o;
    }
    // PAF constructor + Overlap::transmute_ (overlap.cpp:84-114)
    OvlH make_sensitive(uint32_t a, uint32_t b, uint32_t ab, uint32_t ae, uint32_t al, uint32_t bb, uint32_t be,
                        uint32_t length, uint32_t strand) {
        rala::Overlap* o = new rala::Overlap(names[a].c_str(), (uint32_t)names[a].size(), al, ab, ae,
                                             strand ? '-' : '+', names[b].c_str(), (uint32_t)names[b].size(), 0,
                                             bb, be, 0, length, 255);
        o->transmute_(piles, name_to_id);
        return o;
    }
    void free_ovl(OvlH h) { delete h; }
    bool trim(OvlH h) { return h->trim(piles); }
    int type(OvlH h) const {
        switch (h->type(piles)) {
            case rala::OverlapType::kX: return kX;
            case rala::OverlapType::kA: return kA;
            case rala::OverlapType::kB: return kB;
            case rala::OverlapType::kAB: return kAB;
            default: return kBA;
        }
    }
    uint32_t a_id(OvlH h) const { return h->a_id(); }
    uint32_t b_id(OvlH h) const { return h->b_id(); }
    uint32_t a_begin(OvlH h) const { return h->a_begin(); }
    uint32_t a_end(OvlH h) const { return h->a_end(); }
    uint32_t b_begin(OvlH h) const { return h->b_begin(); }
    uint32_t b_end(OvlH h) const { return h->b_end(); }
    uint32_t length(OvlH h) const { return h->length(); }
    uint32_t strand(OvlH h) const { return h->orientation(); }
};

}  // namespace ora
