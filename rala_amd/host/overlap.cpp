#include "overlap.hpp"

#include <stdio.h>
#include <stdlib.h>

#include <algorithm>

#include "../csrc/geom.h"
#include "pile.hpp"

namespace rala {

std::unique_ptr<Overlap> createOverlap(const std::string& a_name, uint32_t a_length, uint32_t a_begin,
    uint32_t a_end, char orientation, const std::string& b_name, uint32_t b_length, uint32_t b_begin,
    uint32_t b_end, uint32_t overlap_length) {
    std::unique_ptr<Overlap> o(new Overlap());
    o->a_name_ = a_name; o->a_begin_ = a_begin; o->a_end_ = a_end; o->a_length_ = a_length;
    o->b_name_ = b_name; o->b_begin_ = b_begin; o->b_end_ = b_end; o->b_length_ = b_length;
    o->length_ = overlap_length;
    o->orientation_ = orientation == '+' ? 0 : 1;
    return o;
}

std::unique_ptr<Overlap> createOverlap(uint64_t a_id, uint64_t b_id, uint32_t a_rc, uint32_t a_begin,
    uint32_t a_end, uint32_t a_length, uint32_t b_rc, uint32_t b_begin, uint32_t b_end, uint32_t b_length) {
    std::unique_ptr<Overlap> o(new Overlap());
    o->a_id_ = a_id - 1; o->a_begin_ = a_begin; o->a_end_ = a_end; o->a_length_ = a_length;
    o->b_id_ = b_id - 1; o->b_begin_ = b_begin; o->b_end_ = b_end; o->b_length_ = b_length;
    o->length_ = std::max(a_end - a_begin, b_end - b_begin);
    o->orientation_ = a_rc == b_rc ? 0 : 1;
    return o;
}

namespace {

// One end of a record while it is being resolved: PAF records arrive with names, MHAP records with
// ids.  The references point into the Overlap that is transmuted.
struct End {
    std::string& name;
    uint64_t& id;
    uint32_t& length;
};

// name -> id, once; the name's storage is given back (an id is all the graph needs from here on)
bool resolve(End e, const std::unordered_map<std::string, uint64_t>& ids) {
    if (e.name.empty()) return true;
    const auto hit = ids.find(e.name);
    if (hit == ids.end()) return false;
    e.id = hit->second;
    e.name.clear();
    e.name.shrink_to_fit();
    return true;
}

const Pile* pile_of(const std::vector<std::unique_ptr<Pile>>& piles, uint64_t id) {
    return id < piles.size() ? piles[id].get() : nullptr;
}

}  // namespace

// reference src/overlap.cpp:36-82: false = drop the record (unknown name, read filtered out),
// fatal when the overlap file and the sequence file disagree on a read's length
bool Overlap::transmute(const std::vector<std::unique_ptr<Pile>>& piles,
    const std::unordered_map<std::string, uint64_t>& name_to_id) {
    if (is_transmuted_) return true;
    for (End e : {End{a_name_, a_id_, a_length_}, End{b_name_, b_id_, b_length_}}) {
        if (!resolve(e, name_to_id)) return false;
        const Pile* pile = pile_of(piles, e.id);
        if (pile == nullptr) return false;
        if (pile->data().size() != e.length) {
            fprintf(stderr, "[rala::Overlap::transmute] error: "
                "unequal lengths in sequence and overlap file for sequence with id %lu!\n", e.id);
            exit(1);
        }
    }
    is_transmuted_ = true;
    return true;
}

// reference src/overlap.cpp:84-114 (sensitive overlaps): no length check, the target side moves
// from valid-region coordinates to read coordinates
bool Overlap::transmute_(const std::vector<std::unique_ptr<Pile>>& piles,
    const std::unordered_map<std::string, uint64_t>& name_to_id) {
    if (is_transmuted_) return true;
    if (!resolve(End{a_name_, a_id_, a_length_}, name_to_id)) return false;
    if (!resolve(End{b_name_, b_id_, b_length_}, name_to_id)) return false;
    const Pile* target = pile_of(piles, b_id_);
    if (target == nullptr) return false;
    const uint32_t shift = target->begin();
    b_begin_ += shift;
    b_end_ += shift;
    b_length_ = (uint32_t)target->data().size();
    is_transmuted_ = true;
    return true;
}

bool Overlap::trim(const std::vector<std::unique_ptr<Pile>>& piles) {
    if (!is_transmuted_) {
        fprintf(stderr, "[rala::Overlap::trim] error: overlap is not transmuted!\n");
        exit(1);
    }
    if (a_id_ >= piles.size() || piles[a_id_] == nullptr || b_id_ >= piles.size() || piles[b_id_] == nullptr) {
        return false;
    }
    const auto& pa = piles[a_id_];
    const auto& pb = piles[b_id_];
    if (pa->begin() > a_length_ || pa->end() > a_length_ || pb->begin() > b_length_ || pb->end() > b_length_) {
        fprintf(stderr, "[rala::Overlap::trim] error: invalid trimmed begin, end coordinates!\n");
        exit(1);
    }
    rala_hip::Coords c = {a_begin_, a_end_, b_begin_, b_end_, length_};
    if (!rala_hip::ovl_trim(c, orientation_, pa->begin(), pa->end(), pb->begin(), pb->end())) return false;
    a_begin_ = c.a_begin; a_end_ = c.a_end; b_begin_ = c.b_begin; b_end_ = c.b_end; length_ = c.length;
    return true;
}

OverlapType Overlap::type(const std::vector<std::unique_ptr<Pile>>& piles) const {
    if (!is_transmuted_) {
        fprintf(stderr, "[rala::Overlap::type] error: overlap is not transmuted!\n");
        exit(1);
    }
    if (a_id_ >= piles.size() || piles[a_id_] == nullptr || b_id_ >= piles.size() || piles[b_id_] == nullptr) {
        fprintf(stderr, "[rala::Overlap::type] error: missing piles!\n");
        exit(1);
    }
    const rala_hip::Coords c = {a_begin_, a_end_, b_begin_, b_end_, length_};
    switch (rala_hip::ovl_type(c, orientation_, piles[a_id_]->begin(), piles[a_id_]->end(), piles[b_id_]->begin(),
                               piles[b_id_]->end())) {
        case rala_hip::kTypeX: return OverlapType::kX;
        case rala_hip::kTypeA: return OverlapType::kA;
        case rala_hip::kTypeB: return OverlapType::kB;
        case rala_hip::kTypeAB: return OverlapType::kAB;
        default: return OverlapType::kBA;
    }
}

}  // namespace rala
