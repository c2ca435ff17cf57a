#include "pile.hpp"

#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <sstream>

#include "../csrc/geom.h"
#include "overlap.hpp"
#include "rala_hip.h"

namespace rala {

namespace {

void die(const char* where, rala_hip_ctx* ctx) {
    fprintf(stderr, "[rala::Pile::%s] error: %s!\n", where, ctx ? rala_hip_last_error(ctx) : "no HIP device");
    exit(1);
}

}  // namespace

std::unique_ptr<Pile> createPile(uint64_t id, uint32_t read_length) {
    return std::unique_ptr<Pile>(new Pile(id, read_length));
}

Pile::Pile(uint64_t id, uint32_t read_length)
        : id_(id), length_(read_length), begin_(0), end_(read_length), p10_(0), median_(0), data_(),
        data_fetched_(false), repeat_hills_(), repeat_hill_coverage_(), chimeric_pits_(), chimeric_pit_min_(),
        chimeric_hills_(), chimeric_hill_coverage_(), ctx_(nullptr), owns_ctx_(false), ctx_read_(0),
        pending_bounds_(), computed_(false), dev_alive_(false), dev_begin_(0), dev_end_(0), dev_median_(0),
        dev_p10_(0) {
}

Pile::~Pile() {
    if (owns_ctx_ && ctx_) rala_hip_destroy(ctx_);
}

// Stand-alone pile: a one-read context; the fused kernel computes the valid region, the
// order statistics, the pits and the hills at once (Graph::initialize order).
void Pile::run_device() {
    if (computed_) return;
    if (ctx_ == nullptr) {
        if (rala_hip_create(0, &ctx_) != RALA_HIP_OK) die("run_device", nullptr);
        owns_ctx_ = true;
        ctx_read_ = 0;
        if (rala_hip_set_reads(ctx_, &length_, 1) != RALA_HIP_OK) die("run_device", ctx_);
    }
    if (!owns_ctx_) {       // views of a Graph's context are computed by the Graph
        computed_ = true;
        return;
    }
    std::vector<uint64_t> tuples(pending_bounds_.size());      // {read 0, bound}
    for (size_t k = 0; k < tuples.size(); ++k) tuples[k] = (uint64_t)pending_bounds_[k] << 32;
    if (rala_hip_set_bound_tuples(ctx_, tuples.data(), tuples.size(), RALA_HIP_MEM_HOST) != RALA_HIP_OK) {
        die("run_device", ctx_);
    }
    const int rc = rala_hip_initialize(ctx_);
    if (rc != RALA_HIP_OK && rc != RALA_HIP_EFILTERED) die("run_device", ctx_);
    uint8_t alive = 0;
    rala_hip_get_piles(ctx_, &dev_begin_, &dev_end_, &dev_median_, &dev_p10_, &alive);
    dev_alive_ = alive != 0;
    dev_pits_.clear(); dev_hills_.clear(); dev_pit_min_.clear();
    if (dev_alive_) {
        for (int kind = 0; kind < 2; ++kind) {
            uint64_t offs[2] = {0, 0};
            rala_hip_get_intervals(ctx_, kind, offs, nullptr, nullptr);
            std::vector<uint32_t> pairs(2 * offs[1] + 2), aux(offs[1] + 1);
            rala_hip_get_intervals(ctx_, kind, offs, pairs.data(), aux.data());
            for (uint64_t k = 0; k < offs[1]; ++k) {
                if (kind == 0) {
                    dev_pits_.emplace_back(pairs[2 * k], pairs[2 * k + 1]);
                    dev_pit_min_.push_back((uint16_t)aux[k]);
                } else {
                    dev_hills_.emplace_back(pairs[2 * k], pairs[2 * k + 1]);
                }
            }
        }
    }
    computed_ = true;
}

const std::vector<uint16_t>& Pile::data() const {
    if (!data_fetched_) {
        data_.assign(length_, 0);
        if (mg_ != nullptr && computed_) {
            if (rala_hip_mg_get_pile_data(mg_, ctx_read_, data_.data()) != RALA_HIP_OK) {
                fprintf(stderr, "[rala::Pile::data] error: %s!\n", rala_hip_mg_last_error(mg_));
                exit(1);
            }
        } else if (ctx_ != nullptr && computed_) {
            if (rala_hip_get_pile_data(ctx_, ctx_read_, data_.data()) != RALA_HIP_OK) die("data", ctx_);
            // Pile::shrink zeroes outside the valid region (reference src/pile.cpp:311-318)
            for (uint32_t i = 0; i < begin_ && i < length_; ++i) data_[i] = 0;
            for (uint32_t i = end_; i < length_; ++i) data_[i] = 0;
        }
        data_fetched_ = true;
    }
    return data_;
}

void Pile::clear() {
    pending_bounds_.clear();
    computed_ = false;
    data_fetched_ = false;
}

void Pile::add_layers(std::vector<uint32_t>& overlap_bounds) {
    if (overlap_bounds.empty()) return;
    std::sort(overlap_bounds.begin(), overlap_bounds.end());
    pending_bounds_.insert(pending_bounds_.end(), overlap_bounds.begin(), overlap_bounds.end());
    computed_ = false;
    data_fetched_ = false;
}

bool Pile::shrink(uint32_t begin, uint32_t end) {
    if (begin > end) {
        fprintf(stderr, "[rala::Pile::shrink] error: invalid begin, end coordinates!\n");
        exit(1);
    }
    if (end - begin < rala_hip::kMinRegion) return false;
    begin_ = begin;
    end_ = end;
    data_fetched_ = false;
    return true;
}

bool Pile::find_valid_region() {
    run_device();
    if (!owns_ctx_) return true;
    if (!dev_alive_) return false;
    return shrink(dev_begin_, dev_end_);
}

void Pile::find_median() {
    run_device();
    if (owns_ctx_) { median_ = dev_median_; p10_ = dev_p10_; }
}

void Pile::find_chimeric_pits() {
    run_device();
    if (owns_ctx_) { chimeric_pits_ = dev_pits_; chimeric_pit_min_ = dev_pit_min_; }
}

void Pile::find_chimeric_hills() {
    run_device();
    if (owns_ctx_) {
        chimeric_hills_ = dev_hills_;
        chimeric_hill_coverage_.assign(chimeric_hills_.size(), 0);
    }
}

// reference src/pile.cpp:366-402; the kernel recorded the minimum coverage inside each pit.  Which piece of the read is
// left: longest_piece (rala_amd/csrc/geom.h), shared with the device's version of this function
bool Pile::break_over_chimeric_pits(uint16_t dataset_median) {
    std::vector<std::pair<uint32_t, uint32_t>> unreal;
    std::vector<uint16_t> unreal_min;
    const rala_hip::Piece keep = rala_hip::longest_piece(begin_, end_, (uint32_t)chimeric_pits_.size(),
        [&](uint32_t k, uint32_t& first, uint32_t& second) { first = chimeric_pits_[k].first; second = chimeric_pits_[k].second; },
        [&](uint32_t k) {
            if ((double)chimeric_pit_min_[k] * 1.84 <= (double)dataset_median) return true;
            unreal.push_back(chimeric_pits_[k]);
            unreal_min.push_back(chimeric_pit_min_[k]);
            return false;
        });
    chimeric_pits_.swap(unreal);
    chimeric_pit_min_.swap(unreal_min);
    return shrink(keep.begin, keep.end);
}

// reference src/pile.cpp:457-469 (begin_ is added to untrimmed coordinates there too)
void Pile::check_chimeric_hills(const std::unique_ptr<Overlap>& overlap) {
    const bool is_a = overlap->a_id() == id_;
    const uint32_t begin = begin_ + (is_a ? overlap->a_begin() : overlap->b_begin());
    const uint32_t end = begin_ + (is_a ? overlap->a_end() : overlap->b_end());
    for (size_t i = 0; i < chimeric_hills_.size(); ++i) {
        if (begin < chimeric_hills_[i].first && end > chimeric_hills_[i].second) ++chimeric_hill_coverage_[i];
    }
}

// reference src/pile.cpp:471-498: a hill that more than three overlaps span is no chimera
bool Pile::break_over_chimeric_hills() {
    const rala_hip::Piece keep = rala_hip::longest_piece(begin_, end_, (uint32_t)chimeric_hills_.size(),
        [&](uint32_t i, uint32_t& first, uint32_t& second) { first = chimeric_hills_[i].first; second = chimeric_hills_[i].second; },
        [&](uint32_t i) { return chimeric_hill_coverage_[i] <= 3; });
    std::vector<std::pair<uint32_t, uint32_t>>().swap(chimeric_hills_);
    std::vector<uint32_t>().swap(chimeric_hill_coverage_);
    return shrink(keep.begin, keep.end);
}

// reference src/pile.cpp:500-566.  Inside Graph::construct the sensitive pass computes the hills of all
// reads at once; a stand-alone pile asks the device for its own (rala_hip_find_repetitive_hills)
void Pile::find_repetitive_hills(uint16_t dataset_median) {
    if (!owns_ctx_ && ctx_ != nullptr) return;       // a Graph's view: already annotated by the Graph
    run_device();
    if (ctx_ == nullptr || !dev_alive_) return;
    if (rala_hip_find_repetitive_hills(ctx_, ctx_read_, begin_, end_, median_, p10_, dataset_median) != RALA_HIP_OK) {
        die("find_repetitive_hills", ctx_);
    }
    uint64_t offs[2] = {0, 0};
    rala_hip_get_intervals(ctx_, 2, offs, nullptr, nullptr);
    std::vector<uint32_t> pairs(2 * offs[1] + 2), aux(offs[1] + 1);
    rala_hip_get_intervals(ctx_, 2, offs, pairs.data(), aux.data());
    repeat_hills_.clear();
    repeat_hill_coverage_.clear();
    for (uint64_t k = 0; k < offs[1]; ++k) {
        repeat_hills_.emplace_back(pairs[2 * k], pairs[2 * k + 1]);
        repeat_hill_coverage_.push_back(false);
    }
}

// reference src/pile.cpp:568-592
void Pile::check_repetitive_hills(const std::unique_ptr<Overlap>& overlap) {
    const uint32_t begin = overlap->b_begin(), end = overlap->b_end(), fuzz = rala_hip::kHillFuzz;
    for (size_t i = 0; i < repeat_hills_.size(); ++i) {
        const auto& h = repeat_hills_[i];
        if (!(begin < h.second && h.first < end)) continue;
        if (h.first < 0.1 * (end_ - begin_) + begin_ && begin - begin_ < end_ - end) {
            if (end >= h.second + fuzz) repeat_hill_coverage_[i] = true;
        } else if (h.second > 0.9 * (end_ - begin_) + begin_ && begin - begin_ > end_ - end) {
            if (begin + fuzz <= h.first) repeat_hill_coverage_[i] = true;
        }
    }
}

void Pile::add_repetitive_region(uint32_t begin, uint32_t end) {
    if (begin > length_ || end > length_) {
        fprintf(stderr, "[rala::Pile::add_repetitive_region] error: [begin,end] out of bounds!\n");
        exit(1);
    }
    repeat_hills_.emplace_back(begin, end);
    repeat_hill_coverage_.push_back(false);
}

// reference src/pile.cpp:605-630
bool Pile::is_valid_overlap(uint32_t begin, uint32_t end) const {
    const uint32_t fuzz = rala_hip::kHillFuzz;
    for (size_t i = 0; i < repeat_hills_.size(); ++i) {
        const auto& it = repeat_hills_[i];
        if (!(begin < it.second && it.first < end)) continue;
        if (it.first < 0.1 * (end_ - begin_) + begin_) {
            if (end < it.second + fuzz && repeat_hill_coverage_[i]) return false;
        } else if (it.second > 0.9 * (end_ - begin_) + begin_) {
            if (begin + fuzz > it.first && repeat_hill_coverage_[i]) return false;
        }
    }
    return true;
}

// reference src/pile.cpp:632-663
std::string Pile::to_json() const {
    const std::vector<uint16_t>& d = data();
    std::stringstream ss;
    ss << "\"" << id_ << "\":{";
    ss << "\"y\":[";
    for (uint32_t i = 0; i < d.size(); ++i) {
        ss << d[i];
        if (i + 1 < d.size()) ss << ",";
    }
    ss << "],";
    ss << "\"b\":" << begin_ << ",";
    ss << "\"e\":" << end_ << ",";
    ss << "\"h\":[";
    for (uint32_t i = 0; i < repeat_hills_.size(); ++i) {
        ss << repeat_hills_[i].first << "," << repeat_hills_[i].second;
        if (i + 1 < repeat_hills_.size()) ss << ",";
    }
    ss << "],";
    ss << "\"m\":" << median_ << ",";
    ss << "\"p10\":" << p10_;
    ss << "}";
    return ss.str();
}

}  // namespace rala
