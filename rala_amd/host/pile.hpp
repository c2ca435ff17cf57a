/*!
 * @file pile.hpp
 *
 * @brief Pile class (interface of rvaser/rala src/pile.hpp:19-170).
 *
 * In this build a pile is computed on the GPU (librala_hip): createPile() +
 * add_layers() only collect the overlap bounds; the first of find_valid_region() /
 * find_median() / find_chimeric_hills() / find_chimeric_pits() runs the fused pile kernel
 * once (Graph::initialize order, reference src/graph.cpp:387-407) and each method then
 * publishes its part of the result.  Piles owned by a Graph are views of the Graph's
 * context.  There is no CPU implementation behind these methods.
 *
 * add_layers() may be called several times; every call must hold whole overlaps (a begin
 * bound and its end bound), which is what Graph::store_overlap_bounds produces
 * (reference src/graph.cpp:311-326).  The reference sweeps each call on its own, so a
 * call with unmatched bounds gives a different pile there than the union does here.
 */

#pragma once

#include <stdint.h>
#include <memory>
#include <string>
#include <utility>
#include <vector>

struct rala_hip_ctx;
struct rala_hip_mg;

namespace rala {

class Overlap;
class Graph;

class Pile;
std::unique_ptr<Pile> createPile(uint64_t id, uint32_t sequence_length);

class Pile {
public:
    ~Pile();

    uint64_t id() const { return id_; }
    /*! @brief begin_ of the valid interval [begin_, end_> */
    uint32_t begin() const { return begin_; }
    uint32_t end() const { return end_; }
    uint16_t p10() const { return p10_; }
    uint16_t median() const { return median_; }

    void find_median();

    /*! @brief coverage vector; fetched from HBM on first use */
    const std::vector<uint16_t>& data() const;

    void clear();
    void add_layers(std::vector<uint32_t>& overlap_bounds);
    bool shrink(uint32_t begin, uint32_t end);
    bool find_valid_region();
    void find_chimeric_pits();
    bool has_chimeric_pit() const { return !chimeric_pits_.empty(); }
    bool break_over_chimeric_pits(uint16_t dataset_median);
    void find_chimeric_hills();
    bool has_chimeric_hill() const { return !chimeric_hills_.empty(); }
    void check_chimeric_hills(const std::unique_ptr<Overlap>& overlap);
    bool break_over_chimeric_hills();
    bool has_chimeric_region() const { return has_chimeric_hill() || has_chimeric_pit(); }
    void find_repetitive_hills(uint16_t dataset_median);
    bool has_repetitive_hills() const { return !repeat_hills_.empty(); }
    void check_repetitive_hills(const std::unique_ptr<Overlap>& overlap);
    void add_repetitive_region(uint32_t begin, uint32_t end);
    bool is_valid_overlap(uint32_t begin, uint32_t end) const;
    std::string to_json() const;

    friend std::unique_ptr<Pile> createPile(uint64_t id, uint32_t sequence_length);
    friend Graph;

private:
    Pile(uint64_t id, uint32_t sequence_length);
    Pile(const Pile&) = delete;
    const Pile& operator=(const Pile&) = delete;

    void run_device();          // standalone piles: one-read context, fused kernel

    uint64_t id_;
    uint32_t length_;
    uint32_t begin_;
    uint32_t end_;
    uint16_t p10_;
    uint16_t median_;
    mutable std::vector<uint16_t> data_;
    mutable bool data_fetched_;
    std::vector<std::pair<uint32_t, uint32_t>> repeat_hills_;
    std::vector<bool> repeat_hill_coverage_;
    std::vector<std::pair<uint32_t, uint32_t>> chimeric_pits_;
    std::vector<uint16_t> chimeric_pit_min_;
    std::vector<std::pair<uint32_t, uint32_t>> chimeric_hills_;
    std::vector<uint32_t> chimeric_hill_coverage_;

    // device side
    rala_hip_ctx* ctx_;         // Graph's context (view) or own one-read context
    rala_hip_mg* mg_ = nullptr; // multi-GPU Graph: the rank that owns this read's coverage
    bool owns_ctx_;
    uint64_t ctx_read_;         // read number inside ctx_
    std::vector<uint32_t> pending_bounds_;
    bool computed_;
    bool dev_alive_;
    uint32_t dev_begin_, dev_end_;
    uint16_t dev_median_, dev_p10_;
    std::vector<std::pair<uint32_t, uint32_t>> dev_pits_, dev_hills_;
    std::vector<uint16_t> dev_pit_min_;
};

}  // namespace rala
