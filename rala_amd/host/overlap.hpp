/*!
 * @file overlap.hpp
 *
 * @brief Overlap class (interface of rvaser/rala src/overlap.hpp:27-117).  The geometry is
 * the same code the HIP kernels run (rala_amd/csrc/geom.h).
 */

#pragma once

#include <stdint.h>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

namespace rala {

class Pile;
class Graph;

enum class OverlapType {
    kX,   // bad overlap
    kA,   // b contained
    kB,   // a contained
    kAB,  // suffix prefix
    kBA   // prefix suffix
};

class Overlap;
/*! @brief PAF record (reference src/overlap.cpp:22-31) */
std::unique_ptr<Overlap> createOverlap(const std::string& a_name, uint32_t a_length, uint32_t a_begin,
    uint32_t a_end, char orientation, const std::string& b_name, uint32_t b_length, uint32_t b_begin,
    uint32_t b_end, uint32_t overlap_length);
/*! @brief MHAP record, 1-based ids (reference src/overlap.cpp:12-20) */
std::unique_ptr<Overlap> createOverlap(uint64_t a_id, uint64_t b_id, uint32_t a_rc, uint32_t a_begin,
    uint32_t a_end, uint32_t a_length, uint32_t b_rc, uint32_t b_begin, uint32_t b_end, uint32_t b_length);

class Overlap {
public:
    ~Overlap() {}

    uint32_t a_id() const { return a_id_; }
    uint32_t a_begin() const { return a_begin_; }
    uint32_t a_end() const { return a_end_; }
    uint32_t a_length() const { return a_length_; }
    uint32_t b_id() const { return b_id_; }
    uint32_t b_begin() const { return b_begin_; }
    uint32_t b_end() const { return b_end_; }
    uint32_t b_length() const { return b_length_; }
    uint32_t length() const { return length_; }
    uint32_t orientation() const { return orientation_; }

    bool transmute(const std::vector<std::unique_ptr<Pile>>& piles,
        const std::unordered_map<std::string, uint64_t>& name_to_id);
    bool transmute_(const std::vector<std::unique_ptr<Pile>>& piles,
        const std::unordered_map<std::string, uint64_t>& name_to_id);
    bool trim(const std::vector<std::unique_ptr<Pile>>& piles);
    OverlapType type(const std::vector<std::unique_ptr<Pile>>& piles) const;

    friend Graph;
    friend std::unique_ptr<Overlap> createOverlap(const std::string&, uint32_t, uint32_t, uint32_t, char,
        const std::string&, uint32_t, uint32_t, uint32_t, uint32_t);
    friend std::unique_ptr<Overlap> createOverlap(uint64_t, uint64_t, uint32_t, uint32_t, uint32_t, uint32_t,
        uint32_t, uint32_t, uint32_t, uint32_t);

private:
    Overlap() {}
    Overlap(const Overlap&) = delete;
    const Overlap& operator=(const Overlap&) = delete;

    std::string a_name_;
    uint64_t a_id_ = 0;
    uint32_t a_begin_ = 0;
    uint32_t a_end_ = 0;
    uint32_t a_length_ = 0;
    std::string b_name_;
    uint64_t b_id_ = 0;
    uint32_t b_begin_ = 0;
    uint32_t b_end_ = 0;
    uint32_t b_length_ = 0;
    uint32_t length_ = 0;
    uint32_t orientation_ = 0;
    bool is_transmuted_ = false;
};

}  // namespace rala
