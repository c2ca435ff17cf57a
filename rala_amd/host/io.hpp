// Minimal readers for the formats the reference takes through its (un-vendored)
// bioparser submodule: FASTA / FASTQ sequences, PAF / MHAP overlaps, plain or gzip.
// Names are cut at the first whitespace.
#pragma once

#include <stdint.h>
#include <functional>
#include <string>

namespace rala {
namespace io {

// name, bases
typedef std::function<void(const std::string&, const std::string&)> SequenceSink;
bool read_fasta(const std::string& path, const SequenceSink& sink);
bool read_fastq(const std::string& path, const SequenceSink& sink);

struct PafRecord {
    std::string q_name, t_name;
    uint32_t q_length, q_begin, q_end, t_length, t_begin, t_end, matching_bases, overlap_length, quality;
    char orientation;
};
struct MhapRecord {
    uint64_t a_id, b_id;
    double error;
    uint32_t minmers, a_rc, a_begin, a_end, a_length, b_rc, b_begin, b_end, b_length;
};
bool read_paf(const std::string& path, const std::function<void(const PafRecord&)>& sink);
bool read_mhap(const std::string& path, const std::function<void(const MhapRecord&)>& sink);

bool has_suffix(const std::string& src, const std::string& suffix);

}  // namespace io
}  // namespace rala
