#include "io.hpp"

#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <vector>

namespace rala {
namespace io {

namespace {

// line reader over gzFile (transparent for uncompressed files)
class Lines {
public:
    explicit Lines(const std::string& path) : f_(gzopen(path.c_str(), "rb")), buf_(1 << 16), pos_(0), len_(0) {
        if (f_) gzbuffer(f_, 1 << 20);
    }
    ~Lines() { if (f_) gzclose(f_); }
    bool ok() const { return f_ != nullptr; }
    bool next(std::string& line) {
        line.clear();
        for (;;) {
            if (pos_ == len_) {
                const int n = gzread(f_, buf_.data(), (unsigned)buf_.size());
                if (n <= 0) return !line.empty();
                pos_ = 0; len_ = (size_t)n;
            }
            const char* p = (const char*)memchr(buf_.data() + pos_, '\n', len_ - pos_);
            if (p) {
                line.append(buf_.data() + pos_, p - (buf_.data() + pos_));
                pos_ = (size_t)(p - buf_.data()) + 1;
                if (!line.empty() && line.back() == '\r') line.pop_back();
                return true;
            }
            line.append(buf_.data() + pos_, len_ - pos_);
            pos_ = len_;
        }
    }
private:
    gzFile f_;
    std::vector<char> buf_;
    size_t pos_, len_;
};

std::string first_token(const std::string& s, size_t from) {
    size_t e = from;
    while (e < s.size() && s[e] != ' ' && s[e] != '\t') ++e;
    return s.substr(from, e - from);
}

void split_tabs(const std::string& line, std::vector<std::string>& out, char sep) {
    out.clear();
    size_t b = 0;
    while (b <= line.size()) {
        size_t e = line.find(sep, b);
        if (e == std::string::npos) e = line.size();
        out.push_back(line.substr(b, e - b));
        b = e + 1;
    }
}

}  // namespace

bool has_suffix(const std::string& src, const std::string& suffix) {
    return src.size() >= suffix.size() && src.compare(src.size() - suffix.size(), suffix.size(), suffix) == 0;
}

bool read_fasta(const std::string& path, const SequenceSink& sink) {
    Lines in(path);
    if (!in.ok()) return false;
    std::string line, name, data;
    bool have = false;
    while (in.next(line)) {
        if (!line.empty() && line[0] == '>') {
            if (have) sink(name, data);
            name = first_token(line, 1);
            data.clear();
            have = true;
        } else if (have) {
            data += line;
        }
    }
    if (have) sink(name, data);
    return true;
}

bool read_fastq(const std::string& path, const SequenceSink& sink) {
    Lines in(path);
    if (!in.ok()) return false;
    std::string head, data, plus, qual;
    while (in.next(head)) {
        if (head.empty()) continue;
        if (!in.next(data) || !in.next(plus)) break;
        // multi-line records: bases until '+', qualities until their length matches
        while (!plus.empty() && plus[0] != '+') {
            data += plus;
            if (!in.next(plus)) break;
        }
        qual.clear();
        std::string q;
        while (qual.size() < data.size() && in.next(q)) qual += q;
        sink(first_token(head, 1), data);
    }
    return true;
}

bool read_paf(const std::string& path, const std::function<void(const PafRecord&)>& sink) {
    Lines in(path);
    if (!in.ok()) return false;
    std::string line;
    std::vector<std::string> f;
    while (in.next(line)) {
        if (line.empty()) continue;
        split_tabs(line, f, '\t');
        if (f.size() < 12) continue;
        PafRecord r;
        r.q_name = first_token(f[0], 0);
        r.q_length = (uint32_t)strtoul(f[1].c_str(), nullptr, 10);
        r.q_begin = (uint32_t)strtoul(f[2].c_str(), nullptr, 10);
        r.q_end = (uint32_t)strtoul(f[3].c_str(), nullptr, 10);
        r.orientation = f[4].empty() ? '+' : f[4][0];
        r.t_name = first_token(f[5], 0);
        r.t_length = (uint32_t)strtoul(f[6].c_str(), nullptr, 10);
        r.t_begin = (uint32_t)strtoul(f[7].c_str(), nullptr, 10);
        r.t_end = (uint32_t)strtoul(f[8].c_str(), nullptr, 10);
        r.matching_bases = (uint32_t)strtoul(f[9].c_str(), nullptr, 10);
        r.overlap_length = (uint32_t)strtoul(f[10].c_str(), nullptr, 10);
        r.quality = (uint32_t)strtoul(f[11].c_str(), nullptr, 10);
        sink(r);
    }
    return true;
}

bool read_mhap(const std::string& path, const std::function<void(const MhapRecord&)>& sink) {
    Lines in(path);
    if (!in.ok()) return false;
    std::string line;
    std::vector<std::string> f;
    while (in.next(line)) {
        if (line.empty()) continue;
        split_tabs(line, f, ' ');
        if (f.size() < 12) continue;
        MhapRecord r;
        r.a_id = strtoull(f[0].c_str(), nullptr, 10);
        r.b_id = strtoull(f[1].c_str(), nullptr, 10);
        r.error = strtod(f[2].c_str(), nullptr);
        r.minmers = (uint32_t)strtoul(f[3].c_str(), nullptr, 10);
        r.a_rc = (uint32_t)strtoul(f[4].c_str(), nullptr, 10);
        r.a_begin = (uint32_t)strtoul(f[5].c_str(), nullptr, 10);
        r.a_end = (uint32_t)strtoul(f[6].c_str(), nullptr, 10);
        r.a_length = (uint32_t)strtoul(f[7].c_str(), nullptr, 10);
        r.b_rc = (uint32_t)strtoul(f[8].c_str(), nullptr, 10);
        r.b_begin = (uint32_t)strtoul(f[9].c_str(), nullptr, 10);
        r.b_end = (uint32_t)strtoul(f[10].c_str(), nullptr, 10);
        r.b_length = (uint32_t)strtoul(f[11].c_str(), nullptr, 10);
        sink(r);
    }
    return true;
}

}  // namespace io
}  // namespace rala
