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

// ---- multi-threaded PAF ingest -----------------------------------------------------------------
#include <fcntl.h>
#include <stdio.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <new>
#include <thread>

namespace rala {
namespace io {

namespace {

inline uint64_t hash_bytes(const char* p, size_t n) {
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) { h ^= (unsigned char)p[i]; h *= 1099511628211ull; }
    return h ^ (h >> 29);
}

inline const char* parse_u32(const char* p, const char* e, uint32_t& v) {
    uint64_t x = 0;
    while (p < e && *p >= '0' && *p <= '9') { x = x * 10 + (uint64_t)(*p - '0'); ++p; }
    v = (uint32_t)x;
    return p;
}

struct alignas(256) Chunk {        // one per thread, appended to on every line: no shared cache lines
    OverlapColumns cols;
    int64_t error_read = -1;        // first line of this chunk with a length mismatch
    // overlap files are grouped by query: the previous line's query name and its id
    const char* last_q = nullptr;
    size_t last_qn = 0;
    uint64_t last_a = ~0ull;
};


// one line [p, e) (no newline); returns false if it is not a 12-column record
inline bool parse_paf_line(const char* p, const char* e, const NameTable& names, const std::vector<uint32_t>& read_len,
                           bool check_lengths, Chunk& c) {
    const char* f[13];
    int nf = 0;
    f[nf++] = p;
    for (const char* q = p; q < e && nf < 13; ++q) {
        if (*q == '\t') f[nf++] = q + 1;
    }
    if (nf < 12) return false;
    auto token_end = [&](int k) {                       // names are cut at the first whitespace
        const char* end = k + 1 < nf ? f[k + 1] - 1 : e;
        const char* q = f[k];
        while (q < end && *q != ' ' && *q != '\t') ++q;
        return q;
    };
    uint32_t ql, qb, qe, tl, tb, te, ol;
    parse_u32(f[1], e, ql); parse_u32(f[2], e, qb); parse_u32(f[3], e, qe);
    parse_u32(f[6], e, tl); parse_u32(f[7], e, tb); parse_u32(f[8], e, te);
    parse_u32(f[10], e, ol);
    const char orientation = f[4] < e && *f[4] != '\t' ? *f[4] : '+';
    const size_t qn = (size_t)(token_end(0) - f[0]);
    if (c.last_q == nullptr || qn != c.last_qn || memcmp(c.last_q, f[0], qn) != 0) {
        c.last_a = names.find(f[0], qn);
        c.last_q = f[0];
        c.last_qn = qn;
    }
    const uint64_t a = c.last_a;
    const uint64_t b = names.find(f[5], (size_t)(token_end(5) - f[5]));
    const uint32_t ia = a == ~0ull ? 0xFFFFFFFFu : (uint32_t)a;
    const uint32_t ib = b == ~0ull ? 0xFFFFFFFFu : (uint32_t)b;
    if (c.error_read < 0) {
        if (check_lengths && ia != 0xFFFFFFFFu && ql != read_len[ia]) c.error_read = ia;
        else if (check_lengths && ia != 0xFFFFFFFFu && ib != 0xFFFFFFFFu && tl != read_len[ib]) c.error_read = ib;
    }
    c.cols.a_id.push_back(ia); c.cols.b_id.push_back(ib);
    c.cols.a_begin.push_back(qb); c.cols.a_end.push_back(qe);
    c.cols.b_begin.push_back(tb); c.cols.b_end.push_back(te);
    c.cols.length.push_back(ol);
    c.cols.strand.push_back(orientation == '+' ? 0 : 1);
    return true;
}

}  // namespace

void NameTable::build(const std::vector<std::string>& names) {
    uint64_t cap = 16;
    while (cap < 2 * names.size() + 2) cap <<= 1;
    mask_ = cap - 1;
    slot_.assign(cap, 0);
    off_.resize(names.size()); len_.resize(names.size());
    arena_.clear();
    for (size_t i = 0; i < names.size(); ++i) {
        off_[i] = (uint32_t)arena_.size();
        len_[i] = (uint32_t)names[i].size();
        arena_ += names[i];
    }
    // a later duplicate name replaces an earlier one, like unordered_map::operator[] in the
    // reference (src/graph.cpp:262)
    for (size_t i = 0; i < names.size(); ++i) {
        uint64_t h = hash_bytes(names[i].data(), names[i].size()) & mask_;
        for (;; h = (h + 1) & mask_) {
            if (slot_[h] == 0) { slot_[h] = i + 1; break; }
            const uint64_t j = slot_[h] - 1;
            if (len_[j] == names[i].size() && memcmp(arena_.data() + off_[j], names[i].data(), len_[j]) == 0) {
                slot_[h] = i + 1;
                break;
            }
        }
    }
}

uint64_t NameTable::find(const char* p, size_t n) const {
    if (slot_.empty()) return ~0ull;
    for (uint64_t h = hash_bytes(p, n) & mask_;; h = (h + 1) & mask_) {
        if (slot_[h] == 0) return ~0ull;
        const uint64_t j = slot_[h] - 1;
        if (len_[j] == n && memcmp(arena_.data() + off_[j], p, n) == 0) return j;
    }
}

namespace {
constexpr size_t kHugeBlock = 1u << 20, kHugePage = 2u << 20;
}

void* allocate_block(size_t bytes) {
    if (bytes >= kHugeBlock && getenv("RALA_IO_NO_HUGEPAGES") == nullptr) {
        const size_t rounded = (bytes + kHugePage - 1) & ~(kHugePage - 1);
        void* p = aligned_alloc(kHugePage, rounded);
        if (!p) throw std::bad_alloc();
        (void)madvise(p, rounded, MADV_HUGEPAGE);
        return p;
    }
    void* p = malloc(bytes ? bytes : 1);
    if (!p) throw std::bad_alloc();
    return p;
}

void free_block(void* p, size_t) { free(p); }

bool read_paf_parallel(const std::string& path, const NameTable& names, const std::vector<uint32_t>& read_len,
    bool check_lengths, uint32_t num_threads, OverlapColumns& out, int64_t* length_error) {
    if (length_error) *length_error = -1;
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); return false; }
    const size_t size = (size_t)st.st_size;
    if (size == 0) { close(fd); return true; }

    // one chunk of the file per thread, read with pread into the thread's own buffer (page faults
    // of a shared mapping serialise on the address-space lock when many threads take them)
    // more chunks than threads, handed out through a counter: threads that are descheduled (a
    // container's CPU quota throttles in bursts) do not hold the others up
    const uint32_t n_thr = (uint32_t)std::max<size_t>(1, std::min<size_t>(std::max(1u, num_threads), size / (1 << 20) + 1));
    const uint32_t T = (uint32_t)std::max<size_t>(n_thr, std::min<size_t>((size_t)n_thr * 8, size / (4 << 20) + 1));
    const auto t_start = std::chrono::steady_clock::now();
    std::vector<Chunk> chunks(T);
    std::vector<int> failed(T, 0);
    std::vector<double> t_read(T, 0.0), t_parse(T, 0.0);
    auto work = [&](uint32_t t) {
        const auto w0 = std::chrono::steady_clock::now();
        const size_t lo = size * t / T, hi = size * (t + 1) / T;
        // a chunk owns the lines that start inside [lo, hi); the last of them may end beyond hi
        const size_t from = lo ? lo - 1 : 0;
        std::vector<char, UninitAllocator<char>> buf;
        size_t have = 0, want = (hi - from) + (1 << 16);
        bool eof = false;
        auto fill = [&](size_t upto) {
            upto = std::min(upto, size - from);
            if (buf.size() < upto) buf.resize(upto);
            while (have < upto) {
                const ssize_t n = pread(fd, buf.data() + have, upto - have, (off_t)(from + have));
                if (n <= 0) { failed[t] = 1; eof = true; return; }
                have += (size_t)n;
            }
            if (from + have >= size) eof = true;
        };
        fill(want);
        if (failed[t]) return;
        const auto w1 = std::chrono::steady_clock::now();
        t_read[t] = std::chrono::duration<double, std::milli>(w1 - w0).count();
        const char* base = buf.data() - from;          // base[x] = byte x of the file
        size_t p = lo;
        if (t > 0) {
            const char* nl = (const char*)memchr(buf.data(), '\n', have);
            // no newline in the whole chunk: the line belongs to an earlier chunk
            if (!nl) { if (!eof) { /* a line longer than the chunk + 64 KiB: not a PAF record */ } return; }
            p = (size_t)(nl - buf.data()) + from + 1;
        }
        Chunk& c = chunks[t];
        const size_t guess = (hi > lo ? hi - lo : 0) / 48 + 16;      // a PAF line is rarely shorter
        c.cols.a_id.reserve(guess); c.cols.b_id.reserve(guess); c.cols.a_begin.reserve(guess);
        c.cols.a_end.reserve(guess); c.cols.b_begin.reserve(guess); c.cols.b_end.reserve(guess);
        c.cols.length.reserve(guess); c.cols.strand.reserve(guess);
        while (p < hi) {
            const char* nl = (const char*)memchr(base + p, '\n', from + have - p);
            while (!nl && !eof) {                      // the line runs past what was read
                fill(have + (1 << 20));
                if (failed[t]) return;
                base = buf.data() - from;
                c.last_q = nullptr;                    // the buffer may have moved
                nl = (const char*)memchr(base + p, '\n', from + have - p);
            }
            const size_t e = nl ? (size_t)(nl - base) : from + have;
            size_t le = e;
            if (le > p && base[le - 1] == '\r') --le;
            if (le > p) parse_paf_line(base + p, base + le, names, read_len, check_lengths, c);
            p = e + 1;
        }
        t_parse[t] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w1).count();
    };
    std::atomic<uint32_t> next(0);
    auto pull = [&](auto&& fn) {
        for (uint32_t t = next.fetch_add(1); t < T; t = next.fetch_add(1)) fn(t);
    };
    std::vector<std::thread> threads;
    for (uint32_t k = 1; k < n_thr; ++k) threads.emplace_back([&] { pull(work); });
    pull(work);
    for (auto& th : threads) th.join();
    close(fd);
    for (uint32_t t = 0; t < T; ++t) if (failed[t]) return false;
    const bool trace = getenv("RALA_IO_TRACE") != nullptr;
    const auto t_parsed = std::chrono::steady_clock::now();

    size_t total = out.size();
    for (const auto& c : chunks) total += c.cols.size();
    std::vector<size_t> at(T);
    size_t run = out.size();
    for (uint32_t t = 0; t < T; ++t) { at[t] = run; run += chunks[t].cols.size(); }
    out.a_id.resize(total); out.b_id.resize(total); out.a_begin.resize(total); out.a_end.resize(total);
    out.b_begin.resize(total); out.b_end.resize(total); out.length.resize(total); out.strand.resize(total);
    auto gather = [&](uint32_t t) {
        const OverlapColumns& c = chunks[t].cols;
        const size_t n = c.size();
        if (!n) return;
        memcpy(out.a_id.data() + at[t], c.a_id.data(), n * 4); memcpy(out.b_id.data() + at[t], c.b_id.data(), n * 4);
        memcpy(out.a_begin.data() + at[t], c.a_begin.data(), n * 4); memcpy(out.a_end.data() + at[t], c.a_end.data(), n * 4);
        memcpy(out.b_begin.data() + at[t], c.b_begin.data(), n * 4); memcpy(out.b_end.data() + at[t], c.b_end.data(), n * 4);
        memcpy(out.length.data() + at[t], c.length.data(), n * 4); memcpy(out.strand.data() + at[t], c.strand.data(), n);
    };
    threads.clear();
    next = 0;
    for (uint32_t k = 1; k < n_thr; ++k) threads.emplace_back([&] { pull(gather); });
    pull(gather);
    for (auto& th : threads) th.join();
    if (length_error) {
        for (const auto& c : chunks) {
            if (c.error_read >= 0) { *length_error = c.error_read; break; }
        }
    }
    if (trace) {
        const auto t_done = std::chrono::steady_clock::now();
        double avg_parse = 0;
        for (double x : t_parse) avg_parse += x / T;
        fprintf(stderr, "[io] %u threads, %u chunks: read + parse %.1f ms (slowest chunk: read %.1f ms, parse %.1f ms; mean parse %.1f ms), gather %.1f ms\n", n_thr, T,
                std::chrono::duration<double, std::milli>(t_parsed - t_start).count(),
                *std::max_element(t_read.begin(), t_read.end()), *std::max_element(t_parse.begin(), t_parse.end()), avg_parse,
                std::chrono::duration<double, std::milli>(t_done - t_parsed).count());
    }
    return true;
}

}  // namespace io
}  // namespace rala
