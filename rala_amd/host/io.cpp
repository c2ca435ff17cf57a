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
#include <sched.h>
#include <stdio.h>
#include <sys/mman.h>
#include <sys/resource.h>
#include <sys/stat.h>
#include <unistd.h>

#include <emmintrin.h>
#include <immintrin.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <new>
#include <thread>

namespace rala {
namespace io {

namespace {

inline uint64_t hash_bytes(const char* p, size_t n) {
    // (rala_amd/csrc/name_table.h: the device tokeniser computes the same)
    return rala_hip::name_hash_with((uint64_t)n, [p](uint64_t k, uint64_t m) {
        uint64_t w = 0;
        memcpy(&w, p + k, (size_t)m);
        return w;
    });
}

inline const char* parse_u32(const char* p, const char* e, uint32_t& v) {
    uint64_t x = 0;
    while (p < e && *p >= '0' && *p <= '9') { x = x * 10 + (uint64_t)(*p - '0'); ++p; }
    v = (uint32_t)x;
    return p;
}

// a line whose numbers are parsed and whose names are located, waiting for its table look-ups
struct PendingLine {
    const char* q; const char* t;       // name tokens
    uint32_t qn, tn;
    uint64_t qh, th;                    // their hashes (qh only when the query differs from the line before)
    uint32_t ql, qb, qe, tl, tb, te, ol;
    uint8_t strand;
    bool new_query;
};

constexpr int kBatch = 16;

struct alignas(256) Chunk {        // one per piece of the file: no shared cache lines
    // where this piece's records go in the final columns (its slot is sized by its line count)
    uint32_t *a_id = nullptr, *b_id = nullptr, *a_begin = nullptr, *a_end = nullptr, *b_begin = nullptr, *b_end = nullptr,
             *length = nullptr;
    uint8_t* strand = nullptr;
    size_t n = 0;                   // records written
    int64_t error_read = -1;        // first line of this chunk with a length mismatch
    // overlap files are grouped by query: the previous line's query name and its id
    const char* last_q = nullptr;
    size_t last_qn = 0;
    uint64_t last_a = ~0ull;
    PendingLine pend[kBatch];
    int n_pend = 0;
};

// the batch's look-ups (their buckets were prefetched while the lines were parsed), in line order
inline void resolve_batch(const NameTable& names, const std::vector<uint32_t>& read_len, bool check_lengths, Chunk& c) {
    for (int k = 0; k < c.n_pend; ++k) {
        const PendingLine& L = c.pend[k];
        if (L.new_query) c.last_a = names.find(L.q, L.qn, L.qh);
        const uint64_t a = c.last_a;
        const uint64_t b = names.find(L.t, L.tn, L.th);
        const uint32_t ia = a == ~0ull ? 0xFFFFFFFFu : (uint32_t)a;
        const uint32_t ib = b == ~0ull ? 0xFFFFFFFFu : (uint32_t)b;
        if (check_lengths && c.error_read < 0) {
            if (ia != 0xFFFFFFFFu && L.ql != read_len[ia]) c.error_read = ia;
            else if (ia != 0xFFFFFFFFu && ib != 0xFFFFFFFFu && L.tl != read_len[ib]) c.error_read = ib;
        }
        const size_t w = c.n++;
        c.a_id[w] = ia; c.b_id[w] = ib;
        c.a_begin[w] = L.qb; c.a_end[w] = L.qe;
        c.b_begin[w] = L.tb; c.b_end[w] = L.te;
        c.length[w] = L.ol;
        c.strand[w] = L.strand;
    }
    c.n_pend = 0;
}

// decimal digits [p, p + d) -> value; d <= 8, eight readable bytes end at p + d (the caller keeps
// eight bytes of slack in front of the first line).  All-digit check, then three multiplications
// (the byte-at-a-time loop with its data-dependent branches was most of a line's cost).
inline bool digits8(const char* p, uint32_t d, uint32_t& v) {
    uint64_t w;
    memcpy(&w, p + d - 8, 8);                           // the number's last digit is the top byte
    const uint64_t keep = d == 8 ? ~0ull : ~0ull << (8 * (8 - d));
    w = (w & keep) | (0x3030303030303030ull & ~keep);
    if ((((w + 0x4646464646464646ull) | (w - 0x3030303030303030ull)) & 0x8080808080808080ull) != 0) return false;
    w -= 0x3030303030303030ull;
    w = (w * 10) + (w >> 8);
    w = (((w & 0x000000FF000000FFull) * 0x000F424000000064ull) + (((w >> 16) & 0x000000FF000000FFull) * 0x0000271000000001ull)) >> 32;
    v = (uint32_t)w;
    return true;
}

inline void field_u32(const char* b, const char* e, uint32_t& v) {
    // same value as parse_u32: the leading digits of the field
    const uint32_t d = (uint32_t)(e - b);
    if (d >= 1 && d <= 8 && digits8(b, d, v)) return;
    parse_u32(b, e, v);
}

// n bytes equal?  Both ranges may be read 16 bytes beyond their end (the caller's slack).
inline bool same_bytes(const char* a, const char* b, size_t n) {
    if (n <= 16) {
        const __m128i x = _mm_loadu_si128((const __m128i*)a), y = _mm_loadu_si128((const __m128i*)b);
        const uint32_t eq = (uint32_t)_mm_movemask_epi8(_mm_cmpeq_epi8(x, y));
        return ((eq | (0xFFFFFFFFu << n)) & 0xFFFFu) == 0xFFFFu;
    }
    return memcmp(a, b, n) == 0;
}

// the fields of a line that starts at p and whose first nt (<= 12) tabs are at tab[]; false if it
// is not a 12-column record
inline bool finish_line(const char* p, const char* const* tab, int nt, bool any_blank, const NameTable& names,
                        const std::vector<uint32_t>& read_len, bool check_lengths, Chunk& c) {
    PendingLine& L = c.pend[c.n_pend];
    if (nt < 11) return false;                       // fewer than 12 columns
    auto name_end = [&](const char* b, const char* e2) {
        if (!any_blank) return e2;
        const char* sp = (const char*)memchr(b, ' ', (size_t)(e2 - b));
        return sp ? sp : e2;
    };
    L.q = p;
    L.qn = (uint32_t)(name_end(p, tab[0]) - p);
    field_u32(tab[0] + 1, tab[1], L.ql);
    field_u32(tab[1] + 1, tab[2], L.qb);
    field_u32(tab[2] + 1, tab[3], L.qe);
    L.strand = (tab[3] + 1 < tab[4] ? tab[3][1] : '+') == '+' ? 0 : 1;
    L.t = tab[4] + 1;
    L.tn = (uint32_t)(name_end(L.t, tab[5]) - L.t);
    field_u32(tab[5] + 1, tab[6], L.tl);
    field_u32(tab[6] + 1, tab[7], L.tb);
    field_u32(tab[7] + 1, tab[8], L.te);
    field_u32(tab[9] + 1, tab[10], L.ol);            // column 11: alignment length
    L.new_query = c.last_q == nullptr || L.qn != c.last_qn || !same_bytes(c.last_q, L.q, L.qn);
    if (L.new_query) {
        L.qh = hash_bytes(L.q, L.qn);
        names.prefetch(L.qh);
        c.last_q = L.q;
        c.last_qn = L.qn;
    }
    L.th = hash_bytes(L.t, L.tn);
    names.prefetch(L.th);
    if (++c.n_pend == kBatch) resolve_batch(names, read_len, check_lengths, c);
    return true;
}

// one line [p, e) (no newline; at least 8 readable bytes in front of p and 16 behind e); returns
// false if it is not a 12-column record.  The tabs come from 16-byte compares (SSE2, baseline of
// x86-64), a name token ends at its first blank, numbers through digits8.
inline bool parse_paf_line(const char* p, const char* e, const NameTable& names, const std::vector<uint32_t>& read_len,
                           bool check_lengths, Chunk& c) {
    const char* tab[12];
    int nt = 0;
    bool any_blank = false;                          // a blank somewhere in front of the 6th tab
    {
        const __m128i tabs = _mm_set1_epi8('\t'), blanks = _mm_set1_epi8(' ');
        for (const char* q = p; q < e && nt < 12; q += 16) {
            const __m128i x = _mm_loadu_si128((const __m128i*)q);
            uint32_t m = (uint32_t)_mm_movemask_epi8(_mm_cmpeq_epi8(x, tabs));
            // names end at their first blank: noted while the bytes are in the register, so that the
            // usual line (no blank in the name columns) needs no second look at them
            if (nt < 6) any_blank |= _mm_movemask_epi8(_mm_cmpeq_epi8(x, blanks)) != 0;
            while (m && nt < 12) {
                const char* at = q + __builtin_ctz(m);
                if (at >= e) { m = 0; break; }
                tab[nt++] = at;
                m &= m - 1;
            }
        }
    }
    return finish_line(p, tab, nt, any_blank, names, read_len, check_lengths, c);
}

// Line starts inside text[0, n): positions j >= 1 whose predecessor is a newline; *lines += those
// that are not newlines themselves, *first_nl = index of the first newline at or below n - 2 (the
// first line start is behind it), or n if there is none.  64 readable bytes behind the text.
__attribute__((target("avx512f,avx512bw")))
void count_line_starts_avx512(const char* text, size_t n, size_t* lines, size_t* first_nl) {
    const __m512i v_nl = _mm512_set1_epi8('\n');
    uint64_t carry = 0;
    size_t count = 0, first = n;
    for (size_t blk = 0; blk < n; blk += 64) {
        const uint64_t valid = n - blk < 64 ? (1ull << (n - blk)) - 1 : ~0ull;
        const uint64_t m = _mm512_cmpeq_epi8_mask(_mm512_loadu_si512((const void*)(text + blk)), v_nl) & valid;
        const uint64_t starts = ((m << 1) | carry) & valid;
        count += (size_t)__builtin_popcountll(starts & ~m);
        if (first == n && starts) first = blk + (size_t)__builtin_ctzll(starts) - 1;
        carry = m >> 63;
    }
    *lines += count;
    *first_nl = first;
}

// The same for a whole read of text with 64-byte compares (AVX-512BW, chosen at run time): tabs,
// newlines and blanks of a block come out of three compares as bit masks, so a line costs one
// pass over its bytes instead of a newline search plus a tab scan.  Lines [from, ...) that end at
// or before text[stop] (a newline, or the end of the file) and start below `limit`; returns the
// offset behind the last line taken.  64 readable bytes behind text[stop] are required.
__attribute__((target("avx512f,avx512bw")))
size_t parse_lines_avx512(const char* text, size_t from, size_t stop, size_t limit, bool ends_at_eof,
                          const NameTable& names, const std::vector<uint32_t>& read_len, bool check_lengths, Chunk& c) {
    const __m512i v_nl = _mm512_set1_epi8('\n'), v_tab = _mm512_set1_epi8('\t'), v_sp = _mm512_set1_epi8(' ');
    const char* tab[12];
    int nt = 0;
    bool blank = false;
    size_t line = from;                               // start of the current line
    if (line >= limit) return line;
    const size_t end = ends_at_eof ? stop : stop + 1; // bytes [from, end) are looked at
    for (size_t blk = from & ~(size_t)63; blk < end; blk += 64) {
        const __m512i x = _mm512_loadu_si512((const void*)(text + blk));
        uint64_t m_nl = _mm512_cmpeq_epi8_mask(x, v_nl), m_tab = _mm512_cmpeq_epi8_mask(x, v_tab);
        uint64_t m_sp = _mm512_cmpeq_epi8_mask(x, v_sp);
        // only bytes [max(from, blk), end)
        uint64_t valid = ~0ull;
        if (blk < from) valid &= ~0ull << (from - blk);
        if (end - blk < 64) valid &= (1ull << (end - blk)) - 1;
        m_nl &= valid; m_tab &= valid; m_sp &= valid;
        if (nt < 6) blank |= (line > blk ? m_sp >> (line - blk) : m_sp) != 0;
        uint64_t ev = m_nl | m_tab;
        while (ev) {
            const unsigned b = (unsigned)__builtin_ctzll(ev);
            ev &= ev - 1;
            const size_t pos = blk + b;
            if ((m_tab >> b) & 1) {
                if (nt < 12) tab[nt++] = text + pos;
                continue;
            }
            // a newline: the line [line, pos)
            size_t le = pos;
            if (le > line && text[le - 1] == '\r') --le;
            if (le > line) {
                // tabs behind a carriage return cannot exist; tabs are all in front of le
                finish_line(text + line, tab, nt, blank, names, read_len, check_lengths, c);
            }
            line = pos + 1;
            nt = 0;
            if (line >= limit) return line;
            // blanks of the new line in the rest of this block
            blank = b < 63 && (m_sp >> (b + 1)) != 0;
        }
    }
    if (ends_at_eof && line < end) {                  // the last line has no newline
        size_t le = end;
        if (le > line && text[le - 1] == '\r') --le;
        if (le > line) finish_line(text + line, tab, nt, blank, names, read_len, check_lengths, c);
        line = end + 1;
    }
    return line;
}


}  // namespace

uint64_t NameTable::hash(const char* p, size_t n) { return hash_bytes(p, n); }

namespace {
// the CPUs this process may run on, one per physical core (the lowest-numbered sibling)
std::vector<int> one_cpu_per_core() {
    std::vector<int> out;
    cpu_set_t allowed;
    if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return out;
    for (int cpu = 0; cpu < CPU_SETSIZE; ++cpu) {
        if (!CPU_ISSET(cpu, &allowed)) continue;
        char path[128];
        snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", cpu);
        FILE* f = fopen(path, "r");
        if (!f) return std::vector<int>();
        int first = -1;
        const int got = fscanf(f, "%d", &first);
        fclose(f);
        if (got != 1) return std::vector<int>();
        if (first == cpu) out.push_back(cpu);
    }
    return out;
}

// While it lives the calling thread runs on the first physical cores only - where the parser threads
// will be pinned - so that what it allocates and touches (the name table) lands in their NUMA node.
struct NearParserCores {
    cpu_set_t before;
    bool active = false;
    NearParserCores() {
        if (getenv("RALA_IO_NO_PIN")) return;
        const std::vector<int> cores = one_cpu_per_core();
        if (cores.size() < 2 || sched_getaffinity(0, sizeof(before), &before) != 0) return;
        cpu_set_t near;
        CPU_ZERO(&near);
        for (size_t k = 0; k < cores.size() && k < 16; ++k) CPU_SET(cores[k], &near);
        active = sched_setaffinity(0, sizeof(near), &near) == 0;
    }
    ~NearParserCores() {
        if (active) (void)sched_setaffinity(0, sizeof(before), &before);
    }
};
}  // namespace

NameTable::~NameTable() {
    if (bucket_) free_block(bucket_, n_bucket_ * sizeof(Bucket));
}

void NameTable::build(const std::vector<std::string>& names) {
    const NearParserCores near;
    uint64_t cap = 16;
    while (cap < 2 * names.size() + 2) cap <<= 1;
    mask_ = cap - 1;
    if (bucket_) free_block(bucket_, n_bucket_ * sizeof(Bucket));
    n_bucket_ = cap;
    bucket_ = (Bucket*)allocate_block(cap * sizeof(Bucket));
    memset((void*)bucket_, 0, cap * sizeof(Bucket));
    arena_.clear();
    // a later duplicate name replaces an earlier one, like unordered_map::operator[] in the
    // reference (src/graph.cpp:262)
    for (size_t i = 0; i < names.size(); ++i) {
        const std::string& s = names[i];
        const uint64_t h = hash_bytes(s.data(), s.size());
        Bucket nb = Bucket();
        nb.hash32 = (uint32_t)(h >> 32); nb.id1 = (uint32_t)(i + 1); nb.len = (uint32_t)s.size(); nb.off = (uint32_t)arena_.size();
        memcpy(nb.head, s.data(), std::min<size_t>(16, s.size()));
        for (uint64_t k = h & mask_;; k = (k + 1) & mask_) {
            Bucket& b = bucket_[k];
            if (b.id1 == 0) { arena_ += s; b = nb; break; }
            if (b.hash32 == nb.hash32 && b.len == nb.len && memcmp(arena_.data() + b.off, s.data(), b.len) == 0) {
                b.id1 = nb.id1;                  // same name again: the later read takes it
                break;
            }
        }
    }
}

uint64_t NameTable::find(const char* p, size_t n, uint64_t h) const {
    if (!bucket_) return ~0ull;
    const uint32_t h32 = (uint32_t)(h >> 32);
    for (uint64_t k = h & mask_;; k = (k + 1) & mask_) {
        const Bucket& b = bucket_[k];
        if (b.id1 == 0) return ~0ull;
        if (b.hash32 != h32 || b.len != n) continue;
        if (n <= 16 ? memcmp(b.head, p, n) == 0 : (memcmp(b.head, p, 16) == 0 && memcmp(arena_.data() + b.off, p, n) == 0)) {
            return (uint64_t)b.id1 - 1;
        }
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
    const bool trace = getenv("RALA_IO_TRACE") != nullptr;
    const auto t_start = std::chrono::steady_clock::now();

    // The file is taken in windows (all of it unless it is very large).  A window is cut into
    // pieces, more pieces than threads, handed out through a counter (threads that are descheduled -
    // a container's CPU quota throttles in bursts - do not hold the others up).  A piece owns the
    // lines that START inside it.  Two passes over the text, both through a buffer of half a
    // megabyte per thread that stays in the core's L2 (the text comes out of the page cache twice;
    // keeping all of it in freshly allocated memory between the passes cost more: 3 GB of pages
    // to fault in, clear and unmap at C3 - 1.7 s of system time):
    //   1. the pieces' line starts are counted (a newline followed by something else);
    //   2. the counts give every piece its slot in the final columns and the pieces are parsed
    //      straight into it - no per-piece vectors, no gather pass.
    // Lines that turn out not to be records leave a gap at the end of the piece's slot; gaps are
    // closed afterwards (they are rare).
    const uint32_t n_thr = (uint32_t)std::max<size_t>(1, std::min<size_t>(std::max(1u, num_threads), size / (1 << 20) + 1));
    const size_t kWindow = (size_t)8 << 30;
    constexpr size_t kFront = 8, kBack = 80;      // slack around the text for the line parser's wide loads
    // 64-byte compares where the CPU has them (RALA_IO_NO_AVX512=1: the 16-byte path)
    const bool wide = __builtin_cpu_supports("avx512bw") && getenv("RALA_IO_NO_AVX512") == nullptr;
    constexpr size_t kText = 512 << 10;           // bytes of text per read
    std::vector<std::thread> threads;
    // The parser threads are pinned, one per physical core, to the FIRST cores the process may use
    // (RALA_IO_NO_PIN=1: left to the scheduler).  On the two-socket EPYC 9575F boxes 16 threads
    // spread over both sockets (which is what the scheduler does, and what a spread pinning does)
    // parse C3 in 535 - 590 ms; on 16 neighbouring cores of one socket in 320 ms: name table and
    // columns then live in one NUMA node and two L3 slices (docs/history/gpurun/r2_ingest_threads.sh).
    const std::vector<int> cores = getenv("RALA_IO_NO_PIN") ? std::vector<int>() : one_cpu_per_core();
    cpu_set_t before;
    const bool pinned = cores.size() >= n_thr && n_thr > 1 && sched_getaffinity(0, sizeof(before), &before) == 0;
    auto pin = [&](uint32_t k) {
        if (!pinned) return;
        cpu_set_t one;
        CPU_ZERO(&one);
        const char* env_span = getenv("RALA_IO_PIN_SPAN");          // diagnostics: spread over the first <n> cores
        const size_t span = env_span ? std::min<size_t>(cores.size(), std::max<size_t>(n_thr, (size_t)atoi(env_span))) : n_thr;
        // several processes of one job (one per GPU) take different groups of cores
        const char* env_rank = getenv("LOCAL_RANK");
        const size_t groups = std::max<size_t>(1, cores.size() / span);
        const size_t first = env_rank ? ((size_t)atoi(env_rank) % groups) * span : 0;
        CPU_SET(cores[first + (size_t)k * span / n_thr], &one);
        (void)sched_setaffinity(0, sizeof(one), &one);
    };
    typedef std::vector<char, UninitAllocator<char>> Text;
    auto run = [&](uint32_t T, auto&& fn) {
        std::atomic<uint32_t> next(0);
        auto pull = [&] {
            Text buf(kFront + kText + kBack);
            memset(buf.data(), 0, kFront);
            for (uint32_t t = next.fetch_add(1); t < T; t = next.fetch_add(1)) fn(t, buf);
        };
        threads.clear();
        for (uint32_t k = 1; k < n_thr; ++k) threads.emplace_back([&, k] { pin(k); pull(); });
        pin(0);
        pull();
        for (auto& th : threads) th.join();
        if (pinned) (void)sched_setaffinity(0, sizeof(before), &before);
    };
    // file bytes [from, from + n) into buf (behind its front slack); false = read error
    auto fetch = [&](Text& buf, size_t from, size_t n) {
        if (buf.size() < kFront + n + kBack) {
            Text bigger(kFront + n + kBack);
            memset(bigger.data(), 0, kFront);
            buf.swap(bigger);
        }
        size_t have = 0;
        while (have < n) {
            const ssize_t got = pread(fd, buf.data() + kFront + have, n - have, (off_t)(from + have));
            if (got <= 0) return false;
            have += (size_t)got;
        }
        return true;
    };
    struct Piece {
        size_t lo = 0, hi = 0;          // file positions this piece owns line starts in
        size_t first = 0;               // file position of the first owned line (== hi: none)
        size_t lines = 0;               // upper bound of the records (non-empty lines)
        bool failed = false;
    };
    double ms_read = 0, ms_parse = 0, ms_close = 0;
    int64_t first_error = -1;
    for (size_t w_lo = 0; w_lo < size; w_lo += kWindow) {
        const size_t w_hi = std::min(size, w_lo + kWindow);
        const size_t w_size = w_hi - w_lo;
        const uint32_t T = (uint32_t)std::max<size_t>(n_thr, std::min<size_t>((size_t)n_thr * 8, w_size / (4 << 20) + 1));
        std::vector<Piece> piece(T);
        std::vector<Chunk> chunks(T);
        const auto p0 = std::chrono::steady_clock::now();
        rusage ru0, ru1, ru2;
        if (trace) getrusage(RUSAGE_SELF, &ru0);
        // ---- pass 1: the owned line starts p in [lo, hi): the byte in front is a newline (or p is
        // the start of the file) and the byte at p is not.  Reads overlap by one byte.
        run(T, [&](uint32_t t, Text& buf) {
            Piece& P = piece[t];
            P.lo = w_lo + w_size * t / T;
            P.hi = w_lo + w_size * (t + 1) / T;
            P.first = P.hi;
            if (P.lo == 0 && P.hi > 0) P.first = 0;
            size_t a = P.lo ? P.lo - 1 : 0;                  // first byte of the next read
            bool start_pending = P.lo == 0;                  // position `a` starts a line (known from the byte in front)
            while (a < P.hi) {
                const size_t n = std::min(kText, P.hi - a);
                if (!fetch(buf, a, n)) { P.failed = true; return; }
                const char* text = buf.data() + kFront;      // text[k] = file byte a + k
                if (start_pending) {
                    if (text[0] != '\n') ++P.lines;
                    start_pending = false;
                }
                // newlines at k <= n - 2 start a line at a + k + 1 < hi, whose first byte is in this read
                if (wide) {
                    size_t first_nl = n;
                    count_line_starts_avx512(text, n, &P.lines, &first_nl);
                    if (first_nl < n && P.first == P.hi) P.first = a + first_nl + 1;
                } else
                for (const char* q = text; n >= 2;) {
                    const char* nl = (const char*)memchr(q, '\n', (size_t)(text + n - 1 - q));
                    if (!nl) break;
                    const size_t p = a + (size_t)(nl - text) + 1;
                    if (p >= P.lo) {
                        if (P.first == P.hi) P.first = p;
                        if (nl[1] != '\n') ++P.lines;
                    }
                    q = nl + 1;
                }
                if (a + n >= P.hi) break;
                a += n - 1;                                  // the last byte again: is it a newline?
            }
        });
        for (const Piece& P : piece) if (P.failed) { close(fd); return false; }
        const auto p1 = std::chrono::steady_clock::now();
        if (trace) getrusage(RUSAGE_SELF, &ru1);
        // ---- slots in the final columns
        std::vector<size_t> at(T + 1);
        at[0] = out.size();
        for (uint32_t t = 0; t < T; ++t) at[t + 1] = at[t] + piece[t].lines;
        const size_t total = at[T];
        out.a_id.resize(total); out.b_id.resize(total); out.a_begin.resize(total); out.a_end.resize(total);
        out.b_begin.resize(total); out.b_end.resize(total); out.length.resize(total); out.strand.resize(total);
        // ---- pass 2: parse into the slots
        run(T, [&](uint32_t t, Text& buf) {
            Piece& P = piece[t];
            Chunk& c = chunks[t];
            c.a_id = out.a_id.data() + at[t]; c.b_id = out.b_id.data() + at[t];
            c.a_begin = out.a_begin.data() + at[t]; c.a_end = out.a_end.data() + at[t];
            c.b_begin = out.b_begin.data() + at[t]; c.b_end = out.b_end.data() + at[t];
            c.length = out.length.data() + at[t]; c.strand = out.strand.data() + at[t];
            size_t p = P.first;                              // start of the next line to parse
            size_t want = kText;
            while (p < P.hi) {
                const size_t n = std::min(want, size - p);
                if (!fetch(buf, p, n)) { P.failed = true; return; }
                const char* text = buf.data() + kFront;      // text[k] = file byte p + k
                // the last complete line of this read ends at `stop`
                size_t stop = n;
                if (p + n < size) {
                    const char* nl = (const char*)memrchr(text, '\n', n);
                    if (!nl) { want *= 2; continue; }        // a line longer than the read
                    stop = (size_t)(nl - text);
                }
                size_t k = 0;
                if (wide) {
                    k = parse_lines_avx512(text, 0, stop, P.hi - p, p + n >= size, names, read_len, check_lengths, c);
                } else
                while (k <= stop && k < n && p + k < P.hi) {
                    const char* nl = (const char*)memchr(text + k, '\n', stop - k);
                    const size_t e = nl ? (size_t)(nl - text) : stop;
                    size_t le = e;
                    if (le > k && text[le - 1] == '\r') --le;
                    if (le > k) parse_paf_line(text + k, text + le, names, read_len, check_lengths, c);
                    k = e + 1;
                }
                // the pending look-ups and the remembered query name point into this read
                resolve_batch(names, read_len, check_lengths, c);
                c.last_q = nullptr;
                if (p + k >= P.hi || p + n >= size) break;
                p += stop + 1;
                want = kText;
            }
        });
        for (const Piece& P : piece) if (P.failed) { close(fd); return false; }
        const auto p2 = std::chrono::steady_clock::now();
        if (trace) getrusage(RUSAGE_SELF, &ru2);
        // ---- close the gaps left by lines that were not records
        size_t w = at[0];
        bool gaps = false;
        for (uint32_t t = 0; t < T; ++t) {
            const size_t n = chunks[t].n;
            if (gaps && n) {
                memmove(out.a_id.data() + w, out.a_id.data() + at[t], n * 4); memmove(out.b_id.data() + w, out.b_id.data() + at[t], n * 4);
                memmove(out.a_begin.data() + w, out.a_begin.data() + at[t], n * 4); memmove(out.a_end.data() + w, out.a_end.data() + at[t], n * 4);
                memmove(out.b_begin.data() + w, out.b_begin.data() + at[t], n * 4); memmove(out.b_end.data() + w, out.b_end.data() + at[t], n * 4);
                memmove(out.length.data() + w, out.length.data() + at[t], n * 4); memmove(out.strand.data() + w, out.strand.data() + at[t], n);
            }
            w += n;
            if (n != piece[t].lines) gaps = true;
            if (first_error < 0 && chunks[t].error_read >= 0) first_error = chunks[t].error_read;
        }
        if (w != total) {
            out.a_id.resize(w); out.b_id.resize(w); out.a_begin.resize(w); out.a_end.resize(w);
            out.b_begin.resize(w); out.b_end.resize(w); out.length.resize(w); out.strand.resize(w);
        }
        const auto p3 = std::chrono::steady_clock::now();
        if (trace) {
            auto cpu = [](const rusage& a, const rusage& b, bool sys) {
                const timeval& x = sys ? a.ru_stime : a.ru_utime;
                const timeval& y = sys ? b.ru_stime : b.ru_utime;
                return (y.tv_sec - x.tv_sec) * 1e3 + (y.tv_usec - x.tv_usec) * 1e-3;
            };
            fprintf(stderr, "[io] cpu time: read + count user %.0f ms sys %.0f ms, parse user %.0f ms sys %.0f ms (page faults %ld + %ld)\n",
                    cpu(ru0, ru1, false), cpu(ru0, ru1, true), cpu(ru1, ru2, false), cpu(ru1, ru2, true),
                    ru1.ru_minflt - ru0.ru_minflt, ru2.ru_minflt - ru1.ru_minflt);
        }
        ms_read += std::chrono::duration<double, std::milli>(p1 - p0).count();
        ms_parse += std::chrono::duration<double, std::milli>(p2 - p1).count();
        ms_close += std::chrono::duration<double, std::milli>(p3 - p2).count();
    }
    close(fd);
    if (length_error) *length_error = first_error;
    if (trace) {
        fprintf(stderr, "[io] %u threads: read + count %.1f ms, parse %.1f ms, close gaps %.1f ms, total %.1f ms\n", n_thr, ms_read,
                ms_parse, ms_close, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count());
    }
    return true;
}

// ---- compressed PAF, and MHAP: one inflating thread, the others parse --------------------------------
// A gzip stream can only be inflated front to back.  One thread does that (zlib; it also reads plain
// files), cutting the text into blocks of a few megabytes that end with a line; the other threads take the
// blocks as they come, count their lines, parse them with the same tokenizer as read_paf_parallel into
// columns of their own, and the blocks' columns are copied into the final ones in file order.  The file's
// rate is then the inflater's (about half a gigabyte of text per second) instead of inflater + parser on one
// thread.  A BGZF file (bgzip) is made of blocks that can be inflated side by side: BgzfSource below.
namespace {

// one MHAP line [p, e): "a_id b_id error minmers a_rc a_begin a_end a_len b_rc b_begin b_end b_len", blank
// separated, ids from 1 (reference src/overlap.cpp:12-20); false if it has fewer than 12 columns
inline bool parse_mhap_line(const char* p, const char* e, const std::vector<uint32_t>& read_len, bool check_lengths, Chunk& c) {
    const char* f[13];
    int nf = 0;
    f[nf++] = p;
    for (const char* q = p; q < e && nf < 13;) {
        const char* sp = (const char*)memchr(q, ' ', (size_t)(e - q));
        if (!sp) break;
        f[nf++] = sp + 1;
        q = sp + 1;
    }
    if (nf < 12) return false;
    auto end_of = [&](int k) { return k + 1 < nf ? f[k + 1] - 1 : e; };
    auto num = [&](int k) { uint32_t v; field_u32(f[k], end_of(k), v); return v; };
    uint64_t ida = 0, idb = 0;
    for (const char* q = f[0]; q < end_of(0) && *q >= '0' && *q <= '9'; ++q) ida = ida * 10 + (uint64_t)(*q - '0');
    for (const char* q = f[1]; q < end_of(1) && *q >= '0' && *q <= '9'; ++q) idb = idb * 10 + (uint64_t)(*q - '0');
    const uint64_t a = ida - 1, b = idb - 1;
    const uint32_t ia = a < read_len.size() ? (uint32_t)a : 0xFFFFFFFFu;
    const uint32_t ib = b < read_len.size() ? (uint32_t)b : 0xFFFFFFFFu;
    const uint32_t a_rc = num(4), ab = num(5), ae = num(6), al = num(7), b_rc = num(8), bb = num(9), be = num(10), bl = num(11);
    if (check_lengths && c.error_read < 0) {
        if (ia != 0xFFFFFFFFu && al != read_len[ia]) c.error_read = ia;
        else if (ia != 0xFFFFFFFFu && ib != 0xFFFFFFFFu && bl != read_len[ib]) c.error_read = ib;
    }
    const size_t w = c.n++;
    c.a_id[w] = ia; c.b_id[w] = ib;
    c.a_begin[w] = ab; c.a_end[w] = ae; c.b_begin[w] = bb; c.b_end[w] = be;
    c.length[w] = std::max(ae - ab, be - bb);
    c.strand[w] = a_rc == b_rc ? 0 : 1;
    return true;
}

struct TextBlock {
    std::vector<char, UninitAllocator<char>> text;      // kFront slack, n bytes of whole lines, kBack slack
    size_t n = 0;
    size_t index = 0;
};

struct BlockColumns {
    OverlapColumns cols;
    size_t n = 0;
    int64_t error_read = -1;
};

// A BGZF file (bgzip, htslib: a series of gzip members of at most 64 KB, each with its compressed size in a "BC" extra
// field - RFC 1952 plus the SAM specification, section 4.1) inflated by several threads: the blocks are read in file order,
// handed to the inflaters, and given out in file order again.  A plain gzip stream has one member and nothing to split; the
// single inflating thread's 0.4 GB/s of text is then the file's rate.
class BgzfSource {
public:
    // the stream starts with a BGZF block header?  (leaves the position at the start)
    static bool is_bgzf(FILE* f) {
        unsigned char h[18];
        const size_t got = fread(h, 1, sizeof(h), f);
        rewind(f);
        if (got < 18) return false;
        return h[0] == 0x1f && h[1] == 0x8b && h[2] == 8 && (h[3] & 4) != 0 && h[10] == 6 && h[11] == 0 && h[12] == 'B' &&
               h[13] == 'C' && h[14] == 2 && h[15] == 0;
    }
    BgzfSource(FILE* f, unsigned n_threads) : f_(f) {
        for (unsigned k = 0; k < std::max(1u, n_threads); ++k) workers_.emplace_back([this] { work(); });
    }
    ~BgzfSource() {
        {
            std::lock_guard<std::mutex> hold(m_);
            stop_ = true;
        }
        cv_job_.notify_all();
        for (auto& t : workers_) t.join();
    }
    // the next block's text in file order; false: the end of the file (`bad` says whether it ended as it should)
    bool next(std::vector<char>& out, bool& bad) {
        bad = false;
        for (;;) {
            while (!file_done_ && inflight_.size() < kWindow) {
                std::unique_ptr<Job> j(new Job);
                const int r = read_block(*j);
                if (r < 0) { broken_ = true; file_done_ = true; break; }
                if (r == 0) { file_done_ = true; break; }
                Job* raw = j.get();
                inflight_.push_back(std::move(j));
                {
                    std::lock_guard<std::mutex> hold(m_);
                    jobs_.push_back(raw);
                }
                cv_job_.notify_one();
            }
            if (inflight_.empty()) { bad = broken_; return false; }
            Job& front = *inflight_.front();
            {
                std::unique_lock<std::mutex> hold(m_);
                cv_done_.wait(hold, [&] { return front.done; });
            }
            if (front.bad) { bad = true; return false; }
            const bool empty = front.out.empty();
            if (!empty) out.swap(front.out);
            inflight_.pop_front();
            if (!empty) return true;                         // (an empty block - the end-of-file marker - is skipped)
        }
    }

private:
    struct Job {
        std::vector<unsigned char> comp;        // the deflate stream, CRC32 and ISIZE of one block
        std::vector<char> out;
        bool done = false, bad = false;
    };
    static constexpr size_t kWindow = 512;      // blocks read ahead of the consumer (32 MB of text)

    // 1: a block read, 0: the end of the file, -1: not a BGZF block
    int read_block(Job& j) {
        unsigned char h[12];
        const size_t got = fread(h, 1, 12, f_);
        if (got == 0) return 0;
        if (got < 12 || h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) return -1;
        const size_t xlen = h[10] | (size_t)h[11] << 8;
        std::vector<unsigned char> extra(xlen);
        if (fread(extra.data(), 1, xlen, f_) != xlen) return -1;
        size_t bsize = 0;
        for (size_t k = 0; k + 4 <= xlen;) {
            const size_t slen = extra[k + 2] | (size_t)extra[k + 3] << 8;
            if (extra[k] == 'B' && extra[k + 1] == 'C' && slen == 2 && k + 6 <= xlen) bsize = (extra[k + 4] | (size_t)extra[k + 5] << 8) + 1;
            k += 4 + slen;
        }
        if (bsize < 12 + xlen + 8) return -1;
        j.comp.resize(bsize - 12 - xlen);
        if (fread(j.comp.data(), 1, j.comp.size(), f_) != j.comp.size()) return -1;
        return 1;
    }
    static void inflate_block(Job& j) {
        const size_t n = j.comp.size();
        const unsigned char* tail = j.comp.data() + n - 8;
        const uint32_t crc = tail[0] | (uint32_t)tail[1] << 8 | (uint32_t)tail[2] << 16 | (uint32_t)tail[3] << 24;
        const uint32_t isize = tail[4] | (uint32_t)tail[5] << 8 | (uint32_t)tail[6] << 16 | (uint32_t)tail[7] << 24;
        if (isize > (1u << 16)) { j.bad = true; return; }
        if (isize == 0) {                                    // (the end-of-file marker; nothing to inflate into)
            std::vector<unsigned char>().swap(j.comp);
            return;
        }
        j.out.resize(isize);
        z_stream z;
        memset(&z, 0, sizeof(z));
        if (inflateInit2(&z, -15) != Z_OK) { j.bad = true; return; }
        z.next_in = j.comp.data(); z.avail_in = (uInt)(n - 8);
        z.next_out = (Bytef*)j.out.data(); z.avail_out = isize;
        const int rc = inflate(&z, Z_FINISH);
        const bool ok = rc == Z_STREAM_END && z.avail_out == 0;
        inflateEnd(&z);
        if (!ok || (uint32_t)crc32(crc32(0L, Z_NULL, 0), (const Bytef*)j.out.data(), isize) != crc) j.bad = true;
        std::vector<unsigned char>().swap(j.comp);
    }
    void work() {
        for (;;) {
            Job* j = nullptr;
            {
                std::unique_lock<std::mutex> hold(m_);
                cv_job_.wait(hold, [&] { return stop_ || !jobs_.empty(); });
                if (jobs_.empty()) return;
                j = jobs_.front();
                jobs_.pop_front();
            }
            inflate_block(*j);
            {
                std::lock_guard<std::mutex> hold(m_);
                j->done = true;
            }
            cv_done_.notify_all();
        }
    }

    FILE* f_;
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_job_, cv_done_;
    std::deque<Job*> jobs_;                              // to inflate
    std::deque<std::unique_ptr<Job>> inflight_;          // in file order (the consumer's side only)
    bool stop_ = false, file_done_ = false, broken_ = false;
};

}  // namespace

bool read_overlaps_streamed(const std::string& path, bool mhap, const NameTable& names, const std::vector<uint32_t>& read_len,
    bool check_lengths, uint32_t num_threads, OverlapColumns& out, int64_t* length_error) {
    if (length_error) *length_error = -1;
    // a BGZF file (bgzip) is inflated by several threads; everything else comes through zlib's gz stream on this thread
    FILE* raw = fopen(path.c_str(), "rb");
    if (!raw) return false;
    const bool bgzf = BgzfSource::is_bgzf(raw) && getenv("RALA_IO_NO_BGZF") == nullptr;
    gzFile in = nullptr;
    std::unique_ptr<BgzfSource> blocks;
    uint32_t n_inflaters = 0;
    if (bgzf) {
        // (inflating a byte takes about one and a half times what parsing it takes)
        n_inflaters = std::max(1u, num_threads > 2 ? (num_threads - 1) * 3 / 5 : 1u);
        blocks.reset(new BgzfSource(raw, n_inflaters));
    } else {
        fclose(raw);
        raw = nullptr;
        in = gzopen(path.c_str(), "rb");
        if (!in) return false;
        gzbuffer(in, 1 << 20);
    }
    constexpr size_t kFront = 8, kBack = 80, kBlockText = 4 << 20;
    const uint32_t n_parsers = std::max(1u, num_threads > 1 ? num_threads - 1 - std::min(n_inflaters, num_threads - 2) : 1u);
    const bool wide = !mhap && __builtin_cpu_supports("avx512bw") && getenv("RALA_IO_NO_AVX512") == nullptr;

    std::mutex m;
    std::condition_variable cv_block, cv_room;
    std::deque<std::unique_ptr<TextBlock>> queue;
    const size_t max_queued = 2 * (size_t)n_parsers + 2;
    bool done = false, failed = false;
    std::vector<std::unique_ptr<BlockColumns>> results;         // by block index (under m when resized)

    auto parse_block = [&](TextBlock& tb) {
        std::unique_ptr<BlockColumns> res(new BlockColumns);
        const char* text = tb.text.data() + kFront;
        const size_t n = tb.n;
        size_t lines = 0;
        for (const char* q = text; q < text + n;) {             // an upper bound of the records
            const char* nl = (const char*)memchr(q, '\n', (size_t)(text + n - q));
            ++lines;
            if (!nl) break;
            q = nl + 1;
        }
        OverlapColumns& c = res->cols;
        c.a_id.resize(lines); c.b_id.resize(lines); c.a_begin.resize(lines); c.a_end.resize(lines);
        c.b_begin.resize(lines); c.b_end.resize(lines); c.length.resize(lines); c.strand.resize(lines);
        std::unique_ptr<Chunk> ck(new Chunk);
        ck->a_id = c.a_id.data(); ck->b_id = c.b_id.data(); ck->a_begin = c.a_begin.data(); ck->a_end = c.a_end.data();
        ck->b_begin = c.b_begin.data(); ck->b_end = c.b_end.data(); ck->length = c.length.data(); ck->strand = c.strand.data();
        if (wide) {
            (void)parse_lines_avx512(text, 0, n, n, true, names, read_len, check_lengths, *ck);
        } else {
            size_t k = 0;
            while (k < n) {
                const char* nl = (const char*)memchr(text + k, '\n', n - k);
                const size_t e = nl ? (size_t)(nl - text) : n;
                size_t le = e;
                if (le > k && text[le - 1] == '\r') --le;
                if (le > k) {
                    if (mhap) parse_mhap_line(text + k, text + le, read_len, check_lengths, *ck);
                    else parse_paf_line(text + k, text + le, names, read_len, check_lengths, *ck);
                }
                k = e + 1;
            }
        }
        if (!mhap) resolve_batch(names, read_len, check_lengths, *ck);
        res->n = ck->n;
        res->error_read = ck->error_read;
        return res;
    };

    std::vector<std::thread> parsers;
    for (uint32_t k = 0; k < n_parsers; ++k) {
        parsers.emplace_back([&] {
            for (;;) {
                std::unique_ptr<TextBlock> tb;
                {
                    std::unique_lock<std::mutex> hold(m);
                    cv_block.wait(hold, [&] { return !queue.empty() || done; });
                    if (queue.empty()) return;
                    tb = std::move(queue.front());
                    queue.pop_front();
                }
                cv_room.notify_one();
                std::unique_ptr<BlockColumns> res = parse_block(*tb);
                std::lock_guard<std::mutex> hold(m);
                if (results.size() <= tb->index) results.resize(tb->index + 1);
                results[tb->index] = std::move(res);
            }
        });
    }

    // this thread inflates (or puts the inflaters' blocks together): blocks end with a line; what follows the last
    // newline opens the next block
    std::vector<char> carry, piece;
    size_t piece_at = 0;
    auto read_text = [&](char* dst, size_t want) -> long {         // bytes, 0 at the end, -1 on a broken file
        if (!bgzf) {
            // (a gzip stream that ends before its trailer - a download cut short - gives a SHORT read, not an error: the
            // state of the stream says which; plain text read through gzread has no trailer to miss)
            const int got = gzread(in, dst, (unsigned)std::min<size_t>(want, (size_t)1 << 20));
            if (got < 0) return -1;
            if ((size_t)got < std::min<size_t>(want, (size_t)1 << 20)) {
                int zerr = Z_OK;
                (void)gzerror(in, &zerr);
                if (zerr != Z_OK && zerr != Z_STREAM_END) return -1;
            }
            return (long)got;
        }
        size_t have = 0;
        while (have < want) {
            if (piece_at == piece.size()) {
                bool bad = false;
                piece_at = 0;
                piece.clear();
                if (!blocks->next(piece, bad)) return bad ? -1 : (long)have;
            }
            const size_t take = std::min(want - have, piece.size() - piece_at);
            memcpy(dst + have, piece.data() + piece_at, take);
            piece_at += take;
            have += take;
        }
        return (long)have;
    };
    size_t n_blocks = 0;
    for (bool eof = false; !eof;) {
        std::unique_ptr<TextBlock> tb(new TextBlock);
        const size_t target = carry.size() + kBlockText;        // (a line longer than a block grows the next one)
        tb->text.resize(kFront + target + kBack);
        memset(tb->text.data(), 0, kFront);
        char* text = tb->text.data() + kFront;
        memcpy(text, carry.data(), carry.size());
        size_t have = carry.size();
        carry.clear();
        while (have < target) {
            const long got = read_text(text + have, target - have);
            if (got < 0) { failed = true; eof = true; break; }
            if (got == 0) { eof = true; break; }
            have += (size_t)got;
        }
        size_t n = have;
        if (!eof) {
            const char* nl = (const char*)memrchr(text, '\n', have);
            if (!nl) {                                          // no line ends in here: all of it opens the next block
                carry.assign(text, text + have);
                continue;
            }
            n = (size_t)(nl - text) + 1;
            carry.assign(text + n, text + have);
        }
        if (n == 0) continue;
        memset(text + n, 0, kBack);
        tb->n = n;
        tb->index = n_blocks++;
        {
            std::unique_lock<std::mutex> hold(m);
            cv_room.wait(hold, [&] { return queue.size() < max_queued; });
            queue.push_back(std::move(tb));
        }
        cv_block.notify_one();
    }
    {
        std::lock_guard<std::mutex> hold(m);
        done = true;
    }
    cv_block.notify_all();
    for (auto& t : parsers) t.join();
    blocks.reset();
    if (in && gzclose(in) != Z_OK) failed = true;              // (Z_BUF_ERROR: the file ended inside the stream)
    if (raw) fclose(raw);
    if (failed) return false;

    // the blocks' columns into the final ones, in file order
    std::vector<size_t> at(n_blocks + 1);
    at[0] = out.size();
    for (size_t b = 0; b < n_blocks; ++b) at[b + 1] = at[b] + (results.size() > b && results[b] ? results[b]->n : 0);
    const size_t total = at[n_blocks];
    out.a_id.resize(total); out.b_id.resize(total); out.a_begin.resize(total); out.a_end.resize(total);
    out.b_begin.resize(total); out.b_end.resize(total); out.length.resize(total); out.strand.resize(total);
    for (size_t b = 0; b < n_blocks; ++b) {                     // the first mismatching line in file order
        if (results.size() > b && results[b] && results[b]->error_read >= 0) {
            if (length_error) *length_error = results[b]->error_read;
            break;
        }
    }
    {
        std::atomic<size_t> next(0);
        auto copy = [&] {
            for (size_t b = next.fetch_add(1); b < n_blocks; b = next.fetch_add(1)) {
                if (results.size() <= b || !results[b]) continue;
                const BlockColumns& r = *results[b];
                const size_t w = at[b], n = r.n;
                memcpy(out.a_id.data() + w, r.cols.a_id.data(), n * 4); memcpy(out.b_id.data() + w, r.cols.b_id.data(), n * 4);
                memcpy(out.a_begin.data() + w, r.cols.a_begin.data(), n * 4); memcpy(out.a_end.data() + w, r.cols.a_end.data(), n * 4);
                memcpy(out.b_begin.data() + w, r.cols.b_begin.data(), n * 4); memcpy(out.b_end.data() + w, r.cols.b_end.data(), n * 4);
                memcpy(out.length.data() + w, r.cols.length.data(), n * 4); memcpy(out.strand.data() + w, r.cols.strand.data(), n);
                results[b].reset();
            }
        };
        std::vector<std::thread> th;
        for (uint32_t k = 1; k < n_parsers; ++k) th.emplace_back(copy);
        copy();
        for (auto& t : th) t.join();
    }
    return true;
}

}  // namespace io
}  // namespace rala
