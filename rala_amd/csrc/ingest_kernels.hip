// PAF (and MHAP) text -> binary overlap columns on the device.
//
// Replaces, for an uncompressed PAF file, the tokeniser in front of Graph::initialize (reference src/graph.cpp:328-352:
// bioparser's PAF parser, one heap Overlap per line, and Overlap::transmute's two hash look-ups, src/overlap.cpp:36-82).
// The host readers (rala_amd/host/io.cpp) parse 50 M lines in 330 ms on 16 threads - 88 % of the time from PAF text to the
// reduced graph; the text itself crosses PCIe in 60 ms, and here 50 M lines are 50 M independent threads.
//
// The text of a RANGE of the file lies in device memory as one buffer: the n bytes whose line starts are this launch's, and
// behind them up to n_avail bytes for the last lines' first eleven columns (n_avail = n at the end of the file, where a
// newline stands in; a rank of a sharded run tokenises its byte range of the file this way - round 5).  `first_is_start`:
// the range's first byte starts a line (it does at the start of the file; elsewhere the byte in front of it says).
// Two passes over the text:
//   count   per chunk of 16 KB the lines that START in it (a position whose predecessor is a newline - or the first byte
//           of the file - and that is no newline itself); an exclusive scan gives every chunk its first row
//   parse   a workgroup per chunk: the chunk and a halo of 2 KB behind it staged in LDS, the line starts listed in order,
//           one thread per line: the first eleven tabs, numbers as their leading digits (mod 2^32, as the host readers take
//           them), names cut at the first blank and looked up in the name table (name_table.h: the host's table as it is),
//           the strand, Overlap::transmute's length check; row = the chunk's first row + the line's place in the chunk
// Exactly the columns rala::io::read_paf_parallel gives for a file of 12-column records.  Anything else - a line with fewer
// columns, a first eleven columns longer than the halo, more lines in a chunk than its list holds - only raises a flag:
// the caller then takes the host reader, which knows what to do with such files (tests/test_gpu_ingest.py).
#include <hip/hip_runtime.h>

#include "device_utils.h"
#include "kernels.h"
#include "name_table.h"

namespace rala_hip {

namespace {

constexpr uint32_t kChunk = 16384;          // bytes of text per workgroup
constexpr uint32_t kHalo = 2048;            // bytes behind the chunk a line that starts in it may use for its first 11 columns
constexpr uint32_t kBlock = 256;
constexpr uint32_t kSeg = kChunk / kBlock;  // 64 bytes per thread
constexpr uint32_t kMaxLines = 1024;        // line starts per chunk (a 12-column record has at least 23 bytes)
static_assert(kSeg == 64, "one 64-bit mask per thread");

// the newline positions among the 64 bytes of 16 words, as a bit mask
__device__ __forceinline__ uint64_t newline_mask(const uint32_t* w) {
    uint64_t m = 0;
#pragma unroll
    for (uint32_t k = 0; k < 16; ++k) {
        const uint32_t x = w[k] ^ 0x0A0A0A0Au;                          // zero bytes where the newlines are
        uint32_t z = ((x - 0x01010101u) & ~x & 0x80808080u);            // 0x80 in those bytes
        // (the borrow trick may also flag a byte above a zero byte: 0x01 ^ ... - confirm byte-wise where anything is flagged)
        if (z) {
            uint32_t exact = 0;
#pragma unroll
            for (uint32_t b = 0; b < 4; ++b) if (((x >> (8 * b)) & 0xFFu) == 0) exact |= 1u << b;
            m |= (uint64_t)exact << (4 * k);
        }
    }
    return m;
}

// line starts among the 64 bytes at text + j0 (j0 a multiple of 64): not a newline, and behind a newline or at the file's
// first byte; bytes at or beyond n are none
__device__ __forceinline__ uint64_t line_starts(const uint32_t* w, uint64_t j0, uint64_t n, bool prev_is_newline) {
    const uint64_t nl = newline_mask(w);
    uint64_t valid = ~0ull;
    if (j0 + 64 > n) valid = j0 >= n ? 0ull : (1ull << (n - j0)) - 1ull;
    const uint64_t behind = (nl << 1) | (prev_is_newline ? 1ull : 0ull);
    return behind & ~nl & valid;
}

__global__ __launch_bounds__(kBlock) void paf_count_kernel(const uint8_t* __restrict__ text, uint64_t n, uint32_t first_is_start,
                                                            uint32_t* __restrict__ chunk_lines) {
    __shared__ uint32_t tmp[kBlock / 64 + 1];
    const uint64_t j0 = (uint64_t)blockIdx.x * kChunk + threadIdx.x * kSeg;
    uint32_t w[16];
    uint64_t starts = 0;
    if (j0 < n) {
        const uint4* p = (const uint4*)(text + j0);             // (the buffer is padded to a whole chunk)
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) {
            const uint4 v = p[k];
            w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w;
        }
        starts = line_starts(w, j0, n, j0 == 0 ? first_is_start != 0 : text[j0 - 1] == '\n');
    }
    const uint32_t total = block_reduce<(int)kBlock>((uint32_t)__popcll(starts), OpAdd(), 0u, tmp);
    if (threadIdx.x == 0) chunk_lines[blockIdx.x] = total;
}

struct NameTableDev {
    const NameBucket* bucket;
    const char* arena;
    uint64_t mask;
};

// eight bytes of the window from any byte offset (three aligned words, funnelled): the names are read by the word
__device__ __forceinline__ uint64_t window8(const uint8_t* win, uint32_t pos) {
    const uint32_t* w = (const uint32_t*)win + (pos >> 2);
    const uint32_t a = w[0], b = w[1], c = w[2];
    const uint32_t sh = (pos & 3u) * 8u;
    const uint32_t lo = sh ? (a >> sh) | (b << (32u - sh)) : a;
    const uint32_t hi = sh ? (b >> sh) | (c << (32u - sh)) : b;
    return (uint64_t)lo | ((uint64_t)hi << 32);
}
__device__ __forceinline__ uint64_t low_bytes(uint64_t w, uint32_t m) { return m >= 8 ? w : w & ((1ull << (8u * m)) - 1ull); }

// rala::io::NameTable::find on the device for the name of n bytes at win + pos
__device__ __forceinline__ uint32_t find_name(const NameTableDev& T, const uint8_t* win, uint32_t pos, uint32_t n) {
    const uint64_t h = name_hash_with((uint64_t)n, [&](uint64_t k, uint64_t m) { return low_bytes(window8(win, pos + (uint32_t)k), (uint32_t)m); });
    const uint32_t h32 = (uint32_t)(h >> 32);
    // the name's first 16 bytes, zero padded, as the buckets keep them
    const uint64_t n0 = low_bytes(window8(win, pos), n), n1 = n > 8 ? low_bytes(window8(win, pos + 8), n - 8) : 0ull;
    for (uint64_t k = h & T.mask;; k = (k + 1) & T.mask) {
        const uint4* b = (const uint4*)(T.bucket + k);
        const uint4 meta = b[0];                                // hash32, id1, len, off
        if (meta.y == 0) return 0xFFFFFFFFu;
        if (meta.x != h32 || meta.z != n) continue;
        const uint4 head = b[1];
        if ((((uint64_t)head.y << 32) | head.x) != n0 || (((uint64_t)head.w << 32) | head.z) != n1) continue;
        bool same = true;
        for (uint32_t i = 16; i < n && same; ++i) same = (uint8_t)T.arena[meta.w + i] == win[pos + i];
        if (same) return meta.y - 1u;
    }
}

// kMhap: the file is MHAP (reference src/overlap.cpp:12-20; bioparser's MhapParser in front of it): twelve blank-separated columns
// "a_id b_id error minmers a_rc a_begin a_end a_length b_rc b_begin b_end b_length", all numbers, ids from 1 - no names to look up;
// id = the column minus one (a value beyond the reads does not resolve), length = the longer of the two spans, strand = a_rc !=
// b_rc: the columns rala::io::read_overlaps_streamed gives for such a file (parse_mhap_line, io.cpp).  Column 12 ends with the line.
template <bool kMhap>
__global__ __launch_bounds__(kBlock) void paf_parse_kernel(const uint8_t* __restrict__ text, uint64_t n, uint64_t n_avail, uint32_t first_is_start,
                                                            const uint32_t* __restrict__ chunk_row,
                                                           NameTableDev names, const uint32_t* __restrict__ read_len, uint32_t n_reads,
                                                           uint32_t check_lengths, PafColumns out, uint32_t* flags,
                                                           unsigned long long* first_bad) {
    __shared__ __align__(16) uint8_t win[kChunk + kHalo + 16];
    __shared__ uint32_t tmp[kBlock / 64 + 2];
    __shared__ uint16_t line_at[kMaxLines];
    const uint64_t c0 = (uint64_t)blockIdx.x * kChunk;
    const uint32_t tid = threadIdx.x;
    // the chunk and its halo; what lies beyond the text reads as newlines
    for (uint32_t v = tid; v < (kChunk + kHalo) / 16; v += kBlock) {
        const uint64_t j = c0 + (uint64_t)v * 16;
        uint4 x = make_uint4(0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au);
        if (j + 16 <= n_avail) {
            x = *(const uint4*)(text + j);
        } else if (j < n_avail) {
            uint8_t b[16];
            for (uint32_t i = 0; i < 16; ++i) b[i] = j + i < n_avail ? text[j + i] : (uint8_t)'\n';
            x = *(const uint4*)b;
        }
        ((uint4*)win)[v] = x;
    }
    __syncthreads();
    // the line starts of this chunk, in order
    uint64_t starts;
    {
        const uint32_t* w = (const uint32_t*)(win + tid * kSeg);
        uint32_t ww[16];
#pragma unroll
        for (uint32_t k = 0; k < 16; ++k) ww[k] = w[k];
        const uint64_t j0 = c0 + tid * kSeg;
        const bool prev_nl = j0 == 0 ? first_is_start != 0 : (tid == 0 ? text[j0 - 1] == '\n' : win[tid * kSeg - 1] == '\n');
        starts = line_starts(ww, j0, n, prev_nl);
    }
    uint32_t total;
    uint32_t at = block_scan_excl<(int)kBlock>((uint32_t)__popcll(starts), OpAdd(), 0u, tmp, total);
    if (total > kMaxLines) {
        if (tid == 0) atomicOr(flags, 2u);          // lines too short to be records
        return;
    }
    while (starts) {
        const uint32_t b = (uint32_t)__builtin_ctzll(starts);
        starts &= starts - 1;
        line_at[at++] = (uint16_t)(tid * kSeg + b);
    }
    __syncthreads();
    const uint32_t row0 = chunk_row[blockIdx.x];
    for (uint32_t li = tid; li < total; li += kBlock) {
        const uint32_t s = line_at[li];
        // the first eleven tabs of the line (it ends at a newline; a carriage return in front of that belongs to no column we read)
        uint32_t tab[11];
        uint32_t nt = 0, p = s;
        const uint32_t limit = kChunk + kHalo;
        while (p < limit && nt < 11) {
            const uint8_t ch = win[p];
            if (ch == '\n') break;
            if (ch == (kMhap ? ' ' : '\t')) tab[nt++] = p;
            ++p;
        }
        if (kMhap && nt == 11) {
            // the twelfth column is read up to the line's end: that end (or the thirteenth column's blank) within the window
            uint32_t q = p;
            while (q < limit && win[q] != '\n' && win[q] != ' ') ++q;
            if (q >= limit) { atomicOr(flags, 4u); continue; }
        }
        if (nt < 11) {
            // fewer than 12 columns, or the first eleven reach beyond the halo: the host reader's business
            atomicOr(flags, p >= limit ? 4u : 1u);
            continue;
        }
        auto number = [&](uint32_t b, uint32_t e) {           // the field's leading digits, mod 2^32 (io.cpp: parse_u32)
            uint64_t x = 0;
            for (uint32_t i = b; i < e; ++i) {
                const uint32_t d = (uint32_t)win[i] - (uint32_t)'0';
                if (d > 9u) break;
                x = x * 10u + d;
            }
            return (uint32_t)x;
        };
        auto name_end = [&](uint32_t b, uint32_t e) {         // names are cut at their first blank
            for (uint32_t i = b; i < e; ++i) if (win[i] == ' ') return i;
            return e;
        };
        if constexpr (kMhap) {
            auto id_of = [&](uint32_t b, uint32_t e) {            // the column's leading digits as a 64-bit number, minus one (ids from 1)
                uint64_t x = 0;
                for (uint32_t i = b; i < e; ++i) {
                    const uint32_t d = (uint32_t)win[i] - (uint32_t)'0';
                    if (d > 9u) break;
                    x = x * 10u + d;
                }
                x -= 1;
                return x < (uint64_t)n_reads ? (uint32_t)x : 0xFFFFFFFFu;
            };
            const uint32_t ia = id_of(s, tab[0]), ib = id_of(tab[0] + 1, tab[1]);
            const uint32_t a_rc = number(tab[3] + 1, tab[4]), ab = number(tab[4] + 1, tab[5]), ae = number(tab[5] + 1, tab[6]);
            const uint32_t al = number(tab[6] + 1, tab[7]), b_rc = number(tab[7] + 1, tab[8]), bb = number(tab[8] + 1, tab[9]);
            const uint32_t be = number(tab[9] + 1, tab[10]), bl = number(tab[10] + 1, limit);
            const uint32_t row = row0 + li;
            out.a_id[row] = ia; out.b_id[row] = ib;
            out.a_begin[row] = ab; out.a_end[row] = ae;
            out.b_begin[row] = bb; out.b_end[row] = be;
            out.length[row] = (ae - ab) > (be - bb) ? ae - ab : be - bb;           // (no alignment length in the file: the longer span, overlap.cpp:18)
            out.strand[row] = a_rc == b_rc ? 0 : 1;
            if (check_lengths) {
                uint32_t bad = 0xFFFFFFFFu;
                if (ia != 0xFFFFFFFFu && al != read_len[ia]) bad = ia;
                else if (ia != 0xFFFFFFFFu && ib != 0xFFFFFFFFu && bl != read_len[ib]) bad = ib;
                if (bad != 0xFFFFFFFFu) atomicMin(first_bad, ((unsigned long long)row << 32) | bad);
            }
            continue;
        }
        const uint32_t qn = name_end(s, tab[0]) - s;
        const uint32_t ql = number(tab[0] + 1, tab[1]);
        const uint32_t qb = number(tab[1] + 1, tab[2]);
        const uint32_t qe = number(tab[2] + 1, tab[3]);
        const uint8_t strand = (tab[3] + 1 < tab[4] ? win[tab[3] + 1] : (uint8_t)'+') == '+' ? 0 : 1;
        const uint32_t t0 = tab[4] + 1;
        const uint32_t tn = name_end(t0, tab[5]) - t0;
        const uint32_t tl = number(tab[5] + 1, tab[6]);
        const uint32_t tb = number(tab[6] + 1, tab[7]);
        const uint32_t te = number(tab[7] + 1, tab[8]);
        const uint32_t ol = number(tab[9] + 1, tab[10]);       // column 11: alignment length
        const uint32_t ia = find_name(names, win, s, qn);
        const uint32_t ib = find_name(names, win, t0, tn);
        const uint32_t row = row0 + li;
        out.a_id[row] = ia; out.b_id[row] = ib;
        out.a_begin[row] = qb; out.a_end[row] = qe;
        out.b_begin[row] = tb; out.b_end[row] = te;
        out.length[row] = ol;
        out.strand[row] = strand;
        if (check_lengths) {
            // Overlap::transmute (overlap.cpp:54-59, 73-78): a first, then b when both names are known
            uint32_t bad = 0xFFFFFFFFu;
            if (ia < n_reads && ql != read_len[ia]) bad = ia;
            else if (ia < n_reads && ib < n_reads && tl != read_len[ib]) bad = ib;
            if (bad != 0xFFFFFFFFu) atomicMin(first_bad, ((unsigned long long)row << 32) | bad);
        }
    }
}

}  // namespace

uint32_t paf_chunk_bytes() { return kChunk; }

uint32_t paf_halo_bytes() { return kHalo + 16; }

void launch_paf_count(const uint8_t* text, uint64_t n, bool first_is_start, uint32_t* chunk_lines, hipStream_t s) {
    const uint32_t chunks = (uint32_t)((n + kChunk - 1) / kChunk);
    if (chunks) hipLaunchKernelGGL(paf_count_kernel, dim3(chunks), dim3(kBlock), 0, s, text, n, first_is_start ? 1u : 0u, chunk_lines);
}

void launch_paf_parse(const uint8_t* text, uint64_t n, uint64_t n_avail, bool first_is_start, const uint32_t* chunk_row, const void* buckets, uint64_t n_buckets,
                      const char* arena, const uint32_t* read_len, uint32_t n_reads, bool check_lengths, const PafColumns& out,
                      uint32_t* flags, unsigned long long* first_bad, hipStream_t s, bool mhap) {
    const uint32_t chunks = (uint32_t)((n + kChunk - 1) / kChunk);
    NameTableDev T;
    T.bucket = (const NameBucket*)buckets; T.arena = arena; T.mask = n_buckets - 1;
    if (chunks && mhap) {
        hipLaunchKernelGGL(paf_parse_kernel<true>, dim3(chunks), dim3(kBlock), 0, s, text, n, n_avail, first_is_start ? 1u : 0u, chunk_row, T, read_len, n_reads,
                           check_lengths ? 1u : 0u, out, flags, first_bad);
    } else if (chunks) {
        hipLaunchKernelGGL(paf_parse_kernel<false>, dim3(chunks), dim3(kBlock), 0, s, text, n, n_avail, first_is_start ? 1u : 0u, chunk_row, T, read_len, n_reads,
                           check_lengths ? 1u : 0u, out, flags, first_bad);
    }
}

}  // namespace rala_hip
