// Minimal readers for the formats the reference takes through its (un-vendored)
// bioparser submodule: FASTA / FASTQ sequences, PAF / MHAP overlaps, plain or gzip.
// Names are cut at the first whitespace.
#pragma once

#include <stdint.h>
#include <functional>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "../csrc/name_table.h"

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

// ---- one-pass, multi-threaded ingest of an uncompressed PAF file into binary columns -------
// (the reference tokenises every line into a heap Overlap twice,
// src/graph.cpp:328,443)

// read name -> id, looked up straight from the bytes of the file (no std::string per field).
// Open addressing over 32-byte buckets: hash, id, length and the first 16 bytes of the name sit
// in ONE cache line, so a lookup of a name of up to 16 characters is one memory access (a table
// of a million names does not fit any cache: with separate offset / length / arena arrays every
// probe was four dependent misses); longer names confirm against the arena.  prefetch() lets a
// caller start the bucket's load while it parses on (io.cpp: lines are resolved in batches).
class NameTable {
public:
    void build(const std::vector<std::string>& names);
    static uint64_t hash(const char* p, size_t n);
    void prefetch(uint64_t h) const { if (bucket_) __builtin_prefetch(&bucket_[h & mask_]); }
    // id of the name [p, p + n) whose hash is h, or ~0ull
    uint64_t find(const char* p, size_t n, uint64_t h) const;
    uint64_t find(const char* p, size_t n) const { return find(p, n, hash(p, n)); }
    // the table as it is, for the device tokeniser (rala_hip_set_name_table)
    const rala_hip::NameBucket* buckets() const { return bucket_; }
    size_t n_buckets() const { return n_bucket_; }
    const std::string& arena() const { return arena_; }
private:
    typedef rala_hip::NameBucket Bucket;
    // (2 MiB-aligned block marked for transparent huge pages: a probe of a 64 MB table would
    // otherwise miss the TLB as well as the caches)
    Bucket* bucket_ = nullptr;
    size_t n_bucket_ = 0;
    std::string arena_;
    uint64_t mask_ = 0;
public:
    NameTable() = default;
    ~NameTable();
    NameTable(const NameTable&) = delete;
    NameTable& operator=(const NameTable&) = delete;
};

// allocator whose resize() leaves new elements uninitialised (the readers overwrite all of them;
// zero-filling 29 bytes per overlap on one thread would cost as much as parsing)
// Large blocks are 2 MiB aligned and marked for transparent huge pages: a 50 M-overlap file puts
// 6 GB of fresh memory behind these vectors - 1.5 M first-touch faults with 4 KiB pages.
void* allocate_block(size_t bytes);
void free_block(void* p, size_t bytes);

template <class T>
struct UninitAllocator : std::allocator<T> {
    template <class U> struct rebind { typedef UninitAllocator<U> other; };
    UninitAllocator() = default;
    template <class U> UninitAllocator(const UninitAllocator<U>&) {}
    T* allocate(size_t n) { return (T*)allocate_block(n * sizeof(T)); }
    T* allocate(size_t n, const void*) { return allocate(n); }
    void deallocate(T* p, size_t n) { free_block(p, n * sizeof(T)); }
    template <class U> void construct(U* p) { ::new ((void*)p) U; }
    template <class U, class... A> void construct(U* p, A&&... a) { ::new ((void*)p) U(std::forward<A>(a)...); }
};

struct OverlapColumns {
    typedef std::vector<uint32_t, UninitAllocator<uint32_t>> U32;
    typedef std::vector<uint8_t, UninitAllocator<uint8_t>> U8;
    U32 a_id, b_id, a_begin, a_end, b_begin, b_end, length;
    U8 strand;
    size_t size() const { return a_id.size(); }
};

// Appends the records of `path` to `out` in file order.  Ids of unknown names are 0xFFFFFFFF.
// Same checks as Overlap::transmute (src/overlap.cpp:36-82): a PAF length that differs from
// the sequence's is fatal - returned as the offending read id in *length_error (the first such
// line in file order), the caller prints the reference's message.  false = cannot open / map.
bool read_paf_parallel(const std::string& path, const NameTable& names, const std::vector<uint32_t>& read_len,
    bool check_lengths, uint32_t num_threads, OverlapColumns& out, int64_t* length_error);

// The same result for a gzip-compressed (or plain) PAF file, or an MHAP file (mhap = true: blank-separated
// columns, 1-based ids, reference src/overlap.cpp:12-20; `names` is not used then): ONE thread inflates -
// a gzip stream cannot be cut - and hands blocks of whole lines to the others, which parse them with the
// tokenizer of read_paf_parallel.
bool read_overlaps_streamed(const std::string& path, bool mhap, const NameTable& names, const std::vector<uint32_t>& read_len,
    bool check_lengths, uint32_t num_threads, OverlapColumns& out, int64_t* length_error);

}  // namespace io
}  // namespace rala
