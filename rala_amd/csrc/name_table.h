// Read name -> id as the tokenisers look it up, on the host (rala_amd/host/io.cpp: NameTable) and on the device
// (ingest_kernels.hip): open addressing over 32-byte buckets - hash, id, length and the first 16 bytes of the name in one
// line - and the names' bytes in an arena for the names that are longer.  The table is built on the host
// (rala::io::NameTable::build) and handed to the device as it is (rala_hip_set_name_table): one definition of the hash and
// of a bucket for both sides.
#pragma once

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define RALA_NT_HD __host__ __device__ inline
#else
#define RALA_NT_HD inline
#endif

namespace rala_hip {

struct alignas(32) NameBucket {
    uint32_t hash32;        // high half of the hash
    uint32_t id1;           // name index + 1, 0 = empty
    uint32_t len;
    uint32_t off;           // start of the name in the arena
    char head[16];          // first min(len, 16) bytes, zero padded
};

// 8 bytes at a time (names are short: a byte-wise FNV chain cost more than the table probe)
template <class Load8>      // load8(k, n) -> the n (<= 8) bytes at offset k, little endian, zero padded
RALA_NT_HD uint64_t name_hash_with(uint64_t n, Load8 load8) {
    uint64_t h = 0x9E3779B97F4A7C15ull ^ (n * 0xFF51AFD7ED558CCDull);
    uint64_t k = 0;
    while (n - k >= 8) {
        h = (h ^ load8(k, 8)) * 0xC2B2AE3D27D4EB4Full;
        h ^= h >> 29;
        k += 8;
    }
    if (n - k) {
        h = (h ^ load8(k, n - k)) * 0xC2B2AE3D27D4EB4Full;
        h ^= h >> 29;
    }
    h *= 0x165667B19E3779F9ull;
    return h ^ (h >> 32);
}

}  // namespace rala_hip
