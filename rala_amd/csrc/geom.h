// Overlap / pile-interval arithmetic shared by the HIP kernels and the host
// side of librala_hip (compiled by hipcc for both).  Plain functions on
// plain values; IEEE double, no contraction (-ffp-contract=off).
//
// Reference behaviour followed (rvaser/rala):
//   Overlap::trim   src/overlap.cpp:117-192
//   Overlap::type   src/overlap.cpp:194-259
//   edge lengths    src/graph.cpp:582-617
//   Pile::shrink / break_over_chimeric_{hills,pits}   src/pile.cpp:299-322, 366-402, 471-498
//   Pile::check_chimeric_hills                        src/pile.cpp:457-469
//   intervalMerge                                     src/pile.cpp:31-52
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#define RALA_HD __host__ __device__ inline
#else
#define RALA_HD inline
#endif

namespace rala_hip {

enum : uint32_t {
    kTypeX = 0,   // internal / bad
    kTypeA = 1,   // b contained in a
    kTypeB = 2,   // a contained in b
    kTypeAB = 3,  // a's suffix overlaps b's prefix
    kTypeBA = 4
};

constexpr uint32_t kMinRegion = 1260;      // pile.cpp:307
constexpr uint32_t kMinOverlapSpan = 84;   // overlap.cpp:176
constexpr uint32_t kHillFuzz = 420;        // pile.cpp:432
constexpr uint32_t kSlopeWindow = 847;     // pile.cpp:68
constexpr uint32_t kMinCoverage = 4;       // pile.cpp:329

struct Coords {
    uint32_t a_begin, a_end, b_begin, b_end, length;
};

RALA_HD uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }
RALA_HD uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }

// Clip an overlap to the valid regions [Ba,Ea) and [Bb,Eb) of its reads.
// false = drop the overlap.  Coordinates stay in untrimmed read space.
RALA_HD bool ovl_trim(Coords& c, uint32_t strand, uint32_t Ba, uint32_t Ea, uint32_t Bb, uint32_t Eb) {
    if (c.a_begin >= Ea || c.a_end <= Ba || c.b_begin >= Eb || c.b_end <= Bb) return false;
    const uint32_t ca0 = c.a_begin < Ba ? Ba - c.a_begin : 0;
    const uint32_t ca1 = c.a_end > Ea ? c.a_end - Ea : 0;
    const uint32_t cb0 = c.b_begin < Bb ? Bb - c.b_begin : 0;
    const uint32_t cb1 = c.b_end > Eb ? c.b_end - Eb : 0;
    uint32_t ab, ae, bb, be;
    if (strand) {
        ab = c.a_begin + cb1; ae = c.a_end - cb0;
        bb = c.b_begin + ca1; be = c.b_end - ca0;
    } else {
        ab = c.a_begin + cb0; ae = c.a_end - cb1;
        bb = c.b_begin + ca0; be = c.b_end - ca1;
    }
    if (ab >= Ea || ae <= Ba || bb >= Eb || be <= Bb) return false;
    ab = umax(ab, Ba); ae = umin(ae, Ea);
    bb = umax(bb, Bb); be = umin(be, Eb);
    if (ab >= ae || ae - ab < kMinOverlapSpan || bb >= be || be - bb < kMinOverlapSpan) return false;
    c.a_begin = ab; c.a_end = ae; c.b_begin = bb; c.b_end = be;
    c.length = umax(ae - ab, be - bb);
    return true;
}

// Overlap geometry rebased to the valid regions; b flipped on the opposite strand.
struct Rebased {
    uint32_t la, a0, a1, lb, b0, b1;
};

RALA_HD Rebased ovl_rebase(const Coords& c, uint32_t strand, uint32_t Ba, uint32_t Ea, uint32_t Bb, uint32_t Eb) {
    Rebased r;
    r.la = Ea - Ba; r.a0 = c.a_begin - Ba; r.a1 = c.a_end - Ba;
    r.lb = Eb - Bb;
    if (strand == 0) {
        r.b0 = c.b_begin - Bb; r.b1 = c.b_end - Bb;
    } else {
        r.b0 = r.lb - c.b_end + Bb; r.b1 = r.lb - c.b_begin + Bb;
    }
    return r;
}

RALA_HD uint32_t ovl_type(const Coords& c, uint32_t strand, uint32_t Ba, uint32_t Ea, uint32_t Bb, uint32_t Eb) {
    const Rebased r = ovl_rebase(c, strand, Ba, Ea, Bb, Eb);
    const uint32_t ta = r.la - r.a1, tb = r.lb - r.b1;
    const uint32_t oh = umin(r.a0, r.b0) + umin(ta, tb);
    const uint32_t sa = r.a1 - r.a0, sb = r.b1 - r.b0;
    if ((double)sa < (double)(uint32_t)(sa + oh) * 0.875 || (double)sb < (double)(uint32_t)(sb + oh) * 0.875) {
        return kTypeX;
    }
    if (r.a0 <= r.b0 && ta <= tb) return kTypeB;
    if (r.a0 >= r.b0 && ta >= tb) return kTypeA;
    const uint32_t ra = c.a_end - c.a_begin, rb = c.b_end - c.b_begin;
    const uint32_t dspan = ra > rb ? ra - rb : rb - ra;
    if ((double)dspan < (double)c.length * 0.01) {
        const uint32_t me = (uint32_t)(0.05 * (double)umax(r.la, r.lb));
        const uint32_t d0 = r.a0 > r.b0 ? r.a0 - r.b0 : r.b0 - r.a0;
        if (d0 < me) return ta >= tb ? kTypeA : kTypeB;
        const uint32_t d1 = ta > tb ? ta - tb : tb - ta;
        if (d1 < me) return r.a0 >= r.b0 ? kTypeA : kTypeB;
    }
    return r.a0 > r.b0 ? kTypeAB : kTypeBA;
}

// The two edges of a dovetail overlap (graph.cpp:594-629).  node ids: read
// rank*2 (+1 = reverse complement); edge e and its twin e^1.
struct EdgePair {
    uint32_t src0, dst0, len0;   // edge
    uint32_t src1, dst1, len1;   // twin
};

RALA_HD bool ovl_edges(const Coords& c, uint32_t strand, uint32_t type, uint32_t node_a, uint32_t node_b_fwd,
                       uint32_t Ba, uint32_t Ea, uint32_t Bb, uint32_t Eb, EdgePair& e) {
    const Rebased r = ovl_rebase(c, strand, Ba, Ea, Bb, Eb);
    const uint32_t na = node_a, nb = node_b_fwd + strand;
    if (type == kTypeAB) {
        e.src0 = na; e.dst0 = nb; e.len0 = r.a0 - r.b0;
        e.src1 = nb ^ 1; e.dst1 = na ^ 1; e.len1 = (r.lb - r.b1) - (r.la - r.a1);
        return true;
    }
    if (type == kTypeBA) {
        e.src0 = nb; e.dst0 = na; e.len0 = r.b0 - r.a0;
        e.src1 = na ^ 1; e.dst1 = nb ^ 1; e.len1 = (r.la - r.a1) - (r.lb - r.b1);
        return true;
    }
    return false;
}

// Pile::break_over_chimeric_pits / _hills (pile.cpp:366-402, 471-498) ask the same question: the read is cut at every
// interval that lies inside its valid region [B, E) and counts (`cuts(i)`: a pit that is real, a hill that too few overlaps
// span) - which piece between two cuts is the longest (the first of the longest)?  That piece is the region Pile::shrink
// is then given.  at(i, first, second) hands out interval i; cuts(i) is asked only about intervals inside the region.
struct Piece {
    uint32_t begin, end;
};
template <class At, class Cuts>
RALA_HD Piece longest_piece(uint32_t B, uint32_t E, uint32_t n, At at, Cuts cuts) {
    Piece best = {0, 0};
    uint32_t from = B;
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t first, second;
        at(i, first, second);
        if (B > first || E < second) continue;
        if (!cuts(i)) continue;
        if ((uint32_t)(first - from) > (uint32_t)(best.end - best.begin)) { best.begin = from; best.end = first; }
        from = second;
    }
    if ((uint32_t)(E - from) > (uint32_t)(best.end - best.begin)) { best.begin = from; best.end = E; }
    return best;
}

// graph.cpp:26-29
RALA_HD bool comparable(double a, double b, double eps) {
    return (a >= b * (1 - eps) && a <= b * (1 + eps)) || (b >= a * (1 - eps) && b <= a * (1 + eps));
}

// Single-pass, non-transitive interval merge (pile.cpp:31-52) on small
// arrays: for each i in input order not yet absorbed, sweep every other j not
// absorbed once; each j strictly overlapping the current (growing) i is
// absorbed; then i is emitted.  An emitted i is never flagged, so a later
// absorber can absorb it again (SURVEY B-T7).  `gone` is scratch of n bytes;
// out_first/out_second must not alias the inputs.  Returns the new count.
RALA_HD uint32_t interval_merge(uint32_t* first, uint32_t* second, uint32_t n, uint8_t* gone,
                                     uint32_t* out_first, uint32_t* out_second) {
    for (uint32_t i = 0; i < n; ++i) gone[i] = 0;
    uint32_t m = 0;
    for (uint32_t i = 0; i < n; ++i) {
        if (gone[i]) continue;
        for (uint32_t j = 0; j < n; ++j) {
            if (j == i || gone[j]) continue;
            if (first[i] < second[j] && second[i] > first[j]) {
                gone[j] = 1;
                first[i] = umin(first[i], first[j]);
                second[i] = umax(second[i], second[j]);
            }
        }
        out_first[m] = first[i];
        out_second[m] = second[i];
        ++m;
    }
    return m;
}

}  // namespace rala_hip
