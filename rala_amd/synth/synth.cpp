// Synthetic read / overlap generator.
//
// Produces, from a seed and with integer arithmetic only (splitmix64), the
// same reads and PAF-shaped overlaps on every machine: N reads of length
// ~N(10000, 1500) clipped at 3000 sampled from a genome of G bases, all-vs-all
// overlaps with genomic intersection >= 500 emitted once per pair
// (query = lower id, grouped by query, ordered by target), +-3 coordinate
// jitter, and the planted artefacts the hot path exists to find:
//   * chimeric reads (two genomic segments) with 5-10 junction-bridging
//     overlaps -> chimeric pits
//   * adapter prefixes with no overlaps -> valid-region trimming
//   * two-copy 3 kb repeats -> repeat-induced overlaps, duplicate pairs
//   * 30-80 deep stacks of short overlaps -> chimeric hills
// Only names and lengths of reads matter to the hot path (reference
// graph.cpp:256-258), so no bases are generated unless a FASTA is requested.
//
// This is bench/test input plumbing, not part of the product path.
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <string>
#include <vector>

namespace {

struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed) {}
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    // uniform in [0, n)
    uint64_t below(uint64_t n) { return n ? (uint64_t)(((__uint128_t)next() * n) >> 64) : 0; }
    // uniform in [lo, hi]
    int64_t range(int64_t lo, int64_t hi) { return lo + (int64_t)below((uint64_t)(hi - lo + 1)); }
    bool chance_permille(uint32_t p) { return below(1000) < p; }
};

struct Segment {
    uint32_t read;
    uint32_t read_off;   // first read coordinate covered by the segment
    uint32_t len;
    int64_t gstart;      // genome start of the segment
    uint8_t rev;         // read is the reverse strand of the genome here
    uint8_t alias;       // repeat copy-2 span re-expressed in copy-1 coordinates
};

struct Rec {
    uint32_t q, t, qb, qe, tb, te, len;
    uint8_t strand;
    uint32_t ord;        // generation order, keeps the sort stable
};

struct Synth {
    uint64_t n_reads, genome;
    std::vector<uint32_t> read_len;
    std::vector<uint8_t> read_kind;       // bit0 chimeric, bit1 adapter, bit2 stack
    std::vector<uint32_t> a_id, b_id, a_begin, a_end, b_begin, b_end, length;
    std::vector<uint8_t> strand;
    // derived sensitive set
    std::vector<uint32_t> s_a_id, s_b_id, s_a_begin, s_a_end, s_b_begin, s_b_end, s_length;
    std::vector<uint8_t> s_strand;
};

inline uint32_t clampu(int64_t v, int64_t lo, int64_t hi) { return (uint32_t)std::max(lo, std::min(hi, v)); }

// read coordinates of the genome interval [x, y) inside segment s
inline void to_read(const Segment& s, int64_t x, int64_t y, int64_t& rb, int64_t& re) {
    if (!s.rev) {
        rb = s.read_off + (x - s.gstart);
        re = s.read_off + (y - s.gstart);
    } else {
        rb = s.read_off + (s.gstart + s.len - y);
        re = s.read_off + (s.gstart + s.len - x);
    }
}

void generate(Synth& S, uint64_t n_reads, uint64_t genome, uint64_t seed, uint32_t plants) {
    Rng rng(seed * 0x2545F4914F6CDD1Dull + 0x1234567);
    S.n_reads = n_reads;
    S.genome = genome;
    S.read_len.resize(n_reads);
    S.read_kind.assign(n_reads, 0);

    const bool plant_chimera = plants & 1, plant_adapter = plants & 2, plant_repeat = plants & 4,
               plant_stack = plants & 8;

    // repeats: per 500 kb block one 3 kb span whose second copy lies elsewhere
    struct Repeat { int64_t p1, p2; };
    std::vector<Repeat> repeats;
    if (plant_repeat) {
        const int64_t block = 500000;
        for (int64_t b = 0; (b + 1) * block <= (int64_t)genome; ++b) {
            Repeat r;
            r.p1 = b * block + rng.range(0, block - 3001);
            r.p2 = rng.range(0, (int64_t)genome - 3001);
            if (r.p2 + 3000 > r.p1 && r.p2 < r.p1 + 3000) continue;   // copies must not touch
            repeats.push_back(r);
        }
        std::sort(repeats.begin(), repeats.end(), [](const Repeat& a, const Repeat& b) { return a.p2 < b.p2; });
    }

    std::vector<Segment> segs;
    segs.reserve(n_reads + n_reads / 8);
    std::vector<uint32_t> chim_cut(n_reads, 0);

    for (uint64_t i = 0; i < n_reads; ++i) {
        int64_t sum = 0;
        for (int k = 0; k < 12; ++k) sum += (int64_t)(rng.next() >> 48);
        int64_t L = 10000 + (1500 * (sum - 393210)) / 65536;
        if (plants & 16) {
            // heavy tail (what nanopore runs look like, not Appendix E's bell): 3 kb + an exponential with a mean of
            // 7 kb, one read in fifty another 30 - 90 kb on top: all length classes of the pile chain in one data set.
            // -ln(u / 2^53) in integers, the same on every machine: ln 2 per halving, linear inside the last octave
            // (-ln x ~ 2 ln 2 (1 - x) on [1/2, 1): close enough for a length).  e = 7000 * 1.024 * (-ln ..)
            uint64_t u = (rng.next() >> 11) | 1ull;              // 53 bits, never zero
            int64_t e = 0;
            while (u < (1ull << 52)) { u <<= 1; e += 4968; }      // 7000 * 1.024 * ln 2
            e += (int64_t)(((1ull << 53) - u) >> 43) * 4968 * 2 / 1024;
            L = 3000 + e * 1000 / 1024;
            if (rng.chance_permille(20)) L += rng.range(30000, 90000);
        }
        // ultra-long reads (bit 5): one read in eighty 100 - 400 kb.  At a coverage of five to seven such a read meets a
        // hundred and more others, every step of its shallow pile is a slope region (6 > int(4 * 1.3)): the lists that
        // the reference keeps in vectors grow with the read's length, not with the coverage.
        if ((plants & 32) && rng.chance_permille(12)) L = rng.range(100000, 400000);
        if (L < 3000) L = 3000;
        if ((uint64_t)L + 16 > genome) L = (int64_t)genome / 2;
        S.read_len[i] = (uint32_t)L;

        uint32_t adapter = 0;
        if (plant_adapter && rng.chance_permille(200)) {
            adapter = (uint32_t)rng.range(50, 400);
            S.read_kind[i] |= 2;
        }
        const bool chim = plant_chimera && rng.chance_permille(30);
        if (chim) {
            S.read_kind[i] |= 1;
            const uint32_t cut = (uint32_t)rng.range((int64_t)(L * 30 / 100), (int64_t)(L * 70 / 100));
            chim_cut[i] = cut;
            Segment s1, s2;
            s1.read = s2.read = (uint32_t)i;
            s1.alias = s2.alias = 0;
            s1.read_off = adapter; s1.len = cut - adapter;
            s2.read_off = cut; s2.len = (uint32_t)L - cut;
            s1.gstart = rng.range(0, (int64_t)genome - s1.len);
            s2.gstart = rng.range(0, (int64_t)genome - s2.len);
            s1.rev = (uint8_t)(rng.next() & 1);
            s2.rev = (uint8_t)(rng.next() & 1);
            segs.push_back(s1);
            segs.push_back(s2);
        } else {
            Segment s;
            s.read = (uint32_t)i; s.alias = 0;
            s.read_off = adapter; s.len = (uint32_t)L - adapter;
            s.gstart = rng.range(0, (int64_t)genome - s.len);
            s.rev = (uint8_t)(rng.next() & 1);
            segs.push_back(s);
        }
        if (plant_stack && !chim && L >= 6000 && rng.chance_permille(12)) S.read_kind[i] |= 4;
    }

    // alias segments for spans that cross a repeat's second copy
    if (!repeats.empty()) {
        const size_t n0 = segs.size();
        for (size_t k = 0; k < n0; ++k) {
            const Segment s = segs[k];
            const int64_t ge = s.gstart + s.len;
            // first repeat with p2 + 3000 > gstart
            size_t lo = 0, hi = repeats.size();
            while (lo < hi) {
                size_t mid = (lo + hi) / 2;
                if (repeats[mid].p2 + 3000 > s.gstart) hi = mid; else lo = mid + 1;
            }
            for (size_t r = lo; r < repeats.size() && repeats[r].p2 < ge; ++r) {
                const int64_t x = std::max<int64_t>(s.gstart, repeats[r].p2);
                const int64_t y = std::min<int64_t>(ge, repeats[r].p2 + 3000);
                if (y - x < 500) continue;
                int64_t rb, re;
                to_read(s, x, y, rb, re);
                Segment a;
                a.read = s.read; a.alias = 1; a.rev = s.rev;
                a.read_off = (uint32_t)rb; a.len = (uint32_t)(y - x);
                a.gstart = repeats[r].p1 + (x - repeats[r].p2);
                segs.push_back(a);
            }
        }
    }

    std::sort(segs.begin(), segs.end(), [](const Segment& a, const Segment& b) {
        if (a.gstart != b.gstart) return a.gstart < b.gstart;
        if (a.read != b.read) return a.read < b.read;
        return a.read_off < b.read_off;
    });

    std::vector<Rec> recs;
    recs.reserve(n_reads * 55);
    auto emit = [&](uint32_t r1, int64_t b1, int64_t e1, uint32_t r2, int64_t b2, int64_t e2, uint8_t strand) {
        // +-3 jitter, clamped to the reads
        const int64_t L1 = S.read_len[r1], L2 = S.read_len[r2];
        uint32_t qb = clampu(b1 + rng.range(-3, 3), 0, L1), qe = clampu(e1 + rng.range(-3, 3), 0, L1);
        uint32_t tb = clampu(b2 + rng.range(-3, 3), 0, L2), te = clampu(e2 + rng.range(-3, 3), 0, L2);
        if (qe < qb + 60 || te < tb + 60) return;      // keep the +-15 bound shrink well defined
        Rec r;
        if (r1 <= r2) { r.q = r1; r.qb = qb; r.qe = qe; r.t = r2; r.tb = tb; r.te = te; }
        else          { r.q = r2; r.qb = tb; r.qe = te; r.t = r1; r.tb = qb; r.te = qe; }
        r.len = std::max(r.qe - r.qb, r.te - r.tb);
        r.strand = strand;
        r.ord = (uint32_t)recs.size();
        recs.push_back(r);
    };

    for (size_t i = 0; i < segs.size(); ++i) {
        const Segment& A = segs[i];
        const int64_t ea = A.gstart + A.len;
        for (size_t j = i + 1; j < segs.size() && segs[j].gstart + 500 <= ea; ++j) {
            const Segment& B = segs[j];
            if (A.read == B.read || (A.alias && B.alias)) continue;
            const int64_t x = B.gstart;                      // B starts inside A
            const int64_t y = std::min<int64_t>(ea, B.gstart + B.len);
            if (y - x < 500) continue;
            int64_t ab, ae, bb, be;
            to_read(A, x, y, ab, ae);
            to_read(B, x, y, bb, be);
            emit(A.read, ab, ae, B.read, bb, be, (uint8_t)(A.rev != B.rev));
        }
    }

    // junction-bridging overlaps for chimeric reads; stacks for hill reads
    for (uint64_t i = 0; i < n_reads && n_reads > 1; ++i) {
        const int64_t L = S.read_len[i];
        if (S.read_kind[i] & 1) {
            const int n = (int)rng.range(5, 10);
            for (int k = 0; k < n; ++k) {
                const int64_t b = std::max<int64_t>(0, (int64_t)chim_cut[i] - rng.range(800, 3000));
                const int64_t e = std::min<int64_t>(L, (int64_t)chim_cut[i] + rng.range(800, 3000));
                uint32_t p = (uint32_t)rng.below(n_reads - 1);
                if (p >= i) ++p;
                const int64_t Lp = S.read_len[p];
                const int64_t span = std::min<int64_t>(e - b, Lp);
                const int64_t pb = rng.range(0, Lp - span);
                emit((uint32_t)i, b, b + span, p, pb, pb + span, (uint8_t)(rng.next() & 1));
            }
        }
        if (S.read_kind[i] & 4) {
            const int64_t centre = rng.range(2500, L - 2500);
            const int n = (int)rng.range(30, 80);
            for (int k = 0; k < n; ++k) {
                const int64_t b = centre - rng.range(150, 400);
                const int64_t e = centre + rng.range(150, 400);
                uint32_t p = (uint32_t)rng.below(n_reads - 1);
                if (p >= i) ++p;
                const int64_t Lp = S.read_len[p];
                const int64_t span = std::min<int64_t>(e - b, Lp);
                const int64_t pb = rng.range(0, Lp - span);
                emit((uint32_t)i, b, b + span, p, pb, pb + span, (uint8_t)(rng.next() & 1));
            }
        }
    }

    // group by query, order by target (then generation order)
    std::sort(recs.begin(), recs.end(), [](const Rec& a, const Rec& b) {
        if (a.q != b.q) return a.q < b.q;
        if (a.t != b.t) return a.t < b.t;
        return a.ord < b.ord;
    });

    const size_t n = recs.size();
    S.a_id.resize(n); S.b_id.resize(n); S.a_begin.resize(n); S.a_end.resize(n);
    S.b_begin.resize(n); S.b_end.resize(n); S.length.resize(n); S.strand.resize(n);
    for (size_t k = 0; k < n; ++k) {
        S.a_id[k] = recs[k].q; S.b_id[k] = recs[k].t;
        S.a_begin[k] = recs[k].qb; S.a_end[k] = recs[k].qe;
        S.b_begin[k] = recs[k].tb; S.b_end[k] = recs[k].te;
        S.length[k] = recs[k].len; S.strand[k] = recs[k].strand;
    }
}

}  // namespace

extern "C" {

// plants: bit0 chimeras, bit1 adapters, bit2 repeats, bit3 stacks (15 = all); bit4: heavy-tailed read lengths; bit5: one read
// in eighty 100 - 400 kb long
void* synth_create(uint64_t n_reads, uint64_t genome_len, uint64_t seed, uint32_t plants) {
    Synth* s = new Synth;
    generate(*s, n_reads, genome_len, seed, plants);
    return s;
}
void synth_destroy(void* p) { delete (Synth*)p; }
uint64_t synth_n_reads(void* p) { return ((Synth*)p)->n_reads; }
uint64_t synth_n_overlaps(void* p) { return ((Synth*)p)->a_id.size(); }
const uint32_t* synth_read_len(void* p) { return ((Synth*)p)->read_len.data(); }
const uint8_t* synth_read_kind(void* p) { return ((Synth*)p)->read_kind.data(); }
// field: 0 a_id 1 b_id 2 a_begin 3 a_end 4 b_begin 5 b_end 6 length ; set: 0 primary, 1 sensitive
const uint32_t* synth_field(void* p, int set, int field) {
    Synth* s = (Synth*)p;
    if (set == 0) {
        switch (field) {
            case 0: return s->a_id.data(); case 1: return s->b_id.data();
            case 2: return s->a_begin.data(); case 3: return s->a_end.data();
            case 4: return s->b_begin.data(); case 5: return s->b_end.data();
            default: return s->length.data();
        }
    }
    switch (field) {
        case 0: return s->s_a_id.data(); case 1: return s->s_b_id.data();
        case 2: return s->s_a_begin.data(); case 3: return s->s_a_end.data();
        case 4: return s->s_b_begin.data(); case 5: return s->s_b_end.data();
        default: return s->s_length.data();
    }
}
const uint8_t* synth_strand(void* p, int set) { return set == 0 ? ((Synth*)p)->strand.data() : ((Synth*)p)->s_strand.data(); }

// Sensitive overlap set (misc/raven.sh second minimap run): targets are the
// surviving TRIMMED reads, queries the original reads.  For every primary
// overlap and each of its reads t that survived with valid region [B, E):
// query = the other read (untrimmed coordinates), target = t with coordinates
// clipped to [B, E) and rebased to B; dropped when the clipped span < 200.
uint64_t synth_make_sensitive(void* p, const uint8_t* alive, const uint32_t* begin, const uint32_t* end) {
    Synth* s = (Synth*)p;
    s->s_a_id.clear(); s->s_b_id.clear(); s->s_a_begin.clear(); s->s_a_end.clear();
    s->s_b_begin.clear(); s->s_b_end.clear(); s->s_length.clear(); s->s_strand.clear();
    auto put = [&](uint32_t q, uint32_t qb, uint32_t qe, uint32_t t, uint32_t tb, uint32_t te, uint8_t st) {
        if (!alive[t]) return;
        const uint32_t cb = std::max(tb, begin[t]), ce = std::min(te, end[t]);
        if (ce <= cb || ce - cb < 200) return;
        s->s_a_id.push_back(q); s->s_b_id.push_back(t);
        s->s_a_begin.push_back(qb); s->s_a_end.push_back(qe);
        s->s_b_begin.push_back(cb - begin[t]); s->s_b_end.push_back(ce - begin[t]);
        s->s_length.push_back(std::max(qe - qb, ce - cb));
        s->s_strand.push_back(st);
    };
    for (size_t k = 0; k < s->a_id.size(); ++k) {
        put(s->a_id[k], s->a_begin[k], s->a_end[k], s->b_id[k], s->b_begin[k], s->b_end[k], s->strand[k]);
        put(s->b_id[k], s->b_begin[k], s->b_end[k], s->a_id[k], s->a_begin[k], s->a_end[k], s->strand[k]);
    }
    return s->s_a_id.size();
}
uint64_t synth_n_sensitive(void* p) { return ((Synth*)p)->s_a_id.size(); }

// Text forms for the CLI path.  Read names are "r<id>"; bases are a fixed
// filler derived from the id (content is irrelevant to the hot path).
int synth_write_fasta(void* p, const char* path) {
    Synth* s = (Synth*)p;
    FILE* f = fopen(path, "w");
    if (!f) return -1;
    static const char acgt[4] = {'A', 'C', 'G', 'T'};
    std::string line;
    for (uint64_t i = 0; i < s->n_reads; ++i) {
        fprintf(f, ">r%lu\n", (unsigned long)i);
        line.resize(s->read_len[i]);
        Rng r(i * 7919 + 13);
        for (uint32_t k = 0; k < s->read_len[i]; k += 32) {
            uint64_t w = r.next();
            for (uint32_t j = 0; j < 32 && k + j < s->read_len[i]; ++j, w >>= 2) line[k + j] = acgt[w & 3];
        }
        fwrite(line.data(), 1, line.size(), f);
        fputc('\n', f);
    }
    fclose(f);
    return 0;
}

// set 0: primary PAF (lengths = full reads).  set 1: sensitive PAF; the
// target length column is the trimmed length and must be supplied.
int synth_write_paf(void* p, int set, const char* path, const uint32_t* target_len_override) {
    Synth* s = (Synth*)p;
    FILE* f = fopen(path, "w");
    if (!f) return -1;
    const size_t n = set == 0 ? s->a_id.size() : s->s_a_id.size();
    for (size_t k = 0; k < n; ++k) {
        const uint32_t q = synth_field(p, set, 0)[k], t = synth_field(p, set, 1)[k];
        const uint32_t len = synth_field(p, set, 6)[k];
        const uint32_t tl = target_len_override ? target_len_override[t] : s->read_len[t];
        fprintf(f, "r%u\t%u\t%u\t%u\t%c\tr%u\t%u\t%u\t%u\t%u\t%u\t255\n", q, s->read_len[q],
                synth_field(p, set, 2)[k], synth_field(p, set, 3)[k], synth_strand(p, set)[k] ? '-' : '+', t, tl,
                synth_field(p, set, 4)[k], synth_field(p, set, 5)[k], (uint32_t)(len * 85ull / 100), len);
    }
    fclose(f);
    return 0;
}

}  // extern "C"
