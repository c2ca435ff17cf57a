// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_core.hpp).
//
// Restatement of the orchestration in the reference's src/graph.cpp for the
// hot path: Graph::initialize (graph.cpp:244-425), Graph::construct pass 2
// (:427-525), both Graph::preprocess overloads (:699-880, :882-1054), the
// node/edge build (:553-632) and Graph::remove_transitive_edges (:1281-1335).
//
// Parity status of THIS file: PARITY UNPINNED.  graph.cpp needs the three
// un-vendored submodules (bioparser, thread_pool, logger; .gitmodules:1-9,
// vendor/* are empty) and therefore cannot be built here; the reference has
// no tests or golden vectors for it.  The driver is a template over a
// backend so that the same glue runs (a) on the flat restatement in
// oracle_core.hpp and (b) on the REAL rala::Pile / rala::Overlap objects
// compiled from /root/reference (oracle/ref_backend.cpp → oracle/_ref/).
// (b) pins every Pile/Overlap call made by the glue; the order of calls is
// this file's reading of graph.cpp.
#pragma once

#include <stdint.h>
#include <algorithm>
#include <atomic>
#include <deque>
#include <future>
#include <memory>
#include <thread>
#include <vector>

// The pool behind the interface of the reference's un-vendored vendor/thread_pool (createThreadPool, submit_task ->
// std::future): the product's header of that name.  With it the fan-out below has the reference's structure - ONE task
// per pile, submitted from the main thread, then waited for (graph.cpp:367-377, 387-407) - which is what bench.py's
// cpu_baseline times (Driver::task_pool); the tests keep the lighter chunked loop.
#include "../rala_amd/host/thread_pool/thread_pool.hpp"

namespace ora {

template <class F>
inline void parallel_for(uint64_t n, uint32_t n_threads, F fn) {
    if (n_threads <= 1 || n < 2) {
        for (uint64_t i = 0; i < n; ++i) fn(i);
        return;
    }
    std::atomic<uint64_t> next(0);
    const uint64_t grain = 16;
    std::vector<std::thread> pool;
    for (uint32_t t = 0; t < n_threads; ++t) {
        pool.emplace_back([&]() {
            for (;;) {
                const uint64_t b = next.fetch_add(grain);
                if (b >= n) break;
                const uint64_t e = std::min(n, b + grain);
                for (uint64_t i = b; i < e; ++i) fn(i);
            }
        });
    }
    for (auto& th : pool) th.join();
}

// graph.cpp:367-377: a task per item through the pool, futures collected and waited for in order
template <class F>
inline void pool_for(uint64_t n, thread_pool::ThreadPool& pool, F fn) {
    std::vector<std::future<void>> futures;
    futures.reserve(n);
    for (uint64_t i = 0; i < n; ++i) {
        futures.emplace_back(pool.submit_task([&fn](uint64_t k) -> void { fn(k); }, i));
    }
    for (const auto& it : futures) it.wait();
}

// Input overlaps, structure of arrays.  id == 0xFFFFFFFF stands for a name
// that is not in the sequence file (Overlap::transmute → false,
// overlap.cpp:44-47,63-66).
struct OvlInput {
    uint64_t n;
    const uint32_t *a_id, *b_id, *a_begin, *a_end, *b_begin, *b_end, *length;
    const uint8_t* strand;
    OvlInput() : n(0), a_id(0), b_id(0), a_begin(0), a_end(0), b_begin(0), b_end(0), length(0), strand(0) {}
};

struct EdgeRec { uint32_t src, dst, len; };

template <class BK>
struct Driver {
    typedef typename BK::OvlH OvlH;
    struct Item { OvlH h; uint64_t src; };

    BK bk;
    uint32_t n_threads;
    std::unique_ptr<thread_pool::ThreadPool> task_pool;     // set: one task per pile like the reference (bench.py's baseline)
    template <class F>
    void parallel_for(uint64_t n, uint32_t, F fn) {
        if (task_pool) pool_for(n, *task_pool, fn);
        else ora::parallel_for(n, n_threads, fn);
    }
    uint64_t n_reads;
    std::vector<uint32_t> read_len;
    OvlInput in;

    // results
    std::vector<uint8_t> valid;                 // is_valid_overlap_ (graph.hpp:168)
    std::vector<Item> overlaps, internals;      // graph.cpp:440
    uint64_t n_prefiltered;
    std::vector<int64_t> read_to_node;          // graph.cpp:553
    std::vector<uint32_t> node_read;
    std::vector<EdgeRec> edges;                 // edge id = index; twin = id ^ 1
    std::vector<uint8_t> edge_marked;
    uint32_t n_transitive;

    Driver() : n_threads(1), n_reads(0), n_prefiltered(0), n_transitive(0) {}

    ~Driver() {
        for (auto& it : overlaps) if (it.h) bk.free_ovl(it.h);
        for (auto& it : internals) if (it.h) bk.free_ovl(it.h);
    }

    void set_reads(const uint32_t* len, uint64_t n) {
        n_reads = n;
        read_len.assign(len, len + n);
        bk.create_piles(len, n);                // graph.cpp:255-259
    }

    bool transmutable(uint64_t i) const {
        return in.a_id[i] < n_reads && in.b_id[i] < n_reads;
    }

    // ---- Graph::initialize, overlap pass 1 (graph.cpp:270-382) ----------
    void pass1_dedupe_and_layers() {
        const uint64_t N = in.n;
        valid.assign(N, 1);
        std::vector<std::vector<uint32_t>> bounds(n_reads);
        std::vector<uint64_t> run;

        auto flush = [&]() {
            // remove_duplicate_overlaps (graph.cpp:273-307)
            for (size_t x = 0; x < run.size(); ++x) {
                const uint64_t i = run[x];
                if (in.a_id[i] == in.b_id[i]) { valid[i] = 0; continue; }
                for (size_t y = x + 1; y < run.size(); ++y) {
                    const uint64_t j = run[y];
                    if (in.b_id[i] != in.b_id[j]) continue;
                    if (in.length[i] > in.length[j]) {
                        valid[j] = 0;
                    } else {
                        valid[i] = 0;
                        break;
                    }
                }
            }
            // store_overlap_bounds (graph.cpp:311-326): every transmuted
            // overlap, valid or not
            for (size_t x = 0; x < run.size(); ++x) {
                const uint64_t i = run[x];
                bounds[in.a_id[i]].push_back((in.a_begin[i] + 15) << 1);
                bounds[in.a_id[i]].push_back((in.a_end[i] - 15) << 1 | 1);
                bounds[in.b_id[i]].push_back((in.b_begin[i] + 15) << 1);
                bounds[in.b_id[i]].push_back((in.b_end[i] - 15) << 1 | 1);
            }
            run.clear();
        };

        for (uint64_t i = 0; i < N; ++i) {
            if (!transmutable(i)) { valid[i] = 0; continue; }      // graph.cpp:337-341
            if (!run.empty() && in.a_id[run[0]] != in.a_id[i]) flush();   // :346-350
            run.push_back(i);
        }
        flush();                                                    // :352-356

        parallel_for(n_reads, n_threads, [&](uint64_t r) {          // :367-377
            bk.add_layers(r, bounds[r]);
            std::vector<uint32_t>().swap(bounds[r]);
        });
    }

    // ---- Graph::initialize, per-pile annotation (graph.cpp:387-425) -----
    // returns false on "filtered all sequences" (graph.cpp:418-421)
    bool annotate() {
        parallel_for(n_reads, n_threads, [&](uint64_t r) {
            if (!bk.find_valid_region(r)) {
                bk.kill(r);
            } else {
                bk.find_median(r);
                bk.find_chimeric_hills(r);
                bk.find_chimeric_pits(r);
            }
        });
        n_prefiltered = 0;
        for (uint64_t r = 0; r < n_reads; ++r) if (!bk.alive(r)) ++n_prefiltered;
        return n_prefiltered != n_reads;
    }

    bool initialize() {
        pass1_dedupe_and_layers();
        return annotate();
    }

    // ---- Graph::construct, overlap pass 2 (graph.cpp:443-518) -----------
    void pass2() {
        const uint64_t N = in.n;
        for (uint64_t i = 0; i < N; ++i) {
            if (!valid[i] || !transmutable(i)) continue;
            const uint32_t a = in.a_id[i], b = in.b_id[i];
            if (!bk.alive(a) || !bk.alive(b)) continue;       // transmute(): overlap.cpp:51,70
            OvlH h = bk.make_ovl(a, b, in.a_begin[i], in.a_end[i], read_len[a], in.b_begin[i], in.b_end[i],
                                 read_len[b], in.length[i], in.strand[i]);
            if (h == nullptr) continue;
            if (!bk.trim(h)) { bk.free_ovl(h); continue; }
            if (bk.has_hill(a)) bk.check_chimeric_hills(a, h);
            if (bk.has_hill(b)) bk.check_chimeric_hills(b, h);
            Item it; it.h = h; it.src = i;
            switch (bk.type(h)) {
                case kX:
                    internals.push_back(it);
                    break;
                case kB:
                    if (!bk.has_chimeric_region(b)) { bk.kill(a); bk.free_ovl(h); }
                    else overlaps.push_back(it);
                    break;
                case kA:
                    if (!bk.has_chimeric_region(a)) { bk.kill(b); bk.free_ovl(h); }
                    else overlaps.push_back(it);
                    break;
                default:
                    overlaps.push_back(it);
                    break;
            }
        }
        drop_dead(overlaps);       // graph.cpp:495-505
        drop_dead(internals);      // :507-514
    }

    void drop_dead(std::vector<Item>& v) {
        size_t w = 0;
        for (size_t k = 0; k < v.size(); ++k) {
            if (!bk.alive(bk.a_id(v[k].h)) || !bk.alive(bk.b_id(v[k].h))) { bk.free_ovl(v[k].h); continue; }
            v[w++] = v[k];
        }
        v.resize(w);
    }

    // trims every item; drops the failures; returns number dropped
    uint64_t retrim(std::vector<Item>& v) {
        size_t w = 0;
        for (size_t k = 0; k < v.size(); ++k) {
            if (!bk.trim(v[k].h)) { bk.free_ovl(v[k].h); continue; }
            v[w++] = v[k];
        }
        const uint64_t dropped = v.size() - w;
        v.resize(w);
        return dropped;
    }

    // Connected components over the current overlaps (graph.cpp:740-773).
    // Only the member SETS matter downstream (a median per component).
    void components(std::vector<std::vector<uint32_t>>& out) {
        std::vector<std::vector<uint32_t>> adj(n_reads);
        for (auto& it : overlaps) {
            adj[bk.a_id(it.h)].push_back(bk.b_id(it.h));
            adj[bk.b_id(it.h)].push_back(bk.a_id(it.h));
        }
        std::vector<char> seen(n_reads, 0);
        out.clear();
        for (uint64_t i = 0; i < n_reads; ++i) {
            if (adj[i].empty() || seen[i]) continue;
            out.resize(out.size() + 1);
            std::deque<uint32_t> que;
            que.push_back((uint32_t)i);
            while (!que.empty()) {
                const uint32_t j = que.front();
                que.pop_front();
                if (seen[j]) continue;
                seen[j] = 1;
                out.back().push_back(j);
                for (uint32_t x : adj[j]) que.push_back(x);
                std::vector<uint32_t>().swap(adj[j]);
            }
        }
    }

    uint16_t component_median(const std::vector<uint32_t>& comp) {
        std::vector<uint16_t> m;                      // graph.cpp:777-783
        for (uint32_t r : comp) m.push_back(bk.median(r));
        std::nth_element(m.begin(), m.begin() + m.size() / 2, m.end());
        return m[m.size() / 2];
    }

    // ---- Graph::preprocess(overlaps, internals) (graph.cpp:699-880) -----
    void preprocess_chimeras() {
        parallel_for(n_reads, n_threads, [&](uint64_t r) {          // :704-720
            if (bk.alive(r) && bk.has_hill(r) && !bk.break_over_chimeric_hills(r)) bk.kill(r);
        });
        retrim(overlaps);                                           // :722-728
        retrim(internals);                                          // :730-736

        for (;;) {                                                  // :738-829
            std::vector<std::vector<uint32_t>> comps;
            components(comps);
            for (auto& comp : comps) {
                const uint16_t med = component_median(comp);
                parallel_for(comp.size(), n_threads, [&](uint64_t k) {
                    if (!bk.break_over_chimeric_pits(comp[k], med)) bk.kill(comp[k]);
                });
            }
            const bool changed = retrim(overlaps) != 0;             // :801-807
            size_t w = 0;                                           // :809-824
            for (size_t k = 0; k < internals.size(); ++k) {
                if (!bk.trim(internals[k].h)) { bk.free_ovl(internals[k].h); continue; }
                const int t = bk.type(internals[k].h);
                if (t == kAB || t == kBA) { overlaps.push_back(internals[k]); continue; }
                internals[w++] = internals[k];
            }
            internals.resize(w);
            if (!changed) break;
        }

        // sequential containment removal, no chimera guard (:831-877)
        auto kill_scan = [&](std::vector<Item>& v) {
            for (auto& it : v) {
                const uint32_t a = bk.a_id(it.h), b = bk.b_id(it.h);
                if (!bk.alive(a) || !bk.alive(b)) { bk.free_ovl(it.h); it.h = nullptr; continue; }
                const int t = bk.type(it.h);
                if (t == kA) { bk.kill(b); bk.free_ovl(it.h); it.h = nullptr; }
                else if (t == kB) { bk.kill(a); bk.free_ovl(it.h); it.h = nullptr; }
            }
        };
        kill_scan(overlaps);
        kill_scan(internals);
        compact(internals);
        for (auto& it : overlaps) {
            if (it.h == nullptr) continue;
            if (!bk.alive(bk.a_id(it.h)) || !bk.alive(bk.b_id(it.h))) { bk.free_ovl(it.h); it.h = nullptr; }
        }
        compact(overlaps);
    }

    void compact(std::vector<Item>& v) {
        size_t w = 0;
        for (size_t k = 0; k < v.size(); ++k) if (v[k].h != nullptr) v[w++] = v[k];
        v.resize(w);
    }

    // ---- Graph::preprocess(overlaps, path) (graph.cpp:882-1054) ---------
    // `sens` are the records of the sensitive overlap file; all ids must
    // resolve and every target (b) pile must be alive, as in the reference
    // (Overlap::transmute_ dereferences piles[b_id_], overlap.cpp:108-110).
    void preprocess_repeats(const OvlInput& sens) {
        std::vector<Item> sov;
        std::vector<std::vector<uint32_t>> bounds(n_reads);
        std::vector<char> is_target(n_reads, 0);
        std::vector<uint32_t> targets;
        for (uint64_t i = 0; i < sens.n; ++i) {                     // :917-939
            const uint32_t a = sens.a_id[i], b = sens.b_id[i];
            // query = original read (untrimmed coordinates), target = trimmed
            // read from the -p run (misc/raven.sh:10-19)
            OvlH h = bk.make_sensitive(a, b, sens.a_begin[i], sens.a_end[i], read_len[a], sens.b_begin[i],
                                       sens.b_end[i], sens.length[i], sens.strand[i]);
            if (!is_target[b]) { is_target[b] = 1; targets.push_back(b); }
            bounds[b].push_back(bk.b_begin(h) << 1);
            bounds[b].push_back(bk.b_end(h) << 1 | 1);
            if (!bk.trim(h)) { bk.free_ovl(h); continue; }
            Item it; it.h = h; it.src = i;
            sov.push_back(it);
        }
        parallel_for(targets.size(), n_threads, [&](uint64_t k) {   // :941-953
            bk.add_layers(targets[k], bounds[targets[k]]);
        });
        parallel_for(targets.size(), n_threads, [&](uint64_t k) {   // :960-969
            bk.find_median(targets[k]);
        });
        std::vector<std::vector<uint32_t>> comps;                   // :971-1026
        components(comps);
        for (auto& comp : comps) {
            const uint16_t med = component_median(comp);
            parallel_for(comp.size(), n_threads, [&](uint64_t k) {
                bk.find_repetitive_hills(comp[k], med);
            });
        }
        for (auto& it : sov) {                                      // :1028-1043
            if (!bk.trim(it.h)) continue;
            const int t = bk.type(it.h);
            if (t == kAB || t == kBA) {
                const uint32_t b = bk.b_id(it.h);
                if (bk.has_rep_hills(b)) bk.check_repetitive_hills(b, it.h);
            }
        }
        for (auto& it : sov) bk.free_ovl(it.h);
        size_t w = 0;                                               // :1045-1051
        for (size_t k = 0; k < overlaps.size(); ++k) {
            OvlH h = overlaps[k].h;
            if (!bk.is_valid_overlap(bk.a_id(h), bk.a_begin(h), bk.a_end(h)) ||
                !bk.is_valid_overlap(bk.b_id(h), bk.b_begin(h), bk.b_end(h))) {
                bk.free_ovl(h);
                continue;
            }
            overlaps[w++] = overlaps[k];
        }
        overlaps.resize(w);
    }

    // ---- node / edge construction (graph.cpp:553-632) -------------------
    void build_graph() {
        read_to_node.assign(n_reads, -1);
        node_read.clear();
        for (uint64_t r = 0; r < n_reads; ++r) {
            if (!bk.alive(r)) continue;
            read_to_node[r] = (int64_t)node_read.size();
            node_read.push_back((uint32_t)r);      // node 2k   (forward)
            node_read.push_back((uint32_t)r);      // node 2k+1 (reverse complement)
        }
        edges.clear();
        for (auto& it : overlaps) {
            OvlH h = it.h;
            const uint32_t a = bk.a_id(h), b = bk.b_id(h);
            const uint32_t na = (uint32_t)read_to_node[a];
            const uint32_t nb = (uint32_t)read_to_node[b] + bk.strand(h);
            const uint32_t Ba = bk.begin(a), Ea = bk.end(a), Bb = bk.begin(b), Eb = bk.end(b);
            const uint32_t la = Ea - Ba, a0 = bk.a_begin(h) - Ba, a1 = bk.a_end(h) - Ba;
            const uint32_t lb = Eb - Bb;
            const uint32_t b0 = bk.strand(h) == 0 ? bk.b_begin(h) - Bb : lb - bk.b_end(h) + Bb;
            const uint32_t b1 = bk.strand(h) == 0 ? bk.b_end(h) - Bb : lb - bk.b_begin(h) + Bb;
            const int t = bk.type(h);
            EdgeRec e, ec;
            if (t == kAB) {
                e.src = na; e.dst = nb; e.len = a0 - b0;
                ec.src = nb ^ 1; ec.dst = na ^ 1; ec.len = (lb - b1) - (la - a1);
            } else if (t == kBA) {
                e.src = nb; e.dst = na; e.len = b0 - a0;
                ec.src = na ^ 1; ec.dst = nb ^ 1; ec.len = (la - a1) - (lb - b1);
            } else {
                continue;
            }
            edges.push_back(e);
            edges.push_back(ec);
        }
        edge_marked.assign(edges.size(), 0);
    }

    // ---- Graph::remove_transitive_edges (graph.cpp:1281-1335) -----------
    uint32_t remove_transitive_edges() {
        const size_t n_nodes = node_read.size();
        // suffix_edges_ per node, in edge-id order (graph.cpp:604-607,622-625)
        std::vector<std::vector<uint32_t>> out(n_nodes);
        for (uint32_t e = 0; e < edges.size(); ++e) out[edges[e].src].push_back(e);
        std::vector<int64_t> cand(n_nodes, -1);
        n_transitive = 0;
        for (size_t a = 0; a < n_nodes; ++a) {
            for (uint32_t e : out[a]) cand[edges[e].dst] = e;            // last writer wins
            for (uint32_t eab : out[a]) {
                const uint32_t b = edges[eab].dst;
                for (uint32_t ebc : out[b]) {
                    const uint32_t c = edges[ebc].dst;
                    if (cand[c] < 0 || edge_marked[cand[c]]) continue;
                    const uint32_t sum = edges[eab].len + edges[ebc].len;
                    if (comparable((double)sum, (double)edges[cand[c]].len, 0.12)) {
                        edge_marked[cand[c]] = 1;
                        edge_marked[cand[c] ^ 1] = 1;
                        ++n_transitive;
                    }
                }
            }
            for (uint32_t e : out[a]) cand[edges[e].dst] = -1;
        }
        return n_transitive;
    }

    bool construct(const OvlInput* sens) {
        if (!initialize()) return false;
        pass2();
        preprocess_chimeras();
        if (sens != nullptr && sens->n != 0) preprocess_repeats(*sens);
        build_graph();
        return true;
    }
};

}  // namespace ora
