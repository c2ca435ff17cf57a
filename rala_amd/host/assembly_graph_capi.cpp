// C surface of rala::AssemblyGraph for the CPU test-suite (tests/test_layout_cpu.py): build a
// graph from arrays, run the clean-up stages, read the result back.  Not part of the product
// boundary (that is include/rala_hip.h + the rala:: classes).
#include <stdio.h>
#include <stdlib.h>
#include <sstream>
#include <stdint.h>
#include <string.h>

#include <string>

#include "assembly_graph.hpp"

namespace {

uint64_t fnv1a(const void* p, size_t n, uint64_t h = 1469598103934665603ull) {
    const unsigned char* c = (const unsigned char*)p;
    for (size_t i = 0; i < n; ++i) { h ^= c[i]; h *= 1099511628211ull; }
    return h;
}

}  // namespace

extern "C" {

void* ag_create() { return new rala::AssemblyGraph(); }
void ag_destroy(void* h) { delete (rala::AssemblyGraph*)h; }

void ag_add_node_pair(void* h, uint64_t sequence_id, const char* name, const char* data, const char* rc) {
    ((rala::AssemblyGraph*)h)->add_sequence_nodes(sequence_id, name, data, rc);
}
void ag_add_edge(void* h, uint32_t begin_node, uint32_t end_node, uint32_t length) {
    ((rala::AssemblyGraph*)h)->add_edge(begin_node, end_node, length);
}
void ag_mark_edge(void* h, uint32_t edge) { ((rala::AssemblyGraph*)h)->mark_edge(edge); }
void ag_note_transitive(void* h) { ((rala::AssemblyGraph*)h)->note_transitive_edges(); }
// engine: the layout steps (rala_hip_layout through a ctypes callback in the GPU tests)
typedef int (*ag_layout_fn)(uint32_t n, double* x, double* y, const uint32_t* adj_off, const uint32_t* adj,
                            uint32_t iterations, double k, double t, double dt);
int ag_postprocess(void* h, uint32_t seed, ag_layout_fn engine) {
    return ((rala::AssemblyGraph*)h)->postprocess(
        [engine](uint32_t n, double* x, double* y, const uint32_t* adj_off, const uint32_t* adj, uint32_t iterations,
                 double k, double t, double dt) { return engine(n, x, y, adj_off, adj, iterations, k, t, dt); },
        seed);
}
void ag_edge_weights(void* h, double* w) {
    auto* g = (rala::AssemblyGraph*)h;
    for (size_t i = 0; i < g->edges().size(); ++i) w[i] = g->edges()[i].alive ? g->edges()[i].weight : 0.0;
}
uint64_t ag_transitive(void* h, uint64_t* pairs) {
    const auto& te = ((rala::AssemblyGraph*)h)->transitive_edges();
    if (pairs) for (size_t i = 0; i < te.size(); ++i) { pairs[2 * i] = te[i].first; pairs[2 * i + 1] = te[i].second; }
    return te.size();
}
void ag_remove_marked(void* h, int remove_nodes) { ((rala::AssemblyGraph*)h)->remove_marked_objects(remove_nodes != 0); }

// op: 0 remove_tips, 1 remove_bubbles, 2 create_unitigs, 3 shrink(arg), 4 remove_long_edges
uint32_t ag_run(void* h, int op, uint32_t arg) {
    auto* g = (rala::AssemblyGraph*)h;
    switch (op) {
        case 0: return g->remove_tips();
        case 1: return g->remove_bubbles();
        case 2: return g->create_unitigs();
        case 3: return g->shrink(arg);
        case 4: return g->remove_long_edges();
    }
    return 0xFFFFFFFFu;
}

void ag_size(void* h, uint64_t* n_nodes, uint64_t* n_edges) {
    auto* g = (rala::AssemblyGraph*)h;
    *n_nodes = g->nodes().size();
    *n_edges = g->edges().size();
}

// per node: alive, length, number of reads, hash of the sequence, hash of the read ids,
// first / last orientation, in / out degree, hash of the adjacency lists (edge ids in order)
void ag_dump_nodes(void* h, uint8_t* alive, uint32_t* length, uint32_t* n_seq, uint64_t* data_hash, uint64_t* ids_hash,
                   uint8_t* first_rc, uint8_t* last_rc, uint32_t* indeg, uint32_t* outdeg, uint64_t* adj_hash) {
    auto* g = (rala::AssemblyGraph*)h;
    for (size_t i = 0; i < g->nodes().size(); ++i) {
        const auto& n = g->nodes()[i];
        alive[i] = n.alive;
        length[i] = n.length();
        n_seq[i] = (uint32_t)n.sequence_ids.size();
        data_hash[i] = fnv1a(n.data.data(), n.data.size());
        ids_hash[i] = fnv1a(n.sequence_ids.data(), n.sequence_ids.size() * 8);
        first_rc[i] = n.is_first_rc; last_rc[i] = n.is_last_rc;
        indeg[i] = n.indegree(); outdeg[i] = n.outdegree();
        uint64_t a = fnv1a(n.prefix_edges.data(), n.prefix_edges.size() * 4);
        adj_hash[i] = fnv1a(n.suffix_edges.data(), n.suffix_edges.size() * 4, a);
        if (!n.alive) first_rc[i] = last_rc[i] = 0;
    }
}

void ag_dump_edges(void* h, uint8_t* alive, uint32_t* begin_node, uint32_t* end_node, uint32_t* length) {
    auto* g = (rala::AssemblyGraph*)h;
    for (size_t i = 0; i < g->edges().size(); ++i) {
        const auto& e = g->edges()[i];
        alive[i] = e.alive;
        begin_node[i] = e.alive ? e.begin_node : 0; end_node[i] = e.alive ? e.end_node : 0;
        length[i] = e.alive ? e.length : 0;
    }
}

// kind 0 csv, 1 gfa, 2 json (piles as "<id>":{} stand-ins); returns the length, copies at most cap bytes
uint64_t ag_print(void* h, int kind, char* dst, uint64_t cap) {
    const rala::AssemblyGraph* g = (const rala::AssemblyGraph*)h;
    std::string s;
    if (kind == 2) {
        std::ostringstream os;
        g->write_json(os, [](uint64_t id) { return "\"" + std::to_string(id) + "\":{}"; });
        s = os.str();
    } else {
        char* buf = nullptr;
        size_t len = 0;
        FILE* f = open_memstream(&buf, &len);
        if (kind == 0) g->write_csv(f); else g->write_gfa(f);
        fclose(f);
        s.assign(buf, len);
        free(buf);
    }
    if (dst) memcpy(dst, s.data(), s.size() < cap ? s.size() : cap);
    return s.size();
}

uint64_t ag_node_data(void* h, uint64_t node, char* dst, uint64_t cap) {
    const auto& d = ((rala::AssemblyGraph*)h)->nodes()[node].data;
    if (dst && cap >= d.size()) memcpy(dst, d.data(), d.size());
    return d.size();
}

}  // extern "C"
