// ORACLE — TEST INFRASTRUCTURE ONLY.  C surface of oracle_layout.hpp, same shape as
// rala_amd/host/assembly_graph_capi.cpp so that tests/test_layout_cpu.py drives both alike.
#include <stdint.h>
#include <string.h>

#include "oracle_layout.hpp"

namespace {

uint64_t fnv1a(const void* p, size_t n, uint64_t h = 1469598103934665603ull) {
    const unsigned char* c = (const unsigned char*)p;
    for (size_t i = 0; i < n; ++i) { h ^= c[i]; h *= 1099511628211ull; }
    return h;
}

uint64_t hash_edges(const std::vector<ora_layout::Edge*>& v, uint64_t h) {
    for (const auto* e : v) {
        const uint32_t id = (uint32_t)e->id;
        h = fnv1a(&id, 4, h);
    }
    return h;
}

}  // namespace

extern "C" {

void* ol_create() { return new ora_layout::Layout(); }
void ol_destroy(void* h) { delete (ora_layout::Layout*)h; }
void ol_add_node_pair(void* h, uint64_t sequence_id, const char* name, const char* data, const char* rc) {
    ((ora_layout::Layout*)h)->add_node_pair(sequence_id, name, data, rc);
}
void ol_add_edge(void* h, uint32_t b, uint32_t e, uint32_t length) { ((ora_layout::Layout*)h)->add_edge(b, e, length); }
void ol_mark_edge(void* h, uint32_t edge) {
    auto* g = (ora_layout::Layout*)h;
    g->mark(g->edges_[edge].get());
}
void ol_note_transitive(void* h) { ((ora_layout::Layout*)h)->note_transitive_edges(); }
void ol_postprocess(void* h, uint32_t seed) { ((ora_layout::Layout*)h)->postprocess(seed); }
void ol_edge_weights(void* h, double* w) {
    auto* g = (ora_layout::Layout*)h;
    for (size_t i = 0; i < g->edges_.size(); ++i) w[i] = g->edges_[i] ? g->edges_[i]->weight : 0.0;
}
uint64_t ol_transitive(void* h, uint64_t* pairs) {
    auto* g = (ora_layout::Layout*)h;
    if (pairs) for (size_t i = 0; i < g->transitive_edges_.size(); ++i) { pairs[2 * i] = g->transitive_edges_[i].first; pairs[2 * i + 1] = g->transitive_edges_[i].second; }
    return g->transitive_edges_.size();
}
void ol_remove_marked(void* h, int remove_nodes) { ((ora_layout::Layout*)h)->remove_marked_objects(remove_nodes != 0); }

uint32_t ol_run(void* h, int op, uint32_t arg) {
    auto* g = (ora_layout::Layout*)h;
    switch (op) {
        case 0: return g->remove_tips();
        case 1: return g->remove_bubbles();
        case 2: return g->create_unitigs();
        case 3: return g->shrink(arg);
        case 4: return g->remove_long_edges();
    }
    return 0xFFFFFFFFu;
}

void ol_size(void* h, uint64_t* n_nodes, uint64_t* n_edges) {
    auto* g = (ora_layout::Layout*)h;
    *n_nodes = g->nodes_.size();
    *n_edges = g->edges_.size();
}

void ol_dump_nodes(void* h, uint8_t* alive, uint32_t* length, uint32_t* n_seq, uint64_t* data_hash, uint64_t* ids_hash,
                   uint8_t* first_rc, uint8_t* last_rc, uint32_t* indeg, uint32_t* outdeg, uint64_t* adj_hash) {
    auto* g = (ora_layout::Layout*)h;
    for (size_t i = 0; i < g->nodes_.size(); ++i) {
        const auto* n = g->nodes_[i].get();
        alive[i] = n != nullptr;
        if (!n) {
            length[i] = n_seq[i] = indeg[i] = outdeg[i] = 0;
            first_rc[i] = last_rc[i] = 0;
            data_hash[i] = ids_hash[i] = adj_hash[i] = fnv1a(nullptr, 0);
            continue;
        }
        length[i] = n->length();
        n_seq[i] = (uint32_t)n->sequence_ids.size();
        data_hash[i] = fnv1a(n->data.data(), n->data.size());
        ids_hash[i] = fnv1a(n->sequence_ids.data(), n->sequence_ids.size() * 8);
        first_rc[i] = n->is_first_rc; last_rc[i] = n->is_last_rc;
        indeg[i] = n->indegree(); outdeg[i] = n->outdegree();
        adj_hash[i] = hash_edges(n->suffix_edges, hash_edges(n->prefix_edges, fnv1a(nullptr, 0)));
    }
}

void ol_dump_edges(void* h, uint8_t* alive, uint32_t* begin_node, uint32_t* end_node, uint32_t* length) {
    auto* g = (ora_layout::Layout*)h;
    for (size_t i = 0; i < g->edges_.size(); ++i) {
        const auto* e = g->edges_[i].get();
        alive[i] = e != nullptr;
        begin_node[i] = e ? (uint32_t)e->begin_node->id : 0;
        end_node[i] = e ? (uint32_t)e->end_node->id : 0;
        length[i] = e ? e->length : 0;
    }
}

// kind 0 csv, 1 gfa, 2 json (piles as "<id>":{} stand-ins); returns the length, copies at most cap bytes
uint64_t ol_print(void* h, int kind, char* dst, uint64_t cap) {
    const ora_layout::Layout* g = (const ora_layout::Layout*)h;
    const std::string s = kind == 0 ? g->print_csv() : kind == 1 ? g->print_gfa()
                          : g->print_json([](uint64_t id) { return "\"" + std::to_string(id) + "\":{}"; });
    if (dst) memcpy(dst, s.data(), s.size() < cap ? s.size() : cap);
    return s.size();
}

uint64_t ol_node_data(void* h, uint64_t node, char* dst, uint64_t cap) {
    const auto* n = ((ora_layout::Layout*)h)->nodes_[node].get();
    if (!n) return 0;
    if (dst && cap >= n->data.size()) memcpy(dst, n->data.data(), n->data.size());
    return n->data.size();
}

}  // extern "C"
