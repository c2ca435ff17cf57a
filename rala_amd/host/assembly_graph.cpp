#include "assembly_graph.hpp"

#include <ostream>

#include <unordered_map>

#include <set>

#include <stdio.h>
#include <stdlib.h>

#include <math.h>

#include <algorithm>
#include <deque>
#include <random>
#include <unordered_set>

namespace rala {

void AssemblyGraph::add_sequence_nodes(uint64_t sequence_id, const std::string& name, const std::string& data,
    const std::string& reverse_complement) {
    for (int rc = 0; rc < 2; ++rc) {
        Node n;
        n.id = nodes_.size();
        n.name = name;
        n.data = rc ? reverse_complement : data;
        n.sequence_ids.assign(1, sequence_id);
        n.is_first_rc = n.is_last_rc = rc != 0;        // graph.cpp:126-131
        n.alive = true;
        nodes_.push_back(std::move(n));
    }
}

void AssemblyGraph::add_edge(uint32_t begin_node, uint32_t end_node, uint32_t length) {
    Edge e;
    e.id = edges_.size();
    e.begin_node = begin_node; e.end_node = end_node; e.length = length;
    e.alive = true;
    nodes_[begin_node].suffix_edges.push_back((uint32_t)e.id);
    nodes_[end_node].prefix_edges.push_back((uint32_t)e.id);
    edges_.push_back(e);
}

void AssemblyGraph::mark_edge(uint32_t edge_id) {
    for (uint32_t e : {edge_id, edge_id ^ 1u}) {
        edges_[e].is_marked = true;
        marked_edges_.push_back(e);
    }
}

// graph.cpp:2118-2151: marked edges leave the adjacency lists (order of the rest kept), nodes
// that end up isolated go too when asked for
void AssemblyGraph::remove_marked_objects(bool remove_nodes) {
    auto drop = [&](std::vector<uint32_t>& v) {
        v.erase(std::remove_if(v.begin(), v.end(), [&](uint32_t e) { return edges_[e].is_marked; }), v.end());
    };
    for (uint32_t e : marked_edges_) {
        if (!edges_[e].alive) continue;             // marked twice
        drop(nodes_[edges_[e].begin_node].suffix_edges);
        drop(nodes_[edges_[e].end_node].prefix_edges);
    }
    for (uint32_t e : marked_edges_) {
        if (!edges_[e].alive) continue;
        if (remove_nodes) {
            for (uint32_t v : {edges_[e].begin_node, edges_[e].end_node}) {
                if (nodes_[v].alive && nodes_[v].outdegree() == 0 && nodes_[v].indegree() == 0) {
                    nodes_[v] = Node();
                }
            }
        }
    }
    for (uint32_t e : marked_edges_) {
        edges_[e].alive = false;
        edges_[e].is_marked = false;
    }
    marked_edges_.clear();
}

// graph.cpp:1322-1332
void AssemblyGraph::note_transitive_edges() {
    for (uint32_t e : marked_edges_) {
        if (!(e & 1)) continue;
        const uint64_t a = (uint64_t)(edges_[e].begin_node >> 1) << 1, b = (uint64_t)(edges_[e].end_node >> 1) << 1;
        transitive_edges_.emplace_back(a, b);
        transitive_edges_.emplace_back(b, a);
    }
    std::sort(transitive_edges_.begin(), transitive_edges_.end());
}

// graph.cpp:1056-1279
int AssemblyGraph::postprocess(const LayoutEngine& engine, uint32_t seed) {
    // duplicates and pairs that lost a node leave the transitive list (:1060-1073)
    if (!transitive_edges_.empty()) {
        std::vector<std::pair<uint64_t, uint64_t>> tmp = {transitive_edges_[0]};
        for (size_t i = 1; i < transitive_edges_.size(); ++i) {
            const auto& te = transitive_edges_[i];
            if (!nodes_[te.first].alive || !nodes_[te.second].alive) continue;
            if (te.first != te.second && te != transitive_edges_[i - 1]) tmp.push_back(te);
        }
        tmp.swap(transitive_edges_);
    }
    // connected components over forward-node ids (:1075-1104), largest first; ties by smallest id
    const size_t n0 = nodes_.size();
    std::vector<std::vector<uint32_t>> components;
    {
        std::vector<bool> is_visited(n0, false);
        std::deque<uint32_t> que;
        for (size_t i = 0; i < n0; ++i) {
            if (!nodes_[i].alive || is_visited[i]) continue;
            components.emplace_back();
            que.assign(1, (uint32_t)i);
            while (!que.empty()) {
                const uint32_t j = que.front();
                que.pop_front();
                if (is_visited[j]) continue;
                is_visited[j] = true;
                is_visited[j ^ 1u] = true;
                components.back().push_back(j & ~1u);
                for (uint32_t e : nodes_[j].prefix_edges) que.push_back(edges_[e].begin_node);
                for (uint32_t e : nodes_[j].suffix_edges) que.push_back(edges_[e].end_node);
            }
            std::sort(components.back().begin(), components.back().end());
        }
    }
    std::sort(components.begin(), components.end(), [](const std::vector<uint32_t>& a, const std::vector<uint32_t>& b) {
        return a.size() != b.size() ? a.size() > b.size() : a[0] < b[0];
    });

    std::mt19937 generator(seed);
    std::uniform_real_distribution<> distribution(0., 1.);
    std::vector<int64_t> member_of(n0, -1);
    for (const auto& component : components) {
        if (component.size() < 6) continue;
        bool has_junctions = false;
        for (uint32_t v : component) if (nodes_[v].is_junction()) { has_junctions = true; break; }
        if (!has_junctions) continue;

        const uint32_t n = (uint32_t)component.size();
        const uint32_t num_iterations = 100;
        const double k = sqrt(1. / static_cast<double>(n));
        const double t = 0.1;
        const double dt = t / static_cast<double>(num_iterations + 1);
        std::vector<double> x(n), y(n);
        for (uint32_t i = 0; i < n; ++i) {
            member_of[component[i]] = i;
            x[i] = distribution(generator);
            y[i] = distribution(generator);
        }
        // attraction partners in the reference's order: prefix edges, suffix edges, transitive
        // edges; a partner outside the component sits at the origin (points_ is zero there)
        std::vector<uint32_t> adj_off(n + 1, 0), adj;
        for (uint32_t i = 0; i < n; ++i) {
            const uint32_t v = component[i];
            auto partner = [&](uint64_t m) { adj.push_back(member_of[m] >= 0 ? (uint32_t)member_of[m] : n); };
            for (uint32_t e : nodes_[v].prefix_edges) partner(edges_[e].begin_node & ~1u);
            for (uint32_t e : nodes_[v].suffix_edges) partner(edges_[e].end_node & ~1u);
            auto lo = std::lower_bound(transitive_edges_.begin(), transitive_edges_.end(),
                std::make_pair((uint64_t)v, (uint64_t)0));
            for (; lo != transitive_edges_.end() && lo->first == v; ++lo) partner(lo->second);
            adj_off[i + 1] = (uint32_t)adj.size();
        }
        // the reference's loop advances its counter twice per pass (:1132, :1225): 50 steps
        const int rc = engine(n, x.data(), y.data(), adj_off.data(), adj.data(), num_iterations / 2, k, t, dt);
        if (rc != 0) return rc;
        for (auto& e : edges_) {
            if (!e.alive || (e.id & 1)) continue;
            const int64_t a = member_of[e.begin_node & ~1u], b = member_of[e.end_node & ~1u];
            if (a < 0 || b < 0) continue;
            const double dx = x[a] - x[b], dy = y[a] - y[b];
            e.weight = sqrt(dx * dx + dy * dy);
            edges_[e.id ^ 1].weight = e.weight;
        }
        for (uint32_t v : component) member_of[v] = -1;
    }
    return 0;
}

// graph.cpp:1337-1366: an out-edge twice as long (in the layout) as a sibling goes
uint32_t AssemblyGraph::remove_long_edges() {
    uint32_t num_long_edges = 0;
    for (const auto& node : nodes_) {
        if (!node.alive || node.suffix_edges.size() < 2) continue;
        for (uint32_t e : node.suffix_edges) {
            for (uint32_t o : node.suffix_edges) {
                if (e == o || edges_[e].is_marked || edges_[o].is_marked) continue;
                if (edges_[e].weight * 2.0 < edges_[o].weight) {
                    mark_edge(o);
                    ++num_long_edges;
                }
            }
        }
    }
    remove_marked_objects();
    return num_long_edges;
}

// A tip (graph.cpp:1368-1438) is a dead-end chain of at most five reads that hangs off a junction.  Stated here as three
// questions about one chain: where does the chain that starts at a tip node stop (Chain), is it short and open enough
// to go, and which edges go with it - the chain's own edges only when every way out of its last node leads into a node that
// has another way in.
namespace {
struct Chain {
    uint32_t last;          // where the walk stopped
    uint32_t reads;         // reads of the nodes walked over
    bool closed;            // came back to its first node
};
}  // namespace

uint32_t AssemblyGraph::remove_tips() {
    const size_t n0 = nodes_.size();
    std::vector<bool> walked(n0, false);

    // follow the only out-edge while neither the node nor its successor branches
    auto follow = [&](uint32_t first) {
        Chain c{first, 0, false};
        for (uint32_t v = first; !nodes_[v].is_junction();) {
            c.last = v;
            c.reads += (uint32_t)nodes_[v].sequence_ids.size();
            walked[v] = walked[v ^ 1u] = true;
            if (nodes_[v].outdegree() == 0) break;
            const uint32_t next = edges_[nodes_[v].suffix_edges[0]].end_node;
            if (nodes_[next].is_junction()) break;
            if (next == first) { c.closed = true; break; }
            v = next;
        }
        return c;
    };
    // the ways out of the chain's end that somebody else also takes
    auto shared_exits = [&](uint32_t v) {
        std::vector<uint32_t> out;
        for (uint32_t e : nodes_[v].suffix_edges) {
            if (nodes_[edges_[e].end_node].indegree() > 1) out.push_back(e);
        }
        return out;
    };

    uint32_t removed = 0;
    for (size_t i = 0; i < n0; ++i) {
        if (!nodes_[i].alive || walked[i] || !nodes_[i].is_tip()) continue;
        const Chain c = follow((uint32_t)i);
        if (c.closed || c.reads > 5 || nodes_[c.last].outdegree() == 0) continue;
        const std::vector<uint32_t> exits = shared_exits(c.last);
        for (uint32_t e : exits) mark_edge(e);
        if (exits.size() == nodes_[c.last].suffix_edges.size()) {
            // nothing holds the chain any more: its own edges go as well
            for (uint32_t v = (uint32_t)i; v != c.last;) {
                const uint32_t e = nodes_[v].suffix_edges[0];
                mark_edge(e);
                v = edges_[e].end_node;
            }
        }
        removed += (uint32_t)exits.size();
        remove_marked_objects(true);
    }
    return removed;
}

uint32_t AssemblyGraph::find_edge(uint32_t src, uint32_t dst) const {
    for (uint32_t e : nodes_[src].suffix_edges) {
        if (edges_[e].end_node == dst) return e;
    }
    fprintf(stderr, "[rala::Graph::find_edge] error: missing edge between nodes %u and %u\n", src, dst);
    exit(1);
}

uint32_t AssemblyGraph::path_length(const std::vector<uint32_t>& path) const {
    if (path.empty()) return 0;
    uint32_t len = nodes_[path.back()].length();
    for (size_t i = 0; i + 1 < path.size(); ++i) {
        for (uint32_t e : nodes_[path[i]].suffix_edges) {
            if (edges_[e].end_node == path[i + 1]) { len += edges_[e].length; break; }
        }
    }
    return len;
}

// graph.cpp:1636-1702
void AssemblyGraph::find_removable_edges(std::vector<uint32_t>& dst, const std::vector<uint32_t>& path) const {
    if (path.empty()) return;
    int64_t pref = -1;                   // first inner node with several in edges
    for (size_t i = 1; i + 1 < path.size(); ++i) {
        if (nodes_[path[i]].indegree() > 1) { pref = (int64_t)i; break; }
    }
    int64_t suff = -1;                   // last inner node with several out edges
    for (size_t i = 1; i + 1 < path.size(); ++i) {
        if (nodes_[path[i]].outdegree() > 1) suff = (int64_t)i;
    }
    auto take = [&](int64_t from, int64_t to) {
        for (int64_t i = from; i < to; ++i) dst.push_back(find_edge(path[i], path[i + 1]));
    };
    const int64_t last = (int64_t)path.size() - 1;
    if (pref == -1 && suff == -1) { take(0, last); return; }
    if (pref != -1 && nodes_[path[pref]].outdegree() > 1) return;
    if (suff != -1 && nodes_[path[suff]].indegree() > 1) return;
    if (pref == -1) take(suff, last);
    else if (suff == -1) take(0, pref);
    else if (suff < pref) take(suff, pref);
}

// Bubbles (graph.cpp:1440-1613): from every node with two ways out, a breadth-first search over out-edges (at most
// 5 Mb away) until some node is reached a second time - the sink; the two ways there are the bubble's sides, and the
// side with fewer reads loses the edges find_removable_edges lets go.  The search state is one frame that is wiped
// after every source (only what was touched).
namespace {
struct BubbleSearch {
    std::vector<uint32_t> reach;            // length of the way found to a node
    std::vector<int64_t> via;               // the node it was first reached from, -1 = not yet
    std::vector<uint32_t> touched;
    std::deque<uint32_t> frontier;
    explicit BubbleSearch(size_t n) : reach(n, 0), via(n, -1) {}
    void wipe() {
        for (uint32_t v : touched) { reach[v] = 0; via[v] = -1; }
        touched.clear();
        frontier.clear();
    }
    // the way from `source` to `node` along the first-reached links
    std::vector<uint32_t> way(uint32_t source, uint32_t node) const {
        std::vector<uint32_t> p;
        for (uint32_t v = node; v != source; v = (uint32_t)via[v]) p.push_back(v);
        p.push_back(source);
        std::reverse(p.begin(), p.end());
        return p;
    }
};
constexpr uint32_t kBubbleReach = 5000000;
}  // namespace

uint32_t AssemblyGraph::remove_bubbles() {
    const size_t n0 = nodes_.size();
    BubbleSearch bfs(n0);

    // sink and the second node it was reached from, if the search from `source` closes a bubble
    auto search = [&](uint32_t source, uint32_t& sink, uint32_t& second) {
        bfs.frontier.push_back(source);
        bfs.touched.push_back(source);
        while (!bfs.frontier.empty()) {
            const uint32_t v = bfs.frontier.front();
            bfs.frontier.pop_front();
            for (uint32_t e : nodes_[v].suffix_edges) {
                const uint32_t w = edges_[e].end_node;
                if (w == source || bfs.reach[v] + edges_[e].length > kBubbleReach) continue;     // a cycle; out of reach
                bfs.reach[w] = bfs.reach[v] + edges_[e].length;
                bfs.touched.push_back(w);
                bfs.frontier.push_back(w);
                if (bfs.via[w] != -1) { sink = w; second = v; return true; }
                bfs.via[w] = v;
            }
        }
        return false;
    };
    auto branches_inside = [&](const std::vector<uint32_t>& side) {
        for (size_t i = 1; i + 1 < side.size(); ++i) if (nodes_[side[i]].is_junction()) return true;
        return false;
    };
    auto reads_on = [&](const std::vector<uint32_t>& side) {
        uint64_t n = 0;
        for (uint32_t v : side) n += nodes_[v].sequence_ids.size();
        return n;
    };
    auto alike = [](uint32_t a, uint32_t b) { return std::min(a, b) >= std::max(a, b) * 0.8; };
    // two sides that share only their ends, no node together with its reverse complement; sides of unlike length
    // must not branch inside
    auto sides_make_a_bubble = [&](const std::vector<uint32_t>& one, const std::vector<uint32_t>& two) {
        if (one.empty() || two.empty()) return false;
        std::unordered_set<uint32_t> seen(one.begin(), one.end());
        seen.insert(two.begin(), two.end());
        if (one.size() + two.size() - 2 != seen.size()) return false;
        for (uint32_t v : one) if (seen.count(v ^ 1u)) return false;
        if (!alike(path_length(one), path_length(two)) && (branches_inside(two) || branches_inside(one))) return false;
        return true;
    };

    uint32_t popped = 0;
    for (size_t s0 = 0; s0 < n0; ++s0) {
        if (!nodes_[s0].alive || nodes_[s0].outdegree() < 2) continue;
        const uint32_t source = (uint32_t)s0;
        uint32_t sink = 0, second = 0;
        if (search(source, sink, second)) {
            const std::vector<uint32_t> one = bfs.way(source, sink);
            std::vector<uint32_t> two = bfs.way(source, second);
            two.push_back(sink);
            if (sides_make_a_bubble(one, two)) {
                const bool one_is_heavier = reads_on(one) > reads_on(two);
                std::vector<uint32_t> doomed;
                find_removable_edges(doomed, one_is_heavier ? two : one);
                if (doomed.empty() && alike(path_length(one), path_length(two))) {
                    find_removable_edges(doomed, one_is_heavier ? one : two);
                }
                for (uint32_t e : doomed) mark_edge(e);
                if (!doomed.empty()) {
                    remove_marked_objects(true);
                    ++popped;
                }
            }
        }
        bfs.wipe();
    }
    return popped;
}

// graph.cpp:133-170 (Node constructor for unitigs)
uint32_t AssemblyGraph::append_unitig(uint32_t begin_node, uint32_t end_node) {
    Node u;
    u.id = nodes_.size();
    u.alive = true;
    u.is_first_rc = nodes_[begin_node].is_first_rc;
    uint32_t node = begin_node;
    while (true) {
        const Edge& edge = edges_[nodes_[node].suffix_edges[0]];
        u.data += nodes_[node].data.substr(0, edge.length);
        u.sequence_ids.insert(u.sequence_ids.end(), nodes_[node].sequence_ids.begin(), nodes_[node].sequence_ids.end());
        u.is_last_rc = nodes_[node].is_last_rc;
        node = edge.end_node;
        if (node == end_node) break;
    }
    if (begin_node != end_node) {
        u.data += nodes_[end_node].data;
        u.sequence_ids.insert(u.sequence_ids.end(), nodes_[end_node].sequence_ids.begin(),
            nodes_[end_node].sequence_ids.end());
        u.is_last_rc = nodes_[end_node].is_last_rc;
    }
    nodes_.push_back(std::move(u));
    return (uint32_t)nodes_.size() - 1;
}

// graph.cpp:1760-1845 (and the same block of shrink, :1934-2012): unitig + complement, the
// edge that enters the chain and the edge that leaves it are re-created on the unitigs, every
// edge of the chain is marked
void AssemblyGraph::splice_unitig(uint32_t begin_node, uint32_t end_node, bool attach) {
    const uint32_t unitig = append_unitig(begin_node, end_node);
    const uint32_t complement = append_unitig(end_node ^ 1u, begin_node ^ 1u);
    auto new_edge = [&](uint32_t from, uint32_t to, uint32_t length) {
        Edge e;
        e.id = edges_.size();
        e.begin_node = from; e.end_node = to; e.length = length;
        e.alive = true;
        edges_.push_back(e);
        return (uint32_t)e.id;
    };
    if (attach) {
        if (nodes_[begin_node].indegree() != 0) {
            const uint32_t edge = nodes_[begin_node].prefix_edges[0];
            const uint32_t from = edges_[edge].begin_node, to_c = edges_[edge ^ 1u].end_node;
            mark_edge(edge);
            const uint32_t ue = new_edge(from, unitig, edges_[edge].length);
            const uint32_t uc = new_edge(complement, to_c,
                edges_[edge ^ 1u].length + nodes_[complement].length() - nodes_[begin_node ^ 1u].length());
            nodes_[from].suffix_edges.push_back(ue);
            nodes_[to_c].prefix_edges.push_back(uc);
            nodes_[unitig].prefix_edges.push_back(ue);
            nodes_[complement].suffix_edges.push_back(uc);
        }
        if (nodes_[end_node].outdegree() != 0) {
            const uint32_t edge = nodes_[end_node].suffix_edges[0];
            const uint32_t to = edges_[edge].end_node, from_c = edges_[edge ^ 1u].begin_node;
            mark_edge(edge);
            const uint32_t ue = new_edge(unitig, to,
                edges_[edge].length + nodes_[unitig].length() - nodes_[end_node].length());
            const uint32_t uc = new_edge(from_c, complement, edges_[edge ^ 1u].length);
            nodes_[unitig].suffix_edges.push_back(ue);
            nodes_[complement].prefix_edges.push_back(uc);
            nodes_[to].prefix_edges.push_back(ue);
            nodes_[from_c].suffix_edges.push_back(uc);
        }
    }
    uint32_t node = begin_node;
    while (true) {
        const uint32_t e = nodes_[node].suffix_edges[0];
        mark_edge(e);
        node = edges_[e].end_node;
        if (node == end_node) break;
    }
}

// graph.cpp:1704-1848
uint32_t AssemblyGraph::create_unitigs() {
    const size_t n0 = nodes_.size();
    std::vector<bool> is_visited(n0, false);
    uint32_t num_unitigs_created = 0;
    for (size_t i = 0; i < n0; ++i) {
        if (!nodes_[i].alive || is_visited[i] || nodes_[i].is_junction()) continue;
        bool is_circular = false;
        uint32_t begin_node = (uint32_t)i;
        while (!nodes_[begin_node].is_junction()) {
            is_visited[begin_node] = true;
            is_visited[begin_node ^ 1u] = true;
            if (nodes_[begin_node].indegree() == 0 ||
                nodes_[edges_[nodes_[begin_node].prefix_edges[0]].begin_node].is_junction()) {
                break;
            }
            begin_node = edges_[nodes_[begin_node].prefix_edges[0]].begin_node;
            if (begin_node == i) { is_circular = true; break; }
        }
        uint32_t end_node = (uint32_t)i;
        while (!nodes_[end_node].is_junction()) {
            is_visited[end_node] = true;
            is_visited[end_node ^ 1u] = true;
            if (nodes_[end_node].outdegree() == 0 ||
                nodes_[edges_[nodes_[end_node].suffix_edges[0]].end_node].is_junction()) {
                break;
            }
            end_node = edges_[nodes_[end_node].suffix_edges[0]].end_node;
            if (end_node == i) { is_circular = true; break; }
        }
        if (!is_circular && begin_node == end_node) continue;
        splice_unitig(begin_node, end_node, begin_node != end_node);
        ++num_unitigs_created;
    }
    remove_marked_objects(true);
    return num_unitigs_created;
}

// graph.cpp:1850-2040
uint32_t AssemblyGraph::shrink(uint32_t epsilon) {
    const size_t n0 = nodes_.size();
    std::vector<bool> is_visited(n0, false);
    std::vector<uint64_t> node_updates(n0, 0);       // forward-node id -> unitig that swallowed it
    uint32_t num_unitigs_created = 0;
    for (size_t i = 0; i < n0; ++i) {
        if (!nodes_[i].alive || is_visited[i] || nodes_[i].is_junction()) continue;
        uint32_t extension = 1;
        bool is_circular = false;
        uint32_t begin_node = (uint32_t)i;
        while (!nodes_[begin_node].is_junction()) {
            is_visited[begin_node] = true;
            is_visited[begin_node ^ 1u] = true;
            if (nodes_[begin_node].indegree() == 0 ||
                nodes_[edges_[nodes_[begin_node].prefix_edges[0]].begin_node].is_junction()) {
                break;
            }
            begin_node = edges_[nodes_[begin_node].prefix_edges[0]].begin_node;
            ++extension;
            if (begin_node == i) { is_circular = true; break; }
        }
        if (is_circular) continue;
        uint32_t end_node = (uint32_t)i;
        while (!nodes_[end_node].is_junction()) {
            is_visited[end_node] = true;
            is_visited[end_node ^ 1u] = true;
            if (nodes_[end_node].outdegree() == 0 ||
                nodes_[edges_[nodes_[end_node].suffix_edges[0]].end_node].is_junction()) {
                break;
            }
            end_node = edges_[nodes_[end_node].suffix_edges[0]].end_node;
            ++extension;
            if (end_node == i) { is_circular = true; break; }
        }
        if (is_circular || begin_node == end_node || extension < 2 * epsilon + 2) continue;
        for (uint32_t k = 0; k < epsilon; ++k) begin_node = edges_[nodes_[begin_node].suffix_edges[0]].end_node;
        for (uint32_t k = 0; k < epsilon; ++k) end_node = edges_[nodes_[end_node].prefix_edges[0]].begin_node;
        // transitive edges follow their nodes into the unitig (:1926-1931; the end node keeps its id)
        for (uint32_t node = begin_node; node != end_node; node = edges_[nodes_[node].suffix_edges[0]].end_node) {
            node_updates[node & ~1u] = nodes_.size();
        }
        splice_unitig(begin_node, end_node, true);
        ++num_unitigs_created;
    }
    remove_marked_objects(true);
    for (auto& te : transitive_edges_) {
        if (node_updates[te.first] != 0) te.first = node_updates[te.first];
        if (node_updates[te.second] != 0) te.second = node_updates[te.second];
    }
    std::sort(transitive_edges_.begin(), transitive_edges_.end());
    return num_unitigs_created;
}

// reference src/graph.cpp:2153-2179
void AssemblyGraph::write_csv(FILE* graph_file) const {
    const auto& nodes = nodes_;
    for (const auto& it : nodes) {
        if (!it.alive || !it.is_rc() || (it.outdegree() == 0 && it.indegree() == 0)) continue;
        const auto& pair = nodes[it.id ^ 1];
        fprintf(graph_file, "%lu LN:i:%u RC:i:%lu,%lu LN:i:%u RC:i:%lu,0,-\n", it.id, it.length(),
            it.sequence_ids.size(), pair.id, pair.length(), pair.sequence_ids.size());
    }
    for (const auto& it : edges_) {
        if (!it.alive) continue;
        const auto& b = nodes[it.begin_node];
        const auto& e = nodes[it.end_node];
        fprintf(graph_file, "%lu LN:i:%u RC:i:%lu,%lu LN:i:%u RC:i:%lu,1,%lu %u %lf\n", b.id, b.length(),
            b.sequence_ids.size(), e.id, e.length(), e.sequence_ids.size(), it.id, it.length, it.weight);
    }
}

// reference src/graph.cpp:2181-2226
void AssemblyGraph::write_gfa(FILE* graph_file) const {
    const auto& nodes = nodes_;
    std::unordered_map<uint64_t, std::string> unitig_name;
    uint32_t unitig_id = 0;
    auto name_of = [&](const Node& n) -> const std::string& {
        return !n.name.empty() ? n.name : unitig_name[n.id];
    };
    for (const auto& it : nodes) {
        if (!it.alive || it.is_rc() || (it.outdegree() == 0 && it.indegree() == 0)) continue;
        if (it.name.empty()) {
            const std::string name = "Utg" + std::to_string(unitig_id++);
            unitig_name[it.id] = name;
            unitig_name[it.id ^ 1] = name;
        }
        fprintf(graph_file, "S\t%s\t%s\tLN:i:%zu\tRC:i:%lu\n", name_of(it).c_str(), it.data.c_str(), it.data.size(),
            it.sequence_ids.size());
    }
    for (const auto& it : edges_) {
        if (!it.alive) continue;
        const auto& b = nodes[it.begin_node];
        const auto& e = nodes[it.end_node];
        fprintf(graph_file, "L\t%s\t%c\t%s\t%c\t%zuM\n", name_of(b).c_str(), b.is_rc() ? '-' : '+',
            name_of(e).c_str(), e.is_rc() ? '-' : '+', b.data.size() - it.length);
    }
}

// reference src/graph.cpp:2228-2297 (the reference walks an unordered_set of sequence ids for the piles;
// here ascending ids)
void AssemblyGraph::write_json(std::ostream& os, const std::function<std::string(uint64_t)>& pile_json) const {
    os << "{\"nodes\":{";
    bool is_first = true;
    const auto& nodes = nodes_;
    const auto& edges = edges_;
    std::set<uint64_t> sequence_ids;
    for (const auto& it : nodes) {
        if (!it.alive || it.is_rc() || !it.is_junction()) continue;
        if (!is_first) os << ",";
        is_first = false;
        os << "\"" << it.sequence_ids.front() << "\":{\"n\":" << it.id << ",";
        os << "\"p\":[";
        sequence_ids.insert(it.sequence_ids.front());
        for (size_t i = 0; i < it.prefix_edges.size(); ++i) {
            const auto& other = nodes[edges[it.prefix_edges[i]].begin_node];
            sequence_ids.insert(other.sequence_ids.back());
            os << "[\"" << other.sequence_ids.back() << "\",\"" << other.id << "\"," << other.is_last_rc << ","
               << other.length() - edges[it.prefix_edges[i]].length << "]";
            if (i + 1 < it.prefix_edges.size()) os << ",";
        }
        os << "],\"s\":[";
        for (size_t i = 0; i < it.suffix_edges.size(); ++i) {
            const auto& other = nodes[edges[it.suffix_edges[i]].end_node];
            sequence_ids.insert(other.sequence_ids.front());
            os << "[\"" << other.sequence_ids.front() << "\",\"" << other.id << "\"," << other.is_first_rc << ","
               << it.length() - edges[it.suffix_edges[i]].length << "]";
            if (i + 1 < it.suffix_edges.size()) os << ",";
        }
        os << "]}";
    }
    os << "}";
    if (sequence_ids.empty()) {
        os << "}";
        return;
    }
    os << ",\"piles\":{";
    is_first = true;
    for (uint64_t id : sequence_ids) {
        const std::string js = pile_json(id);
        if (js.empty()) continue;               // (a filtered read has no pile)
        if (!is_first) os << ",";
        is_first = false;
        os << js;
    }
    os << "}}";
}


}  // namespace rala
