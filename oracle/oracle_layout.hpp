// ORACLE — TEST INFRASTRUCTURE ONLY.  Never included by anything under rala_amd/.
//
// CPU restatement of the clean-up stages that follow transitive reduction in rvaser/rala
// src/graph.cpp: Node / Edge (:56-180), remove_long_edges (:1337-1366), remove_tips
// (:1368-1438), remove_bubbles (:1440-1613), find_edge / find_removable_edges (:1615-1702),
// create_unitigs (:1704-1848), shrink (:1850-2040), remove_marked_objects (:2118-2151),
// shrinkToFit (:30-54).
//
// PARITY UNPINNED: graph.cpp cannot be compiled in this image (its bioparser / thread_pool /
// logger headers are absent and no stand-ins are written), and the reference holds no tests or
// golden vectors for these functions.  The restatement keeps the reference's object model -
// heap nodes and edges that point at each other, nullable slots in nodes_ / edges_, a set of
// marked edge ids, shrinkToFit on the adjacency vectors - so that it is structurally
// independent of the index-based product code (rala_amd/host/assembly_graph.cpp) it checks.
// postprocess (:1056-1279), the force-directed layout behind the edge weights, is restated
// with the two sources of run-to-run variation pinned the way the product pins them: a fixed
// mt19937 seed instead of std::random_device, and ascending node ids wherever the reference
// walks an unordered_set (initial points, the repulsion sum, the order of equally large
// components).
#pragma once

#include <unordered_map>

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <math.h>

#include <algorithm>
#include <deque>
#include <memory>
#include <random>
#include <set>
#include <string>
#include <unordered_set>
#include <vector>

namespace ora_layout {

struct Edge;

struct Node {
    uint64_t id;
    std::string name, data;
    std::vector<Edge*> prefix_edges, suffix_edges;
    std::vector<uint64_t> sequence_ids;
    bool is_first_rc, is_last_rc;
    Node* pair;

    // graph.cpp:126-131
    Node(uint64_t id_, uint64_t sequence_id, const std::string& name_, const std::string& data_)
            : id(id_), name(name_), data(data_), sequence_ids(1, sequence_id), is_first_rc(id_ & 1),
              is_last_rc(id_ & 1), pair(nullptr) {}
    // graph.cpp:133-170
    Node(uint64_t id_, Node* begin_node, Node* end_node);

    bool is_rc() const { return id & 1; }
    uint32_t length() const { return data.size(); }
    uint32_t indegree() const { return prefix_edges.size(); }
    uint32_t outdegree() const { return suffix_edges.size(); }
    bool is_junction() const { return outdegree() > 1 || indegree() > 1; }
    bool is_tip() const { return outdegree() > 0 && indegree() == 0 && sequence_ids.size() < 6; }
};

struct Edge {
    uint64_t id;
    Node* begin_node;
    Node* end_node;
    uint32_t length;
    double weight;
    bool is_marked;
    Edge* pair;
    Edge(uint64_t id_, Node* b, Node* e, uint32_t len)
            : id(id_), begin_node(b), end_node(e), length(len), weight(0), is_marked(false), pair(nullptr) {}
    std::string label() const { return begin_node->data.substr(0, length); }
};

inline Node::Node(uint64_t id_, Node* begin_node, Node* end_node) : id(id_), pair(nullptr) {
    is_first_rc = begin_node->is_first_rc;
    is_last_rc = false;
    Node* node = begin_node;
    while (true) {
        Edge* edge = node->suffix_edges[0];
        data += edge->label();
        sequence_ids.insert(sequence_ids.end(), node->sequence_ids.begin(), node->sequence_ids.end());
        is_last_rc = node->is_last_rc;
        node = edge->end_node;
        if (node == end_node) break;
    }
    if (begin_node != end_node) {
        data += end_node->data;
        sequence_ids.insert(sequence_ids.end(), end_node->sequence_ids.begin(), end_node->sequence_ids.end());
        is_last_rc = end_node->is_last_rc;
    }
}

// graph.cpp:30-54: nulls to the back, order of the rest kept, then cut
template <typename T>
void shrink_to_fit(std::vector<T>& src, uint64_t begin) {
    uint64_t i = begin;
    for (uint64_t j = begin; i < src.size(); ++i) {
        if (src[i] != nullptr) continue;
        j = std::max(j, i);
        while (j < src.size() && src[j] == nullptr) ++j;
        if (j >= src.size()) break;
        if (i != j) std::swap(src[i], src[j]);
    }
    if (i < src.size()) src.resize(i);
}

class Layout {
public:
    std::vector<std::unique_ptr<Node>> nodes_;
    std::vector<std::unique_ptr<Edge>> edges_;
    std::unordered_set<uint32_t> marked_edges_;
    std::vector<std::pair<uint64_t, uint64_t>> transitive_edges_;

    // graph.cpp:553-574: forward and reverse-complement node of one read
    void add_node_pair(uint64_t sequence_id, const std::string& name, const std::string& data, const std::string& rc) {
        const uint64_t id = nodes_.size();
        nodes_.emplace_back(new Node(id, sequence_id, name, data));
        nodes_.emplace_back(new Node(id + 1, sequence_id, name, rc));
        nodes_[id]->pair = nodes_[id + 1].get();
        nodes_[id + 1]->pair = nodes_[id].get();
    }
    // graph.cpp:576-632: edges arrive in twin pairs
    void add_edge(uint32_t b, uint32_t e, uint32_t length) {
        const uint64_t id = edges_.size();
        edges_.emplace_back(new Edge(id, nodes_[b].get(), nodes_[e].get(), length));
        nodes_[b]->suffix_edges.push_back(edges_[id].get());
        nodes_[e]->prefix_edges.push_back(edges_[id].get());
        if (id & 1) {
            edges_[id]->pair = edges_[id - 1].get();
            edges_[id - 1]->pair = edges_[id].get();
        }
    }

    void mark(Edge* edge) {
        edge->is_marked = true;
        edge->pair->is_marked = true;
        marked_edges_.emplace(edge->id);
        marked_edges_.emplace(edge->pair->id);
    }

    // graph.cpp:1322-1332
    void note_transitive_edges() {
        for (const auto& it : marked_edges_) {
            if (it & 1) {
                transitive_edges_.emplace_back((edges_[it]->begin_node->id >> 1) << 1, (edges_[it]->end_node->id >> 1) << 1);
                transitive_edges_.emplace_back(transitive_edges_.back().second, transitive_edges_.back().first);
            }
        }
        std::sort(transitive_edges_.begin(), transitive_edges_.end());
    }

    // graph.cpp:1056-1279 (see the header of this file for what is pinned)
    void postprocess(uint32_t seed) {
        if (transitive_edges_.empty() == false) {
            std::vector<std::pair<uint64_t, uint64_t>> tmp = {transitive_edges_[0]};
            for (uint64_t i = 1; i < transitive_edges_.size(); ++i) {
                if (nodes_[transitive_edges_[i].first] == nullptr || nodes_[transitive_edges_[i].second] == nullptr) continue;
                if (transitive_edges_[i].first != transitive_edges_[i].second &&
                    transitive_edges_[i] != transitive_edges_[i - 1]) {
                    tmp.emplace_back(transitive_edges_[i]);
                }
            }
            tmp.swap(transitive_edges_);
        }
        std::vector<std::set<uint64_t>> components;
        std::vector<bool> is_visited(nodes_.size(), false);
        for (uint64_t i = 0; i < nodes_.size(); ++i) {
            if (nodes_[i] == nullptr || is_visited[i]) continue;
            components.resize(components.size() + 1);
            std::deque<uint64_t> que = {i};
            while (!que.empty()) {
                uint64_t j = que.front();
                que.pop_front();
                if (is_visited[j]) continue;
                const auto& node = nodes_[j];
                is_visited[node->id] = true;
                is_visited[node->pair->id] = true;
                components.back().emplace((node->id >> 1) << 1);
                for (const auto& it : node->prefix_edges) que.emplace_back(it->begin_node->id);
                for (const auto& it : node->suffix_edges) que.emplace_back(it->end_node->id);
            }
        }
        std::sort(components.begin(), components.end(), [](const std::set<uint64_t>& lhs, const std::set<uint64_t>& rhs) {
            return lhs.size() != rhs.size() ? lhs.size() > rhs.size() : *lhs.begin() < *rhs.begin();
        });
        std::mt19937 generator(seed);
        std::uniform_real_distribution<> distribution(0., 1.);
        using point = std::pair<double, double>;
        for (const auto& component : components) {
            if (component.size() < 6) continue;
            bool has_junctions = false;
            for (const auto& it : component) {
                if (nodes_[it]->is_junction()) {
                    has_junctions = true;
                    break;
                }
            }
            if (has_junctions == false) continue;
            uint32_t num_iterations = 100;
            double k = sqrt(1. / static_cast<double>(component.size()));
            double t = 0.1;
            double dt = t / static_cast<double>(num_iterations + 1);
            auto add = [](const point& x, const point& y) { return std::make_pair(x.first + y.first, x.second + y.second); };
            auto substract = [](const point& x, const point& y) { return std::make_pair(x.first - y.first, x.second - y.second); };
            auto multiply = [](const point& x, double s) { return std::make_pair(x.first * s, x.second * s); };
            auto norm = [](const point& x) { return sqrt(x.first * x.first + x.second * x.second); };
            std::vector<point> points(nodes_.size());
            for (const auto& it : component) {
                points[it].first = distribution(generator);
                points[it].second = distribution(generator);
            }
            for (uint32_t i = 0; i < num_iterations; ++i) {
                std::vector<point> displacements(nodes_.size());
                for (const auto& n : component) {
                    point displacement = {0., 0.};
                    for (const auto& m : component) {
                        if (n == m) continue;
                        auto delta = substract(points[n], points[m]);
                        auto distance = norm(delta);
                        if (distance < 0.01) distance = 0.01;
                        displacement = add(displacement, multiply(delta, (k * k) / (distance * distance)));
                    }
                    auto pull = [&](uint64_t m) {
                        auto delta = substract(points[n], points[m]);
                        auto distance = norm(delta);
                        if (distance < 0.01) distance = 0.01;
                        displacement = add(displacement, multiply(delta, -1. * distance / k));
                    };
                    for (const auto& e : nodes_[n]->prefix_edges) pull((e->begin_node->id >> 1) << 1);
                    for (const auto& e : nodes_[n]->suffix_edges) pull((e->end_node->id >> 1) << 1);
                    bool found = false;
                    for (const auto& e : transitive_edges_) {
                        if (e.first != n) {
                            if (found) break;
                            continue;
                        }
                        found = true;
                        pull(e.second);
                    }
                    auto length = norm(displacement);
                    if (length < 0.01) length = 0.1;
                    displacements[n] = add(displacements[n], multiply(displacement, t / length));
                }
                for (const auto& n : component) points[n] = add(points[n], displacements[n]);
                t -= dt;
                ++i;
            }
            for (const auto& it : edges_) {
                if (it == nullptr || it->id & 1) continue;
                auto n = (it->begin_node->id >> 1) << 1;
                auto m = (it->end_node->id >> 1) << 1;
                if (component.find(n) != component.end() && component.find(m) != component.end()) {
                    it->weight = norm(substract(points[n], points[m]));
                    it->pair->weight = it->weight;
                }
            }
        }
    }

    // graph.cpp:2118-2151
    void remove_marked_objects(bool remove_nodes = false) {
        auto delete_edges = [&](std::vector<Edge*>& edges) {
            for (uint32_t i = 0; i < edges.size(); ++i) {
                if (edges[i]->is_marked) edges[i] = nullptr;
            }
            shrink_to_fit(edges, 0);
        };
        std::unordered_set<uint32_t> marked_nodes;
        for (const auto& it : marked_edges_) {
            if (remove_nodes) {
                marked_nodes.emplace(edges_[it]->begin_node->id);
                marked_nodes.emplace(edges_[it]->end_node->id);
            }
            delete_edges(edges_[it]->begin_node->suffix_edges);
            delete_edges(edges_[it]->end_node->prefix_edges);
        }
        if (remove_nodes) {
            for (const auto& it : marked_nodes) {
                if (nodes_[it]->outdegree() == 0 && nodes_[it]->indegree() == 0) nodes_[it].reset();
            }
        }
        for (const auto& it : marked_edges_) edges_[it].reset();
        marked_edges_.clear();
    }

    // graph.cpp:1337-1366
    uint32_t remove_long_edges() {
        uint32_t num_long_edges = 0;
        for (const auto& node : nodes_) {
            if (node == nullptr || node->suffix_edges.size() < 2) continue;
            for (const auto& edge : node->suffix_edges) {
                for (const auto& other_edge : node->suffix_edges) {
                    if (edge->id == other_edge->id || edge->is_marked || other_edge->is_marked) continue;
                    if (edge->weight * 2.0 < other_edge->weight) {
                        mark(other_edge);
                        ++num_long_edges;
                    }
                }
            }
        }
        remove_marked_objects();
        return num_long_edges;
    }

    // graph.cpp:1368-1438
    uint32_t remove_tips() {
        uint32_t num_tip_edges = 0;
        std::vector<bool> is_visited(nodes_.size(), false);
        for (const auto& it : nodes_) {
            if (it == nullptr || is_visited[it->id] || !it->is_tip()) continue;
            bool is_circular = false;
            uint32_t num_reads = 0;
            auto end_node = it.get();
            while (!end_node->is_junction()) {
                num_reads += end_node->sequence_ids.size();
                is_visited[end_node->id] = true;
                is_visited[end_node->pair->id] = true;
                if (end_node->outdegree() == 0 || end_node->suffix_edges[0]->end_node->is_junction()) break;
                end_node = end_node->suffix_edges[0]->end_node;
                if (end_node->id == it->id) {
                    is_circular = true;
                    break;
                }
            }
            if (is_circular || end_node->outdegree() == 0 || num_reads > 5) continue;
            uint32_t num_removed_edges = 0;
            for (const auto& edge : end_node->suffix_edges) {
                if (edge->end_node->indegree() > 1) {
                    mark(edge);
                    ++num_removed_edges;
                }
            }
            if (num_removed_edges == end_node->suffix_edges.size()) {
                auto curr_node = it.get();
                while (curr_node->id != end_node->id) {
                    mark(curr_node->suffix_edges[0]);
                    curr_node = curr_node->suffix_edges[0]->end_node;
                }
            }
            num_tip_edges += num_removed_edges;
            remove_marked_objects(true);
        }
        return num_tip_edges;
    }

    // graph.cpp:1615-1634
    uint64_t find_edge(uint64_t src, uint64_t dst) {
        for (const auto& edge : nodes_[src]->suffix_edges) {
            if (edge->end_node->id == dst) return edge->id;
        }
        fprintf(stderr, "[ora_layout::find_edge] error: missing edge between nodes %lu and %lu\n", src, dst);
        exit(1);
    }

    // graph.cpp:1636-1702
    void find_removable_edges(std::vector<uint64_t>& dst, const std::vector<uint64_t>& path) {
        if (path.empty()) return;
        int64_t pref = -1;
        for (uint64_t i = 1; i < path.size() - 1; ++i) {
            if (nodes_[path[i]]->indegree() > 1) {
                pref = i;
                break;
            }
        }
        int64_t suff = -1;
        for (uint64_t i = 1; i < path.size() - 1; ++i) {
            if (nodes_[path[i]]->outdegree() > 1) suff = i;
        }
        if (pref == -1 && suff == -1) {
            for (uint64_t i = 0; i < path.size() - 1; ++i) dst.emplace_back(find_edge(path[i], path[i + 1]));
            return;
        }
        if (pref != -1 && nodes_[path[pref]]->outdegree() > 1) return;
        if (suff != -1 && nodes_[path[suff]]->indegree() > 1) return;
        if (pref == -1) {
            for (uint64_t i = suff; i < path.size() - 1; ++i) dst.emplace_back(find_edge(path[i], path[i + 1]));
        } else if (suff == -1) {
            for (int64_t i = 0; i < pref; ++i) dst.emplace_back(find_edge(path[i], path[i + 1]));
        } else if (suff < pref) {
            for (int64_t i = suff; i < pref; ++i) dst.emplace_back(find_edge(path[i], path[i + 1]));
        }
    }

    // graph.cpp:1440-1613
    uint32_t remove_bubbles() {
        std::vector<uint32_t> distance(nodes_.size(), 0);
        std::vector<uint64_t> visited(nodes_.size() + 1, 0);
        uint64_t visited_length = 0;
        std::vector<int64_t> predecessor(nodes_.size(), -1);
        std::deque<uint64_t> node_queue;

        auto extract_path = [&](std::vector<uint64_t>& dst, uint64_t source, uint64_t sink) {
            uint64_t curr_id = sink;
            while (curr_id != source) {
                dst.emplace_back(curr_id);
                curr_id = predecessor[curr_id];
            }
            dst.emplace_back(source);
            std::reverse(dst.begin(), dst.end());
        };
        auto calculate_path_length = [&](const std::vector<uint64_t>& path) -> uint32_t {
            if (path.empty()) return 0;
            uint32_t path_length = nodes_[path.back()]->length();
            for (uint64_t i = 0; i < path.size() - 1; ++i) {
                for (const auto& edge : nodes_[path[i]]->suffix_edges) {
                    if (edge->end_node->id == (uint64_t)path[i + 1]) {
                        path_length += edge->length;
                        break;
                    }
                }
            }
            return path_length;
        };
        auto is_valid_bubble = [&](const std::vector<uint64_t>& path, const std::vector<uint64_t>& other_path) -> bool {
            if (path.empty() || other_path.empty()) return false;
            std::unordered_set<uint64_t> node_set;
            for (const auto& it : path) node_set.emplace(it);
            for (const auto& it : other_path) node_set.emplace(it);
            if (path.size() + other_path.size() - 2 != node_set.size()) return false;
            for (const auto& it : path) {
                if (node_set.count(nodes_[it]->pair->id) != 0) return false;
            }
            uint32_t path_length = calculate_path_length(path);
            uint32_t other_path_length = calculate_path_length(other_path);
            if (std::min(path_length, other_path_length) < std::max(path_length, other_path_length) * 0.8) {
                for (uint64_t i = 1; i < other_path.size() - 1; ++i) {
                    if (nodes_[other_path[i]]->is_junction()) return false;
                }
                for (uint64_t i = 1; i < path.size() - 1; ++i) {
                    if (nodes_[path[i]]->is_junction()) return false;
                }
            }
            return true;
        };

        uint32_t num_bubbles_popped = 0;
        for (const auto& node : nodes_) {
            if (node == nullptr || node->outdegree() < 2) continue;
            bool found_sink = false;
            uint64_t sink = 0, sink_other_predecesor = 0;
            uint64_t source = node->id;
            node_queue.emplace_back(source);
            visited[visited_length++] = source;
            while (!node_queue.empty() && !found_sink) {
                uint64_t v = node_queue.front();
                const auto& curr_node = nodes_[v];
                node_queue.pop_front();
                for (const auto& edge : curr_node->suffix_edges) {
                    uint64_t w = edge->end_node->id;
                    if (w == source) continue;
                    if (distance[v] + edge->length > 5000000) continue;
                    distance[w] = distance[v] + edge->length;
                    visited[visited_length++] = w;
                    node_queue.emplace_back(w);
                    if (predecessor[w] != -1) {
                        sink = w;
                        sink_other_predecesor = v;
                        found_sink = true;
                        break;
                    }
                    predecessor[w] = v;
                }
            }
            if (found_sink) {
                std::vector<uint64_t> path;
                extract_path(path, source, sink);
                std::vector<uint64_t> other_path(1, sink);
                extract_path(other_path, source, sink_other_predecesor);
                if (is_valid_bubble(path, other_path)) {
                    uint64_t path_num_reads = 0;
                    for (const auto& it : path) path_num_reads += nodes_[it]->sequence_ids.size();
                    uint64_t other_path_num_reads = 0;
                    for (const auto& it : other_path) other_path_num_reads += nodes_[it]->sequence_ids.size();
                    std::vector<uint64_t> edges_for_removal;
                    if (path_num_reads > other_path_num_reads) {
                        find_removable_edges(edges_for_removal, other_path);
                    } else {
                        find_removable_edges(edges_for_removal, path);
                    }
                    if (edges_for_removal.empty()) {
                        uint32_t path_length = calculate_path_length(path);
                        uint32_t other_path_length = calculate_path_length(other_path);
                        if (std::min(path_length, other_path_length) >= std::max(path_length, other_path_length) * 0.8) {
                            if (path_num_reads > other_path_num_reads) {
                                find_removable_edges(edges_for_removal, path);
                            } else {
                                find_removable_edges(edges_for_removal, other_path);
                            }
                        }
                    }
                    for (const auto& edge_id : edges_for_removal) mark(edges_[edge_id].get());
                    if (!edges_for_removal.empty()) {
                        remove_marked_objects(true);
                        ++num_bubbles_popped;
                    }
                }
            }
            node_queue.clear();
            for (uint64_t i = 0; i < visited_length; ++i) {
                distance[visited[i]] = 0;
                predecessor[visited[i]] = -1;
            }
            visited_length = 0;
        }
        return num_bubbles_popped;
    }

    // the block shared by create_unitigs (graph.cpp:1760-1845) and shrink (:1934-2012)
    void make_unitig(Node* begin_node, Node* end_node, bool attach, uint64_t& node_id, uint64_t& edge_id,
                     std::vector<std::unique_ptr<Node>>& unitigs, std::vector<std::unique_ptr<Edge>>& unitig_edges) {
        std::unique_ptr<Node> unitig(new Node(node_id++, begin_node, end_node));
        std::unique_ptr<Node> unitig_complement(new Node(node_id++, end_node->pair, begin_node->pair));
        unitig->pair = unitig_complement.get();
        unitig_complement->pair = unitig.get();
        if (attach) {
            if (begin_node->indegree() != 0) {
                Edge* edge = begin_node->prefix_edges[0];
                mark(edge);
                std::unique_ptr<Edge> ue(new Edge(edge_id++, edge->begin_node, unitig.get(), edge->length));
                std::unique_ptr<Edge> uc(new Edge(edge_id++, unitig_complement.get(), edge->pair->end_node,
                    edge->pair->length + unitig_complement->length() - begin_node->pair->length()));
                ue->pair = uc.get();
                uc->pair = ue.get();
                edge->begin_node->suffix_edges.emplace_back(ue.get());
                edge->pair->end_node->prefix_edges.emplace_back(uc.get());
                unitig->prefix_edges.emplace_back(ue.get());
                unitig_complement->suffix_edges.emplace_back(uc.get());
                unitig_edges.emplace_back(std::move(ue));
                unitig_edges.emplace_back(std::move(uc));
            }
            if (end_node->outdegree() != 0) {
                Edge* edge = end_node->suffix_edges[0];
                mark(edge);
                std::unique_ptr<Edge> ue(new Edge(edge_id++, unitig.get(), edge->end_node,
                    edge->length + unitig->length() - end_node->length()));
                std::unique_ptr<Edge> uc(new Edge(edge_id++, edge->pair->begin_node, unitig_complement.get(),
                    edge->pair->length));
                ue->pair = uc.get();
                uc->pair = ue.get();
                unitig->suffix_edges.emplace_back(ue.get());
                unitig_complement->prefix_edges.emplace_back(uc.get());
                edge->end_node->prefix_edges.emplace_back(ue.get());
                edge->pair->begin_node->suffix_edges.emplace_back(uc.get());
                unitig_edges.emplace_back(std::move(ue));
                unitig_edges.emplace_back(std::move(uc));
            }
        }
        unitigs.emplace_back(std::move(unitig));
        unitigs.emplace_back(std::move(unitig_complement));
        Node* node = begin_node;
        while (true) {
            Edge* edge = node->suffix_edges[0];
            mark(edge);
            node = edge->end_node;
            if (node == end_node) break;
        }
    }

    // graph.cpp:1704-1848
    uint32_t create_unitigs() {
        std::vector<bool> is_visited(nodes_.size(), false);
        uint64_t node_id = nodes_.size();
        std::vector<std::unique_ptr<Node>> unitigs;
        uint64_t edge_id = edges_.size();
        std::vector<std::unique_ptr<Edge>> unitig_edges;
        uint32_t num_unitigs_created = 0;
        for (const auto& it : nodes_) {
            if (it == nullptr || is_visited[it->id] || it->is_junction()) continue;
            bool is_circular = false;
            auto begin_node = it.get();
            while (!begin_node->is_junction()) {
                is_visited[begin_node->id] = true;
                is_visited[begin_node->pair->id] = true;
                if (begin_node->indegree() == 0 || begin_node->prefix_edges[0]->begin_node->is_junction()) break;
                begin_node = begin_node->prefix_edges[0]->begin_node;
                if (begin_node->id == it->id) {
                    is_circular = true;
                    break;
                }
            }
            auto end_node = it.get();
            while (!end_node->is_junction()) {
                is_visited[end_node->id] = true;
                is_visited[end_node->pair->id] = true;
                if (end_node->outdegree() == 0 || end_node->suffix_edges[0]->end_node->is_junction()) break;
                end_node = end_node->suffix_edges[0]->end_node;
                if (end_node->id == it->id) {
                    is_circular = true;
                    break;
                }
            }
            if (!is_circular && begin_node == end_node) continue;
            make_unitig(begin_node, end_node, begin_node != end_node, node_id, edge_id, unitigs, unitig_edges);
            ++num_unitigs_created;
        }
        for (uint64_t i = 0; i < unitigs.size(); ++i) nodes_.emplace_back(std::move(unitigs[i]));
        for (uint64_t i = 0; i < unitig_edges.size(); ++i) edges_.emplace_back(std::move(unitig_edges[i]));
        remove_marked_objects(true);
        return num_unitigs_created;
    }

    // graph.cpp:1850-2040
    uint32_t shrink(uint32_t epsilon) {
        std::vector<bool> is_visited(nodes_.size(), false);
        std::vector<uint64_t> node_updates(nodes_.size(), 0);
        uint64_t node_id = nodes_.size();
        std::vector<std::unique_ptr<Node>> unitigs;
        uint64_t edge_id = edges_.size();
        std::vector<std::unique_ptr<Edge>> unitig_edges;
        uint32_t num_unitigs_created = 0;
        for (const auto& it : nodes_) {
            if (it == nullptr || is_visited[it->id] || it->is_junction()) continue;
            uint32_t extension = 1;
            bool is_circular = false;
            auto begin_node = it.get();
            while (!begin_node->is_junction()) {
                is_visited[begin_node->id] = true;
                is_visited[begin_node->pair->id] = true;
                if (begin_node->indegree() == 0 || begin_node->prefix_edges[0]->begin_node->is_junction()) break;
                begin_node = begin_node->prefix_edges[0]->begin_node;
                ++extension;
                if (begin_node->id == it->id) {
                    is_circular = true;
                    break;
                }
            }
            if (is_circular) continue;
            auto end_node = it.get();
            while (!end_node->is_junction()) {
                is_visited[end_node->id] = true;
                is_visited[end_node->pair->id] = true;
                if (end_node->outdegree() == 0 || end_node->suffix_edges[0]->end_node->is_junction()) break;
                end_node = end_node->suffix_edges[0]->end_node;
                ++extension;
                if (end_node->id == it->id) {
                    is_circular = true;
                    break;
                }
            }
            if (is_circular || begin_node == end_node || extension < 2 * epsilon + 2) continue;
            for (uint32_t i = 0; i < epsilon; ++i) begin_node = begin_node->suffix_edges[0]->end_node;
            for (uint32_t i = 0; i < epsilon; ++i) end_node = end_node->prefix_edges[0]->begin_node;
            for (auto node = begin_node; node != end_node; node = node->suffix_edges[0]->end_node) {
                node_updates[(node->id >> 1) << 1] = node_id;
            }
            make_unitig(begin_node, end_node, true, node_id, edge_id, unitigs, unitig_edges);
            ++num_unitigs_created;
        }
        for (uint64_t i = 0; i < unitigs.size(); ++i) nodes_.emplace_back(std::move(unitigs[i]));
        for (uint64_t i = 0; i < unitig_edges.size(); ++i) edges_.emplace_back(std::move(unitig_edges[i]));
        remove_marked_objects(true);
        for (auto& it : transitive_edges_) {
            if (node_updates[it.first] != 0) it.first = node_updates[it.first];
            if (node_updates[it.second] != 0) it.second = node_updates[it.second];
        }
        std::sort(transitive_edges_.begin(), transitive_edges_.end());
        return num_unitigs_created;
    }

    // ---- the on-disk formats (graph.cpp:2153-2297), written to a string ------------------------
    // print_csv (graph.cpp:2153-2179)
    std::string print_csv() const {
        std::string out;
        char line[512];
        for (const auto& it : nodes_) {
            if (it == nullptr || !it->is_rc() || (it->outdegree() == 0 && it->indegree() == 0)) continue;
            snprintf(line, sizeof(line), "%lu LN:i:%u RC:i:%lu,%lu LN:i:%u RC:i:%lu,0,-\n", it->id, it->length(),
                     it->sequence_ids.size(), it->pair->id, it->pair->length(), it->pair->sequence_ids.size());
            out += line;
        }
        for (const auto& it : edges_) {
            if (it == nullptr) continue;
            snprintf(line, sizeof(line), "%lu LN:i:%u RC:i:%lu,%lu LN:i:%u RC:i:%lu,1,%lu %u %lf\n", it->begin_node->id,
                     it->begin_node->length(), it->begin_node->sequence_ids.size(), it->end_node->id,
                     it->end_node->length(), it->end_node->sequence_ids.size(), it->id, it->length, it->weight);
            out += line;
        }
        return out;
    }
    // print_gfa (graph.cpp:2181-2226)
    std::string print_gfa() const {
        std::string out;
        std::unordered_map<uint64_t, std::string> unitig_name;
        uint32_t unitig_id = 0;
        auto name_of = [&](const Node* n) -> std::string { return !n->name.empty() ? n->name : unitig_name[n->id]; };
        for (const auto& it : nodes_) {
            if (it == nullptr || it->is_rc() || (it->outdegree() == 0 && it->indegree() == 0)) continue;
            if (it->name.empty()) {
                const std::string u = "Utg" + std::to_string(unitig_id++);
                unitig_name[it->id] = u;
                unitig_name[it->pair->id] = u;
            }
            out += "S\t" + name_of(it.get()) + "\t" + it->data + "\tLN:i:" + std::to_string(it->data.size()) + "\tRC:i:" +
                   std::to_string(it->sequence_ids.size()) + "\n";
        }
        for (const auto& it : edges_) {
            if (it == nullptr) continue;
            out += "L\t" + name_of(nodes_[it->begin_node->id].get()) + "\t" + (it->begin_node->is_rc() ? "-" : "+") + "\t" +
                   name_of(nodes_[it->end_node->id].get()) + "\t" + (it->end_node->is_rc() ? "-" : "+") + "\t" +
                   std::to_string(it->begin_node->data.size() - it->length) + "M\n";
        }
        return out;
    }
    // print_json (graph.cpp:2228-2297); the reference walks an unordered_set of sequence ids for the
    // piles, here (as in the product) ascending ids; pile_json(id) stands for piles_[id]->to_json()
    template <class PileJson>
    std::string print_json(const PileJson& pile_json) const {
        std::string out = "{\"nodes\":{";
        bool is_first = true;
        std::set<uint64_t> sequence_ids;
        for (const auto& it : nodes_) {
            if (it == nullptr || it->is_rc() || !it->is_junction()) continue;
            if (!is_first) out += ",";
            is_first = false;
            out += "\"" + std::to_string(it->sequence_ids.front()) + "\":{\"n\":" + std::to_string(it->id) + ",\"p\":[";
            sequence_ids.insert(it->sequence_ids.front());
            for (uint32_t i = 0; i < it->prefix_edges.size(); ++i) {
                const Node* other = it->prefix_edges[i]->begin_node;
                sequence_ids.insert(other->sequence_ids.back());
                out += "[\"" + std::to_string(other->sequence_ids.back()) + "\",\"" + std::to_string(other->id) + "\"," +
                       std::to_string((int)other->is_last_rc) + "," + std::to_string(other->length() - it->prefix_edges[i]->length) + "]";
                if (i < it->prefix_edges.size() - 1) out += ",";
            }
            out += "],\"s\":[";
            for (uint32_t i = 0; i < it->suffix_edges.size(); ++i) {
                const Node* other = it->suffix_edges[i]->end_node;
                sequence_ids.insert(other->sequence_ids.front());
                out += "[\"" + std::to_string(other->sequence_ids.front()) + "\",\"" + std::to_string(other->id) + "\"," +
                       std::to_string((int)other->is_first_rc) + "," + std::to_string(it->length() - it->suffix_edges[i]->length) + "]";
                if (i < it->suffix_edges.size() - 1) out += ",";
            }
            out += "]}";
        }
        out += "}";
        if (sequence_ids.empty()) return out + "}";
        out += ",\"piles\":{";
        is_first = true;
        for (uint64_t id : sequence_ids) {
            if (!is_first) out += ",";
            is_first = false;
            out += pile_json(id);
        }
        return out + "}}";
    }
};

}  // namespace ora_layout
