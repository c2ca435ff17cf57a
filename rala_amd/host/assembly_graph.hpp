/*!
 * @file assembly_graph.hpp
 *
 * @brief The assembly graph after Graph::construct and the clean-up stages that follow
 * transitive reduction (reference rvaser/rala src/graph.cpp:56-180 Node / Edge,
 * :1337-1366 remove_long_edges, :1368-1438 remove_tips, :1440-1613 remove_bubbles,
 * :1615-1702 find_edge / find_removable_edges, :1704-1848 create_unitigs, :1850-2040 shrink,
 * :2118-2151 remove_marked_objects).
 *
 * postprocess (:1056-1279) is the force-directed layout that gives edges their weights; its
 * O(n^2) iterations run through a caller-supplied engine (the GPU kernel behind
 * rala_hip_layout), everything around it - components, initial points, attraction lists - is
 * here.  The reference seeds it from std::random_device and walks unordered sets; this build
 * fixes the seed and the orders (ascending node ids), which makes runs reproducible.
 *
 * Index based: nodes and edges live in two vectors and refer to each other by position;
 * objects are created in pairs (forward, reverse complement), so the twin of object k is k ^ 1.
 * Plain host code - the graphs are small (about 1 % of the overlaps survive to here).
 */

#pragma once

#include <stdint.h>
#include <stdio.h>

#include <functional>
#include <ostream>
#include <string>
#include <utility>
#include <vector>

namespace rala {

class AssemblyGraph {
public:
    struct Node {
        uint64_t id = 0;
        std::string name;                   // empty for unitigs
        std::string data;
        std::vector<uint32_t> prefix_edges, suffix_edges;       // edge ids, insertion order
        std::vector<uint64_t> sequence_ids;
        bool is_first_rc = false, is_last_rc = false;
        bool alive = false;

        bool is_rc() const { return id & 1; }
        uint32_t length() const { return (uint32_t)data.size(); }
        uint32_t indegree() const { return (uint32_t)prefix_edges.size(); }
        uint32_t outdegree() const { return (uint32_t)suffix_edges.size(); }
        bool is_junction() const { return outdegree() > 1 || indegree() > 1; }
        bool is_tip() const { return outdegree() > 0 && indegree() == 0 && sequence_ids.size() < 6; }
    };
    struct Edge {
        uint64_t id = 0;
        uint32_t begin_node = 0, end_node = 0, length = 0;
        double weight = 0;
        bool is_marked = false;
        bool alive = false;
    };

    /*! @brief appends the node of a read and its reverse complement (ids 2k, 2k + 1) */
    void add_sequence_nodes(uint64_t sequence_id, const std::string& name, const std::string& data,
        const std::string& reverse_complement);
    /*! @brief appends one edge; edges must be added in twin pairs (ids 2k, 2k + 1) */
    void add_edge(uint32_t begin_node, uint32_t end_node, uint32_t length);

    void mark_edge(uint32_t edge_id);       // the edge and its twin
    void remove_marked_objects(bool remove_nodes = false);

    /*! @brief after marking transitive edges, before removing them (graph.cpp:1322-1332) */
    void note_transitive_edges();
    /*!
     * @brief one layout of n points: x, y in / out; the attraction partners of point i are
     * adj[adj_off[i] .. adj_off[i + 1]) (index n = a point fixed at the origin); `iterations`
     * steps of repulsion k^2 / d^2 from every other point + attraction d / k along the lists,
     * step length t, t -= dt after every step.  Returns 0 on success.
     */
    typedef std::function<int(uint32_t n, double* x, double* y, const uint32_t* adj_off, const uint32_t* adj,
        uint32_t iterations, double k, double t, double dt)> LayoutEngine;
    /*! @brief graph.cpp:1056-1279: edge weights from a force-directed layout per component */
    int postprocess(const LayoutEngine& engine, uint32_t seed = 0);
    uint32_t remove_long_edges();
    uint32_t remove_tips();
    uint32_t remove_bubbles();
    uint32_t create_unitigs();
    uint32_t shrink(uint32_t epsilon);

    /*! @brief the on-disk formats of the reference (graph.cpp:2153-2297): debug CSV, GFA, and the
     *  JSON that misc/plotter.py reads; pile_json(sequence id) supplies Pile::to_json */
    void write_csv(FILE* to) const;
    void write_gfa(FILE* to) const;
    void write_json(std::ostream& os, const std::function<std::string(uint64_t)>& pile_json) const;

    const std::vector<Node>& nodes() const { return nodes_; }
    const std::vector<Edge>& edges() const { return edges_; }
    std::vector<Node>& nodes() { return nodes_; }
    std::vector<Edge>& edges() { return edges_; }
    const std::vector<std::pair<uint64_t, uint64_t>>& transitive_edges() const { return transitive_edges_; }

private:
    uint32_t find_edge(uint32_t src, uint32_t dst) const;
    void find_removable_edges(std::vector<uint32_t>& dst, const std::vector<uint32_t>& path) const;
    uint32_t path_length(const std::vector<uint32_t>& path) const;
    // node that merges the chain begin .. end (walking first suffix edges); appended, not linked
    uint32_t append_unitig(uint32_t begin_node, uint32_t end_node);
    // replaces the chain begin .. end by a freshly appended unitig pair, re-attaching the edge
    // that enters the chain and the edge that leaves it
    void splice_unitig(uint32_t begin_node, uint32_t end_node, bool attach);

    std::vector<Node> nodes_;
    std::vector<Edge> edges_;
    std::vector<uint32_t> marked_edges_;
    std::vector<std::pair<uint64_t, uint64_t>> transitive_edges_;   // forward-node ids, both directions, sorted
};

}  // namespace rala
