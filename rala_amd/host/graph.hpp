/*!
 * @file graph.hpp
 *
 * @brief Graph class (interface of rvaser/rala src/graph.hpp:37-180) on top of librala_hip.
 *
 * construct() and remove_transitive_edges() run on the GPU through the C ABI
 * (include/rala_hip.h).  The clean-up after transitive reduction (tips, bubbles, unitigs,
 * shrink; reference src/graph.cpp:1337-2040) is host code on the small surviving graph
 * (assembly_graph.hpp).  The force-directed layout that weighs the edges for
 * remove_long_edges() (reference :1056-1279) runs its O(n^2) steps on the GPU
 * (rala_hip_layout) with fixed seeds; the reference seeds it from std::random_device.
 */

#pragma once

#include <stdint.h>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "assembly_graph.hpp"
#include "io.hpp"

struct rala_hip_ctx;
struct rala_hip_mg;

namespace rala {

class Sequence;
class Pile;
class Overlap;

class Graph;
std::unique_ptr<Graph> createGraph(const std::string& sequences_path, const std::string& overlaps_path,
    uint32_t num_threads);

class Graph {
public:
    ~Graph();

    /*! @brief pile-o-grams, chimera / containment removal, (repeat annotation), graph build */
    void construct(const std::string& sensitive_overlaps_path);
    /*! @brief transitive reduction, then the clean-up stages where available */
    void simplify();
    uint32_t remove_transitive_edges();
    uint32_t remove_long_edges();
    uint32_t remove_tips();
    uint32_t remove_bubbles();
    uint32_t create_unitigs();
    uint32_t shrink(uint32_t epsilon);
    void extract_contigs(std::vector<std::unique_ptr<Sequence>>& dst, bool drop_unassembled_sequences = true);
    void extract_nodes(std::vector<std::unique_ptr<Sequence>>& dst);
    void print_csv(const std::string& path) const;
    void print_gfa(const std::string& path) const;
    void print_json(const std::string& path) const;
    void print_debug(const std::string& prefix) const;

    /*! @brief per-read piles after construct() (nullptr = filtered), as the reference's piles_ */
    const std::vector<std::unique_ptr<Pile>>& piles() const { return piles_; }

    friend std::unique_ptr<Graph> createGraph(const std::string& sequences_path,
        const std::string& overlaps_path, uint32_t num_threads);

private:
    Graph(const std::string& sequences_path, const std::string& overlaps_path, uint32_t num_threads);
    Graph(const Graph&) = delete;
    const Graph& operator=(const Graph&) = delete;

    void initialize();
    void initialize_piles();
    void postprocess();

    void open_devices();

    std::string sequences_path_, overlaps_path_;
    uint32_t num_threads_;
    rala_hip_ctx* ctx_;                 // one GPU: the context; several: rank 0's replicated result
    // several GPUs of this node (RALA_GPUS / rala --gpus): one rank object per device, reads
    // partitioned over them (include/rala_hip.h, rala_hip_mg_*)
    std::vector<rala_hip_mg*> ranks_;
    void* local_group_ = nullptr;

    std::unordered_map<std::string, uint64_t> name_to_id_;
    std::vector<std::string> names_;
    std::vector<uint32_t> read_len_;
    std::vector<std::unique_ptr<Pile>> piles_;
    io::NameTable name_table_;
    io::OverlapColumns overlaps_;       // parsed once

    AssemblyGraph graph_;
    uint32_t layout_seed_ = 0;          // one fixed seed per layout round
};

}  // namespace rala
