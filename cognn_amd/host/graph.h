// Host-side graph preprocessing: partitioned edge list -> per-party index arrays -> flat CSRs.
// Restates SSEdgeCentricAlgoKernel::onPreprocessClient (include/ss_vertex_centric_algo_kernel.h:
// 279-534, "-r 1" no-dummy-edge mode) and the loader's degree accounting (include/graph.h:607-633,
// include/graph_io_util.h:167-177), but emits CSR directly instead of duplicated position vectors.
#pragma once
#include <stdint.h>
#include <string>
#include <vector>

namespace cognn {

// One CSR block "edges of party P whose destination lives in party g":
//   rows   = distinct destination vids in ascending order  (idVecs[g], ss_...h:462-464)
//   col    = source vertex as LOCAL ROW of party P          (updateSrcVertexPos[g], ss_...h:481,496)
// The duplicated updateDstVertexPos[g] of the reference is rows[] repeated (rowptr run lengths).
struct EdgeBlock {
    std::vector<uint64_t> rows_vid;
    std::vector<uint32_t> rowptr;   // rows_vid.size()+1
    std::vector<uint32_t> col;
};

struct PartyGraph {
    int party = 0;
    std::vector<uint64_t> localVertexPos;      // ascending vids = row order of every [n x F] tensor (ss_...h:474)
    std::vector<uint32_t> trueInDeg;           // what onAlgoKernelStart sees (ss_...h:177 runs before :190)
    std::vector<uint32_t> inDeg;               // localVertexInDeg after the dummy-source inflation (ss_...h:411-418,476)
    std::vector<uint32_t> outDeg;
    std::vector<uint8_t> isBorder;             // isLocalVertexBorder (graph_io_util.h:172)
    std::vector<uint8_t> selfDummy;            // isGatherDstVertexDummy[self] (ss_...h:487)
    std::vector<EdgeBlock> out;                // out[g], g = 0..k-1 (g == party: local edges, rows = all local vertices)
};

struct PartitionedGraph {
    int k = 0;
    int64_t num_vertices = 0;
    std::vector<int32_t> tid;                  // vid -> party      (partition file, graph_io_util.h:67-86)
    std::vector<uint32_t> row_of_vid;          // vid -> local row in its party's ordering
    std::vector<PartyGraph> party;
    int64_t num_edges = 0;
};

// Builds all parties' structures. Throws std::runtime_error on malformed input.
PartitionedGraph build_partitioned_graph(int k, int64_t num_vertices, int64_t num_edges, const int64_t* src,
                                         const int64_t* dst, const int32_t* part, bool undirected);

// Vertex ordering only (tid check, localVertexPos / row_of_vid, empty degree arrays): what a single-process run needs on the
// host before the device builds degrees and the CSR from the edge list (cognn_graph_build_colocated).
PartitionedGraph build_vertex_layout(int k, int64_t num_vertices, const int32_t* part);

// Text loaders with the reference's formats (graph_io_util.h:17-22,67-73,121-147; harness.cpp:21-48).
void load_partition_file(const std::string& path, std::vector<int32_t>& part);
void load_edge_list_file(const std::string& path, std::vector<int64_t>& src, std::vector<int64_t>& dst);

// Binary graph container for graphs whose text form is unwieldy (2^20 vertices / 2^24 edges = hundreds of MB of text):
//   "COGNNBG1" | u64 num_vertices | u64 num_edges | i64 src[num_edges] | i64 dst[num_edges] | i32 part[num_vertices]
// Same content as the edge-list + partition text files (graph_io_util.h:67-164); little endian.
bool is_binary_graph_file(const std::string& path);
void load_binary_graph_file(const std::string& path, std::vector<int64_t>& src, std::vector<int64_t>& dst, std::vector<int32_t>& part);
void save_binary_graph_file(const std::string& path, const std::vector<int64_t>& src, const std::vector<int64_t>& dst,
                            const std::vector<int32_t>& part);

}  // namespace cognn
