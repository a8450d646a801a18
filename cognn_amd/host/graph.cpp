#include "graph.h"

#include <algorithm>
#include <cerrno>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <stdexcept>

namespace cognn {

namespace {
struct E2 {
    uint32_t src, dst;
};
}  // namespace

PartitionedGraph build_vertex_layout(int k, int64_t V, const int32_t* part) {
    if (k < 1) throw std::runtime_error("build_partitioned_graph: need at least one party");
    if (V < 0 || V >= (int64_t)1 << 31) throw std::runtime_error("build_partitioned_graph: vertex count out of range");
    PartitionedGraph G;
    G.k = k;
    G.num_vertices = V;
    G.tid.assign(part, part + V);
    G.row_of_vid.assign((size_t)V, 0);
    G.party.resize(k);
    for (int p = 0; p < k; ++p) G.party[p].party = p;
    for (int64_t v = 0; v < V; ++v) {
        const int32_t t = G.tid[v];
        if (t < 0 || t >= k) throw std::runtime_error("partition: tile index out of range for vid " + std::to_string(v));
        auto& pg = G.party[t];
        G.row_of_vid[v] = (uint32_t)pg.localVertexPos.size();
        pg.localVertexPos.push_back((uint64_t)v);      // ascending vid == sorted idVecs[self] (ss_...h:462-474)
    }
    for (auto& pg : G.party) {
        const size_t n = pg.localVertexPos.size();
        pg.trueInDeg.assign(n, 0); pg.inDeg.assign(n, 0); pg.outDeg.assign(n, 0); pg.isBorder.assign(n, 0); pg.selfDummy.assign(n, 0);
    }
    return G;
}

PartitionedGraph build_partitioned_graph(int k, int64_t V, int64_t E, const int64_t* src, const int64_t* dst,
                                         const int32_t* part, bool undirected) {
    PartitionedGraph G = build_vertex_layout(k, V, part);
    // degree accounting + bucket own-tile edges by source party (graph.h:607-633, graph_io_util.h:167-177)
    std::vector<std::vector<E2>> own(k);
    const int64_t total = undirected ? 2 * E : E;
    G.num_edges = total;
    for (int p = 0; p < k; ++p) own[p].reserve((size_t)(total / k + 16));
    auto add_edge = [&](int64_t s, int64_t d) {
        if (s < 0 || s >= V || d < 0 || d >= V) throw std::runtime_error("edge list: vertex id out of range");
        const int ps = G.tid[s], pd = G.tid[d];
        G.party[ps].outDeg[G.row_of_vid[s]]++;
        G.party[pd].trueInDeg[G.row_of_vid[d]]++;
        if (ps != pd) G.party[ps].isBorder[G.row_of_vid[s]] = 1;
        own[ps].push_back(E2{(uint32_t)s, (uint32_t)d});
    };
    for (int64_t e = 0; e < E; ++e) {
        add_edge(src[e], dst[e]);
        if (undirected) add_edge(dst[e], src[e]);      // graph_io_util.h:161-163
    }
    for (int p = 0; p < k; ++p) {
        auto& pg = G.party[p];
        auto& ed = own[p];
        // order: destination party, destination vid, then source vid == the order in which the sorted
        // edge vector (Edge::lessFunc: src then dst) appends sources to each destination (ss_...h:295-314)
        std::sort(ed.begin(), ed.end(), [&](const E2& a, const E2& b) {
            const int ta = G.tid[a.dst], tb = G.tid[b.dst];
            if (ta != tb) return ta < tb;
            if (a.dst != b.dst) return a.dst < b.dst;
            return a.src < b.src;
        });
        pg.out.assign(k, EdgeBlock());
        size_t i = 0;
        for (int g = 0; g < k; ++g) {
            EdgeBlock& blk = pg.out[g];
            if (g == p) {
                // rows = every local vertex; vertices without a local in-edge get the dummy self source, which
                // contributes nothing (isGatherDstVertexDummy) but inflates both degrees (ss_...h:411-418)
                const size_t n = pg.localVertexPos.size();
                blk.rows_vid = pg.localVertexPos;
                blk.rowptr.assign(n + 1, 0);
                size_t j = i;
                while (j < ed.size() && G.tid[ed[j].dst] == g) ++j;
                blk.col.reserve(j - i);
                for (size_t q = i; q < j; ++q) blk.rowptr[G.row_of_vid[ed[q].dst] + 1]++;
                for (size_t r = 0; r < n; ++r) blk.rowptr[r + 1] += blk.rowptr[r];
                for (size_t q = i; q < j; ++q) blk.col.push_back(G.row_of_vid[ed[q].src]);   // already grouped by dst
                i = j;
            } else {
                blk.rowptr.push_back(0);
                while (i < ed.size() && G.tid[ed[i].dst] == g) {
                    const uint32_t d = ed[i].dst;
                    blk.rows_vid.push_back(d);
                    while (i < ed.size() && ed[i].dst == d) {
                        blk.col.push_back(G.row_of_vid[ed[i].src]);
                        ++i;
                    }
                    blk.rowptr.push_back((uint32_t)blk.col.size());
                }
            }
        }
        const size_t n = pg.localVertexPos.size();
        pg.inDeg = pg.trueInDeg;
        const EdgeBlock& self = pg.out[p];
        for (size_t r = 0; r < n; ++r) {
            if (self.rowptr[r + 1] == self.rowptr[r]) {
                pg.selfDummy[r] = 1;
                pg.inDeg[r]++;
                pg.outDeg[r]++;
            }
        }
        std::vector<E2>().swap(ed);
    }
    return G;
}

static bool next_effective_line(std::istream& in, std::string& line) {
    // graph_io_util.h:17-22: skip empty lines and lines starting with '#'
    while (std::getline(in, line)) {
        if (!line.empty() && line[0] != '#') return true;
    }
    return false;
}

void load_partition_file(const std::string& path, std::vector<int32_t>& part) {
    std::ifstream in(path);
    if (!in.is_open()) throw std::runtime_error("cannot open partition file " + path);
    std::string line;
    std::vector<std::pair<uint64_t, int32_t>> rows;
    while (next_effective_line(in, line)) {
        std::istringstream iss(line);
        uint64_t vid = 0; uint32_t tid = 0;
        if (!(iss >> vid >> tid)) throw std::runtime_error("Invalid format in graph topology input files.");
        rows.emplace_back(vid, (int32_t)tid);
    }
    part.assign(rows.size(), -1);
    for (auto& r : rows) {
        if (r.first >= rows.size()) throw std::runtime_error("partition file: vertex ids must be dense in [0, n)");
        if (part[r.first] != -1) throw std::runtime_error("partition file: duplicate vertex " + std::to_string(r.first));
        part[r.first] = r.second;
    }
}

void load_edge_list_file(const std::string& path, std::vector<int64_t>& src, std::vector<int64_t>& dst) {
    std::ifstream in(path);
    if (!in.is_open()) throw std::runtime_error("cannot open edge list file " + path);
    std::string line;
    while (next_effective_line(in, line)) {
        const char* b = line.c_str();
        char* e = nullptr;
        errno = 0;
        const uint64_t s = strtoull(b, &e, 10);
        if (e == b || errno == ERANGE) throw std::runtime_error("Invalid format in graph topology input files.");
        b = e;
        const uint64_t d = strtoull(b, &e, 10);
        if (e == b || errno == ERANGE) throw std::runtime_error("Invalid format in graph topology input files.");
        src.push_back((int64_t)s);
        dst.push_back((int64_t)d);                      // optional weight column is unused by GCN (ss_...h:794)
    }
}

static const char kBinMagic[8] = {'C', 'O', 'G', 'N', 'N', 'B', 'G', '1'};

bool is_binary_graph_file(const std::string& path) {
    std::ifstream in(path, std::ios::binary);
    char m[8];
    return in.is_open() && in.read(m, 8) && std::equal(m, m + 8, kBinMagic);
}

void load_binary_graph_file(const std::string& path, std::vector<int64_t>& src, std::vector<int64_t>& dst, std::vector<int32_t>& part) {
    std::ifstream in(path, std::ios::binary);
    char m[8];
    uint64_t nv = 0, ne = 0;
    if (!in.is_open() || !in.read(m, 8) || !std::equal(m, m + 8, kBinMagic) || !in.read((char*)&nv, 8) || !in.read((char*)&ne, 8) ||
        nv >= (1ull << 31) || ne >= (1ull << 40))
        throw std::runtime_error("Invalid format in graph topology input files.");
    src.resize(ne); dst.resize(ne); part.resize(nv);
    if (!in.read((char*)src.data(), (std::streamsize)(ne * 8)) || !in.read((char*)dst.data(), (std::streamsize)(ne * 8)) ||
        !in.read((char*)part.data(), (std::streamsize)(nv * 4)))
        throw std::runtime_error("Invalid format in graph topology input files.");
}

void save_binary_graph_file(const std::string& path, const std::vector<int64_t>& src, const std::vector<int64_t>& dst,
                            const std::vector<int32_t>& part) {
    std::ofstream out(path, std::ios::binary);
    if (!out.is_open()) throw std::runtime_error("cannot write " + path);
    const uint64_t nv = part.size(), ne = src.size();
    out.write(kBinMagic, 8); out.write((const char*)&nv, 8); out.write((const char*)&ne, 8);
    out.write((const char*)src.data(), (std::streamsize)(ne * 8)); out.write((const char*)dst.data(), (std::streamsize)(ne * 8));
    out.write((const char*)part.data(), (std::streamsize)(nv * 4));
}

}  // namespace cognn
