// gcn-optimize / gcn-inference-optimize — command-line entry point with the reference's interface
// (algo_kernels/common_harness/harness.cpp:50-212, include/harness.h:91-220):
//
//   gcn-optimize -t <parties> -g <tiles> -i <party index> -m <iterations> -p <parts> -s <setting>
//                [-n 1] [-c 1] [-r 1] [-u] <edge list> <vertex list> <partition> <output> <config> [src]
//
// The reference starts one process per party and connects them over TCP; this binary hosts all parties of the
// run on one GPU (in-device share exchange) and prints the log lines of party `-i` (its "::<tag> took X seconds"
// and accuracy lines, tools/plot/*.py).  One-party-per-GPU runs go through the torch.distributed launcher
// (tools/run_cluster.py).  `-r 0` (power-of-two dummy edges, ss_...h:358-398) is not supported: no script of the
// reference uses it and its dummy contributions are never masked.
#include <getopt.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/cognn_engine.h"
#include "graph.h"

namespace {

struct GnnParam {               // include/task/task.h:78-170
    int num_layers = 2, num_labels = 0, input_dim = 0, hidden_dim = 0, num_samples = 0, num_edges = 0;
    double learning_rate = 0, train_ratio = 0, val_ratio = 0, test_ratio = 0;
    void readConfig(const std::string& file) {
        std::ifstream fin(file);
        if (!fin.is_open()) { std::cerr << "Failed to open the file: " << file << std::endl; return; }
        std::string param; char colon;
        while (fin >> param >> colon) {
            if (colon != ':') { std::cerr << "Invalid format: expected a colon after " << param << std::endl; break; }
            if (param == "num_layers") fin >> num_layers;
            else if (param == "num_labels") fin >> num_labels;
            else if (param == "input_dim") fin >> input_dim;
            else if (param == "hidden_dim") fin >> hidden_dim;
            else if (param == "num_samples") fin >> num_samples;
            else if (param == "num_edges") fin >> num_edges;
            else if (param == "learning_rate") fin >> learning_rate;
            else if (param == "train_ratio") fin >> train_ratio;
            else if (param == "val_ratio") fin >> val_ratio;
            else if (param == "test_ratio") fin >> test_ratio;
            else { std::cerr << "Unknown parameter: " << param << std::endl; break; }
        }
    }
};

void printHelp(const char* prog) {
    std::cerr << "Usage: " << prog << " -t <parties> -g <tiles> [options] <edgelistFile> <vertexlistFile> [partitionFile] [outputFile] [GNNConfigFile] [src]\n\n"
              << "Options:\n"
              << "\t-t <threadCount>      Number of parties (tiles handled by this run).\n"
              << "\t-g <graphTileCount>   Total number of graph tiles.\n"
              << "\t-i <tileIndex>        Party whose log lines are printed.\n"
              << "\t-m [maxIter]          Maximum GAS iterations (6 per training epoch, 2 = one inference pass).\n"
              << "\t-p [numParts]         Number of partitions per thread (unused).\n"
              << "\t-s <setting>          Setting string (keys the dealer / offline phase).\n"
              << "\t-n <0|1>              1: skip the offline phase up front (products are dealt on demand).\n"
              << "\t-c <0|1>              Cluster mode (accepted, ignored: there are no sockets).\n"
              << "\t-r <0|1>              1: no dummy edges (required).\n"
              << "\t-u                    Treat the edge list as undirected.\n"
              << "\t-h                    Print this help message.\n";
}

void print_duration(std::chrono::high_resolution_clock::time_point t0, const char* tag) {
    const double sec = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
    printf("::%s took %lf seconds\n", tag, sec);
}

uint64_t fnv1a(const std::string& s) {
    uint64_t h = 0xcbf29ce484222325ull;
    for (unsigned char c : s) { h ^= c; h *= 0x100000001b3ull; }
    return h;
}

}  // namespace

int main(int argc, char* argv[]) {
    const std::string prog = argv[0];
    size_t threadCount = 0, graphTileCount = 0, tileIndex = 0;
    uint64_t maxIters = 1000;     // maxItersDefault (harness.h:22)
    uint32_t numParts = 16;       // numPartsDefault (harness.h:23)
    bool undirected = false, noPreprocess = false, isCluster = false, isNoDummyEdge = false;
    std::string setting;
    int ch;
    opterr = 0;
    while ((ch = getopt(argc, argv, "t:g:i:m:p:s:n:c:r:uh")) != -1) {
        uint32_t flag = 0;
        switch (ch) {
            case 't': std::stringstream(optarg) >> threadCount; break;
            case 'g': std::stringstream(optarg) >> graphTileCount; break;
            case 'i': std::stringstream(optarg) >> tileIndex; break;
            case 'm': std::stringstream(optarg) >> maxIters; break;
            case 'p': std::stringstream(optarg) >> numParts; break;
            case 's': std::stringstream(optarg) >> setting; break;       // (the reference falls through into -n here, harness.h:140-146; harmless)
            case 'n': std::stringstream(optarg) >> flag; if (flag == 1) noPreprocess = true; break;
            case 'c': std::stringstream(optarg) >> flag; if (flag == 1) isCluster = true; break;
            case 'r': std::stringstream(optarg) >> flag; if (flag == 1) isNoDummyEdge = true; break;
            case 'u': undirected = true; break;
            case 'h':
            default: printHelp(argv[0]); return -1;
        }
    }
    if (threadCount == 0 || graphTileCount == 0) {
        std::cerr << "Must specify number of threads and number of graph tiles." << std::endl;
        printHelp(argv[0]);
        return -1;
    }
    if (graphTileCount % threadCount != 0) {
        std::cerr << "Number of threads must be a divisor of number of graph tiles." << std::endl;
        return -1;
    }
    argc -= optind; argv += optind;
    if (argc < 5) {
        std::cerr << "Must specify an input edge list file, a vertex list file, a partition file, an output file and a GNN config file." << std::endl;
        return -1;
    }
    const std::string edgelistFile = argv[0], vertexlistFile = argv[1], partitionFile = argv[2], outputFile = argv[3], configFile = argv[4];
    (void)outputFile; (void)numParts; (void)isCluster;
    if (!isNoDummyEdge) {
        std::cerr << "Only the no-dummy-edge mode (-r 1) is supported." << std::endl;
        return -1;
    }
    if (tileIndex >= threadCount) { std::cerr << "Tile index out of range." << std::endl; return -1; }
    const bool inference = prog.find("inference") != std::string::npos || getenv("COGNN_INFERENCE_VARIANT");

    GnnParam gp;
    gp.readConfig(configFile);
    const int k = (int)threadCount;
    try {
        auto t_pre = std::chrono::high_resolution_clock::now();
        std::vector<int32_t> part;
        std::vector<int64_t> src, dst;
        if (cognn::is_binary_graph_file(edgelistFile)) {
            cognn::load_binary_graph_file(edgelistFile, src, dst, part);            // binary container: edges + partition in one file
        } else {
            cognn::load_partition_file(partitionFile, part);
            cognn::load_edge_list_file(edgelistFile, src, dst);
        }
        for (auto& t : part) { t /= (int32_t)(graphTileCount / threadCount); }      // tileMergeFactor (graph_io_util.h:76)
        cognn_engine_config cfg{};
        cfg.num_parties = k; cfg.rank = 0; cfg.world = 1;
        cfg.variant = inference ? COGNN_VARIANT_OPTIMIZE_GCN_INFERENCE : COGNN_VARIANT_OPTIMIZE_GCN;
        cfg.num_layers = gp.num_layers; cfg.num_labels = gp.num_labels; cfg.input_dim = gp.input_dim; cfg.hidden_dim = gp.hidden_dim;
        cfg.learning_rate = gp.learning_rate; cfg.train_ratio = gp.train_ratio; cfg.val_ratio = gp.val_ratio; cfg.test_ratio = gp.test_ratio;
        cfg.seed = fnv1a(setting); cfg.device = 0; cfg.stream = nullptr; cfg.undirected = undirected; cfg.verbose = 0;
        cognn_engine* e = nullptr;
        if (cognn_engine_create(&cfg, (int64_t)part.size(), (int64_t)src.size(), src.data(), dst.data(), part.data(), &e)) {
            std::cerr << cognn_engine_last_error() << std::endl;
            return -1;
        }
        // vertex data: "<vid> f_0 ... f_{in-1} <label>" (harness.cpp:21-48, kernel_harness.h:37-44)
        std::vector<std::vector<double>> feats(k);
        std::vector<std::vector<int32_t>> labels(k);
        std::vector<int64_t> rows(k);
        std::vector<std::vector<int64_t>> rowOf(k);
        std::vector<int64_t> rowIndex(part.size(), -1);
        for (int p = 0; p < k; ++p) {
            cognn_engine_party_rows(e, p, &rows[p]);
            std::vector<int64_t> vids((size_t)rows[p]);
            cognn_engine_party_vids(e, p, vids.data());
            for (int64_t r = 0; r < rows[p]; ++r) rowIndex[(size_t)vids[r]] = r;
            feats[p].assign((size_t)rows[p] * gp.input_dim, 0.0);
            labels[p].assign((size_t)rows[p], 0);
        }
        {
            std::ifstream in(vertexlistFile);
            if (!in.is_open()) { std::cerr << "cannot open vertex list file " << vertexlistFile << std::endl; return -1; }
            std::string line;
            while (std::getline(in, line)) {
                if (line.empty() || line[0] == '#') continue;
                std::istringstream iss(line);
                uint64_t vid = 0;
                if (!(iss >> vid)) { std::cerr << "Invalid format in vertex list file." << std::endl; return -1; }
                if (vid >= part.size()) continue;
                const int p = part[vid];
                const int64_t r = rowIndex[vid];
                for (int j = 0; j < gp.input_dim; ++j) iss >> feats[p][(size_t)r * gp.input_dim + j];
                iss >> labels[p][(size_t)r];
            }
        }
        for (int p = 0; p < k; ++p)
            if (cognn_engine_set_party_data(e, p, feats[p].data(), labels[p].data())) { std::cerr << cognn_engine_last_error() << std::endl; return -1; }
        std::cout << "Graph loaded from " << edgelistFile << " and " << partitionFile << " with " << graphTileCount << " graph tiles, into "
                  << threadCount << " tiles. Treated as " << (undirected ? "undirected" : "directed") << " graph.Current tile is the No."
                  << tileIndex << " tile." << std::endl;
        std::cout << tileIndex << " Initialize graph algo kernel" << std::endl;
        if (cognn_engine_start(e)) { std::cerr << cognn_engine_last_error() << std::endl; return -1; }
        print_duration(t_pre, "preprocess");
        // offline phase and its cache: ./preprocess/<setting>/ is written by a run without -n and reused by `-n 1`
        // (README.md:215-216, 306-307 of the reference); what is missing is dealt on demand
        std::string cacheDir = "preprocess/" + setting;
        for (auto& ch : cacheDir) if (ch == ' ') ch = '_';
        if (!noPreprocess) {
            auto t_om = std::chrono::high_resolution_clock::now();
            if (cognn_engine_offline(e, 0, (int64_t)maxIters)) { std::cerr << cognn_engine_last_error() << std::endl; return -1; }
            print_duration(t_om, "preprocess_OM");
            if (!getenv("COGNN_NO_PREPROCESS_CACHE")) {
                const std::string cmd = "mkdir -p '" + cacheDir + "'";
                if (system(cmd.c_str()) == 0 && cognn_engine_offline_save(e, cacheDir.c_str()))
                    std::cerr << "warning: offline cache not written: " << cognn_engine_last_error() << std::endl;
            }
        } else {
            int64_t loaded = 0;
            if (cognn_engine_offline_load(e, cacheDir.c_str(), 0, (int64_t)maxIters, &loaded) == 0)
                std::cout << tileIndex << " Reused " << loaded << " offline products from " << cacheDir << std::endl;
        }
        std::cout << tileIndex << " Begin algo kernel iteration" << std::endl;
        const int epoch = 3 * gp.num_layers;
        for (uint64_t it = 0; it < maxIters; ++it) {
            printf("tid-> %lld, iteration-> %lld\n", (long long)tileIndex, (long long)it);
            auto t_it = std::chrono::high_resolution_clock::now();
            if (cognn_engine_run(e, (int64_t)it, (int64_t)it + 1)) { std::cerr << cognn_engine_last_error() << std::endl; return -1; }
            if ((int)(it % epoch) == gp.num_layers - 1) {                  // prediction layer: gcn.h:619-632
                double m[8];
                if (cognn_engine_get_metrics(e, (int32_t)tileIndex, m)) { std::cerr << cognn_engine_last_error() << std::endl; return -1; }
                printf("--------\n");
                printf("cross-entropy-loss = %lf\n", m[5]);
                printf("full set accuracy = %lf\n", m[0]);
                printf("training set accuracy = %lf\n", m[1]);
                printf("border training set accuracy = %lf\n", m[2]);
                printf("test set accuracy = %lf\n", m[3]);
                printf("border test set accuracy = %lf\n", m[4]);
                printf("the number of vertices is %lu, the number of border vertices is %lu\n", (unsigned long)m[6], (unsigned long)m[7]);
            } else {
                double m[8];
                (void)m;
                int64_t r, c;
                cognn_engine_get_shares(e, (int32_t)tileIndex, 0, nullptr, &r, &c);   // forces completion for the timing line
            }
            print_duration(t_it, "iteration");
        }
        std::cout << tileIndex << " Finish algo kernel" << std::endl;
        cognn_engine_destroy(e);
    } catch (const std::exception& ex) {
        std::cerr << ex.what() << std::endl;
        return -1;
    }
    return 0;
}
