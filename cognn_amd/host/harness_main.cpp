// gcn-optimize / gcn-inference-optimize / gcn-original — command-line entry point with the reference's interface
// (algo_kernels/common_harness/harness.cpp:50-212, include/harness.h:91-220):
//
//   gcn-optimize -t <parties> -g <tiles> -i <party index> -m <iterations> -p <parts> -s <setting>
//                [-n 1] [-c 1] [-r 1] [-u] <edge list> <vertex list> <partition> <output> <config> [src]
//
// The reference starts one process per party and connects them over TCP (include/engine.h:157-201).  This binary runs
//   * by default all parties of the run on one GPU (in-device share exchange), printing the log lines of party `-i`
//     (its "::<tag> took X seconds" and accuracy lines, tools/plot/*.py);
//   * with `-c 1` (the reference's cluster switch) or WORLD_SIZE > 1 in the environment as ONE RANK of a multi-GPU run:
//     the k parties are mapped to `world` processes in contiguous blocks, one process per GPU, shares travel over RCCL
//     p2p (include/cognn_exchange.h).  `-c 1` alone means one party per GPU: world = k, rank = `-i`, exactly the
//     reference's k command lines; RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT override (torchrun style).
// `-r 0` (power-of-two dummy edges, ss_...h:358-398) is not supported: no script of the reference uses it and its dummy
// contributions are never masked.
#include <getopt.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
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
#ifndef COGNN_NO_RCCL        /* the CPU test build (oracle/gcn-optimize-cpuref) has no device and no communicator */
#include "../../include/cognn_exchange.h"
#endif
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
              << "\t-m [maxIter]          Maximum GAS iterations (6 per training epoch, 2 = one inference pass; gcn-original: 4 per epoch).\n"
              << "\t-p [numParts]         Number of partitions per thread (unused).\n"
              << "\t-s <setting>          Setting string (keys the dealer / offline phase).\n"
              << "\t-n <0|1>              1: skip the offline phase up front (products are dealt on demand).\n"
              << "\t-c <0|1>              1: this process is one rank of a multi-GPU run (one party per GPU unless WORLD_SIZE says otherwise).\n"
              << "\t-r <0|1>              1: no dummy edges (required).\n"
              << "\t-u                    Treat the edge list as undirected.\n"
              << "\t-h                    Print this help message.\n";
}

// One log per hosted party: the party named by -i goes to stdout (the launcher redirects it into gcn_test_<dataset>_<i>.log,
// tmp_run_cluster.py:146); when a rank hosts several parties and COGNN_LOG_PREFIX is set, every other hosted party p gets
// <prefix><p>.log with the same lines and its own metrics, so that the reference's one-log-per-party layout is complete.
struct PartyLog { int party; FILE* f; };
std::vector<PartyLog> g_logs{{0, stdout}};
#define LOGF(...) do { for (auto& L_ : g_logs) fprintf(L_.f, __VA_ARGS__); } while (0)

void print_duration(std::chrono::high_resolution_clock::time_point t0, const char* tag) {
    const double sec = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
    LOGF("::%s took %lf seconds\n", tag, sec);
}

void print_seconds(double sec, const char* tag) { LOGF("::%s took %lf seconds\n", tag, sec); }

// mkdir -p without a shell (the setting string comes from the command line)
bool make_dirs(const std::string& path) {
    for (size_t i = 1; i <= path.size(); ++i) {
        if (i != path.size() && path[i] != '/') continue;
        const std::string sub = path.substr(0, i);
        if (mkdir(sub.c_str(), 0777) != 0 && errno != EEXIST) return false;
    }
    return true;
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
    (void)outputFile; (void)numParts;
    if (!isNoDummyEdge) {
        std::cerr << "Only the no-dummy-edge mode (-r 1) is supported." << std::endl;
        return -1;
    }
    if (tileIndex >= threadCount) { std::cerr << "Tile index out of range." << std::endl; return -1; }
    const bool inference = prog.find("inference") != std::string::npos || getenv("COGNN_INFERENCE_VARIANT");
    // bin/gcn-original = the unoptimised kernel (algo_kernels/vertex_centric/original-gcn, tools/tmp_run_cluster.py:285-286)
    const bool original = !inference && (prog.find("gcn-original") != std::string::npos || getenv("COGNN_ORIGINAL_VARIANT"));

    GnnParam gp;
    gp.readConfig(configFile);
    const int k = (int)threadCount;
    // rank layout of a multi-GPU run
    auto env_int = [](const char* name, int dflt) { const char* v = getenv(name); return v && *v ? atoi(v) : dflt; };
    int world = env_int("WORLD_SIZE", isCluster ? k : 1);
    if (world < 1 || k % world != 0) { std::cerr << "The number of parties must be a multiple of the number of ranks." << std::endl; return -1; }
    const int perRank = k / world;
    const int rank = world > 1 ? env_int("RANK", (int)tileIndex / perRank) : 0;
    if (rank < 0 || rank >= world) { std::cerr << "Rank out of range." << std::endl; return -1; }
    const int device = world > 1 ? env_int("LOCAL_RANK", rank) : 0;
    if (world > 1 && ((int)tileIndex < rank * perRank || (int)tileIndex >= (rank + 1) * perRank)) tileIndex = (size_t)(rank * perRank);
    g_logs[0].party = (int)tileIndex;
    if (const char* prefix = getenv("COGNN_LOG_PREFIX")) {
        for (int p = rank * perRank; p < (rank + 1) * perRank; ++p) {
            if (p == (int)tileIndex) continue;
            const std::string path = std::string(prefix) + std::to_string(p) + ".log";
            FILE* f = fopen(path.c_str(), "w");
            if (!f) { std::cerr << "cannot write " << path << std::endl; return -1; }
            g_logs.push_back(PartyLog{p, f});
        }
    }
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
        cfg.num_parties = k; cfg.rank = rank; cfg.world = world;
        cfg.variant = inference ? COGNN_VARIANT_OPTIMIZE_GCN_INFERENCE : original ? COGNN_VARIANT_ORIGINAL_GCN : COGNN_VARIANT_OPTIMIZE_GCN;
        cfg.num_layers = gp.num_layers; cfg.num_labels = gp.num_labels; cfg.input_dim = gp.input_dim; cfg.hidden_dim = gp.hidden_dim;
        cfg.learning_rate = gp.learning_rate; cfg.train_ratio = gp.train_ratio; cfg.val_ratio = gp.val_ratio; cfg.test_ratio = gp.test_ratio;
        cfg.seed = fnv1a(setting); cfg.device = device; cfg.stream = nullptr; cfg.undirected = undirected; cfg.verbose = 1;
        // which rank holds which share (multi-rank runs): the reference's command line has no switch for it, so the CLI - and only
        // the CLI - reads it from the environment; anything but the two names is an error, not a silent default
        if (const char* pl = getenv("COGNN_PLACEMENT")) {
            if (!strcmp(pl, "party") || !strcmp(pl, "0")) cfg.placement = COGNN_PLACE_PARTY;
            else if (!strcmp(pl, "vertex-set") || !strcmp(pl, "1")) cfg.placement = COGNN_PLACE_VERTEX_SET;
            else { std::cerr << "COGNN_PLACEMENT must be 'party' or 'vertex-set' (got '" << pl << "')." << std::endl; return -1; }
        }
        cognn_engine* e = nullptr;
        if (cognn_engine_create(&cfg, (int64_t)part.size(), (int64_t)src.size(), src.data(), dst.data(), part.data(), &e)) {
            std::cerr << cognn_engine_last_error() << std::endl;
            return -1;
        }
        if (inference && maxIters <= (uint64_t)gp.num_layers) cognn_engine_set_option(e, COGNN_OPT_FORWARD_ONLY, 1);   // -m 2: one inference pass
#ifdef COGNN_NO_RCCL
        if (world > 1) { std::cerr << "This build has no RCCL transport." << std::endl; return -1; }
#else
        cognn_rccl_exchange* xch = nullptr;
        if (world > 1) {                                         // communicator bootstrap: the counterpart of engine.h:166-201
            unsigned char id[COGNN_RCCL_ID_BYTES];
            const char* addr = getenv("MASTER_ADDR");
            if (cognn_rccl_rendezvous_tcp(addr && *addr ? addr : "127.0.0.1", env_int("MASTER_PORT", 0), rank, world, 120.0, id) ||
                cognn_rccl_exchange_create(id, rank, world, device, nullptr, &xch) || cognn_engine_set_exchange_rccl(e, xch)) {
                std::cerr << cognn_exchange_last_error() << std::endl;
                return -1;
            }
        }
#endif
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
            if (p / perRank != rank) continue;                  // only the parties hosted here need their vertex data
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
                if (p / perRank != rank) continue;
                const int64_t r = rowIndex[vid];
                for (int j = 0; j < gp.input_dim; ++j) iss >> feats[p][(size_t)r * gp.input_dim + j];
                iss >> labels[p][(size_t)r];
            }
        }
        for (int p = rank * perRank; p < (rank + 1) * perRank; ++p)
            if (cognn_engine_set_party_data(e, p, feats[p].data(), labels[p].data())) { std::cerr << cognn_engine_last_error() << std::endl; return -1; }
        for (auto& L : g_logs) {
            fprintf(L.f, "Graph loaded from %s and %s with %zu graph tiles, into %zu tiles. Treated as %s graph.Current tile is the No.%d tile.\n",
                    edgelistFile.c_str(), partitionFile.c_str(), (size_t)graphTileCount, (size_t)threadCount, undirected ? "undirected" : "directed", L.party);
            fprintf(L.f, "%d Initialize graph algo kernel\n", L.party);
        }
        if (cognn_engine_start(e)) { std::cerr << cognn_engine_last_error() << std::endl; return -1; }
        print_duration(t_pre, "preprocess");
        // Offline phase and its cache: ./preprocess/<setting>/ is written by a run without -n and reused by `-n 1`
        // (README.md:215-216, 306-307 of the reference); what is missing is dealt on demand.  Products are dealt one epoch
        // ahead and released once consumed, so device memory does not grow with -m.
        std::string cacheDir = "preprocess/" + setting;
        for (auto& ch : cacheDir) if (ch == ' ') ch = '_';
        const bool useCache = !getenv("COGNN_NO_PREPROCESS_CACHE");
        bool cacheDirOk = false;
        if (!noPreprocess && useCache) {
            cacheDirOk = make_dirs(cacheDir);
            if (!cacheDirOk) std::cerr << "warning: cannot create " << cacheDir << ": offline cache not written" << std::endl;
        }
        const int epoch = (original ? 2 : 3) * gp.num_layers;     // getEpochLayerNum
        int64_t reused = 0;
        auto deal_epoch = [&](uint64_t it0) -> bool {
            const int64_t it1 = (int64_t)std::min<uint64_t>(it0 + (uint64_t)epoch, maxIters);
            auto t_om = std::chrono::high_resolution_clock::now();
            if (!noPreprocess) {
                if (cognn_engine_offline(e, (int64_t)it0, it1)) return false;
                if (cacheDirOk && cognn_engine_offline_save(e, cacheDir.c_str()))
                    std::cerr << "warning: offline cache not written: " << cognn_engine_last_error() << std::endl;
            } else {
                int64_t loaded = 0;
                if (cognn_engine_offline_load(e, cacheDir.c_str(), (int64_t)it0, it1, &loaded)) return false;
                reused += loaded;
            }
            if (cognn_engine_sync(e)) return false;
            if (it0 == 0) {
                if (!noPreprocess) print_duration(t_om, "preprocess_OM");
                else for (auto& L : g_logs) fprintf(L.f, "%d Reused %lld offline products from %s\n", L.party, (long long)reused, cacheDir.c_str());
            }
            return true;
        };
        for (auto& L : g_logs) fprintf(L.f, "%d Begin algo kernel iteration\n", L.party);
        for (uint64_t it = 0; it < maxIters; ++it) {
            if (it % (uint64_t)epoch == 0 && !deal_epoch(it)) { std::cerr << cognn_engine_last_error() << std::endl; return -1; }
            for (auto& L : g_logs) fprintf(L.f, "tid-> %lld, iteration-> %lld\n", (long long)L.party, (long long)it);
            auto t_it = std::chrono::high_resolution_clock::now();
            if (cognn_engine_run(e, (int64_t)it, (int64_t)it + 1) || cognn_engine_sync(e)) { std::cerr << cognn_engine_last_error() << std::endl; return -1; }
            const int ei = (int)(it % epoch);
            const bool applyOnly = ei != 0 && ei % gp.num_layers == 0;             // ss_...h:709
            double ph[6];
            if (cognn_engine_get_phase_seconds(e, ph)) { std::cerr << cognn_engine_last_error() << std::endl; return -1; }
            if (!applyOnly) {
                // The tags of the reference's print_duration sites (ss_...h:745,765,802,808,822,856,881,897; parsed by
                // tools/plot/plot_duration_breakdown_and_comm.py:23-46,99), from HIP-event timers.  OEP, ScatterComp, the prefix
                // aggregation and the masked additions are ONE fused CSR launch pair here: its time is reported as
                // "premerging" (the aggregation) and the steps that no longer exist separately as 0.
                print_seconds(ph[0], "PreScatterComp Client");
                print_seconds(ph[0], "PreScatterComp Server");
                print_seconds(0.0, "Scatter_preparation");
                print_seconds(0.0, "Scatter_computation");
                print_seconds(ph[1], "premerging");
                print_seconds(0.0, "premerged_extraction");
                print_seconds(0.0, "Gather_preparation");
                print_seconds(ph[2], "Gather_computation");
            }
            if (ei == gp.num_layers - 1) {                                         // prediction layer: gcn.h:619-632
                for (auto& L : g_logs) {
                    double m[8];
                    if (cognn_engine_get_metrics(e, (int32_t)L.party, m)) { std::cerr << cognn_engine_last_error() << std::endl; return -1; }
                    fprintf(L.f, "--------\n");
                    fprintf(L.f, "cross-entropy-loss = %lf\n", m[5]);
                    // sci::accuracy reports per cent (README.md:226-236: "full set accuracy = 19.188192" = 104 of 542 vertices)
                    fprintf(L.f, "full set accuracy = %lf\n", 100.0 * m[0]);
                    fprintf(L.f, "training set accuracy = %lf\n", 100.0 * m[1]);
                    fprintf(L.f, "border training set accuracy = %lf\n", 100.0 * m[2]);
                    fprintf(L.f, "test set accuracy = %lf\n", 100.0 * m[3]);
                    fprintf(L.f, "border test set accuracy = %lf\n", 100.0 * m[4]);
                    fprintf(L.f, "the number of vertices is %lu, the number of border vertices is %lu\n", (unsigned long)m[6], (unsigned long)m[7]);
                }
            }
            if (!applyOnly) print_seconds(ph[3] + ph[4], "Apply_computation");     // incl. weight averaging (inside ApplyComp, gcn.h:747-802)
            print_duration(t_it, "iteration");
        }
        for (auto& L : g_logs) fprintf(L.f, "%d Finish algo kernel\n", L.party);
#ifndef COGNN_NO_RCCL
        if (xch) {                                               // sendFinish / recvFinish (ss_...h:270-272)
            int64_t rounds = 0, sent = 0, recvd = 0;
            double comm_ms = 0;
            cognn_rccl_exchange_stats(xch, &rounds, &sent, &recvd);
            cognn_rccl_exchange_time(xch, &comm_ms);
            for (auto& L : g_logs)
                fprintf(L.f, "%d exchange rounds %lld, sent %.2fMB, received %.2fMB, %.3f ms on the communication stream\n", L.party, (long long)rounds,
                        sent / 1048576.0, recvd / 1048576.0, comm_ms);
            if (cognn_rccl_exchange_barrier(xch)) std::cerr << cognn_exchange_last_error() << std::endl;
        }
        cognn_engine_destroy(e);
        cognn_rccl_exchange_destroy(xch);
#else
        cognn_engine_destroy(e);
#endif
    } catch (const std::exception& ex) {
        std::cerr << ex.what() << std::endl;
        return -1;
    }
    for (size_t i = 1; i < g_logs.size(); ++i) fclose(g_logs[i].f);
    return 0;
}
