// Host engine, the offline (dealer) phase: product shares of the Beaver products, their on-disk cache.
// (One of the translation units of the engine: engine_internal.h has the shared state and declarations.)
#include "engine_internal.h"

namespace cognn_eng {


// the Beaver product a side runs in GAS iteration `it` (at most one: PreScatter in forward iterations, Apply in backward ones)
bool gemm_of_iteration(cognn_engine* E, Side& s, int64_t it, GemmSpec& g) {
    const IterInfo I = iter_info(E, it);
    if (!I.apply_only && I.fwd) { g = prescatter_spec(E, s, I.layer); return true; }
    if (!I.fwd && (I.e - I.f) % 2 == 0 && I.layer == I.f - 1) {
        g = GemmSpec{s.n, E->hid(), E->lab(), 0, COGNN_OP_AP_GEMM, COGNN_OP_AP_GEMM_TRUNC};
        return true;
    }
    if (!I.fwd && (I.e - I.f) % 2 == 1) { g = wgrad_spec(E, s, I.layer, it); return true; }
    return false;
}

// dealer phase: product shares of every Beaver GEMM in [it0,it1).  The products of one shape (N, K) - all sides' triples of a protocol
// phase, and the same phase of later iterations - go to the grouped MFMA launch (cognn_dealer_gemm_c1_group_u64: operands
// generated in registers, nothing materialised), up to 16 per launch; the weight-gradient triples (transposed left operand,
// K = #vertices) stay on the per-triple path (a K-split MFMA kernel of its own).
void run_offline(cognn_engine* E, int64_t it0, int64_t it1) {
    // recorded epochs deal their product shares inside the recording (on demand), unless the caller replays ONE epoch and keeps them
    if (E->graph_epochs && !E->retain_offline) return;
    if (original(E)) return;                               // original-gcn: every product share is dealt when its product runs
    struct Pending { std::vector<cognn_dealer_job> jobs; };
    std::map<std::pair<int64_t, int64_t>, Pending> groups;  // (N, K) -> jobs waiting for a launch
    auto flush = [&](std::pair<int64_t, int64_t> nk, Pending& p) {
        if (p.jobs.empty()) return;
        BE(cognn_dealer_gemm_c1_group_u64(E->ctx, p.jobs.data(), (int32_t)p.jobs.size(), nk.first, nk.second));
        p.jobs.clear();
    };
    std::vector<cognn_dealer_tn_job> tn_jobs;
    for (int64_t it = it0; it < it1; ++it) {
        const u64 salt_before = E->salt_now;
        set_salt(E, it);
        if (E->graph_epochs && E->salt_now != salt_before)  // (recorded epochs: the salt lives on the device - jobs of two epochs cannot share a launch)
            for (auto& g : groups) flush(g.first, g.second);
        for (auto& s : E->sides) {
            GemmSpec g;
            if (s.p != 1 || !gemm_of_iteration(E, s, it, g) || s.c1.count({it, g.op})) continue;
            cognn_keys k = gemm_keys(E, s, it, g);
            u64* c = c1_alloc(E, g.M * g.N);
            s.c1[{it, g.op}] = Side::C1{c, g.M * g.N};
            if (g.transA == 0 && E->dealer_group && E->be->cognn_dealer_gemm_c1_groupable(g.N, g.K)) {
                Pending& p = groups[{g.N, g.K}];
                cognn_dealer_job j;
                j.C1 = c; j.keys = k; j.M = g.M;
                p.jobs.push_back(j);
                if (p.jobs.size() == 16) flush({g.N, g.K}, p);
            } else if (g.transA != 0 && E->dealer_group) {  // (every side has scratch of its own: the jobs of one iteration share their fills)
                cognn_dealer_tn_job j;
                j.C1 = c; j.keys = k; j.M = g.M; j.N = g.N; j.K = g.K; j.transA = g.transA; j.scratchA = s.scratch; j.scratchB = s.scratch + g.M * g.K;
                tn_jobs.push_back(j);
            } else {
                BE(cognn_dealer_gemm_c1_u64(E->ctx, c, &k, g.M, g.N, g.K, g.transA, s.scratch, s.scratch + g.M * g.K));
            }
        }
        for (size_t b = 0; b < tn_jobs.size(); b += 16)
            BE(cognn_dealer_gemm_c1_tn_group_u64(E->ctx, tn_jobs.data() + b, (int32_t)std::min<size_t>(16, tn_jobs.size() - b)));
        tn_jobs.clear();
    }
    for (auto& g : groups) flush(g.first, g.second);
    set_salt_value(E, 0);
}

// identifies the run a cached product share belongs to: parties, ranks, variant, dimensions, graph size, rows of the owner
u64 run_fingerprint(cognn_engine* E, const Side& s) {
    u64 h = 0xcbf29ce484222325ull;
    auto mix = [&](u64 v) { for (int b = 0; b < 8; ++b) { h ^= (v >> (8 * b)) & 0xff; h *= 0x100000001b3ull; } };
    mix((u64)E->k); mix((u64)E->world); mix((u64)E->rank); mix((u64)E->cfg.variant);
    mix((u64)E->in()); mix((u64)E->hid()); mix((u64)E->lab());
    mix((u64)E->G.num_edges); mix((u64)E->G.row_of_vid.size()); mix((u64)s.owner); mix((u64)s.n);
    return h;
}

}  // namespace cognn_eng

using namespace cognn_eng;

extern "C" {

int cognn_engine_offline(cognn_engine* E, int64_t it0, int64_t it1) {
    return guard([&] {
        if (!E || !E->started) throw EngineError("cognn_engine_offline: engine not started");
        run_offline(E, it0, it1);
    });
}

static std::string c1_path(cognn_engine* E, const char* dir, const Side& s, int64_t it, int op) {
    char buf[512];
    snprintf(buf, sizeof(buf), "%s/c1_r%d_o%d_i%lld_op%d.bin", dir, E->rank, s.owner, (long long)it, op);
    return buf;
}

int cognn_engine_offline_save(cognn_engine* E, const char* dir) {
    return guard([&] {
        if (!E || !dir) throw EngineError("cognn_engine_offline_save: bad arguments");
        for (auto& s : E->sides)
            for (auto& kv : s.c1) {
                GemmSpec g;
                if (!gemm_of_iteration(E, s, kv.first.first, g) || g.op != kv.first.second || g.M * g.N != kv.second.elems)
                    throw EngineError("cognn_engine_offline_save: inconsistent product share table");
                const int64_t elems = kv.second.elems;
                std::vector<u64> host((size_t)elems);
                BE(cognn_memcpy_d2h(E->ctx, host.data(), kv.second.ptr, (size_t)elems * 8));
                const std::string path = c1_path(E, dir, s, kv.first.first, kv.first.second);
                FILE* f = fopen(path.c_str(), "wb");
                if (!f) throw EngineError("cognn_engine_offline_save: cannot write " + path);
                const u64 hdr[7] = {kC1Magic, E->cfg.seed, (u64)g.M, (u64)g.N, (u64)g.K, (u64)g.transA, run_fingerprint(E, s)};
                const bool ok = fwrite(hdr, 8, 7, f) == 7 && fwrite(host.data(), 8, (size_t)elems, f) == (size_t)elems;
                fclose(f);
                if (!ok) throw EngineError("cognn_engine_offline_save: short write to " + path);
            }
    });
}

int cognn_engine_offline_discard(cognn_engine* E, int64_t it0, int64_t it1, int64_t* discarded) {
    return guard([&] {
        if (!E) throw EngineError("cognn_engine_offline_discard: null engine");
        int64_t n = 0;
        for (auto& s : E->sides)
            for (auto f = s.c1.begin(); f != s.c1.end();) {
                if (f->first.first < it0 || f->first.first >= it1) { ++f; continue; }
                E->c1_pool[f->second.elems].push_back(f->second.ptr);   // (one stream: a later deal into the buffer is ordered after this one)
                f = s.c1.erase(f);
                ++n;
            }
        if (discarded) *discarded = n;
    });
}

int cognn_engine_offline_load(cognn_engine* E, const char* dir, int64_t it0, int64_t it1, int64_t* loaded) {
    return guard([&] {
        if (!E || !dir || !E->started) throw EngineError("cognn_engine_offline_load: bad arguments or engine not started");
        int64_t n = 0;
        for (auto& s : E->sides) {
            if (s.p != 1) continue;
            for (int64_t it = it0; it < it1; ++it) {
                GemmSpec g;
                if (!gemm_of_iteration(E, s, it, g) || s.c1.count({it, g.op})) continue;
                FILE* f = fopen(c1_path(E, dir, s, it, g.op).c_str(), "rb");
                if (!f) continue;
                u64 hdr[7];
                // only a file written for exactly this product of exactly this run is accepted; anything else is dealt on demand
                const bool match = fread(hdr, 8, 7, f) == 7 && hdr[0] == kC1Magic && hdr[1] == E->cfg.seed && hdr[2] == (u64)g.M &&
                                   hdr[3] == (u64)g.N && hdr[4] == (u64)g.K && hdr[5] == (u64)g.transA && hdr[6] == run_fingerprint(E, s);
                if (match) {
                    std::vector<u64> host((size_t)(g.M * g.N));
                    if (fread(host.data(), 8, host.size(), f) == host.size() && fgetc(f) == EOF) {
                        u64* c = c1_alloc(E, g.M * g.N);
                        BE(cognn_memcpy_h2d(E->ctx, c, host.data(), host.size() * 8));
                        s.c1[{it, g.op}] = Side::C1{c, g.M * g.N};
                        ++n;
                    }
                }
                fclose(f);
            }
        }
        if (loaded) *loaded = n;
    });
}


}  // extern "C"
