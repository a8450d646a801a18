"""Host-side mirror of the reference's harness entry (algo_kernels/common_harness/harness.cpp:50-212)
over the C++ engine: load a partitioned graph, hand every hosted party its vertex data, start, run
GAS iterations.  All compute happens in libcognn_hip.so."""
import ctypes

import numpy as np

from . import capi
from .engine_api import EngineConfig

VARIANTS = {"optimize-gcn": 0, "optimize-gcn-inference": 1, "original-gcn": 2}     # original-gcn across ranks: party placement (cognn_engine.h)


class GnnParam:
    """GNNParam (include/task/task.h:78-170)."""

    def __init__(self, num_layers=2, num_labels=7, input_dim=1433, hidden_dim=16, num_samples=2708, num_edges=0,
                 learning_rate=0.5, train_ratio=0.2, val_ratio=0.2, test_ratio=0.6):
        self.num_layers = num_layers; self.num_labels = num_labels; self.input_dim = input_dim
        self.hidden_dim = hidden_dim; self.num_samples = num_samples; self.num_edges = num_edges
        self.learning_rate = learning_rate; self.train_ratio = train_ratio
        self.val_ratio = val_ratio; self.test_ratio = test_ratio

    @staticmethod
    def read_config(path):
        """`key : value` lines (task.h:106-169); parsing stops at the first malformed or unknown key."""
        g = GnnParam()
        toks = open(path).read().split()
        i = 0
        while i + 2 < len(toks):
            key, colon, val = toks[i], toks[i + 1], toks[i + 2]
            if colon != ":":
                break
            if key in ("num_layers", "num_labels", "input_dim", "hidden_dim", "num_samples", "num_edges"):
                setattr(g, key, int(val))
            elif key in ("learning_rate", "train_ratio", "val_ratio", "test_ratio"):
                setattr(g, key, float(val))
            else:
                break
            i += 3
        return g


def _check(rc):
    if rc != 0:
        raise capi.CognnError(capi.load().cognn_engine_last_error().decode())


class Engine:
    def __init__(self, num_parties, src, dst, part, param, seed=0xC06A11, variant="optimize-gcn", rank=0, world=1,
                 device=0, stream=None, undirected=False, verbose=False, placement="party"):
        """placement (world > 1): "party" - a rank holds what its parties hold (own shares + the co-shares of the previous party's
        vertices); "vertex-set" - a rank holds both shares of its parties' vertex sets (cognn_engine_config::placement)."""
        self.lib = capi.load()
        self.k = num_parties; self.rank = rank; self.world = world; self.param = param
        if stream is None:
            import torch
            stream = torch.cuda.current_stream(device).cuda_stream if torch.cuda.is_available() else 0
        cfg = EngineConfig(num_parties, rank, world, VARIANTS[variant], param.num_layers, param.num_labels,
                           param.input_dim, param.hidden_dim, param.learning_rate, param.train_ratio, param.val_ratio,
                           param.test_ratio, seed, device, ctypes.c_void_p(stream), int(undirected), int(verbose),
                           {"party": 0, "vertex-set": 1}[placement])
        src = np.ascontiguousarray(src, dtype=np.int64); dst = np.ascontiguousarray(dst, dtype=np.int64)
        part = np.ascontiguousarray(part, dtype=np.int32)
        h = ctypes.c_void_p()
        _check(self.lib.cognn_engine_create(ctypes.byref(cfg), len(part), len(src), src.ctypes.data, dst.ctypes.data,
                                            part.ctypes.data, ctypes.byref(h)))
        self.h = h
        self._keep = []
        m = num_parties // world
        self.hosted = list(range(rank * m, (rank + 1) * m))

    def close(self):
        if self.h:
            self.lib.cognn_engine_destroy(self.h)
            self.h = None

    def party_rows(self, party):
        n = ctypes.c_int64()
        _check(self.lib.cognn_engine_party_rows(self.h, party, ctypes.byref(n)))
        return n.value

    def party_vids(self, party):
        out = np.empty(self.party_rows(party), dtype=np.int64)
        _check(self.lib.cognn_engine_party_vids(self.h, party, out.ctypes.data))
        return out

    def party_degrees(self, party):
        n = self.party_rows(party)
        t = np.empty(n, dtype=np.int64); i = np.empty(n, dtype=np.int64); b = np.empty(n, dtype=np.uint8)
        _check(self.lib.cognn_engine_party_degrees(self.h, party, t.ctypes.data, i.ctypes.data, b.ctypes.data))
        return t, i, b

    def set_party_data(self, party, features, labels):
        f = np.ascontiguousarray(features, dtype=np.float64); l = np.ascontiguousarray(labels, dtype=np.int32)
        _check(self.lib.cognn_engine_set_party_data(self.h, party, f.ctypes.data, l.ctypes.data))

    def set_global_data(self, features, labels):
        """Convenience: slice global [V x in] features / labels by each hosted party's vids."""
        for p in self.hosted:
            vids = self.party_vids(p)
            self.set_party_data(p, np.asarray(features)[vids], np.asarray(labels)[vids])

    def set_weights(self, w0, w1):
        a = np.ascontiguousarray(w0, dtype=np.float64); b = np.ascontiguousarray(w1, dtype=np.float64)
        _check(self.lib.cognn_engine_set_weights(self.h, a.ctypes.data, b.ctypes.data))

    def set_exchange(self, fn):
        """fn: the callback of cognn_amd.dist.make_exchange, or the (begin, wait[, wait_round]) tuple of make_exchange_async."""
        self._xfn = fn                                     # keep the ctypes callbacks alive
        if isinstance(fn, tuple) and len(fn) == 3:
            _check(self.lib.cognn_engine_set_exchange_async2(self.h, fn[0], fn[1], fn[2], None))
        elif isinstance(fn, tuple):
            _check(self.lib.cognn_engine_set_exchange_async(self.h, fn[0], fn[1], None))
        else:
            _check(self.lib.cognn_engine_set_exchange(self.h, fn, None))

    def start(self):
        _check(self.lib.cognn_engine_start(self.h))

    def offline(self, it0, it1):
        _check(self.lib.cognn_engine_offline(self.h, it0, it1))

    def offline_discard(self, it0, it1):
        """Returns the unconsumed product shares of iterations [it0, it1) to the buffer pool; how many there were."""
        n = ctypes.c_int64()
        _check(self.lib.cognn_engine_offline_discard(self.h, it0, it1, ctypes.byref(n)))
        return n.value

    def offline_save(self, directory):
        _check(self.lib.cognn_engine_offline_save(self.h, str(directory).encode()))

    def offline_load(self, directory, it0, it1):
        n = ctypes.c_int64()
        _check(self.lib.cognn_engine_offline_load(self.h, str(directory).encode(), it0, it1, ctypes.byref(n)))
        return n.value

    def run(self, it0, it1):
        _check(self.lib.cognn_engine_run(self.h, it0, it1))

    def sync(self):
        _check(self.lib.cognn_engine_sync(self.h))

    def retain_offline(self, on=True):
        """Keep dealt product shares after use (replaying the same iterations, e.g. bench.py); default: release them."""
        _check(self.lib.cognn_engine_set_option(self.h, 1, int(on)))

    def pair_fusion(self, on=True):
        """Co-located share-holders run their two-party steps as pair chains (default) or through the per-side kernels."""
        _check(self.lib.cognn_engine_set_option(self.h, 2, int(on)))

    def forward_only(self, on=True):
        """Promise that no backward iteration follows (inference): stores that only the backward pass reads are skipped."""
        _check(self.lib.cognn_engine_set_option(self.h, 3, int(on)))

    def dealer_streams(self, on=True):
        """Measurement mode (needs retain_offline): the co-located pairs' dealer values are materialised in HBM at first use and
        read from there instead of being regenerated from the counter PRNG (cognn_engine.h: COGNN_OPT_DEALER_STREAMS)."""
        _check(self.lib.cognn_engine_set_option(self.h, 5, int(on)))

    def dealer_minimal(self, on=True):
        """The corrections-only dealt form (COGNN_OPT_DEALER_STREAMS = 2): every party regenerates what it derives from its own seed and reads
        from HBM only what a PRG-compressed dealer must send it - c_1, r_1, r'_1 and the ReLU's published g."""
        _check(self.lib.cognn_engine_set_option(self.h, 5, 2 if on else 0))

    def graph_epochs(self, on=True):
        """Whole epochs per run() call are recorded once (hipGraph) and replayed (cognn_engine.h: COGNN_OPT_GRAPH_EPOCHS)."""
        _check(self.lib.cognn_engine_set_option(self.h, 6, int(on)))

    def exchange_chunks(self, chunks):
        """The element-wise open -> exchange -> close steps of share-holders whose peer is on another rank run in `chunks` row
        chunks, each chunk's messages in flight behind the next chunk's kernels (cognn_engine.h: COGNN_OPT_EXCHANGE_CHUNKS)."""
        _check(self.lib.cognn_engine_set_option(self.h, 7, int(chunks)))

    def packed_openings(self, on=True):
        """Opened truncation / ReLU-product shares cross ranks as 6 bytes per element (cognn_engine.h: COGNN_OPT_PACKED_OPENINGS)."""
        _check(self.lib.cognn_engine_set_option(self.h, 8, int(on)))

    def public_openings(self, on=True):
        """Share-holders outside pair chains derive the opening that follows a truncation themselves (default) instead of
        exchanging it as two shares (cognn_engine.h: COGNN_OPT_PUBLIC_OPENINGS)."""
        _check(self.lib.cognn_engine_set_option(self.h, 4, int(on)))

    def phase_seconds(self):
        """Device time per phase of the last iteration (engine created with verbose=True), see cognn_engine_get_phase_seconds."""
        out = np.zeros(6, dtype=np.float64)
        _check(self.lib.cognn_engine_get_phase_seconds(self.h, out.ctypes.data))
        return dict(zip(["prescatter", "message_passing", "gather_scale", "apply", "weight_average", "rounds"], out.tolist()))

    def memory(self):
        n = ctypes.c_int64(); b = ctypes.c_int64()
        _check(self.lib.cognn_engine_get_memory(self.h, ctypes.byref(n), ctypes.byref(b)))
        return n.value, b.value

    def shares(self, owner, side):
        r = ctypes.c_int64(); c = ctypes.c_int64()
        _check(self.lib.cognn_engine_get_shares(self.h, owner, side, None, ctypes.byref(r), ctypes.byref(c)))
        out = np.zeros((r.value, c.value), dtype=np.uint64)
        if out.size:
            _check(self.lib.cognn_engine_get_shares(self.h, owner, side, out.ctypes.data, None, None))
        return out

    def weight(self, owner, side, layer):
        p = self.param
        shape = (p.input_dim, p.hidden_dim) if layer == 0 else (p.hidden_dim, p.num_labels)
        out = np.zeros(shape, dtype=np.uint64)
        _check(self.lib.cognn_engine_get_weight(self.h, owner, side, layer, out.ctypes.data))
        return out

    def metrics(self, party):
        out = np.zeros(8, dtype=np.float64)
        _check(self.lib.cognn_engine_get_metrics(self.h, party, out.ctypes.data))
        keys = ["full", "train", "border_train", "test", "border_test", "loss", "n", "n_border"]
        return dict(zip(keys, out.tolist()))

    def enable_timing(self, on=True):
        _check(self.lib.cognn_engine_enable_timing(self.h, int(on)))

    def timing(self, kind):
        n = ctypes.c_int64(); ms = ctypes.c_double(); algo = ctypes.c_double()
        _check(self.lib.cognn_engine_get_timing(self.h, kind, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(algo)))
        return n.value, ms.value, algo.value

    def workload(self):
        out = np.zeros(6, dtype=np.int64)
        _check(self.lib.cognn_engine_get_workload(self.h, out.ctypes.data))
        return dict(zip(["agg_edges", "agg_rows", "part_edges", "part_rows", "num_edges", "table_rows"], out.tolist()))


def print_metrics(m):
    """The reference's log lines (gcn.h:620-632), parsed by tools/plot/plot_accuracy.py:17-24.  Accuracies are printed in
    per cent like the reference's (README.md:226-236); the metrics dictionary holds fractions."""
    print("--------")
    print("cross-entropy-loss = %f" % m["loss"])
    print("full set accuracy = %f" % (100.0 * m["full"]))
    print("training set accuracy = %f" % (100.0 * m["train"]))
    print("border training set accuracy = %f" % (100.0 * m["border_train"]))
    print("test set accuracy = %f" % (100.0 * m["test"]))
    print("border test set accuracy = %f" % (100.0 * m["border_test"]))
    print("the number of vertices is %d, the number of border vertices is %d" % (int(m["n"]), int(m["n_border"])))
