"""N>1 path on CPU: world_size-2 (and 3) engine runs over torch.distributed/gloo, the same engine.cpp and the
same cognn_amd/dist.py exchange callback the GPU box uses with RCCL, only the arithmetic backend is the plain-C++
reference one (oracle/cpu_backend.cpp).  Every rank's shares are compared bit for bit with the oracle."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import cognn_oracle as co

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def _build_cpu_engine():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _run(cfg, world, tmp_path):
    cfg = dict(cfg, out=str(tmp_path / "shares"))
    port = _free_port()
    procs = []
    if cfg.get("inproc"):                                  # the ranks as threads of ONE process over a mailbox transport (tests/inproc_worker.py)
        env = dict(os.environ, OMP_NUM_THREADS="1", **cfg.get("env", {}))
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "inproc_worker.py"), json.dumps(dict(cfg, world=world))], env=env,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        world_procs = 0
    else:
        world_procs = world
    for r in range(world_procs):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "multirank_worker.py"), json.dumps(cfg)], env=env))
    for p in procs:
        assert p.wait(timeout=300) == 0
    got = {}
    for r in range(world):
        with np.load(cfg["out"] + ".rank%d.npz" % r) as z:
            for key in z.files:
                if key in ("hostile_rounds", "exchange_stats"):
                    got.setdefault(key, []).append(z[key])
                    continue
                assert key not in got
                got[key] = z[key]
    return got


def _check(cfg, world, tmp_path):
    k, V = cfg["k"], cfg["V"]
    src, dst = co.synth_graph(V, cfg["Eu"], cfg["gseed"])
    part = [v % k for v in range(V)]
    feats, labels = co.synth_features(V, cfg["in"], cfg["lab"], cfg["gseed"] + 1, density=cfg.get("density", 0.2))
    p = co.GnnParam(**dict(dict(num_labels=cfg["lab"], input_dim=cfg["in"], hidden_dim=cfg["hid"], num_samples=V, learning_rate=0.5), **cfg.get("param", {})))
    if cfg["variant"] == "original-gcn":
        import original_gcn
        o = original_gcn.OriginalOracleEngine(k, src, dst, part, feats, labels, p, seed=cfg["seed"])
    else:
        o = co.OracleEngine(k, src, dst, part, feats, labels, p, seed=cfg["seed"], variant=cfg["variant"])
    got = _run(cfg, world, tmp_path)
    for it in range(cfg["iters"]):
        o.iteration(it)
        if ("it%d_o0_s0" % it) not in got:                 # whole_epochs: only the last iteration of every call was saved
            continue
        for P in range(k):
            a, b = o.shares(P)
            assert np.array_equal(got["it%d_o%d_s0" % (it, P)], a), (it, P)
            assert np.array_equal(got["it%d_o%d_s1" % (it, P)], b), (it, P)
            c = (P + 1) % k
            for l in range(2):
                assert np.array_equal(got["it%d_o%d_s0_w%d" % (it, P, l)], o.states[P].localWeight[l]), (it, P, l)
                assert np.array_equal(got["it%d_o%d_s1_w%d" % (it, P, l)], o.states[c].remoteWeight[l]), (it, P, l)


BASE = dict(V=48, Eu=120, gseed=3, seed=17, hid=6, lab=4, variant="optimize-gcn", iters=6)
BASE["in"] = 10


def test_two_parties_two_ranks_training(tmp_path):
    _check(dict(BASE, k=2), 2, tmp_path)


def test_four_parties_two_ranks_training(tmp_path):
    _check(dict(BASE, k=4), 2, tmp_path)


def test_four_parties_two_ranks_training_blocking_exchange(tmp_path):
    _check(dict(BASE, k=4, blocking_exchange=True), 2, tmp_path)


def test_four_parties_two_ranks_training_exchanged_openings(tmp_path):
    """COGNN_OPT_PUBLIC_OPENINGS off: the openings that follow a truncation travel as two shares (two more rounds per layer)."""
    _check(dict(BASE, k=4, exchanged_openings=True), 2, tmp_path)


def test_four_parties_two_ranks_training_no_pair_chains(tmp_path):
    """co-located share-holders through the per-side kernels too (public openings between them, in-process hand-off)."""
    _check(dict(BASE, k=4, pair_fusion=False), 2, tmp_path)


def test_three_parties_three_ranks_training(tmp_path):
    _check(dict(BASE, k=3), 3, tmp_path)


def test_four_parties_four_ranks_inference(tmp_path):
    _check(dict(BASE, k=4, variant="optimize-gcn-inference", iters=2), 4, tmp_path)


def test_whole_epochs_per_call(tmp_path):
    """run(it0, it0 + 6): the cross-iteration paths (deferred ReLU' selection of the co-located pairs, the opening the ReLU leaves
    for the next product) on one rank and on two."""
    _check(dict(BASE, k=3, iters=12, whole_epochs=True), 1, tmp_path)
    _check(dict(BASE, k=4, iters=12, whole_epochs=True), 2, tmp_path)


@pytest.mark.parametrize("k,world,variant,iters,extra", [(2, 2, "optimize-gcn", 6, {}), (4, 2, "optimize-gcn", 6, {}), (3, 3, "optimize-gcn", 12, {}),
                                                          (4, 4, "optimize-gcn-inference", 2, {}), (4, 2, "optimize-gcn", 12, {"whole_epochs": True}),
                                                          (4, 2, "optimize-gcn", 6, {"pair_fusion": False}), (6, 3, "optimize-gcn", 6, {"V": 61, "Eu": 200})])
def test_vertex_set_placement(tmp_path, k, world, variant, iters, extra):
    """COGNN_PLACE_VERTEX_SET: every rank holds both shares of its parties' vertex sets - all two-party steps are pair chains, only
    the Gather's replicas / partial sums and the weight average travel - and ends in the oracle's shares and weights like the
    party placement."""
    _check(dict(BASE, k=k, variant=variant, iters=iters, placement="vertex-set", **extra), world, tmp_path)


def test_single_rank_host_logic_three_parties(tmp_path):
    """world == 1 through the same worker: the engine's co-located path (in-device hand-off) on the CPU backend."""
    _check(dict(BASE, k=3, iters=12), 1, tmp_path)


@pytest.mark.parametrize("k,extra", [(2, {}), (3, {}), (4, {}), (2, {"pair_fusion": 0}), (3, {"V": 9, "Eu": 5}), (5, {"V": 61, "Eu": 300, "hid": 5, "lab": 3})])
def test_original_gcn_single_process(tmp_path, k, extra):
    """BASELINE config 1's kernel (original-gcn: per-edge two-normaliser Scatter, aggregate-then-transform, 4 GAS iterations per
    epoch, weight average after both backward iterations) on the engine - world 1, plain-C++ backend - against
    oracle/original_gcn.py after every GAS iteration of two epochs: vertex shares (incl. the empty tensor after the first layer's
    backward Apply) and both weight shares of every party; sparse graphs with dummy self entries, multi-edges, k up to 5."""
    _check(dict(BASE, k=k, variant="original-gcn", iters=8, **extra), 1, tmp_path)


@pytest.mark.parametrize("k,world,extra", [(2, 2, {}), (3, 3, {}), (4, 2, {}), (4, 4, {}), (4, 4, {"hostile": 1}), (6, 3, {"V": 61, "Eu": 200}),
                                            (3, 3, {"V": 9, "Eu": 5}), (5, 5, {"V": 61, "Eu": 300, "hid": 5, "lab": 3, "inproc": True}),
                                            (4, 2, {"packed_openings": True}), (4, 2, {"V": 71, "Eu": 260, "in": 33, "hid": 5, "lab": 7})])
def test_original_gcn_across_ranks(tmp_path, k, world, extra):
    """The unoptimised kernel with its parties on different ranks (the reference's deployment of it: k processes): the client and
    the server of a Scatter instance exchange the openings of the two per-edge scales, the client's results for destination owners
    whose co-party lives elsewhere travel pre-summed - same shares and weights as the oracle after every GAS iteration of two
    epochs, also over the hostile transport and with sparse graphs (dummy self entries, parties without edges to each other)."""
    _check(dict(BASE, k=k, variant="original-gcn", iters=8, **extra), world, tmp_path)


@pytest.mark.parametrize("seed", range(4 * int(os.environ.get("COGNN_FUZZ_SCALE", "1"))))
def test_original_gcn_across_ranks_random_configuration(tmp_path, seed):
    """Random party / rank counts, graph sizes (down to fewer vertices than parties have neighbours: empty instances, dummy self
    entries everywhere) and odd widths for the unoptimised kernel across ranks; every second one as threads of one process."""
    rng = np.random.default_rng(7700 + seed)
    world = int(rng.choice([2, 3, 4]))
    k = world * int(rng.integers(1, 3))
    V = int(rng.integers(k, 160))
    cfg = dict(BASE, k=k, V=V, Eu=int(min(V * (V - 1) // 2, rng.integers(1, 3 * V + 1))), gseed=int(rng.integers(1, 1000)),
               seed=int(rng.integers(1, 1 << 30)), hid=int(rng.choice([3, 8, 16])), lab=int(rng.choice([2, 5, 7])),
               variant="original-gcn", iters=8, inproc=bool(seed % 2))
    cfg["in"] = int(rng.choice([5, 16, 33]))
    _check(cfg, world, tmp_path)


@pytest.mark.parametrize("seed", [1, 2])
@pytest.mark.parametrize("k,world,variant,iters,extra", [(4, 2, "optimize-gcn", 6, {}), (4, 4, "optimize-gcn", 6, {}), (4, 4, "optimize-gcn-inference", 2, {}),
                                                        (4, 2, "optimize-gcn", 6, {"exchanged_openings": True}), (6, 2, "optimize-gcn", 12, {"whole_epochs": True}),
                                                        (4, 4, "optimize-gcn", 6, {"placement": "vertex-set"}), (6, 2, "optimize-gcn", 12, {"placement": "vertex-set", "whole_epochs": True})])
def test_hostile_transport(tmp_path, seed, k, world, variant, iters, extra):
    """Late reads of the outboxes, poisoned inboxes until wait(), rounds and messages completed in shuffled order
    (tests/hostile_transport.py): every exchange_wait the engine owes - before it overwrites a buffer it handed to a round,
    before it reads a received one, with two rounds in flight during the message passing - is exercised at world 2 and 4."""
    _check(dict(BASE, k=k, variant=variant, iters=iters, hostile=seed, **extra), world, tmp_path)


@pytest.mark.parametrize("chunks,k,world,variant,iters,extra", [
    (2, 4, 2, "optimize-gcn", 6, {}), (3, 4, 4, "optimize-gcn", 6, {}), (4, 4, 4, "optimize-gcn-inference", 2, {}),
    (3, 4, 2, "optimize-gcn", 6, {"exchanged_openings": True}), (2, 4, 2, "optimize-gcn", 6, {"pair_fusion": 0}),
    (8, 3, 3, "optimize-gcn", 6, {}), (2, 4, 2, "optimize-gcn", 6, {"per_round": False}), (3, 4, 2, "optimize-gcn", 6, {"blocking_exchange": True}),
    (3, 4, 2, "optimize-gcn", 6, {"placement": "vertex-set"})])
def test_chunked_exchange_pipeline(tmp_path, chunks, k, world, variant, iters, extra):
    """COGNN_OPT_EXCHANGE_CHUNKS: the open -> exchange -> close steps of the cross-rank sides in row chunks (chunk c's round in
    flight behind chunk c+1's kernels, closes waiting round by round) give the oracle's shares bit for bit - over gloo with the
    per-round wait, with the wait-everything fallback (per_round False), with the blocking callback, with more chunks than some
    tensors have element pairs (V = 48: 12-16 rows per party)."""
    _check(dict(BASE, k=k, variant=variant, iters=iters, chunks=chunks, **extra), world, tmp_path)


@pytest.mark.parametrize("seed,chunks,k,world,variant,iters", [(1, 2, 4, 2, "optimize-gcn", 6), (2, 3, 4, 4, "optimize-gcn", 6),
                                                              (3, 4, 4, 2, "optimize-gcn-inference", 2), (4, 3, 6, 2, "optimize-gcn", 12)])
def test_chunked_exchange_pipeline_hostile_transport(tmp_path, seed, chunks, k, world, variant, iters):
    """The chunked pipeline over the hostile transport with per-round completion: a close that runs before ITS round has been
    waited for reads poison, a chunk that is re-opened before its round completed ships the wrong bytes, and the rounds of the
    later chunks really are still undelivered while an earlier chunk closes (left_inflight > 0)."""
    got_cfg = dict(BASE, k=k, variant=variant, iters=iters, chunks=chunks, hostile=seed)
    _check(got_cfg, world, tmp_path)
    left = 0
    for r in range(world):
        with np.load(str(tmp_path / "shares") + ".rank%d.npz" % r) as z:
            left = max(left, int(z["hostile_rounds"][2]))
    assert left > 0


def test_hostile_transport_detects_a_missing_wait(tmp_path):
    """The transport's own sensitivity: when the first wait() completes nothing (= the engine consumed a round it never waited
    for), the poisoned inboxes reach the arithmetic and the shares no longer match the oracle."""
    with pytest.raises(AssertionError):
        _check(dict(BASE, k=4, variant="optimize-gcn-inference", iters=2, hostile=1, hostile_skip_waits=1), 2, tmp_path)


def test_hostile_transport_detects_a_missing_round_wait(tmp_path):
    """... and with the chunked pipeline: a per-round wait that completes nothing leaves that chunk's inbox poisoned."""
    with pytest.raises(AssertionError):
        _check(dict(BASE, k=4, variant="optimize-gcn-inference", iters=2, hostile=1, hostile_skip_waits=2, chunks=3), 2, tmp_path)


# ---- world 8, k = 8, one party per rank: the north-star layout (the reference's k x k mesh, include/engine.h:157-201, and its
# per-peer threads, ss_...h:702-704,926-928) ----
W8 = dict(BASE, k=8, V=96, Eu=260)


@pytest.mark.parametrize("variant,iters,extra", [("optimize-gcn", 12, {}), ("optimize-gcn-inference", 2, {"placement": "vertex-set"}),
                                                 ("optimize-gcn-inference", 2, {"inproc": True}), ("optimize-gcn", 12, {"placement": "vertex-set", "inproc": True}),
                                                 ("optimize-gcn", 12, {"whole_epochs": True, "inproc": True}), ("optimize-gcn", 6, {"exchanged_openings": True, "inproc": True}),
                                                 ("optimize-gcn", 6, {"chunks": 3})])
def test_eight_parties_eight_ranks(tmp_path, variant, iters, extra):
    """One party per rank at world 8 over gloo (plain-C++ backend): every owner / co-party pair crosses a rank boundary, every
    owner's co-share is replicated to six other ranks, partial sums travel between all 56 ordered rank pairs, the weight average
    gathers from six ranks onto ranks 0 and 1 (gcn.h:747-802) - two training epochs / an inference pass, both placements, every
    party's two shares and weight shares against the oracle after every GAS iteration.  (Three cases as eight gloo processes, the
    others as eight threads of one process over the mailbox transport - same engine code, a fraction of the start-up time.)"""
    _check(dict(W8, variant=variant, iters=iters, **extra), 8, tmp_path)


@pytest.mark.parametrize("variant,iters,extra", [("optimize-gcn", 12, {}), ("optimize-gcn", 12, {"placement": "vertex-set", "whole_epochs": True}),
                                                 ("optimize-gcn-inference", 2, {"chunks": 2})])
def test_eight_parties_eight_ranks_in_one_process(tmp_path, variant, iters, extra):
    """The same layout with the eight ranks as threads of ONE process (tests/inproc_worker.py): eight engines and contexts side by
    side in one address space, two epochs - nothing process-global (the epoch salt, error strings) may leak from one to another."""
    _check(dict(W8, variant=variant, iters=iters, inproc=True, **extra), 8, tmp_path)


@pytest.mark.parametrize("seed,variant,iters,extra", [(1, "optimize-gcn", 6, {}), (2, "optimize-gcn-inference", 2, {}),
                                                      (3, "optimize-gcn", 6, {"placement": "vertex-set"}), (4, "optimize-gcn", 6, {"chunks": 2})])
def test_eight_parties_eight_ranks_hostile_transport(tmp_path, seed, variant, iters, extra):
    """The same layout over the hostile asynchronous transport (late reads, poisoned inboxes, shuffled completion)."""
    _check(dict(W8, variant=variant, iters=iters, hostile=seed, **extra), 8, tmp_path)


def test_mismatched_configuration_fails_fast(tmp_path):
    """Two ranks launched with different placements: start() swaps a configuration fingerprint with every other rank, so both
    stop with an error instead of hanging later in mismatched send / receive lists."""
    cfg = dict(BASE, k=4, placement_by_rank=["party", "vertex-set"], out=str(tmp_path / "x"))
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "multirank_worker.py"), json.dumps(cfg)], env=env,
                                      stdout=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=120)[0] for p in procs]
    assert [p.returncode for p in procs] == [7, 7], outs
    assert all("runs a different configuration" in o for o in outs), outs


def test_engines_starting_side_by_side_get_the_reference_weights(tmp_path):
    """Four engines started at the same moment in one process (threads): the Glorot weights come from libc's srand(42) / rand()
    (gcn.h:838-852), ONE generator state per process - the engine runs each matrix's sequence under a lock, so every rank starts from
    the weights the oracle has (wide matrices: tens of thousands of rand() calls per matrix, which interleaved before the lock)."""
    cfg = dict(BASE, k=4, V=64, Eu=150, hid=64, lab=8, variant="optimize-gcn-inference", iters=2, inproc=True)
    cfg["in"] = 400
    _check(cfg, 4, tmp_path)


@pytest.mark.parametrize("k,world,variant,iters,extra", [
    (4, 2, "optimize-gcn", 12, {}), (4, 4, "optimize-gcn-inference", 2, {}), (8, 8, "optimize-gcn", 6, {"V": 96, "Eu": 260}),
    (8, 8, "optimize-gcn-inference", 2, {"V": 96, "Eu": 260, "inproc": True}), (4, 2, "optimize-gcn", 6, {"chunks": 3}), (4, 2, "optimize-gcn", 6, {"exchanged_openings": True}),
    (4, 2, "optimize-gcn", 6, {"blocking_exchange": True}), (4, 4, "optimize-gcn", 6, {"hostile": 5}), (8, 8, "optimize-gcn", 6, {"V": 96, "Eu": 260, "hostile": 6, "chunks": 2}),
    (6, 3, "optimize-gcn", 6, {"V": 61, "Eu": 200, "pair_fusion": False})])
def test_packed_openings(tmp_path, k, world, variant, iters, extra):
    """COGNN_OPT_PACKED_OPENINGS: the opened shares of every truncation and of the ReLU's masked product cross ranks as 6 bytes per
    element (both parties form the opened value from the top 48 bits of the two shares, so the low 16 never matter): the same shares as
    the oracle bit for bit - over gloo, the blocking callback, the chunked pipeline, the hostile transport (inboxes poisoned until the
    wait: a restore before the round's completion would unpack poison), worlds 2 / 3 / 4 / 8."""
    _check(dict(BASE, k=k, variant=variant, iters=iters, packed_openings=True, **extra), world, tmp_path)
