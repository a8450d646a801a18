"""N>1 path on the HIP backend: 2 and 4 engine ranks share the one GPU of the box (every rank runs the real kernels on
cuda:0), messages travel as host copies over gloo (cognn_amd/dist.py host_staged) because two RCCL ranks cannot share a
device.  Exercises what the CPU multi-rank tests cannot: the rank-aware share-table layout, the partial-sum launch and the
inbox/outbox segments with the HIP kernels.  Every rank's shares are compared bit for bit with the oracle."""
import os

import numpy as np
import pytest

from test_multirank_cpu import BASE, _check

pytestmark = pytest.mark.gpu


def test_four_parties_two_ranks_training_hip(tmp_path):
    _check(dict(BASE, k=4, backend="hip"), 2, tmp_path)


def test_four_parties_four_ranks_inference_hip(tmp_path):
    _check(dict(BASE, k=4, variant="optimize-gcn-inference", iters=2, backend="hip"), 4, tmp_path)


def test_three_parties_three_ranks_training_hip(tmp_path):
    _check(dict(BASE, k=3, backend="hip"), 3, tmp_path)


def test_eight_parties_two_ranks_wide_rows_hip(tmp_path):
    # wider rows and more vertices: 16-byte gather path, MFMA GEMM shapes, k >= 3 replication across ranks
    cfg = dict(BASE, k=8, V=4096, Eu=16384, hid=16, lab=8, variant="optimize-gcn-inference", iters=2, backend="hip")
    cfg["in"] = 32
    _check(cfg, 2, tmp_path)


def test_four_parties_two_ranks_training_wide_rows_hip(tmp_path):
    # one training epoch at kernel-relevant widths: wave-specialised NN / TN Beaver products, batched element-wise launches,
    # mask/opening reuse of both weight gradients, all across a rank boundary
    cfg = dict(BASE, k=4, V=2048, Eu=8192, hid=64, lab=16, variant="optimize-gcn", iters=6, backend="hip")
    cfg["in"] = 128
    _check(cfg, 2, tmp_path)


def test_four_parties_four_ranks_training_hip(tmp_path):
    cfg = dict(BASE, k=4, V=600, Eu=2000, hid=16, lab=7, variant="optimize-gcn", iters=12, backend="hip")
    cfg["in"] = 33
    _check(cfg, 4, tmp_path)


@pytest.mark.parametrize("k,world,variant,iters,shape", [(4, 2, "optimize-gcn", 12, dict(V=2048, Eu=8192, hid=64, lab=16, inn=128)),
                                                          (8, 4, "optimize-gcn-inference", 2, dict(V=4096, Eu=16384, hid=16, lab=8, inn=32)),
                                                          (3, 3, "optimize-gcn", 6, dict(V=600, Eu=2000, hid=16, lab=7, inn=33))])
def test_vertex_set_placement_hip(tmp_path, k, world, variant, iters, shape):
    """COGNN_PLACE_VERTEX_SET with the HIP kernels: both shares of a vertex set on one rank (pair chains for every two-party step),
    replicas and partial sums across ranks - kernel-relevant widths, one and two training epochs."""
    cfg = dict(BASE, k=k, V=shape["V"], Eu=shape["Eu"], hid=shape["hid"], lab=shape["lab"], variant=variant, iters=iters, backend="hip",
               placement="vertex-set")
    cfg["in"] = shape["inn"]
    _check(cfg, world, tmp_path)


@pytest.mark.parametrize("k,extra", [(2, {}), (3, {"pair_fusion": 0}), (4, {"V": 700, "Eu": 2500, "in": 33, "hid": 16, "lab": 7}),
                                     (2, {"V": 2708, "Eu": 5278, "in": 300, "hid": 16, "lab": 7}), (3, {"V": 9, "Eu": 5})])
def test_original_gcn_single_process_hip(tmp_path, k, extra):
    # BASELINE config 1's kernel on the HIP engine (world 1): the fused per-edge Scatter + Gather launch, forward products after the
    # aggregation (grouped / split-K kernels, two opened-share streams), weight-gradient products with freshly masked ah_t^T, the
    # one-pass weight update + average - two epochs, every GAS iteration against oracle/original_gcn.py; the 2708-vertex case has
    # Cora's vertex and edge counts (narrower features so that the oracle stays in test time)
    _check(dict(BASE, k=k, variant="original-gcn", iters=8, backend="hip", **extra), 1, tmp_path)


@pytest.mark.parametrize("k,world,extra", [(2, 2, {}), (4, 2, {"V": 700, "Eu": 2500, "in": 33, "hid": 16, "lab": 7}), (3, 3, {"V": 9, "Eu": 5}),
                                            (5, 5, {"V": 1433, "Eu": 5000, "in": 65, "hid": 16, "lab": 7, "inproc": True})])
def test_original_gcn_across_ranks_hip(tmp_path, k, world, extra):
    """The unoptimised kernel with its parties on different ranks (the reference's deployment of BASELINE config 1: one process per
    party), HIP kernels: client / server roles of every Scatter instance exchange their openings - the oracle's shares and weights
    after every GAS iteration of two epochs (odd row widths, dummy self entries, five parties as five engine threads)."""
    _check(dict(BASE, k=k, variant="original-gcn", iters=8, backend="hip", **extra), world, tmp_path)


@pytest.mark.parametrize("chunks,world", [(2, 2), (3, 4), (4, 2)])
def test_chunked_exchange_pipeline_hip(tmp_path, chunks, world):
    # COGNN_OPT_EXCHANGE_CHUNKS on the HIP kernels: the chunk window of the element-wise launches (batched and single), the
    # truncation opening of a product deferred into the chunk loop, per-round waits; one training epoch at kernel-relevant widths
    cfg = dict(BASE, k=4, V=1500, Eu=6000, hid=16, lab=7, variant="optimize-gcn", iters=6, backend="hip", chunks=chunks)
    cfg["in"] = 64
    _check(cfg, world, tmp_path)


@pytest.mark.parametrize("seed", range(3 * int(os.environ.get("COGNN_FUZZ_SCALE", "1"))))
def test_multirank_random_configuration_hip(tmp_path, seed):
    rng = np.random.default_rng(9000 + seed)
    world = int(rng.choice([2, 3, 4]))
    k = world * int(rng.integers(1, 3))
    V = int(rng.integers(k, 400)) if seed % 4 else int(rng.integers(300 * k, 700 * k))   # every 4th: rows enough for the MFMA kernels
    cfg = dict(BASE, k=k, V=V, Eu=int(min(V * (V - 1) // 2, rng.integers(1, 4 * V + 1))), gseed=int(rng.integers(1, 1000)),
               seed=int(rng.integers(1, 1 << 30)), hid=int(rng.choice([3, 8, 16, 33])), lab=int(rng.choice([2, 5, 16])),
               variant="optimize-gcn" if seed % 2 else "optimize-gcn-inference", iters=6 if seed % 2 else 2, backend="hip",
               chunks=int(rng.choice([1, 1, 2, 3, 5])))
    cfg["in"] = int(rng.choice([5, 16, 40]))
    if seed % 3 == 2:                                        # every third: both shares of a vertex set on one rank, whole epochs per call
        cfg["placement"] = "vertex-set"
        if cfg["variant"] == "optimize-gcn":
            cfg.update(iters=12, whole_epochs=True)
    _check(cfg, world, tmp_path)


@pytest.mark.parametrize("variant,iters", [("optimize-gcn", 12), ("original-gcn", 8)])
def test_two_parties_per_rank_products_per_side_hip(tmp_path, variant, iters):
    """Two parties per rank, the per-side product path (COGNN_GEMM_PER_SIDE), no offline call, product-sized shapes: the two p = 1 sides
    of a rank follow each other and deal their product shares where they are used, into buffers the sides before them released."""
    cfg = dict(BASE, k=4, V=2400, Eu=9000, hid=64, lab=16, variant=variant, iters=iters, backend="hip", inproc=True, density=0.05,
               env={"COGNN_GEMM_PER_SIDE": "1"})
    cfg["in"] = 128
    _check(cfg, 2, tmp_path)


@pytest.mark.parametrize("seed", range(2 * int(os.environ.get("COGNN_FUZZ_SCALE", "1"))))
def test_original_gcn_across_ranks_random_configuration_hip(tmp_path, seed):
    rng = np.random.default_rng(7800 + seed)
    world = int(rng.choice([2, 3, 4]))
    k = world * int(rng.integers(1, 3))
    V = int(rng.integers(k, 600))
    cfg = dict(BASE, k=k, V=V, Eu=int(min(V * (V - 1) // 2, rng.integers(1, 4 * V + 1))), gseed=int(rng.integers(1, 1000)),
               seed=int(rng.integers(1, 1 << 30)), hid=int(rng.choice([3, 8, 16])), lab=int(rng.choice([2, 5, 7])),
               variant="original-gcn", iters=8, backend="hip", inproc=bool(seed % 2))
    cfg["in"] = int(rng.choice([5, 16, 33, 65]))
    _check(cfg, world, tmp_path)


# ---- the north-star layout on the HIP kernels: 8 parties, one per rank, world 8.  A GPU box admits at most 6 processes on its card,
# so the eight ranks run as threads of one process over a mailbox transport (tests/inproc_worker.py): eight engines / contexts on
# the one GPU, every owner / co-party pair across a rank boundary ----
SMALL = dict(BASE, k=8, V=1 << 14, Eu=1 << 17, hid=64, lab=16, density=0.01, backend="hip", inproc=True)   # bench.py's `small` workload shape
SMALL["in"] = 128


@pytest.mark.parametrize("variant,iters,extra", [("optimize-gcn-inference", 2, {}), ("optimize-gcn-inference", 2, {"placement": "vertex-set"}),
                                                 ("optimize-gcn", 6, {}), ("optimize-gcn", 6, {"placement": "vertex-set", "whole_epochs": True})])
def test_eight_parties_eight_ranks_small_workload_hip(tmp_path, variant, iters, extra):
    """Inference pass and training epoch of the `small` workload shape (2^14 vertices / 2^18 directed edges, in = 128, hid = 64,
    labels = 16: grouped MFMA products, 16-byte Gather lanes) at world 8, both placements, bit-exact vs the oracle."""
    _check(dict(SMALL, variant=variant, iters=iters, **extra), 8, tmp_path)


@pytest.mark.parametrize("variant,iters,extra", [("optimize-gcn", 12, {}), ("optimize-gcn", 12, {"placement": "vertex-set"}),
                                                 ("optimize-gcn", 12, {"chunks": 3}), ("optimize-gcn-inference", 2, {"exchanged_openings": True})])
def test_eight_parties_eight_ranks_hip(tmp_path, variant, iters, extra):
    """Two training epochs (weight average across eight ranks, epoch salt past epoch 0 in eight contexts at once) at hid = 16."""
    cfg = dict(BASE, k=8, V=4096, Eu=16384, hid=16, lab=8, variant=variant, iters=iters, backend="hip", inproc=True, **extra)
    cfg["in"] = 32
    _check(cfg, 8, tmp_path)


# ---- BASELINE.json configs[2] and [3] in their stated layouts: 2-party CiteSeer-shaped on 2 ranks, 4-party PubMed-shaped on 4 ranks
# (one party per rank; HIP kernels, ranks share the GPU, gloo host-staged), one training epoch vs the oracle ----
@pytest.mark.parametrize("name,k,V,E,inn,lab,lr,tr,placement", [
    ("citeseer-2p", 2, 3312, 10016, 3703, 6, 0.8, 0.2, "party"), ("pubmed-4p", 4, 19717, 128146, 500, 3, 8.0, 0.05, "party"),
    ("citeseer-2p", 2, 3312, 10016, 3703, 6, 0.8, 0.2, "vertex-set"), ("pubmed-4p", 4, 19717, 128146, 500, 3, 8.0, 0.05, "vertex-set")])
def test_baseline_configs_3_and_4_one_party_per_rank_hip(tmp_path, name, k, V, E, inn, lab, lr, tr, placement):
    cfg = dict(BASE, k=k, V=V, Eu=E // 2, hid=16, lab=lab, density=0.01, variant="optimize-gcn", iters=6, backend="hip", placement=placement,
               param=dict(learning_rate=lr, train_ratio=tr, val_ratio=0.2 if tr == 0.2 else 0.15, test_ratio=0.6 if tr == 0.2 else 0.8))
    cfg["in"] = inn
    _check(cfg, k, tmp_path)


@pytest.mark.parametrize("cfg_extra,world", [(dict(k=4, V=2048, Eu=8192, hid=64, lab=16, variant="optimize-gcn", iters=6, inn=128), 2),
                                             (dict(k=4, V=1500, Eu=6000, hid=16, lab=7, variant="optimize-gcn", iters=6, inn=64, chunks=3), 4),
                                             (dict(k=8, V=1 << 14, Eu=1 << 17, hid=64, lab=16, density=0.01, variant="optimize-gcn-inference", iters=2, inn=128, inproc=True), 8),
                                             (dict(k=8, V=4096, Eu=16384, hid=16, lab=8, variant="optimize-gcn", iters=12, inn=32, inproc=True), 8)])
def test_packed_openings_hip(tmp_path, cfg_extra, world):
    """COGNN_OPT_PACKED_OPENINGS on the HIP kernels: pack / unpack launches around the exchange of every opened truncation share and
    of the ReLU's opened product (6 bytes per element on the wire), chunk windows included; one party per rank at world 8."""
    extra = dict(cfg_extra)
    inn = extra.pop("inn")
    cfg = dict(BASE, backend="hip", packed_openings=True, **extra)
    cfg["in"] = inn
    _check(cfg, world, tmp_path)
    if cfg.get("inproc"):                                   # the wire really is shorter: bytes counted by the mailbox transport of every rank
        import numpy as np
        packed_bytes = sum(int(np.load(str(tmp_path / "shares") + ".rank%d.npz" % r)["exchange_stats"][1]) for r in range(world))
        import shutil
        plain = tmp_path / "plain"; plain.mkdir()
        _check(dict(cfg, packed_openings=False), world, plain)
        plain_bytes = sum(int(np.load(str(plain / "shares") + ".rank%d.npz" % r)["exchange_stats"][1]) for r in range(world))
        assert packed_bytes < 0.96 * plain_bytes, (packed_bytes, plain_bytes)     # (all bytes of the run: set-up, the feature opening, replicas and partial sums included)
