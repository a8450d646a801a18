"""cognn_graph_build_colocated: degrees, dummy-source rule and the aggregate CSR built on the device from the edge list,
against the oracle's restatement of onPreprocessClient (ss_...h:295-534) and the host builder's CSR semantics.  The entry order
inside a CSR row is unspecified (uint64 addition commutes), so rows are compared as multisets."""
import ctypes

import numpy as np
import pytest

import cognn_oracle as co
from gpu_util import dev, dev_empty, host, ptr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from cognn_amd import capi
    c = capi.Context(0)
    yield c
    c.close()


def _layout(k, part):
    V = len(part)
    row_of = np.zeros(V, dtype=np.uint32)
    n = [0] * k
    for v in range(V):
        row_of[v] = n[part[v]]; n[part[v]] += 1
    a_off, b_off, off = [0] * k, [0] * k, 0
    for p in range(k):
        off = (off + 1) & ~1; a_off[p] = off; off += n[p]
    for p in range(k):                                   # co-hosted owners in the order of their co-parties (engine layout)
        o = (p + k - 1) % k
        off = (off + 1) & ~1; b_off[o] = off; off += n[o]
    off = (off + 1) & ~1
    return row_of, np.array(a_off, dtype=np.int64), np.array(b_off, dtype=np.int64), off


@pytest.mark.parametrize("k,V,Eu,undirected,uneven", [(2, 40, 90, False, False), (3, 301, 900, False, True), (4, 1000, 0, False, False),
                                                        (5, 257, 700, True, True), (8, 5000, 40000, False, False)])
def test_device_graph_build_matches_oracle_preprocess(ctx, k, V, Eu, undirected, uneven):
    rng = np.random.default_rng(V + Eu)
    src, dst = co.synth_graph(V, Eu, 11) if Eu else (np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64))
    if undirected:                                       # keep one direction per pair: -u generates the other
        keep = src < dst
        src, dst = src[keep], dst[keep]
    if len(src):                                         # some duplicate edges and a self loop, as a text edge list may hold them
        src = np.concatenate([src, src[:5], [3]]); dst = np.concatenate([dst, dst[:5], [3]])
    part = (rng.integers(0, k, size=V) if uneven else np.arange(V) % k).astype(np.int32)
    row_of, a_off, b_off, rows = _layout(k, part)
    E = len(src)
    total = 2 * E if undirected else E
    rowptr = dev_empty(rows + 1, "u32"); col = dev_empty(max(2 * total, 1), "u32")
    tin, din, dout = dev_empty(V, "u32"), dev_empty(V, "u32"), dev_empty(V, "u32")
    border, dummy = dev_empty(V, "u8"), dev_empty(V, "u8")
    scratch = dev_empty(V + 2 * rows + 2, "u32")
    ctx.call("cognn_graph_build_colocated", V, E, int(undirected), ptr(dev(src.astype(np.int64))), ptr(dev(dst.astype(np.int64))),
             ptr(dev(part)), ptr(dev(row_of.view(np.int32))), ptr(dev(a_off)), ptr(dev(b_off)), rows, ptr(rowptr), ptr(col), ptr(tin),
             ptr(din), ptr(dout), ptr(border), ptr(dummy), ptr(scratch))
    fs, fd = (np.concatenate([src, dst]), np.concatenate([dst, src])) if undirected else (src, dst)
    # degrees and flags vs the oracle's per-party preprocessing
    for P in range(k):
        gs = co.preprocess_party(P, k, fs, fd, list(part))
        vids = np.array(gs.localVertexPos, dtype=np.int64)
        if len(vids) == 0:
            continue
        assert list(host(din, np.uint32)[vids]) == gs.localVertexInDeg
        assert list(host(tin, np.uint32)[vids]) == [gs.true_in_deg[v] for v in vids]
        assert list(host(border, np.uint8)[vids].astype(bool)) == gs.isLocalVertexBorder
        assert list(host(dummy, np.uint8)[vids].astype(bool)) == gs.isGatherDstVertexDummy[P]
    # CSR: every directed edge u -> v adds (A row of v <- own-party A row of u or the source party's B row) and the mirrored
    # entry for the B row
    want = [[] for _ in range(rows)]
    A = lambda v: int(a_off[part[v]] + row_of[v]); B = lambda v: int(b_off[part[v]] + row_of[v])
    for u, v in zip(fs, fd):
        same = part[u] == part[v]
        want[A(v)].append(A(u) if same else B(u)); want[B(v)].append(B(u) if same else A(u))
    rp = host(rowptr, np.uint32); cl = host(col, np.uint32)
    assert rp[0] == 0 and rp[-1] == 2 * total
    for r in range(rows):
        assert sorted(cl[rp[r]:rp[r + 1]].tolist()) == sorted(want[r]), r


def test_device_graph_build_orders_hub_rows_like_short_ones(ctx):
    """A skewed graph: two hubs (degree ~3000 and ~12000: the LDS and the in-place sort of row_order_big_kernel) among
    low-degree vertices.  Every CSR row - hub or not - holds the expected multiset in the one fixed order (ascending by the
    hash of the source row), so the build is deterministic and no single lane walks a hub row."""
    rng = np.random.default_rng(5)
    k, V = 4, 20000
    hubs = {7: 3000, 1234: 12000}
    src = [rng.integers(0, V, size=30000)]; dst = [rng.integers(0, V, size=30000)]
    for h, d in hubs.items():
        src.append(rng.integers(0, V, size=d)); dst.append(np.full(d, h))
    src = np.concatenate(src).astype(np.int64); dst = np.concatenate(dst).astype(np.int64)
    part = (np.arange(V) % k).astype(np.int32)
    row_of, a_off, b_off, rows = _layout(k, part)
    E = len(src)
    bufs = {}
    for run in range(2):
        rowptr = dev_empty(rows + 1, "u32"); col = dev_empty(2 * E, "u32")
        tin, din, dout = dev_empty(V, "u32"), dev_empty(V, "u32"), dev_empty(V, "u32")
        border, dummy = dev_empty(V, "u8"), dev_empty(V, "u8")
        ctx.call("cognn_graph_build_colocated", V, E, 0, ptr(dev(src)), ptr(dev(dst)), ptr(dev(part)), ptr(dev(row_of.view(np.int32))), ptr(dev(a_off)),
                 ptr(dev(b_off)), rows, ptr(rowptr), ptr(col), ptr(tin), ptr(din), ptr(dout), ptr(border), ptr(dummy), ptr(dev_empty(V + 2 * rows + 2, "u32")))
        bufs[run] = (host(rowptr, np.uint32).copy(), host(col, np.uint32).copy())
    rp, cl = bufs[0]
    assert np.array_equal(rp, bufs[1][0]) and np.array_equal(cl, bufs[1][1]), "two builds of the same edge list differ"
    want = [[] for _ in range(rows)]
    A = lambda v: int(a_off[part[v]] + row_of[v]); B = lambda v: int(b_off[part[v]] + row_of[v])
    for u, v in zip(src, dst):
        same = part[u] == part[v]
        want[A(v)].append(A(u) if same else B(u)); want[B(v)].append(B(u) if same else A(u))

    def key(c):
        c = np.uint32(c)
        with np.errstate(over="ignore"):
            c ^= c >> np.uint32(16); c *= np.uint32(0x7feb352d); c ^= c >> np.uint32(15); c *= np.uint32(0x846ca68b); c ^= c >> np.uint32(16)
        return int(c)
    for h in hubs:
        for r in (A(h), B(h)):
            got = cl[rp[r]:rp[r + 1]].tolist()
            assert len(got) >= hubs[h] and got == sorted(want[r], key=lambda c: (key(c), c)), "hub row %d" % r
    for r in range(0, rows, 97):
        got = cl[rp[r]:rp[r + 1]].tolist()
        assert got == sorted(want[r], key=lambda c: (key(c), c)), r


def test_device_graph_build_rejects_bad_vertex_ids(ctx):
    from cognn_amd import capi
    V, k = 10, 2
    part = (np.arange(V) % k).astype(np.int32)
    row_of, a_off, b_off, rows = _layout(k, part)
    src = np.array([0, 11], dtype=np.int64); dst = np.array([1, 2], dtype=np.int64)
    bufs = [dev_empty(rows + 1, "u32"), dev_empty(4, "u32")] + [dev_empty(V, "u32") for _ in range(3)] + [dev_empty(V, "u8"), dev_empty(V, "u8")]
    with pytest.raises(capi.CognnError, match="vertex id out of range"):
        ctx.call("cognn_graph_build_colocated", V, 2, 0, ptr(dev(src)), ptr(dev(dst)), ptr(dev(part)), ptr(dev(row_of.view(np.int32))),
                 ptr(dev(a_off)), ptr(dev(b_off)), rows, *[ptr(b) for b in bufs], ptr(dev_empty(V + 2 * rows + 2, "u32")))
