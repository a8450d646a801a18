"""bench.attach_transport: which exchange a multi-rank bench run ends up with - the native RCCL transport, the torch.distributed
callbacks when ANY rank could not create it (agreed through an all-reduce, so that no rank is left behind with the other transport),
the host-staged callbacks for the gloo rehearsal.  The real transports need several GPUs; here their constructors are stand-ins and only
the decision logic runs."""
import os
import sys
import types

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


class FakeTorch:
    int32 = torch.int32

    @staticmethod
    def device(kind, index=0):
        return torch.device("cpu")

    @staticmethod
    def tensor(v, dtype=None, device=None):
        return torch.tensor(v, dtype=dtype)


class FakeEngine:
    def __init__(self):
        self.exchange = None

    def set_exchange(self, x):
        self.exchange = x


def fakes(native_fails_here, native_fails_elsewhere):
    calls = []
    dist = types.SimpleNamespace(ReduceOp=types.SimpleNamespace(MIN="min"))

    def all_reduce(t, op=None):
        assert op == "min"
        if native_fails_elsewhere:
            t.fill_(0)                                         # (what the MIN over the ranks would give)
    dist.all_reduce = all_reduce
    cdist = types.SimpleNamespace()

    def attach_rccl(eng, local_rank):
        calls.append("rccl")
        if native_fails_here:
            raise RuntimeError("ncclCommInitRank: unhandled system error")
        eng.exchange = "native"
        return "RCCL"
    cdist.attach_rccl = attach_rccl

    def make_exchange_async(device, host_staged=False, per_round=False):
        calls.append(("callbacks", host_staged, per_round))
        return ("callbacks", host_staged, per_round)
    cdist.make_exchange_async = make_exchange_async
    return dist, cdist, calls


def test_native_transport_when_every_rank_has_it():
    dist, cdist, calls = fakes(False, False)
    eng = FakeEngine()
    xch, desc, err = bench.attach_transport(eng, FakeTorch, dist, cdist, "nccl", 0, False)
    assert xch == "RCCL" and err is None and "native RCCL" in desc and eng.exchange == "native" and calls == ["rccl"]


@pytest.mark.parametrize("here,elsewhere", [(True, True), (False, True)])
def test_every_rank_falls_back_together(here, elsewhere):
    dist, cdist, calls = fakes(here, elsewhere)
    eng = FakeEngine()
    xch, desc, err = bench.attach_transport(eng, FakeTorch, dist, cdist, "nccl", 3, True)
    assert xch is None and "FALLBACK" in desc and err
    assert ("unhandled system error" in err) == here
    assert eng.exchange == ("callbacks", False, True) and calls[-1] == ("callbacks", False, True)   # device-direct callbacks, per-round waits kept


def test_gloo_rehearsal_is_host_staged():
    dist, cdist, calls = fakes(False, False)
    eng = FakeEngine()
    xch, desc, err = bench.attach_transport(eng, FakeTorch, dist, cdist, "gloo", 0, False)
    assert xch is None and err is None and "host-staged" in desc and eng.exchange == ("callbacks", True, False) and "rccl" not in calls
