"""Communicator bootstrap of one-process-per-GPU runs (include/cognn_exchange.h, cognn_rccl_rendezvous_tcp): rank 0 creates the
RCCL unique id and serves it over TCP, the other ranks fetch it - the counterpart of the reference's fixed-port session setup
(include/engine.h:166-201).  Needs no GPU (the id is host data); what follows it, ncclCommInitRank, does."""
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "rendezvous_worker.py")


def _port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def test_every_rank_receives_rank0s_id():
    import __graft_entry__ as ge
    ge.build()
    world, port = 4, _port()
    procs = [subprocess.Popen([sys.executable, WORKER, str(r), str(world), str(port), "30"], stdout=subprocess.PIPE, text=True)
             for r in (2, 1, 3)]                                  # clients first: they retry until rank 0 listens
    procs.append(subprocess.Popen([sys.executable, WORKER, "0", str(world), str(port), "30"], stdout=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=90)[0].strip().splitlines()[-1] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    ids = {o.split()[1] for o in outs}
    assert len(ids) == 1 and len(ids.pop()) == 256 and all(o.startswith("ID ") for o in outs)


def test_missing_ranks_time_out_with_a_message():
    port = _port()
    r0 = subprocess.run([sys.executable, WORKER, "0", "2", str(port), "1.5"], capture_output=True, text=True, timeout=60)
    assert r0.returncode == 3 and "0 of 1 ranks connected" in r0.stdout
    r1 = subprocess.run([sys.executable, WORKER, "1", "2", str(_port()), "0.5"], capture_output=True, text=True, timeout=60)
    assert r1.returncode == 3 and "not reachable" in r1.stdout
