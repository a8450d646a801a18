"""A deliberately unhelpful asynchronous exchange for the multi-rank tests (CPU backend, host buffers, gloo): what a real
asynchronous transport is ALLOWED to do, taken to the extreme -
  * begin() sends nothing and copies nothing: an outbox is read only when the round completes, so an engine that overwrites a
    buffer it has handed to begin() before calling wait() ships the overwritten bytes;
  * begin() fills every inbox of the round with a poison pattern, and the data lands only in wait(): an engine that reads a
    received buffer without waiting reads poison;
  * wait() completes the rounds in flight in shuffled order, and the messages of a round in shuffled order (messages between a
    pair of ranks are matched by their per-pair sequence number, as RCCL matches them by issue order).
  * wait_round(r) (per_round=True; the chunked pipelines of COGNN_OPT_EXCHANGE_CHUNKS) completes the rounds up to r only: the
    later ones stay undelivered - their inboxes poisoned, their outboxes unread - until their own wait.
gloo's blocking host-staged delivery (cognn_amd/dist.py) hides all three; a missing exchange_wait shows up here as a share
mismatch against the oracle."""
import ctypes
import random

import numpy as np
import torch
import torch.distributed as dist

from cognn_amd.engine_api import EXCHANGE_FN, EXCHANGE_WAIT_FN, EXCHANGE_WAIT_ROUND_FN


def make_hostile_exchange(seed, group=None, skip_waits=0, per_round=False):
    """skip_waits: the first n calls of wait() return without completing anything (the rounds complete at a later wait) - what an
    engine with a MISSING exchange_wait looks like from the data's point of view; used to show that the transport detects it."""
    rank = dist.get_rank(group)
    rng = random.Random(seed * 1000 + rank)
    pending = []                                             # rounds in flight: lists of (is_send, ptr, nbytes, peer, tag)
    seq = {}                                                 # (peer, is_send) -> messages issued so far
    stats = {"rounds": 0, "max_inflight": 0}

    def _view(ptr, nbytes):
        buf = (ctypes.c_uint8 * nbytes).from_address(ptr)
        return torch.from_numpy(np.frombuffer(buf, dtype=np.uint8))

    def _begin(user, xfers, n):
        try:
            rnd = []
            for i in range(n):
                x = xfers[i]
                key = (int(x.peer), int(x.is_send))
                tag = seq.get(key, 0)
                seq[key] = tag + 1
                if not x.is_send:
                    ctypes.memset(x.ptr, 0xA5, x.bytes)      # nothing has arrived yet
                rnd.append((int(x.is_send), int(x.ptr), int(x.bytes), int(x.peer), tag))
            if rnd:
                pending.append((stats["rounds"], rnd))
                stats["rounds"] += 1
                stats["max_inflight"] = max(stats["max_inflight"], len(pending))
            return 0
        except Exception as ex:  # noqa: BLE001
            print("hostile exchange (begin) failed: %r" % (ex,), flush=True)
            return 1

    def _complete(upto):
        try:
            if stats.setdefault("waits", 0) < skip_waits:
                stats["waits"] += 1
                return 0
            stats["waits"] += 1
            rounds = [r for i, r in pending if i <= upto]
            later = [(i, r) for i, r in pending if i > upto]
            pending[:] = later
            stats["left_inflight"] = max(stats.get("left_inflight", 0), len(later))
            rng.shuffle(rounds)
            works, keep = [], []
            for rnd in rounds:
                msgs = rnd[:]
                rng.shuffle(msgs)
                for is_send, ptr, nbytes, peer, tag in msgs:
                    t = _view(ptr, nbytes)                   # a send reads its buffer NOW, not when begin() was called
                    keep.append(t)
                    works.append(dist.isend(t, peer, group=group, tag=tag) if is_send else dist.irecv(t, peer, group=group, tag=tag))
            for w in works:
                w.wait()
            return 0
        except Exception as ex:  # noqa: BLE001
            print("hostile exchange (wait) failed: %r" % (ex,), flush=True)
            return 1

    def _wait(user):
        return _complete(1 << 62)

    def _wait_round(user, rnd):
        return _complete(int(rnd))

    if per_round:
        return (EXCHANGE_FN(_begin), EXCHANGE_WAIT_FN(_wait), EXCHANGE_WAIT_ROUND_FN(_wait_round)), stats
    return (EXCHANGE_FN(_begin), EXCHANGE_WAIT_FN(_wait)), stats
