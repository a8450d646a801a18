"""GPU-side plumbing of the exchange callback: a raw device pointer must be aliased (not copied) by the torch tensor
that torch.distributed sends/receives (cognn_amd/dist.py).  The multi-rank protocol itself is covered on CPU over gloo
(tests/test_multirank_cpu.py); >1 GPU is not available to the tests."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_device_pointer_wrap_aliases_memory():
    import torch
    from cognn_amd import dist as cdist
    t = torch.arange(1024, dtype=torch.int64, device="cuda")
    w = cdist._wrap(t.data_ptr(), t.numel() * 8, torch.device("cuda", 0))
    assert w.dtype == torch.uint8 and w.numel() == 8192 and w.data_ptr() == t.data_ptr()
    w.view(torch.int64)[5] = -7
    torch.cuda.synchronize()
    assert int(t[5].item()) == -7
    sub = cdist._wrap(t.data_ptr() + 64, 128, torch.device("cuda", 0))          # interior pointer (segment of a table)
    assert sub.data_ptr() == t.data_ptr() + 64
    assert np.array_equal(sub.view(torch.int64).cpu().numpy(), t[8:24].cpu().numpy())


def test_single_rank_nccl_group_and_exchange_noop():
    """world_size 1 over the nccl backend: bench.py's distributed branch (barrier, all_reduce of the step time) works and an
    empty exchange list is a no-op."""
    import os
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29517")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        x = torch.tensor([1.5], dtype=torch.float64, device="cuda")
        dist.all_reduce(x, op=dist.ReduceOp.MAX)
        dist.barrier()
        assert float(x.item()) == 1.5
        from cognn_amd import dist as cdist
        fn = cdist.make_exchange(torch.device("cuda", 0))
        assert fn(None, None, 0) == 0
    finally:
        dist.destroy_process_group()
