"""CPU: schedule, shard arithmetic and the flat-bucket all-reduce over gloo (world_size 2)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load_golden


def test_cosine_schedule_and_lambda_quirk():
    import isd_amd
    g = load_golden("g8_cosine.npz")
    t = isd_amd.cosine_scheduler(1, 0.1, 200, 5, warmup_epochs=10)
    np.testing.assert_allclose(t, g["schedule"], rtol=0, atol=1e-15)
    assert isd_amd.lr_multiplier(t, 0) == t[-1]        # trainer.py:52: first step reads index -1
    assert isd_amd.lr_multiplier(t, 1) == t[0] == 0.0
    assert isd_amd.lr_multiplier(t, 51) == 1.0


def test_flat_param_packing_keeps_names_and_aliases():
    from isd_amd.classifier import _FeatureModel
    m = _FeatureModel(12, 16, 5, 4)
    flat = m.flat_params()
    assert flat.numel() == sum(p.numel() for p in m.parameters())
    names = [k for k, _ in m.named_parameters()]
    assert names[:3] == ["net.cnn.cnn1.weight", "net.cnn.cnn1.bias", "net.cnn.cnn2.weight"]
    flat.zero_()
    assert all(float(p.abs().sum()) == 0 for p in m.parameters())      # same storage
    g = m.flat_grads()
    g.fill_(2.0)
    assert all(float(p.grad.min()) == 2.0 for p in m.parameters())
    sd = {k: torch.ones_like(v) for k, v in m.state_dict().items()}
    m.load_state_dict(sd)                                               # copy_ in place keeps the aliasing
    assert float(m.flat_params().min()) == 1.0
    assert m.net.cnn.flat_params().data_ptr() == m.flat_params().data_ptr()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import isd_amd
    b = isd_amd.GradientBucket()
    lo, hi = b.shard(10)
    # each rank's local gradient is already divided by the GLOBAL batch: the SUM is the global mean gradient
    full = torch.arange(10, dtype=torch.float32)
    g = torch.zeros(4)
    g[0] = full[lo:hi].sum() / 10.0
    g[1] = float(rank + 1)
    b.all_reduce_(g)
    h = torch.full((2,), float(rank + 1))
    work = b.all_reduce_start(h)                   # the split form bench.py pipelines behind the next extraction
    b.all_reduce_wait(work)
    assert h.tolist() == [3.0, 3.0]
    params = torch.full((3,), float(rank))
    b.broadcast_(params)
    q.put((rank, lo, hi, g.tolist(), params.tolist(), b.world_size))
    dist.destroy_process_group()


def test_gradient_bucket_allreduce_and_shards_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, lo0, hi0, g0, p0, w0), (r1, lo1, hi1, g1, p1, w1) = res
    assert (lo0, hi0, lo1, hi1) == (0, 5, 5, 10) and w0 == w1 == 2
    assert g0 == g1
    assert abs(g0[0] - 4.5) < 1e-6          # mean of 0..9
    assert g0[1] == 3.0                      # 1 + 2
    assert p0 == p1 == [0.0, 0.0, 0.0]       # parameters broadcast from rank 0


def test_shard_remainder_goes_to_first_ranks():
    import isd_amd

    class B(isd_amd.GradientBucket):
        def __init__(self, w, r):
            self._w, self._r, self.dist, self.group = w, r, None, None
        world_size = property(lambda s: s._w)
        rank = property(lambda s: s._r)
    spans = [B(4, r).shard(10) for r in range(4)]
    assert spans == [(0, 3), (3, 6), (6, 8), (8, 10)]
