"""GPU: data-parallel equivalence.  Two ranks (gloo rendezvous, both on cuda:0) each train on half of every batch
with the flat-gradient all-reduce; the parameters must equal a single process training on the whole batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make(seed=3):
    import isd_amd
    from isd_amd.classifier import _FeatureModel
    torch.manual_seed(seed)
    fx = isd_amd.FeatureExtractor(512, 256.0, isd_amd.BANDS_9)
    model = _FeatureModel(9 * 8, 32, 5, 4).cuda()
    g = torch.Generator().manual_seed(11)
    x = torch.randn(64, 8, 512, generator=g).cuda()
    y = torch.randint(0, 5, (64,), generator=g).cuda()
    return isd_amd, fx, model, x, y


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    isd_amd, fx, model, x, y = _make()
    if rank == 1:                                   # replicas must start from rank 0's parameters
        with torch.no_grad():
            model.flat_params().add_(0.5)
    tr = isd_amd.Trainer(model, lr=5e-4, weight_decay=1e-2)
    lo, hi = tr.bucket.shard(x.shape[0])
    losses = []
    f = fx(x[lo:hi])
    for _ in range(3):
        # bench.py's N > 1 order: the next batch's features are extracted between the start of the all-reduce
        # and the optimizer step
        out = tr.step_begin(f.view(hi - lo, -1, f.shape[-1]), y[lo:hi], global_batch=x.shape[0])
        f = fx(x[lo:hi], out=f)
        tr.step_finish()
        loss = out["loss"].clone()
        tr.bucket.all_reduce_(loss)                 # local shares of the global-mean loss add up
        losses.append(float(loss))
    q.put((rank, losses, model.flat_params().cpu().numpy()))
    dist.destroy_process_group()


def test_two_rank_data_parallel_equals_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    isd_amd, fx, model, x, y = _make()
    tr = isd_amd.Trainer(model, lr=5e-4, weight_decay=1e-2)
    ref_losses = []
    for _ in range(3):
        f = fx(x)
        ref_losses.append(float(tr.step(f.view(x.shape[0], -1, f.shape[-1]), y)["loss"]))
    ref = model.flat_params().cpu().numpy()
    np.testing.assert_allclose(res[0][1], ref_losses, rtol=2e-5)
    np.testing.assert_allclose(res[1][1], ref_losses, rtol=2e-5)
    assert np.abs(res[0][2] - res[1][2]).max() == 0.0                    # replicas stay bit-identical
    assert np.abs(res[0][2] - ref).max() < 2e-5 * np.abs(ref).max()
