"""GPU, more than one device: the N > 1 path on RCCL itself (BASELINE config 4).

One rank per GPU with the 'nccl' backend (= RCCL over xGMI), as bench.py launches them: contiguous batch shards,
forward/backward -> asynchronous flat-gradient all-reduce on RCCL's stream -> the next batch's feature extraction
queued under it -> stream-side wait + AdamW (Trainer.step_begin / step_finish), and the synchronised-BatchNorm
exchange of the EEGNet head (int64 accumulator words and fp64 sums through ncclAllReduce).  Every rank must end on
the parameters of a single process training on the whole batch.

The builder's boxes have one GPU, so these tests are SKIPPED there; they exist so that the first multi-GPU box the
suite runs on proves or breaks the RCCL path.  The same equalities are exercised on one card
through gloo in test_dp_gpu.py / test_syncbn_gpu.py.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

MAX_RANKS = 8                      # one rank per GPU of a whole node; 2, 4 and 8 all divide the test batches (64, 24)
# ISD_TEST_DP_REHEARSE=1: run the same workers with two gloo ranks sharing cuda:0 (checks the test's own logic on a
# one-GPU box; RCCL refuses two ranks on one device)
REHEARSE = bool(os.environ.get("ISD_TEST_DP_REHEARSE"))
needs_two = pytest.mark.skipif(torch.cuda.device_count() < 2 and not REHEARSE,
                               reason="needs at least two GPUs (RCCL with more than one rank)")


def _ranks_for(n_dev):
    """2, 4 or 8 ranks: the largest of them the devices allow (all divide the test batches)."""
    n = min(n_dev, MAX_RANKS)
    return 8 if n >= 8 else 4 if n >= 4 else 2 if n >= 2 else 1


def _n_ranks():
    return 2 if REHEARSE else _ranks_for(torch.cuda.device_count())


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


def _init(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    if REHEARSE:
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        return
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))


def _all_reduce(t, op=dist.ReduceOp.SUM):
    """dist.all_reduce on a device tensor (gloo rehearsal: through the host)."""
    if REHEARSE:
        h = t.cpu()
        dist.all_reduce(h, op=op)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op)


def _dp_worker(rank, world, port, q):
    try:
        _init(rank, world, port)
        isd_amd, fx, model, x, y = _make()
        if rank > 0:                                  # replicas must start from rank 0's parameters
            with torch.no_grad():
                model.flat_params().add_(0.25 * rank)
        tr = isd_amd.Trainer(model, lr=5e-4, weight_decay=1e-2)
        lo, hi = tr.bucket.shard(x.shape[0])
        f = fx(x[lo:hi])
        losses = []
        for _ in range(3):
            out = tr.step_begin(f.view(hi - lo, -1, f.shape[-1]), y[lo:hi], global_batch=x.shape[0])
            assert REHEARSE or tr._pending[0] is not None   # the all-reduce is in flight on RCCL's stream
            f = fx(x[lo:hi], out=f)                   # bench.py's order: the next extraction queued under it
            tr.step_finish()
            loss = out["loss"].clone()
            tr.bucket.all_reduce_(loss)               # local shares of the global-mean loss add up
            losses.append(float(loss))
        # bench.py's timing reduction: MAX over ranks of a float64 device scalar
        t = torch.tensor([float(rank)], dtype=torch.float64, device="cuda")
        _all_reduce(t, dist.ReduceOp.MAX)
        torch.cuda.synchronize()
        q.put((rank, losses, model.flat_params().cpu().numpy(), float(t)))
        dist.destroy_process_group()
    except Exception as e:                            # report instead of leaving the parent to time out
        import traceback
        q.put((rank, None, f"{type(e).__name__}: {e}\n{traceback.format_exc()}", None))


def _spawn(target, world, extra=()):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port) + tuple(extra) + (q,)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
    for r in res:
        assert r[1] is not None, f"rank {r[0]} failed:\n{r[2]}"
    for p in procs:
        assert p.exitcode == 0
    return sorted(res, key=lambda r: r[0])


@needs_two
def test_rccl_data_parallel_step_equals_single_process_whole_batch():
    world = _n_ranks()
    res = _spawn(_dp_worker, world)
    isd_amd, fx, model, x, y = _make()
    tr = isd_amd.Trainer(model, lr=5e-4, weight_decay=1e-2)
    ref_losses = []
    for _ in range(3):
        f = fx(x)
        ref_losses.append(float(tr.step(f.view(x.shape[0], -1, f.shape[-1]), y)["loss"]))
    ref = model.flat_params().cpu().numpy()
    for r in res:
        np.testing.assert_allclose(r[1], ref_losses, rtol=2e-5)
        assert np.array_equal(r[2], res[0][2])                           # replicas stay bit-identical
        assert r[3] == float(world - 1)
    assert np.abs(res[0][2] - ref).max() < 2e-5 * np.abs(ref).max()


def _head(kind):
    import isd_amd.nn as inn
    torch.manual_seed(7)
    m = (inn.EEGNet_Encoder(6, 16, dropout=0.0) if kind == "eegnet" else inn.HeadConv_Paper_Version(6, 16)).cuda()
    with torch.no_grad():
        for bn in m._bns():
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.uniform_(-0.3, 0.3)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(24, 6, 250, generator=g).cuda()
    w = torch.randn(24, 16, generator=g).cuda()
    return m, x, w


def _run_head(m, x, w):
    m.train()
    y = m(x)
    (y * w).sum().backward()
    grads = torch.cat([p.grad.reshape(-1) for p in m._ordered_params()])
    stats = torch.cat([t.reshape(-1) for bn in m._bns() for t in (bn.running_mean, bn.running_var)])
    return y.detach(), grads, stats


def _syncbn_worker(rank, world, port, kind, q):
    try:
        _init(rank, world, port)
        m, x, w = _head(kind)
        n = x.shape[0] // world
        y, grads, stats = _run_head(m, x[rank * n:(rank + 1) * n].contiguous(), w[rank * n:(rank + 1) * n].contiguous())
        _all_reduce(grads)                                     # the data-parallel gradient all-reduce (SUM), on RCCL
        torch.cuda.synchronize()
        q.put((rank, y.cpu().numpy(), grads.cpu().numpy(), stats.cpu().numpy()))
        dist.destroy_process_group()
    except Exception as e:
        import traceback
        q.put((rank, None, f"{type(e).__name__}: {e}\n{traceback.format_exc()}", None))


@needs_two
@pytest.mark.parametrize("kind", ["eegnet", "paper"])
def test_rccl_synchronised_batchnorm_equals_single_process(kind):
    world = _n_ranks()
    res = _spawn(_syncbn_worker, world, (kind,))
    m, x, w = _head(kind)
    y, grads, stats = _run_head(m, x, w)
    y, grads, stats = y.cpu().numpy(), grads.cpu().numpy(), stats.cpu().numpy()
    got_y = np.concatenate([r[1] for r in res])
    assert np.abs(got_y - y).max() < 1e-5 * np.abs(y).max()
    for r in res:
        assert np.abs(r[3] - stats).max() < 1e-6 * max(np.abs(stats).max(), 1.0)
        assert np.abs(r[2] - grads).max() < 2e-4 * np.abs(grads).max()
        assert np.array_equal(r[2], res[0][2])


def test_rank_count_rule():
    """The helper that picks the number of ranks divides the test batches (64 and 24) for every device count."""
    for n_dev, want in ((1, 1), (2, 2), (3, 2), (4, 4), (6, 4), (8, 8), (16, 8)):
        n = _ranks_for(n_dev)
        assert n == want and 64 % n == 0 and 24 % n == 0
