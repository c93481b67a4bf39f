"""GPU: synchronised BatchNorm and the RCCL path.

* Two ranks (gloo rendezvous, both on cuda:0) each hold half of a batch; EEGNet_Encoder / CVBlock /
  HeadConv_Paper_Version with the fp64 sum
  blocks all-reduced between the stages must reproduce the single-process result on the whole batch (SURVEY.md 8e;
  the reference's heads use nn.BatchNorm2d on one device, fast.py:46-63,133-159): outputs, running statistics, and
  parameter gradients after the gradient all-reduce.
* One rank with the 'nccl' backend (= RCCL): Trainer.step_begin -> extract -> step_finish through the asynchronous
  all-reduce and the stream-side wait -- the part of bench.py's N > 1 path that can run on one GPU.
"""
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


def _head(kind):
    import isd_amd.nn as inn
    torch.manual_seed(7)
    m = (inn.EEGNet_Encoder(6, 16, dropout=0.0) if kind == "eegnet" else
         inn.CVBlock(6, 16, dropout=0.0) if kind == "cvblock" else inn.HeadConv_Paper_Version(6, 16)).cuda()
    with torch.no_grad():
        for bn in m._bns():
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.uniform_(-0.3, 0.3)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(12, 6, 250, generator=g).cuda()
    w = torch.randn(12, 16, generator=g).cuda()
    return m, x, w


def _run_head(m, x, w):
    m.train()
    y = m(x)
    (y * w).sum().backward()
    grads = torch.cat([p.grad.reshape(-1) for p in m._ordered_params()])
    stats = torch.cat([t.reshape(-1) for bn in m._bns() for t in (bn.running_mean, bn.running_var)])
    return y.detach(), grads, stats


def _syncbn_worker(rank, world, port, kind, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m, x, w = _head(kind)
    n = x.shape[0] // world
    y, grads, stats = _run_head(m, x[rank * n:(rank + 1) * n].contiguous(), w[rank * n:(rank + 1) * n].contiguous())
    h = grads.cpu()
    dist.all_reduce(h)                                         # the data-parallel gradient all-reduce (SUM)
    q.put((rank, y.cpu().numpy(), h.numpy(), stats.cpu().numpy()))
    dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["eegnet", "cvblock", "paper"])
def test_synchronised_batchnorm_two_ranks_equal_single_process(kind):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_syncbn_worker, args=(r, 2, port, kind, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    m, x, w = _head(kind)
    y, grads, stats = _run_head(m, x, w)
    y, grads, stats = y.cpu().numpy(), grads.cpu().numpy(), stats.cpu().numpy()
    got_y = np.concatenate([res[0][1], res[1][1]])
    assert np.abs(got_y - y).max() < 1e-5 * np.abs(y).max()
    for r in res:
        assert np.abs(r[3] - stats).max() < 1e-6 * max(np.abs(stats).max(), 1.0)      # same running statistics everywhere
        # BN1's affine gradients are cancellation residue (scale invariance): compared on the gradient scale
        assert np.abs(r[2] - grads).max() < 2e-4 * np.abs(grads).max()
    assert np.array_equal(res[0][2], res[1][2])
    # and without the exchange the shards do NOT reproduce it (the test would pass vacuously otherwise)
    m2, _, _ = _head(kind)
    m2.sync_bn = False
    y_half, _, _ = _run_head(m2, x[:6].contiguous(), w[:6].contiguous())
    assert np.abs(y_half.cpu().numpy() - y[:6]).max() > 1e-3 * np.abs(y).max()


def _rccl_worker(port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    import isd_amd
    from isd_amd.classifier import _FeatureModel
    torch.manual_seed(3)
    fx = isd_amd.FeatureExtractor(512, 256.0, isd_amd.BANDS_9)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(64, 8, 512, generator=g).cuda()
    y = torch.randint(0, 5, (64,), generator=g).cuda()
    out = {}
    for tag, always in (("rccl", True), ("plain", False)):
        torch.manual_seed(3)
        model = _FeatureModel(9 * 8, 32, 5, 4).cuda()
        tr = isd_amd.Trainer(model, lr=5e-4, weight_decay=1e-2,
                             bucket=isd_amd.GradientBucket(always_collective=always))
        f = fx(x)
        losses = []
        for _ in range(3):
            o = tr.step_begin(f.view(64, -1, f.shape[-1]), y, global_batch=64)     # forward/backward + async all-reduce
            assert (tr._pending[0] is not None) == always
            f = fx(x, out=f)                                                       # queued under the collective
            tr.step_finish()                                                       # stream-side wait + AdamW
            losses.append(float(o["loss"]))
        out[tag] = (losses, model.flat_params().cpu().numpy())
    # a collective on a device tensor of another dtype (what the SyncBN exchange and bench.py's timing reduction issue)
    t = torch.arange(8, dtype=torch.float64, device="cuda")
    dist.all_reduce(t)
    torch.cuda.synchronize()
    q.put((out, t.cpu().numpy()))
    dist.destroy_process_group()


def test_rccl_backend_drives_the_pipelined_step_on_one_gpu():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q))
    p.start()
    out, t = q.get(timeout=300)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert out["rccl"][0] == out["plain"][0] and np.array_equal(out["rccl"][1], out["plain"][1])
    assert np.array_equal(t, np.arange(8.0))
