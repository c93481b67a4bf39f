"""GPU: isd_adamw_step (csrc/adamw.hip) against torch.optim.AdamW on the host.

The reference trains with ``optim.AdamW(self.parameters(), lr=0.0005)`` (src/fast/train/trainer.py:49); the Trainer
applies the same update to its flat parameter block with one HIP kernel.  Checked here: a 60-step trajectory on random
gradients against torch's own (CPU, fp32, single-tensor) implementation -- block lengths that are and are not a
multiple of the kernel's four-element vectors, a changing learning rate, weight decay on and off -- and the device-side
learning rate / step count used for graph replay.  (G9, tests/test_classifier_gpu.py, pins the Trainer as a whole to
the reference's own three-step trajectory.)
"""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _step(L, p, g, m, v, lr, wd, t, lr_dev=None, step_dev=None, betas=(0.9, 0.999), eps=1e-8):
    from isd_amd import _lib
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(L.isd_adamw_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr, betas[0],
                                betas[1], eps, wd, t, lr_dev.data_ptr() if lr_dev is not None else None,
                                step_dev.data_ptr() if step_dev is not None else None, st))


@pytest.mark.parametrize("n", [1, 3, 4, 1021, 4096, 600_005])
@pytest.mark.parametrize("wd", [1e-2, 0.0])
def test_adamw_trajectory_matches_torch(n, wd):
    from isd_amd import _lib
    L = _lib.lib()
    g0 = torch.Generator().manual_seed(n)
    ref = torch.nn.Parameter(torch.randn(n, generator=g0))
    opt = torch.optim.AdamW([ref], lr=5e-4, weight_decay=wd, foreach=False)
    p = ref.detach().clone().cuda()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    steps = 60 if n < 100_000 else 12
    gmax = torch.zeros(n)
    for t in range(1, steps + 1):
        lr = 5e-4 * (0.5 + 0.5 * np.cos(t / 7.0))                # a schedule: the rate changes every step
        # gradients spanning many magnitudes, some exactly zero
        g = torch.randn(n, generator=g0) * torch.exp(4.0 * torch.randn(n, generator=g0))
        g[torch.rand(n, generator=g0) < 0.05] = 0.0
        gmax = torch.maximum(gmax, g.abs())
        for grp in opt.param_groups:
            grp["lr"] = lr
        ref.grad = g.clone()
        opt.step()
        _step(L, p, g.cuda(), m, v, lr, wd, t)
    st = opt.state[ref]
    # same operations in the same order: what is left is the rounding of 1 - beta^t and of the two divisions
    # (the first moment cancels: its error is relative to the gradients that went into it)
    assert ((m.cpu() - st["exp_avg"]).abs() <= 1e-6 * (st["exp_avg"].abs() + gmax)).all()
    assert torch.allclose(v.cpu(), st["exp_avg_sq"], rtol=1e-6, atol=0)
    err = (p.cpu() - ref.detach()).abs().max().item()
    assert err < 2e-6 * max(1.0, ref.detach().abs().max().item()), err


def test_adamw_device_side_rate_and_step_equal_the_by_value_path():
    from isd_amd import _lib
    L = _lib.lib()
    n = 5003
    g0 = torch.Generator().manual_seed(1)
    pa = torch.randn(n, generator=g0).cuda()
    pb = pa.clone()
    ma, va, mb, vb = (torch.zeros_like(pa) for _ in range(4))
    lr_dev = torch.zeros(1, device="cuda")
    step_dev = torch.zeros(1, dtype=torch.int64, device="cuda")
    for t in range(1, 20):
        g = torch.randn(n, generator=g0).cuda()
        lr = 1e-3 / t
        _step(L, pa, g, ma, va, lr, 1e-2, t)
        lr_dev.fill_(lr)
        step_dev.add_(1)
        _step(L, pb, g, mb, vb, 123.0, 1e-2, 0, lr_dev=lr_dev, step_dev=step_dev)   # by-value rate / step are ignored
    assert torch.equal(ma, mb) and torch.equal(va, vb)
    # the host forms 1 - lr wd in fp32 from a double lr, the kernel from the fp32 device value: last-bit agreement
    assert (pa - pb).abs().max().item() < 1e-6


def test_adamw_rejects_bad_arguments():
    from isd_amd import _lib
    L = _lib.lib()
    p = torch.zeros(16, device="cuda")
    g, m, v = torch.zeros_like(p), torch.zeros_like(p), torch.zeros_like(p)
    with pytest.raises(_lib.IsdError):
        _step(L, p, g, m, v, 1e-3, 1e-2, 0)                          # the first step is 1
    with pytest.raises(_lib.IsdError):
        _step(L, p[1:], g[1:], m[1:], v[1:], 1e-3, 1e-2, 1)          # blocks must be 16-byte aligned
    with pytest.raises(_lib.IsdError):
        _step(L, p, g, m, v, 1e-3, 1e-2, 1, betas=(1.0, 0.999))


def _param_list(seed, sizes):
    g0 = torch.Generator().manual_seed(seed)
    return [torch.randn(n, generator=g0) for n in sizes]


@pytest.mark.parametrize("capturable", [False, True])
def test_fused_adamw_list_of_tensors_matches_torch(capturable):
    """isd_amd.FusedAdamW (one launch over a list of tensors; device-side rate and self-advancing step count when
    capturable) against torch.optim.AdamW on the host: odd sizes, a 2-D tensor, a tensor that never gets a gradient."""
    import isd_amd
    sizes = [1, 3, 5, 64, 1030, 4097, 70_001, 7]
    init = _param_list(0, sizes)
    ref = [torch.nn.Parameter(t.clone()) for t in init]
    ref[3] = torch.nn.Parameter(init[3].clone().view(8, 8))
    dev = [torch.nn.Parameter(r.detach().clone().cuda()) for r in ref]
    opt_ref = torch.optim.AdamW(ref, lr=5e-4, weight_decay=1e-2, foreach=False)
    lr = torch.tensor(5e-4, device="cuda") if capturable else 5e-4
    opt = isd_amd.FusedAdamW(dev, lr=lr, weight_decay=1e-2, capturable=capturable)
    g0 = torch.Generator().manual_seed(1)
    for t in range(1, 26):
        rate = 5e-4 * (1.0 - t / 40.0)
        for grp in opt_ref.param_groups:
            grp["lr"] = rate
        if capturable:
            lr.fill_(rate)
        else:
            opt.param_groups[0]["lr"] = rate
        opt_ref.zero_grad(set_to_none=True)
        opt.zero_grad(set_to_none=True)
        for i, (r, d) in enumerate(zip(ref, dev)):
            if i == 7:
                continue                                             # never trained
            g = torch.randn(r.shape, generator=g0)
            r.grad = g.clone()
            d.grad = g.cuda()
        opt_ref.step()
        opt.step()
    if capturable:
        assert opt.state["flat"]["step"].tolist()[:2] == [25, 0]     # advanced by the kernel itself
    for i, (r, d) in enumerate(zip(ref, dev)):
        err = (d.detach().cpu() - r.detach()).abs().max().item()
        assert err < 2e-6 * max(1.0, r.detach().abs().max().item()), (i, err)
    assert torch.equal(dev[7].detach().cpu(), init[7])
    sd = opt.state_dict()
    assert sd["step"] == 25 and sd["exp_avg"].numel() == sum((n + 3) & ~3 for n in sizes)


def test_fused_adamw_more_tensors_than_one_launch_holds():
    """More than 96 tensors: several launches per step, the device step count advanced once.  (Equal-sized small
    tensors take the two-dimensional grid; one 2.1 M-element tensor among them sends its launch to the one-dimensional
    grid with the bisection.)"""
    import isd_amd
    sizes = [17 + 3 * i for i in range(230)]
    sizes[40] = 2_100_003
    init = _param_list(2, sizes)
    ref = [torch.nn.Parameter(t.clone()) for t in init]
    dev = [torch.nn.Parameter(t.clone().cuda()) for t in init]
    opt_ref = torch.optim.AdamW(ref, lr=1e-3, foreach=False)
    opt = isd_amd.FusedAdamW(dev, lr=torch.tensor(1e-3, device="cuda"), capturable=True)
    g0 = torch.Generator().manual_seed(3)
    for _ in range(5):
        for r, d in zip(ref, dev):
            g = torch.randn(r.shape, generator=g0)
            r.grad, d.grad = g.clone(), g.cuda()
        opt_ref.step()
        opt.step()
    assert opt.state["flat"]["step"].tolist()[:2] == [5, 0]
    for r, d in zip(ref, dev):
        assert (d.detach().cpu() - r.detach()).abs().max().item() < 2e-6 * max(1.0, r.detach().abs().max().item())


def test_graphed_step_with_fused_adamw_follows_torch_adamw():
    """The captured FAST step (isd_amd.graph) with FusedAdamW against the same capture with torch's capturable AdamW:
    same initialisation, same batches, dropout off."""
    import isd_amd
    import isd_amd.nn as inn
    from isd_amd.graph import GraphedTrainStep
    rng = np.random.default_rng(0)
    X = torch.from_numpy(rng.standard_normal((32, 64, 800)).astype(np.float32)).cuda()
    y = torch.from_numpy(rng.integers(0, 5, 32).astype(np.uint8)).cuda()
    finals, losses = [], []
    for fused in (False, True):
        torch.manual_seed(4)
        m = inn.FAST(inn.fast_config(dropout=0.0)).cuda().train()
        lr = torch.tensor(0.0, device="cuda")
        opt = (isd_amd.FusedAdamW(m.parameters(), lr=lr, capturable=True) if fused
               else torch.optim.AdamW(m.parameters(), lr=lr, capturable=True))
        gs = GraphedTrainStep(m, opt, X, y, 16)
        ls = []
        for t in range(8):
            idx = torch.arange(16, device="cuda") + 16 * (t % 2)
            gs.loss_sum.zero_()
            gs.step(idx, 5e-4)
            ls.append(float(gs.loss_sum) / 16)
        losses.append(ls)
        finals.append({k: v.detach().clone() for k, v in m.named_parameters()})
    np.testing.assert_allclose(losses[1], losses[0], rtol=2e-4)
    # parameters: 1e-3 = two steps of the learning rate.  The key bias of the attention (a third of in_proj_bias) has
    # NO gradient mathematically -- softmax is invariant to it -- so its computed gradient is rounding noise, which
    # AdamW normalises to full-size steps: the two optimizers' last bits decide its walk (observed 1.9e-4 to 2.3e-4
    # after eight steps); every other tensor agrees to ~1e-5.
    for k, v in finals[0].items():
        assert float((v - finals[1][k]).abs().max()) < 1e-3 * max(1.0, float(v.abs().max())), k
    tight = [k for k in finals[0] if "in_proj_bias" not in k]
    assert max(float((finals[0][k] - finals[1][k]).abs().max()) for k in tight) < 1e-4


def test_graphed_step_with_fused_adamw_advances_the_dropout_masks():
    """FusedAdamW(capturable) shares its device-side step count with the captured step as the dropout counter
    (graph.GraphedTrainStep) -- the production pairing (experiment.train_one_fold).  Replays of one batch at learning
    rate 0 must draw new masks each time, the counter must read the number of completed steps (as the own-counter path
    with torch's AdamW does), and a ragged eager batch and a step without gradients advance it."""
    import isd_amd
    import isd_amd.nn as inn
    from isd_amd.graph import GraphedTrainStep
    rng = np.random.default_rng(1)
    X = torch.from_numpy(rng.standard_normal((32, 64, 800)).astype(np.float32)).cuda()
    y = torch.from_numpy(rng.integers(0, 5, 32).astype(np.uint8)).cuda()
    idx = torch.arange(16, device="cuda")
    first = {}
    for fused in (True, False):
        torch.manual_seed(0)
        m = inn.FAST(inn.fast_config(dropout=0.3)).cuda().train()
        lr = torch.tensor(0.0, device="cuda")
        opt = (isd_amd.FusedAdamW(m.parameters(), lr=lr, capturable=True) if fused
               else torch.optim.AdamW(m.parameters(), lr=lr, capturable=True))
        g = GraphedTrainStep(m, opt, X, y, 16)
        assert g._own_counter == (not fused) and int(m.seed_dev) == 0
        losses = []
        for _ in range(3):
            g.loss_sum.zero_()
            g.step(idx, 0.0)
            losses.append(float(g.loss_sum) / 16)
        assert len(set(losses)) == 3 and all(np.isfinite(losses)), losses
        assert int(m.seed_dev) == 3
        g.step(idx[:5], 0.0)                                       # ragged batch: eager, same code
        assert int(m.seed_dev) == 4
        first[fused] = losses[0]
        if fused:
            opt.zero_grad(set_to_none=True)
            opt.step()                                             # no gradient anywhere: the step still counts
            assert int(m.seed_dev) == 5
            sd = opt.state_dict()
            assert sd["step"] == 5
            opt.step()
            opt.load_state_dict(sd)                                # the mask stream goes back with the optimizer state
            assert int(m.seed_dev) == 5
            with pytest.raises(ValueError):
                opt.load_state_dict(torch.optim.AdamW(m.parameters(), lr=1e-3).state_dict())
    assert all(np.isfinite(v) for v in first.values())      # (the two models' masks differ: the seed mixes in the module instance)


def test_fused_adamw_is_a_torch_optimizer_driven_by_the_reference_schedule():
    """The reference wraps its optimizer in LambdaLR with the per-step cosine multiplier (trainer.py:48-54): FusedAdamW is
    a torch.optim.Optimizer, so the same scheduler object drives it, and the run equals torch's AdamW under it."""
    import isd_amd
    table = isd_amd.cosine_scheduler(1, 0.1, 4, 5, warmup_epochs=1)
    init = _param_list(5, [33, 1000])
    ref = [torch.nn.Parameter(t.clone()) for t in init]
    dev = [torch.nn.Parameter(t.clone().cuda()) for t in init]
    opts = [torch.optim.AdamW(ref, lr=5e-4, foreach=False), isd_amd.FusedAdamW(dev, lr=5e-4)]
    assert isinstance(opts[1], torch.optim.Optimizer)
    scheds = [torch.optim.lr_scheduler.LambdaLR(o, lambda step: float(table[max(step - 1, 0)])) for o in opts]
    g0 = torch.Generator().manual_seed(6)
    for _ in range(12):
        for r, d in zip(ref, dev):
            g = torch.randn(r.shape, generator=g0)
            r.grad, d.grad = g.clone(), g.cuda()
        for o, sc in zip(opts, scheds):
            o.step()
            sc.step()
        assert abs(opts[0].param_groups[0]["lr"] - opts[1].param_groups[0]["lr"]) < 1e-12
        opts[1].zero_grad()
        assert all(d.grad is None for d in dev)
    for r, d in zip(ref, dev):
        assert (d.detach().cpu() - r.detach()).abs().max().item() < 2e-6 * max(1.0, r.detach().abs().max().item())
    sd = opts[1].state_dict()
    fresh = isd_amd.FusedAdamW(dev, lr=5e-4)
    fresh.load_state_dict(sd)
    assert fresh.state_dict()["step"] == 12 and torch.equal(fresh.state["flat"]["exp_avg"], opts[1].state["flat"]["exp_avg"])


def test_fused_adamw_capturable_state_round_trip_resumes_the_trajectory():
    """state_dict() / load_state_dict() with the device-side step block: a run interrupted after 7 steps and resumed in
    a fresh optimizer ends where the uninterrupted torch run ends (the block's running powers beta^t are restored)."""
    import isd_amd
    init = _param_list(8, [257, 4100])
    ref = [torch.nn.Parameter(t.clone()) for t in init]
    dev = [torch.nn.Parameter(t.clone().cuda()) for t in init]
    opt_ref = torch.optim.AdamW(ref, lr=1e-3, foreach=False)
    opt = isd_amd.FusedAdamW(dev, lr=torch.tensor(1e-3, device="cuda"), capturable=True)
    g0 = torch.Generator().manual_seed(9)
    for t in range(15):
        if t == 7:
            sd = opt.state_dict()
            opt = isd_amd.FusedAdamW(dev, lr=torch.tensor(1e-3, device="cuda"), capturable=True)
            opt.load_state_dict(sd)
        for r, d in zip(ref, dev):
            g = torch.randn(r.shape, generator=g0)
            r.grad, d.grad = g.clone(), g.cuda()
        opt_ref.step()
        opt.step()
    assert opt.state["flat"]["step"].tolist()[:2] == [15, 0]
    for r, d in zip(ref, dev):
        assert (d.detach().cpu() - r.detach()).abs().max().item() < 2e-6 * max(1.0, r.detach().abs().max().item())
