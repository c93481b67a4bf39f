"""GPU parity: factorised HIP EEGNet_Encoder vs the golden vectors captured from the reference and the oracle."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from oracle import cnn as ocnn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def inn():
    import isd_amd.nn as m
    assert torch.cuda.is_available()
    return m


def _sd(g, prefix):
    return {k[len(prefix):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(prefix)}


def _grad_err(got, want, scale):
    return float(np.abs(np.asarray(got, np.float64) - np.asarray(want, np.float64)).max() / scale)


@pytest.mark.parametrize("tag,C,T", [("z6", 6, 250), ("c128", 128, 96)])
def test_eegnet_matches_reference_golden(inn, tag, C, T):
    g = load_golden("g7_eegnet.npz")
    m = inn.EEGNet_Encoder(C, 32, dropout=0.0).cuda()
    m.load_state_dict(_sd(g, f"{tag}.sd."))
    x = torch.from_numpy(g[f"{tag}.x"]).cuda()
    m.eval()
    with torch.no_grad():
        y_eval = m(x)
    assert y_eval.shape == (x.shape[0], 32)
    assert rel_err(y_eval.cpu(), g[f"{tag}.y_eval"]) < 1e-4
    m.train()
    y = m(x)
    assert rel_err(y.detach().cpu(), g[f"{tag}.y_train"]) < 1e-4
    y.square().sum().backward()
    # BN1's gamma/beta gradients are ~0 by scale invariance (BN2 follows): compare against the gradient scale
    scale = max(float(np.abs(g[k]).max()) for k in g.files if k.startswith(f"{tag}.grad."))
    for k, p in m.named_parameters():
        want = g[f"{tag}.grad.{k}"]
        tol = 1e-4 * max(float(np.abs(want).max()), 1e-3 * scale)
        assert np.abs(p.grad.cpu().numpy() - want).max() < tol + 1e-7, k
    sd_after = _sd(g, f"{tag}.sd_after.")
    for k, v in m.state_dict().items():
        if "running" in k:
            assert rel_err(v.cpu(), sd_after[k]) < 1e-4, k
        if "num_batches_tracked" in k:
            assert int(v) == int(sd_after[k])


@pytest.mark.parametrize("C,T,K,B", [(4, 64, 64, 3), (9, 333, 32, 5), (64, 512, 64, 4), (300, 40, 16, 2),
                                     # rows of at most 79 samples take the one-Gram-matrix statistics path
                                     (5, 65, 64, 6), (3, 79, 64, 5), (7, 30, 16, 9), (6, 80, 64, 5), (130, 47, 32, 3),
                                     # long rows: several segments of the backward correlation kernel
                                     (3, 2100, 64, 2), (2, 1024, 64, 2),
                                     # wide inputs: whole-row spatial product (5-tile groups), spatial weight gradient
                                     # in 80-step chunks with dword-aligned 16-byte loads and a one-step last chunk
                                     (200, 81, 32, 2), (515, 100, 16, 2), (129, 161, 64, 2)])
def test_eegnet_vs_oracle_shapes(inn, C, T, K, B):
    torch.manual_seed(C + T)
    m = inn.EEGNet_Encoder(C, 16, kernel_length=K, dropout=0.0).cuda()
    with torch.no_grad():
        for bn in m._bns():
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.uniform_(-0.3, 0.3)
    p = {k: v.detach().cpu().clone().double() for k, v in m.state_dict().items()}
    for k, v in p.items():
        if "running" not in k and "num_batches" not in k:
            v.requires_grad_()
    x = torch.randn(B, C, T)
    w = torch.randn(B, 16)
    m.train()
    y = m(x.cuda())
    (y * w.cuda()).sum().backward()
    yr = ocnn.eegnet_encoder(x.double(), p, training=True, kernel_length=K)
    (yr * w.double()).sum().backward()
    assert rel_err(y.detach().cpu(), yr.detach()) < 1e-4
    scale = max(float(v.grad.abs().max()) for v in p.values() if v.grad is not None)
    for k, q in m.named_parameters():
        want = p[k].grad
        # BN1's gamma/beta gradients vanish up to eps effects (BN2 renormalises): they are fp32 cancellation
        # residue ~1e-6 of the gradient scale in the reference too, so they are compared on that scale
        floor = 5e-2 if k.startswith("temporal_conv.1.") else 1e-3
        tol = 1e-4 * max(float(want.abs().max()), floor * scale)
        assert float((q.grad.cpu().double() - want).abs().max()) < tol + 1e-7, k


def test_eegnet_dropout_is_unbiased_and_deterministic_per_seed(inn):
    torch.manual_seed(0)
    m = inn.EEGNet_Encoder(8, 32, dropout=0.25).cuda().train()
    x = torch.randn(64, 8, 256, device="cuda")
    m.p = 0.0
    ref = m(x).detach()
    m.p = 0.25
    outs = torch.stack([m(x).detach() for _ in range(24)])
    assert not torch.equal(outs[0], outs[1])                       # fresh mask per call
    assert float((outs.mean(0) - ref).abs().mean() / ref.abs().mean()) < 0.2
    with pytest.raises(Exception):
        inn.EEGNet_Encoder(8, 32).cuda()(torch.randn(2, 8, 20, device="cuda"))   # too short for the pooling


@pytest.mark.parametrize("kind,C,T,K,B", [("eegnet", 6, 250, 64, 3), ("eegnet", 9, 333, 32, 5), ("eegnet", 3, 700, 64, 2),
                                          ("eegnet", 5, 65, 64, 4), ("cvblock", 7, 250, 0, 3), ("paper", 6, 250, 0, 4),
                                          ("paper", 17, 300, 0, 2)])
@pytest.mark.parametrize("train", [True, False])
def test_bn_heads_input_gradient_vs_oracle(inn, kind, C, T, K, B, train):
    """d loss / d x of EEGNet_Encoder, CVBlock and HeadConv_Paper_Version (the head contract of fast.py:203-210 asks for an autograd-differentiable
    encoder; the attribution scripts differentiate w.r.t. the trials): train mode -- BatchNorm's batch-mean / variance
    paths included -- and eval mode (running statistics: the backward follows an eval-mode forward that kept its
    activations), with the parameter gradients of both, against the oracle's autograd in fp64."""
    torch.manual_seed(C + T + int(train))
    if kind == "eegnet":
        m = inn.EEGNet_Encoder(C, 16, kernel_length=K, dropout=0.0).cuda()
        ref = lambda xx, pp, tr: ocnn.eegnet_encoder(xx, pp, training=tr, kernel_length=K)
    elif kind == "cvblock":
        m = inn.CVBlock(C, 16, dropout=0.0).cuda()
        ref = lambda xx, pp, tr: ocnn.cvblock(xx, pp, training=tr)
    else:
        m = inn.HeadConv_Paper_Version(C, 16).cuda()
        ref = lambda xx, pp, tr: ocnn.headconv_paper(xx, pp, training=tr)
    with torch.no_grad():
        for bn in m._bns():
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.uniform_(-0.3, 0.3)
            bn.running_mean.uniform_(-0.2, 0.2)
            bn.running_var.uniform_(0.5, 2.0)
    p = {k: v.detach().cpu().clone().double() for k, v in m.state_dict().items()}
    for k, v in p.items():
        if "running" not in k and "num_batches" not in k:
            v.requires_grad_()
    x = torch.randn(B, C, T)
    w = torch.randn(B, 16)
    m.train(train)
    xg = x.cuda().requires_grad_()
    y = m(xg)
    (y * w.cuda()).sum().backward()
    xr = x.double().requires_grad_()
    yr = ref(xr, p, train)
    (yr * w.double()).sum().backward()
    assert rel_err(y.detach().cpu(), yr.detach()) < 1e-4
    assert rel_err(xg.grad.cpu(), xr.grad) < 2e-4
    scale = max(float(v.grad.abs().max()) for v in p.values() if v.grad is not None)
    for k, q in m.named_parameters():
        assert _grad_err(q.grad.cpu().numpy(), p[k].grad.numpy(), scale) < 2e-4, k
    # the parameter-only backward (no input gradient requested) is unchanged by the new path
    m.zero_grad(set_to_none=True)
    if train:
        (m(x.cuda()) * w.cuda()).sum().backward()
        for k, q in m.named_parameters():
            assert _grad_err(q.grad.cpu().numpy(), p[k].grad.numpy(), scale) < 2e-4, k
