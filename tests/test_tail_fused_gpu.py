"""GPU parity of the one-launch transformer tail (csrc/tailfused.hip: cls/pos embedding -> AttentionBlocks -> cls
dropout -> last_layer, forward and backward) against the per-operator HIP path, the torch modules it replaces
(nn.LayerNorm / nn.MultiheadAttention / nn.Linear, fast.py:10-29, 260-268) and, with dropout on, its own directional
derivative.  The reference goldens of this mode (G5 gradients, G6 logits) run through it in test_cnn_gpu.py."""
import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def inn():
    import isd_amd.nn as m
    assert torch.cuda.is_available()
    return m


def _torch_tail(m, feature):
    """fast.py:260-268 with the stock torch modules on the same parameters (fp64 on the CPU)."""
    import torch.nn.functional as F
    sd = {k: v.detach().cpu().double() for k, v in m.state_dict().items()}
    p = {k: v.clone().requires_grad_() for k, v in sd.items()}
    B, N, Z, Fd = feature.shape
    x = feature.detach().cpu().double().reshape(B, N, Z * Fd)
    tok = F.gelu(F.linear(x, p["input_layer.0.weight"], p["input_layer.0.bias"]))
    tok = torch.cat([p["cls_token"].expand(B, -1, -1), tok], dim=1) + p["pos_embedding"][:, :N + 1]
    D, H = m.config.dim_token, m.config.num_heads
    for l in range(len(m.transformer)):
        q = f"transformer.{l}."
        h = F.layer_norm(tok, (D,), p[q + "layer_norm_1.weight"], p[q + "layer_norm_1.bias"])
        a, _ = F.multi_head_attention_forward(
            h.transpose(0, 1), h.transpose(0, 1), h.transpose(0, 1), D, H, p[q + "attn.in_proj_weight"],
            p[q + "attn.in_proj_bias"], None, None, False, 0.0, p[q + "attn.out_proj.weight"],
            p[q + "attn.out_proj.bias"], training=False, need_weights=False)
        tok = tok + a.transpose(0, 1)
        h = F.layer_norm(tok, (D,), p[q + "layer_norm_2.weight"], p[q + "layer_norm_2.bias"])
        h = F.gelu(F.linear(h, p[q + "linear.0.weight"], p[q + "linear.0.bias"]))
        tok = tok + F.linear(h, p[q + "linear.3.weight"], p[q + "linear.3.bias"])
    return F.linear(tok[:, 0], p["last_layer.weight"], p["last_layer.bias"]), p


@pytest.mark.parametrize("B,N,kw", [
    (5, 5, {}),                                                          # production: D = 32, 4 blocks, 8 heads, 6 tokens
    (23, 7, dict(seq_len=1000)),                                         # 8 tokens: a wave holds 8 trials, 3 waves
    (9, 2, dict(dim_token=16, num_heads=2, num_layers=2, n_classes=3)),  # head_dim 8, D = 16
    (3, 0, {}),                                                          # the cls token alone
])
def test_fused_tail_matches_torch_modules_and_per_operator_path(inn, B, N, kw):
    torch.manual_seed(B)
    m = inn.FAST(inn.fast_config(dropout=0.0, **kw)).cuda()
    assert N <= m.n_tokens
    feature = torch.randn(B, N, 8, 32, device="cuda")
    y = torch.randint(0, m.config.n_classes, (B,), device="cuda")
    want, p = _torch_tail(m, feature)
    torch.nn.functional.cross_entropy(want, y.cpu()).backward()
    grads = {}
    for fused in (True, False):
        m.fuse_tail = fused
        m.zero_grad(set_to_none=True)
        f = feature.clone().requires_grad_()
        logits = m.forward_transformer(f)
        assert logits.shape == (B, m.config.n_classes)
        assert rel_err(logits.detach().cpu(), want.detach()) < 1e-4, fused
        torch.nn.functional.cross_entropy(logits, y).backward()
        grads[fused] = {k: q.grad.detach().cpu().clone() for k, q in m.named_parameters() if q.grad is not None}
        grads[fused]["feature"] = f.grad.detach().cpu().clone()
        tail = [k for k in grads[fused] if not k.startswith("head.") and k != "feature"]
        assert len(tail) == 2 + 2 + 2 + 12 * len(m.transformer)
        for k in tail:
            assert rel_err(grads[fused][k], p[k].grad) < 2e-4, (fused, k)
    if N:
        assert rel_err(grads[True]["feature"], grads[False]["feature"]) < 1e-4
    if N < m.n_tokens:                                                   # unused positional rows get zero gradient
        assert float(grads[True]["pos_embedding"][:, N + 1:].abs().max()) == 0.0
    with torch.no_grad():                                                # inference: same launch without the record
        m.fuse_tail = True
        assert rel_err(m.forward_transformer(feature).cpu(), want.detach()) < 1e-4


def test_fused_tail_full_batch_properties(inn):
    """B = 4096 (512 waves, one partial block each): trials are independent, the gradient of the batch is the mean of
    its halves' gradients, and repeated runs are bitwise equal (the slabs are summed in a fixed order)."""
    torch.manual_seed(1)
    m = inn.FAST(inn.fast_config(dropout=0.0)).cuda()
    B = 4096
    feature = torch.randn(B, 5, 8, 32, device="cuda")
    y = torch.randint(0, 5, (B,), device="cuda")

    def run(sl):
        m.zero_grad(set_to_none=True)
        logits = m.forward_transformer(feature[sl])
        torch.nn.functional.cross_entropy(logits, y[sl]).backward()
        return logits.detach(), torch.cat([q.grad.reshape(-1) for q in m._tail_params()]).clone()
    lg, gfull = run(slice(0, B))
    lg2, gfull2 = run(slice(0, B))
    assert torch.equal(lg, lg2) and torch.equal(gfull, gfull2)
    l0, g0 = run(slice(0, B // 2))
    l1, g1 = run(slice(B // 2, B))
    assert torch.equal(lg[:B // 2], l0) and torch.equal(lg[B // 2:], l1)
    assert float((gfull - 0.5 * (g0 + g1)).abs().max() / gfull.abs().max()) < 1e-4
    want, _ = _torch_tail(m, feature[1000:1003])
    assert rel_err(lg[1000:1003].cpu(), want.detach()) < 1e-4


def test_fused_tail_dropout_masks_and_directional_derivative(inn):
    """Training mode, dropout 0.3: masks are counter-based -- the same call index reproduces the logits, the next one
    does not -- and the analytic gradient equals the central
    difference of the loss along the gradient direction (same masks on both sides)."""
    torch.manual_seed(2)
    m = inn.FAST(inn.fast_config(dropout=0.3)).cuda().train()
    B = 64
    feature = torch.randn(B, 5, 8, 32, device="cuda")
    y = torch.randint(0, 5, (B,), device="cuda")
    ps = m._tail_params() + [m.input_layer[0].weight, m.input_layer[0].bias]

    def loss_at(call):
        m._tail_calls = call
        return torch.nn.functional.cross_entropy(m.forward_transformer(feature).double(), y)
    m.zero_grad(set_to_none=True)
    l0 = loss_at(7)
    l0.backward()
    g = [q.grad.detach().clone() for q in ps]
    with torch.no_grad():
        assert float(loss_at(7)) == float(l0) and float(loss_at(8)) != float(l0)
        m.eval()
        le = float(loss_at(7))
        m.train()
        assert le != float(l0)
        gnorm = float(torch.sqrt(sum((a.double() ** 2).sum() for a in g)))
        v = [a / gnorm for a in g]                       # unit step along the gradient: the derivative there is |g|
        eps = 1e-2
        for q, d in zip(ps, v):
            q.add_(eps * d)
        lp = float(loss_at(7))
        for q, d in zip(ps, v):
            q.sub_(2 * eps * d)
        lm = float(loss_at(7))
        for q, d in zip(ps, v):
            q.add_(eps * d)
    fd = (lp - lm) / (2 * eps)
    an = float(sum((a.double() * d.double()).sum() for a, d in zip(g, v)))
    assert abs(fd - an) < 2e-2 * abs(an) + 1e-5, (fd, an)


def test_fused_tail_rejects_bad_arguments(inn):
    import isd_amd._lib as L
    lib = L.lib()
    assert lib.isd_tail_fused_supported(5, 32, 8, 4, 64, 5) == 1
    assert lib.isd_tail_fused_supported(8, 32, 8, 4, 64, 5) == 0        # 9 tokens
    assert lib.isd_tail_fused_supported(5, 64, 8, 4, 128, 5) == 0       # dim_token 64: per-operator path
    assert lib.isd_tail_fused_supported(5, 32, 2, 4, 64, 5) == 0        # head_dim 16
    assert lib.isd_tail_fused_param_count(6, 32, 4, 5) == 6 * 32 + 32 + 4 * 8544 + 5 * 32 + 5
    t = torch.zeros(8, device="cuda")
    with pytest.raises(L.IsdError):
        L.check(lib.isd_tail_fused_forward(t.data_ptr(), t.data_ptr(), t.data_ptr(), 0, 0, 1, 8, 9, 32, 8, 4, 64, 5,
                                           0.0, 0.0, 0.0, 0, 0, 0))
    with pytest.raises(L.IsdError):
        L.check(lib.isd_tail_fused_forward(t.data_ptr(), t.data_ptr(), t.data_ptr(), 0, 0, 1, 5, 6, 32, 8, 4, 64, 5,
                                           1.0, 0.0, 0.0, 0, 0, 0))
    # a model outside the fused kernel's shapes keeps working through the per-operator path
    m = inn.FAST(inn.fast_config(dim_token=64, num_heads=8)).cuda()
    assert not m._tail_fusable(torch.zeros(2, 5, 64, device="cuda"))
    assert m.forward_transformer(torch.randn(2, 5, 8, 32, device="cuda")).shape == (2, 5)
