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


def _host_masks(seed, B, S, D, H, L, p_blk, p_cls):
    """The counter-based dropout masks of csrc/tailfused.hip (tf_drop / tf_keep) recomputed on the host: per layer the
    attention-probability mask [B, H, S, S], the two MLP masks [B, S, 2D] / [B, S, D], and the cls mask [B, D]; every
    entry is 0 or 1 / (1 - p) in fp32, exactly what the kernels multiply by."""
    M64 = (1 << 64) - 1

    def key(layer, site):
        v = (seed + 0x9E3779B97F4A7C15 * (layer * 8 + site + 1)) & M64
        v ^= v >> 30
        v = (v * 0xBF58476D1CE4E5B9) & M64
        v ^= v >> 27
        v = (v * 0x94D049BB133111EB) & M64
        v ^= v >> 31
        return (v ^ (v >> 32)) & 0xFFFFFFFF

    def keep(k, e, p):
        e = e.astype(np.uint64)
        lo, hi = (e & np.uint64(0xFFFFFFFF)).astype(np.uint32), (e >> np.uint64(32)).astype(np.uint32)
        h = ((lo ^ (hi * np.uint32(0x27D4EB2F))) * np.uint32(0x9E3779B1)) ^ np.uint32(k)
        h ^= h >> np.uint32(16)
        h = h * np.uint32(0x85EBCA6B)
        h ^= h >> np.uint32(13)
        h = h * np.uint32(0xC2B2AE35)
        h ^= h >> np.uint32(16)
        u = (h >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
        inv = np.float32(1.0) / (np.float32(1.0) - np.float32(p))
        return np.where(u >= np.float32(p), inv, np.float32(0.0)).astype(np.float64)

    tok = (np.arange(B)[:, None] * S + np.arange(S)[None, :]).astype(np.int64)          # [B, S]
    out = []
    with np.errstate(over="ignore"):
        for l in range(L):
            e = (tok[:, None, :, None] * H + np.arange(H)[None, :, None, None]) * 8 + np.arange(S)[None, None, None, :]
            out.append((keep(key(l, 0), e, p_blk),
                        keep(key(l, 1), tok[:, :, None] * (2 * D) + np.arange(2 * D)[None, None, :], p_blk),
                        keep(key(l, 2), tok[:, :, None] * D + np.arange(D)[None, None, :], p_blk)))
        cls = keep(key(8, 0), np.arange(B)[:, None] * D + np.arange(D)[None, :], p_cls)
    return out, cls


def _torch_tail_masked(m, feature, masks, cls_mask):
    """fast.py:10-29, 260-268 written out in fp64 torch with the dropout masks given: attention probabilities,
    MLP hidden, MLP output, cls token (the four nn.Dropout / MultiheadAttention(dropout=) sites of the reference)."""
    import torch.nn.functional as F
    p = {k: v.detach().cpu().double().clone().requires_grad_() for k, v in m.state_dict().items()}
    B, N, Z, Fd = feature.shape
    x = feature.detach().cpu().double().reshape(B, N, Z * Fd).requires_grad_()
    tok = F.gelu(F.linear(x, p["input_layer.0.weight"], p["input_layer.0.bias"]))
    tok = torch.cat([p["cls_token"].expand(B, -1, -1), tok], dim=1) + p["pos_embedding"][:, :N + 1]
    D, H = m.config.dim_token, m.config.num_heads
    S, dh = N + 1, D // H
    for l, (ma, m1, m2) in enumerate(masks):
        q = f"transformer.{l}."
        h = F.layer_norm(tok, (D,), p[q + "layer_norm_1.weight"], p[q + "layer_norm_1.bias"])
        qkv = F.linear(h, p[q + "attn.in_proj_weight"], p[q + "attn.in_proj_bias"]).view(B, S, 3, H, dh)
        qq, kk, vv = (qkv[:, :, t].transpose(1, 2) for t in range(3))                    # [B, H, S, dh]
        pr = torch.softmax(qq @ kk.transpose(-1, -2) / dh ** 0.5, dim=-1) * torch.from_numpy(ma)
        ctx = (pr @ vv).transpose(1, 2).reshape(B, S, D)
        tok = tok + F.linear(ctx, p[q + "attn.out_proj.weight"], p[q + "attn.out_proj.bias"])
        h = F.layer_norm(tok, (D,), p[q + "layer_norm_2.weight"], p[q + "layer_norm_2.bias"])
        h = F.gelu(F.linear(h, p[q + "linear.0.weight"], p[q + "linear.0.bias"])) * torch.from_numpy(m1)
        tok = tok + F.linear(h, p[q + "linear.3.weight"], p[q + "linear.3.bias"]) * torch.from_numpy(m2)
    cls = tok[:, 0] * torch.from_numpy(cls_mask)
    return F.linear(cls, p["last_layer.weight"], p["last_layer.bias"]), p, x


@pytest.mark.parametrize("B,N,kw", [
    (64, 5, {}),                                                         # production shape, 10 trials per workgroup
    (23, 7, dict(seq_len=1000)),                                         # 8 tokens, ragged last workgroup
    (9, 2, dict(dim_token=16, num_heads=2, num_layers=2, n_classes=3)),  # head width 8, D = 16
])
def test_fused_tail_under_dropout_matches_torch_autograd_with_the_same_masks(inn, B, N, kw):
    """Training mode, dropout 0.3: the kernels' counter-based masks are recomputed on the host
    from the launch's seed, the reference network is evaluated in fp64 torch with exactly those masks, and the logits,
    EVERY parameter gradient of the tail (52 tensors at 4 blocks), the token-projection gradients and the gradient
    w.r.t. the zone features must agree per tensor -- the same bound as the dropout-free comparison above."""
    from isd_amd.nn import _dropout_seed
    torch.manual_seed(20 + B)
    m = inn.FAST(inn.fast_config(dropout=0.3, **kw)).cuda().train()
    feature = torch.randn(B, N, 8, 32, device="cuda")
    y = torch.randint(0, m.config.n_classes, (B,), device="cuda")
    m.zero_grad(set_to_none=True)
    m._tail_calls = 40
    f = feature.clone().requires_grad_()
    logits = m.forward_transformer(f)
    torch.nn.functional.cross_entropy(logits, y).backward()
    seed = _dropout_seed(m._tail_stream, m._tail_calls)
    c = m.config
    masks, cls_mask = _host_masks(seed, B, N + 1, c.dim_token, c.num_heads, len(m.transformer), 0.3, 0.3)
    frac = float(np.mean([float((a == 0).mean()) for ms in masks for a in ms]))
    assert 0.25 < frac < 0.35                                           # the masks do drop ~30 %
    want, p, xr = _torch_tail_masked(m, feature, masks, cls_mask)
    torch.nn.functional.cross_entropy(want, y.cpu()).backward()
    assert rel_err(logits.detach().cpu(), want.detach()) < 1e-4
    named = dict(m.named_parameters())
    tail = [k for k in named if not k.startswith("head.")]
    assert len(tail) == 2 + 2 + 2 + 12 * len(m.transformer)
    for k in tail:
        assert rel_err(named[k].grad.cpu(), p[k].grad) < 2e-4, k
    assert rel_err(f.grad.cpu().reshape(B, N, -1), xr.grad) < 2e-4
    # eval mode draws nothing
    m.eval()
    with torch.no_grad():
        le = m.forward_transformer(feature)
    ones = [tuple(np.ones_like(a) for a in ms) for ms in masks]
    we, _, _ = _torch_tail_masked(m, feature, ones, np.ones_like(cls_mask))
    assert rel_err(le.cpu(), we.detach()) < 1e-4


def test_fused_tail_dropout_masks_and_directional_derivative(inn):
    """Training mode, dropout 0.3: masks are counter-based -- the same call index reproduces the logits, the next one
    does not -- and the analytic gradient equals the central difference of the loss along RANDOM directions (same masks
    on both sides; fp64 loss, a step small enough that the cubic term is below the bound)."""
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
        gen = torch.Generator(device="cuda").manual_seed(11)
        for trial in range(3):
            # random direction, each tensor's step relative to its own scale (a uniform absolute step on every
            # parameter is dominated by the curvature of the smallest tensors)
            v = [torch.randn(q.shape, device="cuda", generator=gen) * q.abs().mean().clamp_min(1e-3) for q in ps]
            errs = []
            for eps in (4e-3, 2e-3):
                for q, d in zip(ps, v):
                    q.add_(eps * d)
                lp = float(loss_at(7))
                for q, d in zip(ps, v):
                    q.sub_(2 * eps * d)
                lm = float(loss_at(7))
                for q, d in zip(ps, v):
                    q.add_(eps * d)
                errs.append((lp - lm) / (2 * eps))
            an = float(sum((a.double() * d.double()).sum() for a, d in zip(g, v)))
            fd = (4 * errs[1] - errs[0]) / 3                               # Richardson: removes the eps^2 term
            assert abs(fd - an) < 2e-2 * abs(an) + 2e-5, (trial, errs, fd, an)


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
