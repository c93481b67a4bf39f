"""GPU parity: HIP Conv4Layers stack / FC head / softmax-CE (through the C ABI) vs the oracle and
the golden vectors captured from the reference model."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from oracle import cnn as ocnn

pytestmark = pytest.mark.gpu

TOL = 1e-4      # north star: logits/features within 1e-4 rel fp32


@pytest.fixture(scope="module")
def inn():
    import isd_amd.nn as m
    assert torch.cuda.is_available()
    return m


def _sd(g, prefix):
    return {k[len(prefix):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(prefix)}


# ------------------------------------------------------------------ Conv4Layers (G4, from the reference)
@pytest.mark.parametrize("cz", [6, 15])
def test_conv4layers_matches_reference_golden(inn, cz):
    g = load_golden("g4_conv4layers.npz")
    m = inn.Conv4Layers(cz, 32).cuda()
    m.load_state_dict(_sd(g, f"c{cz}.sd."))
    x = torch.from_numpy(g[f"c{cz}.x"]).cuda()
    y = m(x)
    assert y.shape == (3, 32)
    assert rel_err(y.detach().cpu(), g[f"c{cz}.y"]) < TOL
    y.square().sum().backward()
    for k, p in m.named_parameters():
        assert rel_err(p.grad.cpu(), g[f"c{cz}.grad.{k}"]) < TOL, k
    # the gradient w.r.t. the input (the head contract asks for an autograd-differentiable encoder, fast.py:203-210)
    m.zero_grad(set_to_none=True)
    xg = x.clone().requires_grad_()
    m(xg).square().sum().backward()
    assert rel_err(xg.grad.cpu(), g[f"c{cz}.dx"]) < TOL
    for k, p in m.named_parameters():
        assert rel_err(p.grad.cpu(), g[f"c{cz}.grad.{k}"]) < TOL, k      # same parameter gradients on this path


def test_input_gradient_through_windows_and_zones_vs_oracle(inn):
    """dL/dx of FAST's CNN head on the reference zones: overlapping 250-sample windows add, zones own their channels."""
    import isd_amd
    torch.manual_seed(12)
    h = inn.Head("Conv4Layers", isd_amd.ELECTRODES, isd_amd.ZONES, 32).cuda()
    x = torch.randn(2, 64, 512)
    w = torch.randn(2 * 3, 8, 32)
    xg = x.cuda().requires_grad_()
    (h.forward_windows(xg, 250, 125) * w.cuda()).sum().backward()
    p = {"head." + k: v.detach().cpu().double().requires_grad_() for k, v in h.state_dict().items()}
    xr = x.double().requires_grad_()
    ref = ocnn.forward_head(xr, p, list(ocnn.ZONES), ocnn.zone_index_lists(), 250, 125)
    (ref.reshape(6, 8, 32) * w.double()).sum().backward()
    assert rel_err(xg.grad.cpu(), xr.grad) < TOL
    for k, q in h.named_parameters():
        assert rel_err(q.grad.cpu(), p["head." + k].grad) < TOL, k
    # the whole model, trained mode: logits -> input
    m = inn.FAST(inn.fast_config(dropout=0.0)).cuda().eval()
    x8 = torch.randn(2, 64, 800, device="cuda", requires_grad=True)
    m(x8).sum().backward()
    assert x8.grad is not None and x8.grad.shape == x8.shape and torch.isfinite(x8.grad).all() and x8.grad.abs().sum() > 0


@pytest.mark.parametrize("channels,T,dim,n_layers,B", [(4, 250, 16, 4, 5), (40, 64, 32, 4, 3), (70, 21, 32, 2, 9),
                                                      (576, 17, 32, 4, 20), (3, 300, 32, 4, 2), (1, 9, 16, 2, 1),
                                                      # wide inputs (LDS-DMA kernels): 16 filters, partial last chunk /
                                                      # channel group, 2-layer stack, one item, many items per workgroup
                                                      (128, 17, 16, 4, 7), (100, 20, 32, 2, 33), (320, 9, 32, 4, 1),
                                                      (68, 17, 16, 2, 300), (200, 40, 32, 4, 5)])
def test_conv4layers_vs_oracle_shapes(inn, channels, T, dim, n_layers, B):
    p = ocnn.init_conv4_params(channels, dim, seed=channels + T, n_layers=n_layers)
    m = inn.Conv4Layers(channels, dim, n_layers).cuda()
    m.load_state_dict(p)
    x = torch.randn(B, channels, T, generator=torch.Generator().manual_seed(1))
    w = torch.randn(B, dim, generator=torch.Generator().manual_seed(2))
    y = m(x.cuda())
    (y * w.cuda()).sum().backward()
    pr = {k: v.clone().double().requires_grad_() for k, v in p.items()}
    yr = ocnn.conv4layers(x.double(), pr, n_layers=n_layers)
    (yr * w.double()).sum().backward()
    assert rel_err(y.detach().cpu(), yr.detach()) < TOL
    for k, q in m.named_parameters():
        assert rel_err(q.grad.cpu(), pr[k].grad) < TOL, k
    # input gradient on the same shapes (layer-wise backward + conv5_dx_kernel)
    xg = x.cuda().requires_grad_()
    (m(xg) * w.cuda()).sum().backward()
    xr = x.double().requires_grad_()
    (ocnn.conv4layers(xr, {k: v.detach() for k, v in pr.items()}, n_layers=n_layers) * w.double()).sum().backward()
    assert rel_err(xg.grad.cpu(), xr.grad) < TOL


# ------------------------------------------------------------------ FAST train_head (G5 / G6, from the reference)
def _small_cfg(inn):
    electrodes = ["Fp1", "Fp2", "F3", "F4", "C3", "C4", "O1", "O2"]
    zones = {"Frontal": ["Fp1", "Fp2", "F3", "F4"], "Central": ["C3", "C4"], "Occipital": ["O1", "O2"]}
    return inn.fast_config(electrodes, zones, dim_cnn=16, dim_token=16, seq_len=500, n_classes=3, num_layers=1,
                           num_heads=4, dropout=0.0)


def test_fast_small_train_head_matches_reference_golden(inn):
    g = load_golden("g5_fast_small.npz")
    m = inn.FAST(_small_cfg(inn)).cuda()
    m.load_state_dict(_sd(g, "sd."))                         # strict: names/shapes equal the reference's
    assert m.n_tokens == 3
    x = torch.from_numpy(g["x"]).cuda()
    feat = m.forward_head(x)
    assert feat.shape == (2, 3, 3, 16)
    assert rel_err(feat.detach().cpu(), g["features"]) < TOL
    lt = m.token_logits(x)
    loss = inn.token_mean_cross_entropy(lt, torch.from_numpy(g["labels"]).cuda())
    loss.backward()
    assert rel_err(m(x, forward_mode="train_head").detach().cpu(), g["train_head.logits"]) < TOL
    assert abs(float(loss) - float(g["train_head.loss"])) < 1e-5
    for k, p in m.named_parameters():
        if f"train_head.grad.{k}" in g.files:
            # measured 4.7e-7 worst (tools/grad_slack_probe.py, profiles/r04_grad_slack.txt); the golden's own fp32
            # error against an fp64 run of the reference is 5e-7.  (2e-4 until round 4: slack nobody had measured.)
            assert rel_err(p.grad.cpu(), g[f"train_head.grad.{k}"]) < 1e-5, k
    # forward_mode='default': the mode the reference trains (trainer.py:58) -- transformer tail included
    m.zero_grad(set_to_none=True)
    logits = m(x)
    loss = torch.nn.functional.cross_entropy(logits, torch.from_numpy(g["labels"]).long().cuda())
    loss.backward()
    assert rel_err(logits.detach().cpu(), g["default.logits"]) < TOL
    assert abs(float(loss.detach()) - float(g["default.loss"])) < 1e-5
    for k, p in m.named_parameters():
        assert rel_err(p.grad.cpu(), g[f"default.grad.{k}"]) < 1e-5, k          # measured 6.1e-7 worst (5e-4 until round 4)
    # 'train_transformer': the CNN head is frozen (no gradient reaches it)
    m.zero_grad(set_to_none=True)
    m(x, forward_mode="train_transformer").sum().backward()
    assert all(p.grad is None for k, p in m.named_parameters() if k.startswith("head."))
    assert m.transformer[0].attn.in_proj_weight.grad is not None


def _bf16_small_run(inn, g):
    """FAST(small_config, act_dtype='bf16') on G5's parameters and input: forward_head features, and logits / loss /
    gradients of both trained modes."""
    cfg = _small_cfg(inn)
    cfg.act_dtype = "bf16"
    m = inn.FAST(cfg).cuda()
    m.load_state_dict(_sd(g, "sd."))
    x = torch.from_numpy(g["x"]).cuda()
    y = torch.from_numpy(g["labels"]).long().cuda()
    out = {"features": m.forward_head(x).detach().cpu().numpy()}
    for mode in ("train_head", "default"):
        m.zero_grad(set_to_none=True)
        logits = m(x, forward_mode=mode)
        loss = torch.nn.functional.cross_entropy(logits, y)
        loss.backward()
        out[f"{mode}.logits"], out[f"{mode}.loss"] = logits.detach().cpu().numpy(), float(loss.detach())
        for k, p in m.named_parameters():
            if p.grad is not None:
                out[f"{mode}.grad.{k}"] = p.grad.detach().cpu().numpy()
    return out


def test_fast_small_bf16_vs_reference_autocast_golden(inn):
    """BASELINE config 3 for the modes the reference trains: FAST(small_config) with bf16 zone-CNN
    activations against G16 = the reference under torch.autocast(bfloat16) (scripts/train_fast.py:277) and against its
    fp32 run G5.  The reference's own autocast run deviates from its fp32 run by 5.7e-3 (features), 2.0e-3 (logits) and
    up to 1.3e-2 (gradients, of each tensor's largest magnitude); the stated tolerance here is twice that, against
    either reference.  (The HIP path keeps the transformer tail, the token projection and every parameter in fp32:
    it is closer to the fp32 run than autocast, which also rounds those to bf16.)"""
    g5, g16 = load_golden("g5_fast_small.npz"), load_golden("g16_fast_small_autocast.npz")
    got = _bf16_small_run(inn, g5)
    for name, ref in (("autocast", g16), ("fp32", g5)):
        assert 0 < rel_err(got["features"], ref["features"]) < 1.2e-2, name
        for mode in ("train_head", "default"):
            assert rel_err(got[f"{mode}.logits"], ref[f"{mode}.logits"]) < 4e-3, (name, mode)
            assert abs(got[f"{mode}.loss"] - float(ref[f"{mode}.loss"])) < 3e-3, (name, mode)
            keys = [k for k in ref.files if k.startswith(f"{mode}.grad.")]
            assert len(keys) >= 19 and all(k in got for k in keys)
            for k in keys:
                assert rel_err(got[k], ref[k]) < 2.6e-2, (name, k)
            assert np.array_equal(got[f"{mode}.logits"].argmax(1), ref[f"{mode}.logits"].argmax(1))


def test_fast_prod_eval_logits_bitexact_argmax(inn):
    g = load_golden("g6_fast_prod.npz")
    m = inn.FAST(inn.fast_config()).cuda().eval()
    m.load_state_dict(_sd(g, "sd."))
    x = torch.from_numpy(np.random.default_rng(6).standard_normal((4, 64, 800)).astype(np.float32)).cuda()
    with torch.no_grad():
        feat = m.forward_head(x)
        logits = m(x, forward_mode="train_head")
        lm, pred = inn.token_mean_predict(m.token_logits(x))
        logits_default = m(x)                                             # forward_mode='default' (transformer tail)
    assert feat.shape == (4, 5, 8, 32)
    assert rel_err(feat.cpu(), g["features"]) < TOL
    assert rel_err(logits.cpu(), g["train_head_logits"]) < TOL
    assert np.array_equal(pred.cpu().numpy(), g["train_head_pred"])       # class indices bit-exact
    assert pred.dtype == torch.int64
    assert rel_err(logits_default.cpu(), g["default_logits"]) < TOL
    assert np.array_equal(logits_default.argmax(1).cpu().numpy(), g["default_pred"])
    with pytest.raises(NotImplementedError):
        m(x, forward_mode="nonexistent")


def test_head_matches_oracle_on_reference_zones(inn):
    # Head contract: [B', 64, 250] -> [B', 8, 32], zone order = dict order
    g = load_golden("g6_fast_prod.npz")
    sd = {k[len("head."):]: v for k, v in _sd(g, "sd.").items() if k.startswith("head.")}
    import isd_amd
    h = inn.Head("Conv4Layers", isd_amd.ELECTRODES, isd_amd.ZONES, 32).cuda()
    h.load_state_dict(sd)
    x = torch.randn(3, 64, 250, generator=torch.Generator().manual_seed(3))
    y = h(x.cuda())
    p = {"head." + k: v for k, v in sd.items()}
    ref = ocnn.head_forward(x, p, list(ocnn.ZONES), ocnn.zone_index_lists())
    assert y.shape == (3, 8, 32)
    assert rel_err(y.detach().cpu(), ref) < TOL
    assert [v.tolist() for v in h.index_dict.values()] == ocnn.zone_index_lists()


def test_head_production_shape_gradients_vs_oracle(inn):
    """Reference zones, F = 32, 250-sample windows sliding by 125: the shape the fused forward / fused backward
    kernels serve.  Features and every parameter gradient against fp64 autograd over the oracle."""
    import isd_amd
    torch.manual_seed(11)
    h = inn.Head("Conv4Layers", isd_amd.ELECTRODES, isd_amd.ZONES, 32).cuda()
    x = torch.randn(3, 64, 512)
    w = torch.randn(3 * 3, 8, 32)
    feat = h.forward_windows(x.cuda(), 250, 125)
    (feat * w.cuda()).sum().backward()
    p = {"head." + k: v.detach().cpu().double().requires_grad_() for k, v in h.state_dict().items()}
    ref = ocnn.forward_head(x.double(), p, list(ocnn.ZONES), ocnn.zone_index_lists(), 250, 125)
    (ref.reshape(9, 8, 32) * w.double()).sum().backward()
    assert feat.shape == (9, 8, 32) and rel_err(feat.detach().cpu(), ref.reshape(9, 8, 32).detach()) < TOL
    for k, q in h.named_parameters():
        assert rel_err(q.grad.cpu(), p["head." + k].grad) < TOL, k


# ------------------------------------------------------------------ FC head and loss
@pytest.mark.parametrize("M,K,N,act", [(37, 256, 32, True), (4096 * 3, 256, 32, True), (100, 32, 5, False),
                                       (1, 7, 3, False), (130, 70, 64, True)])
def test_linear_forward_backward_vs_torch(inn, M, K, N, act):
    gen = torch.Generator().manual_seed(M + K)
    x = torch.randn(M, K, generator=gen)
    w = torch.randn(N, K, generator=gen) / K ** 0.5
    b = torch.randn(N, generator=gen)
    dy = torch.randn(M, N, generator=gen)
    xr, wr, br = (t.clone().double().requires_grad_() for t in (x, w, b))
    yr = torch.nn.functional.linear(xr, wr, br)
    yr = torch.nn.functional.gelu(yr) if act else yr
    (yr * dy.double()).sum().backward()
    xg, wg, bg = (t.clone().cuda().requires_grad_() for t in (x, w, b))
    y = inn.linear(xg, wg, bg, act=act)
    (y * dy.cuda()).sum().backward()
    assert rel_err(y.detach().cpu(), yr.detach()) < 1e-5
    assert rel_err(xg.grad.cpu(), xr.grad) < 1e-5
    assert rel_err(wg.grad.cpu(), wr.grad) < 1e-5
    assert rel_err(bg.grad.cpu(), br.grad) < 1e-5


@pytest.mark.parametrize("B,n_tok,n_cls,dtype", [(2, 3, 3, torch.uint8), (4096, 1, 5, torch.int64), (300, 5, 5, torch.uint8)])
def test_softmax_ce_argmax_vs_torch(inn, B, n_tok, n_cls, dtype):
    gen = torch.Generator().manual_seed(B)
    lt = torch.randn(B, n_tok, n_cls, generator=gen)
    lt[0] = 0.25                                              # all-equal row: argmax tie -> index 0
    y = torch.randint(0, n_cls, (B,), generator=gen).to(dtype)
    ltr = lt.clone().double().requires_grad_()
    lr = ocnn.cross_entropy(ltr.mean(1), y)
    lr.backward()
    ltg = lt.clone().cuda().requires_grad_()
    loss = inn.token_mean_cross_entropy(ltg, y.cuda())
    loss.backward()
    assert abs(float(loss) - float(lr)) < 1e-5
    assert rel_err(ltg.grad.cpu(), ltr.grad) < 1e-5
    lm, pred = inn.token_mean_predict(lt.cuda())
    assert rel_err(lm.cpu(), lt.mean(1)) < 1e-6
    assert np.array_equal(pred.cpu().numpy(), ocnn.predict(lm.cpu()).numpy())
    assert int(pred[0]) == 0


# ------------------------------------------------------------------ feature classifier
@pytest.mark.parametrize("n_layers", [2, 4])
def test_feature_cnn_vs_oracle(inn, n_layers):
    nbC, J, B = 9 * 8, 17, 6
    m = inn.FeatureCNN(nbC, 32, 5, n_layers).cuda()
    p = {k: v.detach().cpu().clone().double().requires_grad_() for k, v in m.state_dict().items()}
    feats = torch.randn(B, 9, 8, J, generator=torch.Generator().manual_seed(4))
    y = torch.tensor([0, 1, 2, 3, 4, 1], dtype=torch.uint8)
    loss = inn.token_mean_cross_entropy(m.token_logits(feats.cuda()), y.cuda())
    loss.backward()
    lr = ocnn.cross_entropy(ocnn.feature_cnn_logits(feats.double(), p, n_layers), y)
    lr.backward()
    assert abs(float(loss) - float(lr)) < 1e-5
    for k, q in m.named_parameters():
        assert rel_err(q.grad.cpu(), p[k].grad) < TOL, k


def test_full_size_feature_cnn_gradient_is_mean_of_shards(inn):
    # BASELINE config 2 size: [4096, 9*64, 17]; property: grad(full batch) == mean of the two half-batch grads
    torch.manual_seed(0)
    m = inn.FeatureCNN(576, 32, 5).cuda()
    feats = torch.randn(4096, 9, 64, 17, device="cuda")
    y = torch.randint(0, 5, (4096,), device="cuda")

    def grads(sl):
        m.zero_grad(set_to_none=True)
        inn.token_mean_cross_entropy(m.token_logits(feats[sl]), y[sl]).backward()
        return torch.cat([p.grad.reshape(-1) for p in m.parameters()]).clone()
    full = grads(slice(0, 4096))
    half = 0.5 * (grads(slice(0, 2048)) + grads(slice(2048, 4096)))
    assert torch.isfinite(full).all()
    assert float((full - half).abs().max() / full.abs().max()) < 1e-4
    # and a 16-trial slice against the oracle
    p = {k: v.detach().cpu().clone().double().requires_grad_() for k, v in m.state_dict().items()}
    ocnn.cross_entropy(ocnn.feature_cnn_logits(feats[:16].cpu().double(), p), y[:16].cpu()).backward()
    got = grads(slice(0, 16))
    ref = torch.cat([p[k].grad.reshape(-1) for k, _ in m.named_parameters()])
    assert rel_err(got.cpu(), ref) < TOL


# ------------------------------------------------------------------ BASELINE config 3: bf16 activations / grads
def test_bf16_activation_mode_tolerance_vs_fp32(inn):
    """bf16 storage of activations + activation gradients, operands rounded to bf16, fp32 accumulate.
    Stated tolerance vs the fp32 path: logits 2e-2 relative, parameter gradients 5e-2 relative."""
    torch.manual_seed(0)
    m32 = inn.FeatureCNN(9 * 8, 32, 5).cuda()
    m16 = inn.FeatureCNN(9 * 8, 32, 5, act_dtype="bf16").cuda()
    m16.load_state_dict(m32.state_dict())
    feats = torch.randn(64, 9, 8, 17, device="cuda") * 2 - 5
    y = torch.randint(0, 5, (64,), device="cuda")
    outs = []
    for m in (m32, m16):
        loss = inn.token_mean_cross_entropy(m.token_logits(feats), y)
        loss.backward()
        outs.append((m(feats).detach(), torch.cat([p.grad.reshape(-1) for p in m.parameters()])))
    (l32, g32), (l16, g16) = outs
    assert 0 < rel_err(l16.cpu(), l32.cpu()) < 2e-2            # different arithmetic, bounded deviation
    assert rel_err(g16.cpu(), g32.cpu()) < 5e-2
    # the same on the zone-wise raw-EEG head
    h32 = inn.Head("Conv4Layers", ocnn.ELECTRODES, ocnn.ZONES, 32).cuda()
    h16 = inn.Head("Conv4Layers", ocnn.ELECTRODES, ocnn.ZONES, 32, act_dtype="bf16").cuda()
    h16.load_state_dict(h32.state_dict())
    x = torch.randn(4, 64, 250, device="cuda")
    a, b = h32(x).detach(), h16(x).detach()
    assert 0 < rel_err(b.cpu(), a.cpu()) < 2e-2


def test_bf16_matrix_core_step_vs_reference_autocast_at_576_channels(inn):
    """BASELINE config 3 at the cfg2 width: the classifier step on the bf16 matrix cores (conv5_fwd_bf16_kernel,
    featcnn_tail_kernel<.., true>, conv5_wgrad_wide_bf16_kernel behind isd_featcnn_step) against G15 = the reference's
    Conv4Layers(576, 32) + Linear under torch.autocast(bfloat16) on CPU (scripts/train_fast.py:277), and against its
    fp32 run.  The reference's own autocast run deviates from its fp32 run by 4.0e-3 (logits) / 7.6e-3 (gradients) of
    the tensor's largest magnitude here; the stated tolerance is TWICE that against either run: logits 8e-3, parameter
    gradients 1.6e-2 (measured, tools/bf16_deviation.py: 4.0e-3 / 7.9e-3 against the autocast run, 2.3e-3 / 3.7e-3
    against the fp32 run -- the bf16 tail of round 3 on v_mfma_f32_16x16x32_bf16)."""
    from test_oracle import g15_params
    import isd_amd
    from isd_amd.classifier import _FeatureModel
    import isd_amd._lib as L
    g = load_golden("g15_bf16_autocast.npz")
    p, x, y = g15_params()
    outs = {}
    for mode in ("f32", "bf16"):
        m = _FeatureModel(576, 32, 5, 4, mode).cuda()
        m.net.cnn.load_state_dict({k: v for k, v in p.items() if k.startswith("cnn")})
        m.net.fc.load_state_dict({k[3:]: v for k, v in p.items() if k.startswith("fc.")})
        xd = x.cuda().contiguous()
        assert L.lib().isd_featcnn_supported(m.conv_plan(xd)._h, 8, 17, 5) == 1
        out = isd_amd.HotPath(m).forward(xd, y.cuda(), want_grad=True)
        grads = {k: q.grad.detach().cpu().numpy().copy() for k, q in m.net.named_parameters()}
        outs[mode] = (out["logits"].cpu().numpy(), float(out["loss"]), grads)
    lg32, loss32, g32 = outs["f32"]
    lg16, loss16, g16 = outs["bf16"]
    assert rel_err(lg32, g["fp32.logits"]) < 1e-4 and abs(loss32 - float(g["fp32.loss"])) < 1e-5
    for ref in ("bf16", "fp32"):
        assert 0 < rel_err(lg16, g[f"{ref}.logits"]) < 8e-3, ref
        assert abs(loss16 - float(g[f"{ref}.loss"])) < 1e-3, ref
        for k, got in g16.items():
            want = g[f"{ref}." + ("fc.grad." + k[3:] if k.startswith("fc.") else "cnn.grad." + k[4:])]
            got = got[:, :, ::9] if k == "cnn.cnn2.weight" else got
            assert rel_err(got, want) < 1.6e-2, (ref, k)
    assert np.array_equal(lg16.argmax(1), g["bf16.logits"].argmax(1))


@pytest.mark.parametrize("B,C,T", [(64, 72, 17), (37, 576, 17), (5, 64, 9), (130, 128, 16)])
def test_bf16_matrix_core_step_shapes_vs_fp32_path(inn, B, C, T):
    """Ragged batches (not a multiple of the items per workgroup / the item pairs of the weight gradient), a partial
    last channel chunk, rows shorter than a DPP row: the bf16 step stays within its stated tolerance of the fp32 step;
    inference, loss-only and training calls agree with each other."""
    import isd_amd
    from isd_amd.classifier import _FeatureModel
    torch.manual_seed(B + C)
    m32 = _FeatureModel(C, 32, 5, 4, "f32").cuda()
    m16 = _FeatureModel(C, 32, 5, 4, "bf16").cuda()
    m16.load_state_dict(m32.state_dict())
    x = (torch.randn(B, C, T, device="cuda") * 2 - 5).contiguous()
    y = torch.randint(0, 5, (B,), device="cuda")
    o32 = isd_amd.HotPath(m32).forward(x, y, want_grad=True)
    hp16 = isd_amd.HotPath(m16)
    o16 = hp16.forward(x, y, want_grad=True)
    assert 0 < rel_err(o16["logits"].cpu(), o32["logits"].cpu()) < 8e-3
    assert rel_err(m16.flat_grads().cpu(), m32.flat_grads().cpu()) < 1.6e-2
    inf = hp16.forward(x)
    ev = hp16.forward(x, y)
    assert torch.equal(inf["logits"], o16["logits"]) and torch.equal(inf["pred"], o16["pred"])
    assert abs(float(ev["loss"]) - float(o16["loss"])) < 1e-6


@pytest.mark.parametrize("B,C", [(64, 72), (37, 576), (4096, 576)])
def test_bf16_feature_map_gives_the_bits_of_the_fp32_map(inn, B, C):
    """BASELINE config 3, round 3: the extractor can write its feature map as bf16 (isd_features_fused_bf16) and the bf16
    classifier step reads it directly (isd_featcnn_step_bf16).  The first layer rounds an fp32 map to bf16 (RNE) as it
    packs its matrix-core operands, so (1) the bf16 map is the RNE rounding of the fp32 map, bit for bit, and (2) the
    step on the bf16 map returns exactly the logits, loss and gradients of the step on the fp32 map of the same values."""
    import isd_amd
    from isd_amd.classifier import _FeatureModel
    nb = 9
    fx = isd_amd.FeatureExtractor(512, 256.0, isd_amd.BANDS_9)
    gen = torch.Generator(device="cuda").manual_seed(B + C)
    x = torch.randn(B, C // nb, 512, device="cuda", generator=gen)
    f32 = fx(x)
    f16 = fx(x, out_dtype=torch.bfloat16)
    assert f16.dtype == torch.bfloat16 and f16.shape == f32.shape
    assert torch.equal(f16, f32.to(torch.bfloat16))                     # torch's conversion is round-to-nearest-even
    out = torch.empty_like(f16)
    assert fx(x, out=out) is out and torch.equal(out, f16)
    torch.manual_seed(1)
    m = _FeatureModel(C, 32, 5, 4, "bf16").cuda()
    y = torch.randint(0, 5, (B,), device="cuda", generator=gen)
    hp = isd_amd.HotPath(m)
    a = hp.forward(f16.float().view(B, C, 17).contiguous(), y, want_grad=True)
    ga = m.flat_grads().clone()
    b = hp.forward(f16.view(B, C, 17), y, want_grad=True)
    gb = m.flat_grads().clone()
    assert torch.equal(a["logits"], b["logits"]) and torch.equal(a["pred"], b["pred"])
    assert float(a["loss"]) == float(b["loss"]) and torch.equal(ga, gb)
    assert torch.equal(hp.forward(f16.view(B, C, 17))["logits"], b["logits"])          # inference call
    # a model without bf16 activations refuses the bf16 map; so does the layer-wise path
    m32 = _FeatureModel(C, 32, 5, 4, "f32").cuda()
    with pytest.raises(TypeError, match="act_dtype"):
        isd_amd.HotPath(m32).forward(f16.view(B, C, 17), y)


def test_bf16_fused_raw_eeg_head_vs_fp32(inn):
    """The reference-native shape (8 zones, windows of 250 samples) with bf16 activations runs the fused forward /
    backward on the bf16 matrix cores (conv4_fused_fwd_bf16_kernel / conv4_fused_bwd_bf16_kernel: [time][channel] bf16
    tiles, v_mfma_f32_16x16x32_bf16, transposed LDS reads for the weight gradients).  Stated tolerance against the
    fp32 kernels on the same parameters: features 8e-3, parameter gradients 2e-2 of the tensor's largest magnitude --
    about twice what the reference's own autocast run deviates from its fp32 run (G16: 5.7e-3 / 1.3e-2; measured here:
    2.4e-3 / 7.4e-3, tools/bf16_deviation.py; scripts/train_fast.py:277 trains under bf16-mixed autocast)."""
    torch.manual_seed(3)
    h32 = inn.Head("Conv4Layers", ocnn.ELECTRODES, ocnn.ZONES, 32).cuda()
    h16 = inn.Head("Conv4Layers", ocnn.ELECTRODES, ocnn.ZONES, 32, act_dtype="bf16").cuda()
    h16.load_state_dict(h32.state_dict())
    for B, T in ((3, 250), (37, 512), (2, 800)):             # 1, 3 and 5 windows per trial; items not a multiple of anything
        x = torch.randn(B, 64, T, device="cuda")
        n_win = (T - 250) // 125 + 1
        w = torch.randn(B * n_win, 8, 32, device="cuda")
        outs = []
        for h in (h32, h16):
            h.zero_grad(set_to_none=True)
            f = h.forward_windows(x, 250, 125)
            assert f.shape == (B * n_win, 8, 32)
            (f * w).sum().backward()
            outs.append((f.detach().cpu(), {k: p.grad.detach().cpu().clone() for k, p in h.named_parameters()}))
        (f32, g32), (f16, g16) = outs
        assert 0 < rel_err(f16, f32) < 8e-3, (B, T)
        for k in g32:
            assert rel_err(g16[k], g32[k]) < 2e-2, (B, T, k)
    # other window lengths: 260 samples (T1 = 256: the x fetch takes its fifth 64-step chunk, 16 full column tiles),
    # 100 samples (6 column tiles, 3 blocks of 32 steps in the weight gradients), many more items than workgroups
    for B, T, wl, st in ((5, 520, 260, 130), (9, 300, 100, 50), (700, 250, 250, 125)):
        x = torch.randn(B, 64, T, device="cuda")
        n_win = (T - wl) // st + 1
        w = torch.randn(B * n_win, 8, 32, device="cuda")
        outs = []
        for h in (h32, h16):
            h.zero_grad(set_to_none=True)
            f = h.forward_windows(x, wl, st)
            (f * w).sum().backward()
            outs.append((f.detach().cpu(), {k: p.grad.detach().cpu().clone() for k, p in h.named_parameters()}))
        (f32, g32), (f16, g16) = outs
        assert 0 < rel_err(f16, f32) < 8e-3, (B, T, wl)
        for k in g32:
            assert rel_err(g16[k], g32[k]) < 2e-2, (B, T, wl, k)
        with torch.no_grad():                                # inference keeps no activations: same features
            assert torch.equal(h16.forward_windows(x, wl, st).cpu(), f16)
    # the oracle (fp64) agrees with the bf16 features within the same tolerance
    x = torch.randn(2, 64, 512)
    p = {"head." + k: v.detach().cpu().double() for k, v in h32.state_dict().items()}
    ref = ocnn.forward_head(x.double(), p, list(ocnn.ZONES), ocnn.zone_index_lists(), 250, 125)
    got = h16.forward_windows(x.cuda(), 250, 125).detach().cpu()
    assert rel_err(got, ref.reshape(got.shape)) < 8e-3
