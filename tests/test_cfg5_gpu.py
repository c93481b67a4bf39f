"""GPU parity for BASELINE config 5 (high-resolution stress: 128 ch, 4 s @ 1024 Hz, 40 bands, 1024-point STFT with
hop 64, EEGNet-style depthwise CNN, batch 2048) at ITS OWN shapes: the head alone (G13), the composed pipeline
(G14) and the full batch through size-independent properties."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from oracle import cnn as ocnn, dsp as odsp

pytestmark = pytest.mark.gpu

FS, T, C, NPERSEG, NOVERLAP = 1024.0, 4096, 128, 1024, 960


@pytest.fixture(scope="module")
def isd():
    import isd_amd
    assert torch.cuda.is_available()
    return isd_amd


def _sd(g, prefix):
    return {k[len(prefix):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(prefix)}


def _check_grads(named, want_of, scale, rel=1e-4):
    for k, p in named:
        want = want_of(k)
        # BN1's gamma / beta gradients vanish up to eps effects (BN2 renormalises): compared on the gradient scale
        floor = 5e-2 if k.startswith("temporal_conv.1.") else 1e-3
        tol = rel * max(float(np.abs(want).max()), floor * scale)
        assert np.abs(p.grad.cpu().numpy() - want).max() < tol + 1e-7, k


@pytest.mark.parametrize("tag", ["f5120", "r128"])
def test_eegnet_head_at_stress_shapes_matches_reference_golden(isd, tag):
    """EEGNet_Encoder(5120, 32) on [2, 5120, 65] and EEGNet_Encoder(128, 32) on [2, 128, 4096] (fast.py:122-167)."""
    import isd_amd.nn as inn
    g = load_golden("g13_eegnet_cfg5.npz")
    Cc, Tt, B, _ = (int(v) for v in g[f"{tag}.cfg"])
    m = inn.EEGNet_Encoder(Cc, 32, dropout=0.0).cuda()
    m.load_state_dict(_sd(g, f"{tag}.sd."))
    x = torch.from_numpy(np.random.default_rng(13).standard_normal((B, Cc, Tt)).astype(np.float32)).cuda()
    m.eval()
    with torch.no_grad():
        assert rel_err(m(x).cpu(), g[f"{tag}.y_eval"]) < 1e-4
    m.train()
    y = m(x)
    assert rel_err(y.detach().cpu(), g[f"{tag}.y_train"]) < 1e-4
    y.square().sum().backward()
    scale = max(float(np.abs(g[k]).max()) for k in g.files if k.startswith(f"{tag}.grad."))
    _check_grads(m.named_parameters(), lambda k: g[f"{tag}.grad.{k}"], scale)
    for k, v in m.state_dict().items():
        if "running" in k:
            assert rel_err(v.cpu(), g[f"{tag}.sd_after.{k}"]) < 1e-4, k
        if "num_batches_tracked" in k:
            assert int(v) == int(g[f"{tag}.sd_after.{k}"])


def test_stress_configuration_composed_matches_golden(isd):
    """[2, 128, 4096] -> all 40 bands -> features -> EEGNet_Encoder(5120, 32) -> Linear(32, 5) -> CE -> every gradient,
    through the estimator's own step (EEGNetPath), against scipy + the reference module (G14)."""
    from isd_amd.classifier import _EEGNetFeatureModel
    g = load_golden("g14_cfg5_composed.npz")
    B = int(g["cfg"][0])
    x = torch.from_numpy(np.random.default_rng(14).standard_normal((B, C, T)).astype(np.float32)).cuda()
    fx = isd.FeatureExtractor(T, FS, isd.BANDS_40, nperseg=NPERSEG, noverlap=NOVERLAP)
    assert fx.can_fuse and fx.n_frames == 65 and fx.n_bands == 40
    feat = fx(x)                                                        # fused path: one kernel per precision set
    f = feat.cpu().numpy().astype(np.float64)
    sub = g["feat_sub"].astype(np.float64)
    assert (np.abs(f[:, ::3, ::16] - sub) <= 1e-4 * np.maximum(1.0, np.abs(sub))).all()     # 1e-4 relative gate
    np.testing.assert_allclose(f.sum(axis=(2, 3)), g["feat_sum"], rtol=2e-6)
    two = fx(x, fused=False)                                            # materialising pair agrees
    assert float((two - feat).abs().max()) < 2e-4
    m = _EEGNetFeatureModel(40 * C, 32, 5, dropout=0.0).cuda()
    m.net.enc.load_state_dict(_sd(g, "enc.sd."))
    m.net.fc.load_state_dict(_sd(g, "fc.sd."))
    out = m.make_path().forward(feat.view(B, 40 * C, 65), torch.from_numpy(g["labels"]).cuda(), want_grad=True)
    assert rel_err(out["logits"].cpu(), g["logits"]) < 1e-4
    assert abs(float(out["loss"]) - float(g["loss"])) < 1e-5
    assert np.array_equal(out["pred"].cpu().numpy(), g["logits"].argmax(1))
    scale = max(float(np.abs(g[k]).max()) for k in g.files if k.startswith("enc.grad."))
    # measured (tools/grad_slack_probe.py, profiles/r04_grad_slack.txt): 1.7e-5 worst in this metric (BN1's bias), with
    # HIP's features and with scipy's alike (they differ by 4.6e-5 in the log domain: not where the error comes from);
    # fc 1.1e-6.  The bound was 1e-3 until round 4.
    _check_grads(m.net.enc.named_parameters(), lambda k: g[f"enc.grad.{k}"], scale, rel=1e-4)
    assert rel_err(m.net.fc.weight.grad.cpu(), g["fc.grad.weight"]) < 1e-5
    assert rel_err(m.net.fc.bias.grad.cpu(), g["fc.grad.bias"]) < 1e-5
    # the autograd modules run the same kernels: same loss, same gradient block
    g_path = m.flat_grads().clone()
    m.zero_grad(set_to_none=True)
    m.net.enc.temporal_conv[1].num_batches_tracked.zero_()
    import isd_amd.nn as inn
    loss = inn.token_mean_cross_entropy(m.net.token_logits(feat), torch.from_numpy(g["labels"]).cuda())
    loss.backward()
    g_auto = torch.cat([p.grad.reshape(-1) for p in m._ordered_params()])
    assert abs(float(loss) - float(out["loss"])) < 1e-6 and rel_err(g_path.cpu(), g_auto.cpu()) < 1e-5


def test_stress_configuration_full_batch_properties(isd):
    """B = 2048 (BASELINE config 5's batch): trials are independent through the extractor, so sampled trials of the
    full batch equal the oracle's features of those trials alone; one optimisation step of the estimator's trainer
    runs at that size, lowers nothing to NaN and moves every parameter tensor."""
    from isd_amd.classifier import _EEGNetFeatureModel
    B = 2048
    gen = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(B, C, T, device="cuda", generator=gen)
    y = torch.randint(0, 5, (B,), device="cuda", generator=gen).to(torch.uint8)
    fx = isd.FeatureExtractor(T, FS, isd.BANDS_40, nperseg=NPERSEG, noverlap=NOVERLAP)
    feat = fx(x)
    assert feat.shape == (B, 40, C, 65) and bool(torch.isfinite(feat).all())
    pick = [0, 1023, 2047]
    ref = odsp.extract_features_scipy(x[pick][:, ::32].cpu().numpy(), fs=FS, bands=odsp.BANDS_40, nperseg=NPERSEG,
                                      noverlap=NOVERLAP).astype(np.float64)          # 4 channels per picked trial
    got = feat[pick][:, :, ::32].cpu().numpy().astype(np.float64)
    assert (np.abs(got - ref) <= 1e-4 * np.maximum(1.0, np.abs(ref))).all()
    alone = fx(x[1023:1024].contiguous())
    assert torch.equal(alone[0], feat[1023])                                          # batch position does not matter
    torch.manual_seed(0)
    m = _EEGNetFeatureModel(40 * C, 32, 5, dropout=0.25).cuda()
    tr = isd.Trainer(m, lr=5e-4, weight_decay=1e-2)
    before = m.flat_params().clone()
    f2 = feat.view(B, 40 * C, 65)
    losses = [float(tr.step(f2, y)["loss"]) for _ in range(3)]
    assert all(np.isfinite(losses)) and abs(losses[0] - np.log(5.0)) < 0.5
    moved = (m.flat_params() - before).abs()
    off = 0
    for p in m._ordered_params():
        assert float(moved[off:off + p.numel()].max()) > 0.0
        off += p.numel()
    assert int(m.net.enc.temporal_conv[1].num_batches_tracked) == 3
    ev = tr.path.forward(f2[:64].contiguous(), y[:64].contiguous())                   # eval mode: running statistics
    assert np.isfinite(float(ev["loss"])) and ev["pred"].shape == (64,)


def test_filterbank_eegnet_classifier_learns_synthetic_task(isd):
    """fit / predict of the config-5 estimator on a reduced synthetic task (48 trials x 128 ch x 4096 samples)."""
    X, y = odsp.synth_trials(48, C, T, FS, seed=2)
    clf = isd.FilterbankEEGNetClassifier(max_epochs=60, batch_size=48, warmup_epochs=2, seed=3, dropout=0.0, lr=5e-3)
    assert clf.fit(X, y) is clf
    assert clf.history_[-1] < 1.3                                                     # from ln 5 = 1.61
    first_loss = clf.history_[-1]
    first_params = clf.model_.flat_params().clone()
    clf.fit(X, y)                                                                     # a second fit starts over
    # ... and repeats the first one bit for bit: every batch sum of the BatchNorm layers is accumulated exactly
    # (csrc/exact.h), every other reduction in a fixed order -- the reference's seed_all(deterministic=True),
    # src/fast/utils.py:104-114
    assert len(clf.history_) == 1 and clf.history_[-1] == first_loss
    assert torch.equal(clf.model_.flat_params(), first_params)
    pred = clf.predict(X)
    assert pred.shape == (48,) and pred.dtype == np.int64 and (pred == y).mean() > 0.4
