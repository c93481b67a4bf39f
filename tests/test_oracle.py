"""The oracle restatements against golden vectors captured from the reference / scipy (CPU)."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from oracle import cnn as ocnn, dsp as odsp


# ------------------------------------------------------------------ DSP (scipy-pinned)
@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_stft_matches_scipy_golden(tag):
    g = load_golden("g1_stft.npz")
    T, fs, nperseg = [int(v) for v in g[f"{tag}_cfg"]]
    f, t, Z = odsp.stft(g[f"{tag}_x"], fs, nperseg, nperseg // 2)
    assert Z.shape == g[f"{tag}_Z"].shape and Z.dtype == np.complex64
    np.testing.assert_allclose(f, g[f"{tag}_f"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(t, g[f"{tag}_t"], rtol=0, atol=1e-12)
    assert np.abs(Z - g[f"{tag}_Z"]).max() <= 2e-6 * np.abs(g[f"{tag}_Z"]).max()
    bm = odsp.band_magnitude(Z, fs, nperseg, odsp.BANDS_5)
    np.testing.assert_allclose(bm, g[f"{tag}_band5"], rtol=1e-5, atol=1e-7)


def test_stft_frame_counts():
    # J = 26 @ T=800, 17 @ T=512, 9 @ T=4096 / nperseg 1024
    assert odsp.stft_frames(800, 64, 32)[0] == 26
    assert odsp.stft_frames(512, 64, 32)[0] == 17
    assert odsp.stft_frames(4096, 1024, 512)[0] == 9
    assert odsp.stft_frames(4096, 1024, 960)[0] == 65


@pytest.mark.parametrize("tag", ["b5", "b9", "b40"])
def test_sosfilt_matches_scipy_golden(tag):
    g = load_golden("g2_sos.npz")
    sos, x, sel = g[f"{tag}_sos"], g[f"{tag}_x"], g[f"{tag}_sel"]
    bands = {"b5": odsp.BANDS_5, "b9": odsp.BANDS_9, "b40": odsp.BANDS_40}[tag]
    for j, b in enumerate(sel):
        _, lo, hi = bands[b]
        np.testing.assert_allclose(odsp.butter_bandpass_sos(4, lo, hi, float(g[f"{tag}_fs"])), sos[b],
                                   rtol=1e-13, atol=0)
        y = odsp.sosfilt(sos[b], x)
        assert rel_err(y, g[f"{tag}_y"][:, j]) < 1e-11


@pytest.mark.parametrize("tag,bands", [("c1", odsp.BANDS_5), ("c2", odsp.BANDS_9), ("c5", odsp.BANDS_40[:6]),
                                       ("c800", odsp.BANDS_9)])
def test_spec_s_features_match_scipy_golden(tag, bands):
    g = load_golden("g3_features.npz")
    B, C, T, fs, nperseg, nov, nb = g[f"{tag}_cfg"]
    B, C, T, nperseg, nov = int(B), int(C), int(T), int(nperseg), int(nov)
    x = g[f"{tag}_x"] if f"{tag}_x" in g else np.random.default_rng(3).standard_normal((B, C, T)).astype(np.float32)
    feat = odsp.extract_features(x[:2], fs=float(fs), bands=bands, nperseg=nperseg, noverlap=nov)
    np.testing.assert_allclose(feat, g[f"{tag}_feat"][:2], rtol=0, atol=2e-5)


def test_scipy_backed_features_equal_the_numpy_restatement():
    x = np.random.default_rng(9).standard_normal((2, 3, 512)).astype(np.float32)
    a = odsp.extract_features(x, fs=256.0, bands=odsp.BANDS_9)
    b = odsp.extract_features_scipy(x, fs=256.0, bands=odsp.BANDS_9)
    np.testing.assert_allclose(a, b, rtol=0, atol=1e-5)


# ------------------------------------------------------------------ CNN (reference-pinned)
def _t(g, prefix):
    return {k[len(prefix):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(prefix)}


@pytest.mark.parametrize("cz", [6, 15])
def test_conv4layers_matches_reference_golden(cz):
    g = load_golden("g4_conv4layers.npz")
    p = {k: v.requires_grad_() for k, v in _t(g, f"c{cz}.sd.").items()}
    x = torch.from_numpy(g[f"c{cz}.x"]).requires_grad_()
    y = ocnn.conv4layers(x, p)
    y.square().sum().backward()
    assert rel_err(y.detach(), g[f"c{cz}.y"]) < 1e-6
    assert rel_err(x.grad, g[f"c{cz}.dx"]) < 1e-5
    for k, v in p.items():
        assert rel_err(v.grad, g[f"c{cz}.grad.{k}"]) < 1e-5, k


def _small_zones():
    electrodes = ["Fp1", "Fp2", "F3", "F4", "C3", "C4", "O1", "O2"]
    zones = {"Frontal": ["Fp1", "Fp2", "F3", "F4"], "Central": ["C3", "C4"], "Occipital": ["O1", "O2"]}
    return list(zones), ocnn.zone_index_lists(electrodes, zones)


def test_fast_small_train_head_matches_reference_golden():
    g = load_golden("g5_fast_small.npz")
    names, idx = _small_zones()
    p = {k: v.requires_grad_() for k, v in _t(g, "sd.").items()}
    x = torch.from_numpy(g["x"])
    feat = ocnn.forward_head(x, p, names, idx)
    assert feat.shape == (2, 3, 3, 16)
    assert rel_err(feat.detach(), g["features"]) < 1e-6
    logits = ocnn.train_head_logits(x, p, names, idx)
    loss = ocnn.cross_entropy(logits, g["labels"])
    loss.backward()
    assert rel_err(logits.detach(), g["train_head.logits"]) < 1e-6
    assert abs(float(loss) - float(g["train_head.loss"])) < 1e-6
    for k in g.files:
        if k.startswith("train_head.grad."):
            name = k[len("train_head.grad."):]
            assert rel_err(p[name].grad, g[k]) < 2e-5, name


def test_fast_small_default_mode_matches_reference_golden():
    g = load_golden("g5_fast_small.npz")
    names, idx = _small_zones()
    p = {k: v.requires_grad_() for k, v in _t(g, "sd.").items()}
    logits = ocnn.default_logits(torch.from_numpy(g["x"]), p, names, idx, num_heads=4, num_layers=1)
    loss = ocnn.cross_entropy(logits, g["labels"])
    loss.backward()
    assert rel_err(logits.detach(), g["default.logits"]) < 1e-5
    assert abs(float(loss.detach()) - float(g["default.loss"])) < 1e-6
    for k in g.files:
        if k.startswith("default.grad."):
            name = k[len("default.grad."):]
            assert rel_err(p[name].grad, g[k]) < 5e-5, name


def test_fast_prod_default_mode_eval_matches_reference_golden():
    g = load_golden("g6_fast_prod.npz")
    p = _t(g, "sd.")
    x = torch.from_numpy(np.random.default_rng(6).standard_normal((4, 64, 800)).astype(np.float32))
    with torch.no_grad():
        logits = ocnn.default_logits(x, p, list(ocnn.ZONES), ocnn.zone_index_lists(), num_heads=8, num_layers=4)
    assert rel_err(logits, g["default_logits"]) < 1e-5
    assert np.array_equal(ocnn.predict(logits).numpy(), g["default_pred"])


def test_fast_prod_eval_logits_and_argmax_match_reference_golden():
    g = load_golden("g6_fast_prod.npz")
    p = _t(g, "sd.")
    x = torch.from_numpy(np.random.default_rng(6).standard_normal((4, 64, 800)).astype(np.float32))
    names, idx = list(ocnn.ZONES), ocnn.zone_index_lists()
    with torch.no_grad():
        feat = ocnn.forward_head(x, p, names, idx)
        logits = ocnn.train_head_logits(x, p, names, idx)
    assert feat.shape == (4, 5, 8, 32)
    assert rel_err(feat, g["features"]) < 1e-6
    assert rel_err(logits, g["train_head_logits"]) < 1e-6
    assert np.array_equal(ocnn.predict(logits).numpy(), g["train_head_pred"])


@pytest.mark.parametrize("tag", ["z6", "c128"])
def test_eegnet_matches_reference_golden(tag):
    g = load_golden("g7_eegnet.npz")
    p = _t(g, f"{tag}.sd.")
    for k, v in p.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_()
    x = torch.from_numpy(g[f"{tag}.x"]).requires_grad_()
    with torch.no_grad():
        assert rel_err(ocnn.eegnet_encoder(x, p, training=False), g[f"{tag}.y_eval"]) < 1e-5
    y = ocnn.eegnet_encoder(x, p, training=True)
    y.square().sum().backward()
    assert rel_err(y.detach(), g[f"{tag}.y_train"]) < 1e-5
    assert rel_err(x.grad, g[f"{tag}.dx"]) < 1e-4
    for k in g.files:
        if k.startswith(f"{tag}.grad."):
            assert rel_err(p[k[len(tag) + 6:]].grad, g[k]) < 1e-4, k
    for name in ("temporal_conv.1", "spatial_conv.1", "separable_conv.2"):
        for buf in ("running_mean", "running_var"):
            assert rel_err(p[f"{name}.{buf}"], g[f"{tag}.sd_after.{name}.{buf}"]) < 1e-6


@pytest.mark.parametrize("fname,fn,bns", [
    ("g10_cvblock.npz", ocnn.cvblock, ("bn1", "bn2", "bn3")),
    ("g11_paperhead.npz", ocnn.headconv_paper, ("norm1", "norm2", "norm3", "norm4"))])
@pytest.mark.parametrize("tag", ["z6", "z15"])
def test_bn_heads_match_reference_golden(fname, fn, bns, tag):
    g = load_golden(fname)
    p = _t(g, f"{tag}.sd.")
    for k, v in p.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_()
    x = torch.from_numpy(g[f"{tag}.x"]).requires_grad_()
    with torch.no_grad():
        assert rel_err(fn(x, p, training=False), g[f"{tag}.y_eval"]) < 1e-5
    y = fn(x, p, training=True)
    y.square().sum().backward()
    assert rel_err(y.detach(), g[f"{tag}.y_train"]) < 1e-5
    assert rel_err(x.grad, g[f"{tag}.dx"]) < 1e-4
    for k in g.files:
        if k.startswith(f"{tag}.grad."):
            assert rel_err(p[k[len(tag) + 6:]].grad, g[k]) < 1e-4, k
    for name in bns:
        for buf in ("running_mean", "running_var"):
            assert rel_err(p[f"{name}.{buf}"], g[f"{tag}.sd_after.{name}.{buf}"]) < 1e-6


def test_cosine_schedule_matches_golden_and_quirk():
    g = load_golden("g8_cosine.npz")
    s = ocnn.cosine_scheduler(1, 0.1, 200, 5, warmup_epochs=10)
    np.testing.assert_allclose(s, g["schedule"], rtol=0, atol=1e-15)
    assert len(s) == 1000 and s[0] == 0 and s[49] == 1 and s[50] == 1
    assert abs(s[-1] - 0.1000025) < 1e-6
    assert ocnn.lr_multiplier(s, 0) == s[-1]          # trainer.py:52 quirk


def test_adamw_trajectory_matches_reference_golden():
    g5, g9 = load_golden("g5_fast_small.npz"), load_golden("g9_adamw.npz")
    names, idx = _small_zones()
    keys = [k[len("final."):] for k in g9.files if k.startswith("final.")]
    p = _t(g5, "sd.")
    params = [p[k].requires_grad_() for k in keys]
    opt = torch.optim.AdamW(params, lr=5e-4)
    x = torch.from_numpy(g5["x"])
    losses = []
    for _ in range(3):
        opt.zero_grad()
        loss = ocnn.cross_entropy(ocnn.train_head_logits(x, p, names, idx), g5["labels"])
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    np.testing.assert_allclose(losses, g9["losses"], rtol=1e-5)
    for k in keys:
        assert rel_err(p[k].detach(), g9["final." + k]) < 1e-5, k


# ---------------------------------------------------------------- BASELINE config 5 (stress) shapes
@pytest.mark.parametrize("tag", ["f5120", "r128"])
def test_eegnet_at_stress_shapes_matches_reference_golden(tag):
    """G13: EEGNet_Encoder(5120, 32) on [2, 5120, 65] and EEGNet_Encoder(128, 32) on [2, 128, 4096]."""
    g = load_golden("g13_eegnet_cfg5.npz")
    C, T, B, stride = (int(v) for v in g[f"{tag}.cfg"])
    p = _t(g, f"{tag}.sd.")
    for k, v in p.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_()
    x = torch.from_numpy(np.random.default_rng(13).standard_normal((B, C, T)).astype(np.float32)).requires_grad_()
    with torch.no_grad():
        assert rel_err(ocnn.eegnet_encoder(x, p, training=False), g[f"{tag}.y_eval"]) < 1e-5
    y = ocnn.eegnet_encoder(x, p, training=True)
    y.square().sum().backward()
    assert rel_err(y.detach(), g[f"{tag}.y_train"]) < 1e-5
    assert rel_err(x.grad[:, ::stride], g[f"{tag}.dx_sub"]) < 1e-4
    for k in g.files:
        if k.startswith(f"{tag}.grad."):
            assert rel_err(p[k[len(tag) + 6:]].grad, g[k]) < 1e-4, k
    for name in ("temporal_conv.1", "spatial_conv.1", "separable_conv.2"):
        for buf in ("running_mean", "running_var"):
            assert rel_err(p[f"{name}.{buf}"], g[f"{tag}.sd_after.{name}.{buf}"]) < 1e-6


def test_stress_configuration_composed_matches_golden():
    """G14: [2, 128, 4096] @ 1024 Hz -> 40-band spec S (1024 / 960) -> EEGNet_Encoder(5120, 32) -> Linear -> CE, the
    oracle's restatement against scipy + the reference module."""
    g = load_golden("g14_cfg5_composed.npz")
    B, C, T, fs, nperseg, nov, nb = (int(v) for v in g["cfg"])
    x = np.random.default_rng(14).standard_normal((B, C, T)).astype(np.float32)
    feat = odsp.extract_features(x, fs=float(fs), bands=odsp.BANDS_40, nperseg=nperseg, noverlap=nov)
    assert feat.shape == (B, nb, C, 65)
    assert np.abs(feat[:, ::3, ::16] - g["feat_sub"]).max() < 1e-5            # log domain
    np.testing.assert_allclose(feat.astype(np.float64).sum(axis=(2, 3)), g["feat_sum"], rtol=1e-6)
    p = _t(g, "enc.sd.")
    w, b = torch.from_numpy(g["fc.sd.weight"]).requires_grad_(), torch.from_numpy(g["fc.sd.bias"]).requires_grad_()
    for k, v in p.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_()
    h = ocnn.eegnet_encoder(torch.from_numpy(feat).reshape(B, nb * C, -1), p, training=True)
    logits = torch.nn.functional.linear(h, w, b)
    loss = ocnn.cross_entropy(logits, torch.from_numpy(g["labels"]))
    loss.backward()
    assert rel_err(logits.detach(), g["logits"]) < 1e-4 and abs(float(loss) - float(g["loss"])) < 1e-5
    scale = max(float(np.abs(g[k]).max()) for k in g.files if k.startswith("enc.grad."))
    for k in g.files:
        if k.startswith("enc.grad."):
            tol = 1e-3 * max(float(np.abs(g[k]).max()), 1e-3 * scale)        # fp32 features in, fp32 autograd
            assert float((p[k[9:]].grad - torch.from_numpy(g[k])).abs().max()) < tol, k
    assert rel_err(w.grad, g["fc.grad.weight"]) < 1e-3 and rel_err(b.grad, g["fc.grad.bias"]) < 1e-3


def g15_params():
    """G15's parameters and inputs, rebuilt exactly as tests/golden/make_golden.py::g15_inputs does."""
    rng = np.random.default_rng(15)
    shapes = (("cnn1.weight", (32, 1, 1, 5), 5), ("cnn1.bias", (32,), 5), ("cnn2.weight", (32, 32, 576, 1), 32 * 576),
              ("cnn3.weight", (32, 32, 1, 5), 160), ("cnn4.weight", (32, 32, 1, 5), 160), ("fc.weight", (5, 32), 32),
              ("fc.bias", (5,), 32))
    p = {k: torch.from_numpy((rng.uniform(-1, 1, shp).astype(np.float32) / np.sqrt(fan)).astype(np.float32))
         for k, shp, fan in shapes}
    x = torch.from_numpy((rng.standard_normal((8, 576, 17)) * 2 - 5).astype(np.float32))
    y = torch.from_numpy(rng.integers(0, 5, 8))
    return p, x, y


def test_feature_classifier_at_576_channels_matches_reference_golden():
    """G15 (fp32 half): the reference's Conv4Layers(576, 32) + Linear(32, 5) on a [8, 576, 17] feature map; the bf16
    half of the fixture (the reference under torch.autocast) is what the bf16 GPU path is compared with."""
    g = load_golden("g15_bf16_autocast.npz")
    p, x, y = g15_params()
    assert np.array_equal(y.numpy().astype(np.uint8), g["labels"])
    for v in p.values():
        v.requires_grad_()
    cnn = {k: v for k, v in p.items() if k.startswith("cnn")}
    logits = torch.nn.functional.linear(ocnn.conv4layers(x, cnn), p["fc.weight"], p["fc.bias"])
    loss = ocnn.cross_entropy(logits, y)
    loss.backward()
    assert rel_err(logits.detach(), g["fp32.logits"]) < 1e-5 and abs(float(loss) - float(g["fp32.loss"])) < 1e-6
    for k, v in p.items():
        want = g["fp32." + ("fc.grad." + k[3:] if k.startswith("fc.") else "cnn.grad." + k)]
        got = v.grad.numpy()
        got = got[:, :, ::9] if k == "cnn2.weight" else got
        assert rel_err(got, want) < 1e-4, k


def test_g16_autocast_fixture_is_the_fp32_fixture_within_bf16_noise():
    """G16 (the reference's FAST(small_config) under torch.autocast(bfloat16)) carries G5's tensors, and deviates from G5
    -- the same reference in fp32, which the oracle reproduces above -- by what the GPU test's tolerance is derived
    from: features 5.7e-3, logits 2e-3, gradients 1.3e-2 of each tensor's largest magnitude."""
    g5, g16 = load_golden("g5_fast_small.npz"), load_golden("g16_fast_small_autocast.npz")
    assert set(g16.files) == {k for k in g5.files if not k.startswith("sd.") and k not in ("x", "labels")}
    assert 1e-3 < rel_err(g16["features"], g5["features"]) < 6e-3
    for mode in ("train_head", "default"):
        assert 1e-4 < rel_err(g16[f"{mode}.logits"], g5[f"{mode}.logits"]) < 2.1e-3
        worst = max(rel_err(g16[k], g5[k]) for k in g16.files if k.startswith(f"{mode}.grad."))
        assert 5e-3 < worst < 1.35e-2, (mode, worst)
