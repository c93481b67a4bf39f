#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ (run in the build container).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Imports the *reference itself* (``/root/reference/src/fast/models/fast.py``,
read-only) and scipy.signal, runs them on seeded inputs and stores inputs +
outputs as small ``.npz`` files.  The reference never travels to the GPU box;
these vectors do.  IDs follow SURVEY.md 8c (G1..G9); G10..G16 were added by the build (BN heads, FIR,
the stress configuration's shapes, the bf16-autocast runs).
"""
import os
import sys
import types

import numpy as np
import scipy.signal as ss
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("ISD_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(REF, "src"))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from fast.models.fast import (FAST, Conv4Layers, CVBlock, EEGNet_Encoder,  # noqa: E402  (the reference)
                              HeadConv_Paper_Version)
from oracle import cnn as ocnn, dsp as odsp  # noqa: E402  (constants / band tables only)


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez(path, **arrs)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def sd_np(module, prefix=""):
    return {prefix + k: v.detach().cpu().numpy().copy() for k, v in module.state_dict().items()}


def grads_np(module, prefix="grad."):
    return {prefix + k: p.grad.detach().numpy().copy() for k, p in module.named_parameters()}


# ---------------------------------------------------------------- G1: scipy stft
def g1():
    out = {}
    for tag, (T, fs, nperseg) in {"a": (800, 250, 64), "b": (512, 256, 64), "c": (4096, 1024, 1024)}.items():
        x = np.random.default_rng(0).standard_normal((2, 3, T)).astype(np.float32)
        f, t, Z = ss.stft(x, fs=fs, nperseg=nperseg, noverlap=nperseg // 2)
        out[f"{tag}_x"], out[f"{tag}_f"], out[f"{tag}_t"], out[f"{tag}_Z"] = x, f, t, Z
        out[f"{tag}_cfg"] = np.array([T, fs, nperseg])
        # scripts/global_shap_analysis.py:151-156 band aggregation on |Z|
        S = np.abs(Z)
        bm = []
        for _, lo, hi in odsp.BANDS_5:
            idx = np.where((f >= lo) & (f <= hi))[0]
            bm.append(S[..., idx, :].mean(axis=-2) if len(idx) else np.zeros(S.shape[:-2] + S.shape[-1:]))
        out[f"{tag}_band5"] = np.stack(bm, axis=-2)
    save("g1_stft.npz", **out)


# ---------------------------------------------------------------- G2: butter sos + sosfilt
def g2():
    out = {}
    for tag, bands, fs, T in (("b5", odsp.BANDS_5, 256.0, 512), ("b9", odsp.BANDS_9, 256.0, 512),
                              ("b40", odsp.BANDS_40, 1024.0, 4096)):
        sos = np.stack([ss.butter(4, (lo, hi), "bandpass", fs=fs, output="sos") for _, lo, hi in bands])
        out[f"{tag}_sos"] = sos
        out[f"{tag}_fs"] = np.array(fs)
        x = np.random.default_rng(1).standard_normal((2 if T <= 512 else 1, 2, T)).astype(np.float32)
        out[f"{tag}_x"] = x
        sel = list(range(len(bands))) if len(bands) <= 9 else [0, 1, 20, 39]
        out[f"{tag}_sel"] = np.array(sel)
        out[f"{tag}_y"] = np.stack([ss.sosfilt(sos[b], x.astype(np.float64), axis=-1) for b in sel], axis=1)
    save("g2_sos.npz", **out)


# ---------------------------------------------------------------- G3: spec-S features via scipy
def g3():
    out = {}
    for tag, bands, fs, T, C, nperseg, nov in (("c1", odsp.BANDS_5, 256.0, 512, 64, 64, 32),
                                                ("c2", odsp.BANDS_9, 256.0, 512, 64, 64, 32),
                                                ("c5", odsp.BANDS_40[:6], 1024.0, 4096, 4, 1024, 960),
                                                # the reference-native trial: 800 samples @ 250 Hz (preprocess.py:62)
                                                ("c800", odsp.BANDS_9, 250.0, 800, 8, 64, 32)):
        B = 4 if C == 64 else 2
        x = np.random.default_rng(3).standard_normal((B, C, T)).astype(np.float32)
        feats = []
        for _, lo, hi in bands:
            sos = ss.butter(4, (lo, hi), "bandpass", fs=fs, output="sos")
            y = ss.sosfilt(sos, x.astype(np.float64), axis=-1)
            f, _, Z = ss.stft(y, fs=fs, nperseg=nperseg, noverlap=nov)
            idx = np.where((f >= lo) & (f <= hi))[0]
            P = (np.abs(Z[..., idx, :]) ** 2).mean(axis=-2)
            feats.append(np.log(P + 1e-10))
        out[f"{tag}_feat"] = np.stack(feats, axis=1).astype(np.float32)
        out[f"{tag}_cfg"] = np.array([B, C, T, fs, nperseg, nov, len(bands)])
        if C != 64:
            out[f"{tag}_x"] = x       # 64-ch inputs are regenerated from the seed
    save("g3_features.npz", **out)


# ---------------------------------------------------------------- G4: Conv4Layers fwd/bwd
def g4():
    out = {}
    for cz in (6, 15):
        torch.manual_seed(0)
        m = Conv4Layers(cz, 32)
        x = torch.randn(3, cz, 250, requires_grad=True)
        y = m(x)
        y.square().sum().backward()
        out.update(sd_np(m, f"c{cz}.sd."))
        out.update(grads_np(m, f"c{cz}.grad."))
        out[f"c{cz}.x"], out[f"c{cz}.y"], out[f"c{cz}.dx"] = x.detach().numpy(), y.detach().numpy(), x.grad.numpy()
    save("g4_conv4layers.npz", **out)


def small_config():
    # values of tests/conftest.py:32-54 (dropout 0 -> deterministic in train mode)
    electrodes = ["Fp1", "Fp2", "F3", "F4", "C3", "C4", "O1", "O2"]
    zones = {"Frontal": ["Fp1", "Fp2", "F3", "F4"], "Central": ["C3", "C4"], "Occipital": ["O1", "O2"]}
    return types.SimpleNamespace(electrodes=electrodes, zone_dict=zones, dim_cnn=16, dim_token=16, seq_len=500,
                                 window_len=250, slide_step=125, head="Conv4Layers", n_classes=3, num_layers=1,
                                 num_heads=4, dropout=0.0)


def prod_config():
    # values of tests/conftest.py:12-29 / scripts/train_fast.py:293-307
    return types.SimpleNamespace(electrodes=ocnn.ELECTRODES, zone_dict=ocnn.ZONES, dim_cnn=32, dim_token=32,
                                 seq_len=800, window_len=250, slide_step=125, head="Conv4Layers", n_classes=5,
                                 num_layers=4, num_heads=8, dropout=0.1)


# ---------------------------------------------------------------- G5 + G9: FAST(small) train_head / default, AdamW
def g5_g9():
    out = {}
    cfg = small_config()
    torch.manual_seed(0)
    m = FAST(cfg)
    m.train()
    x = torch.randn(2, 8, 500)
    labels = torch.tensor([0, 2])
    out.update(sd_np(m, "sd."))
    out["x"], out["labels"] = x.numpy(), labels.numpy().astype(np.uint8)
    feat = m.forward_head(x)
    out["features"] = feat.detach().numpy()
    for mode in ("train_head", "default"):
        m.zero_grad()
        logits = m(x, forward_mode=mode)
        loss = torch.nn.CrossEntropyLoss()(logits, labels)
        loss.backward()
        out[f"{mode}.logits"], out[f"{mode}.loss"] = logits.detach().numpy(), loss.detach().numpy()
        for k, p in m.named_parameters():
            if p.grad is not None:
                out[f"{mode}.grad.{k}"] = p.grad.detach().numpy().copy()
    save("g5_fast_small.npz", **out)

    # G9: 3 AdamW steps (lr 5e-4, torch defaults) in train_head mode, params not used there excluded
    torch.manual_seed(0)
    m = FAST(cfg)
    m.train()
    names = [k for k, _ in m.named_parameters() if k.startswith(("head.", "input_layer.", "last_layer."))]
    params = [dict(m.named_parameters())[k] for k in names]
    opt = torch.optim.AdamW(params, lr=5e-4)
    losses = []
    for _ in range(3):
        opt.zero_grad()
        loss = torch.nn.CrossEntropyLoss()(m(x, forward_mode="train_head"), labels)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    o9 = {"losses": np.array(losses, dtype=np.float64)}
    o9.update({"final." + k: dict(m.named_parameters())[k].detach().numpy().copy() for k in names})
    save("g9_adamw.npz", **o9)


# ---------------------------------------------------------------- G6: production config, eval logits + argmax
def g6():
    cfg = prod_config()
    torch.manual_seed(0)
    m = FAST(cfg)
    m.eval()
    # input regenerated from this seed by the tests (not stored)
    x = torch.from_numpy(np.random.default_rng(6).standard_normal((4, 64, 800)).astype(np.float32))
    with torch.no_grad():
        feat = m.forward_head(x)
        lg_head = m(x, forward_mode="train_head")
        lg_def = m(x, forward_mode="default")
    keep = {k: v for k, v in sd_np(m, "sd.").items()}
    save("g6_fast_prod.npz", features=feat.numpy(), train_head_logits=lg_head.numpy(),
         train_head_pred=lg_head.argmax(1).numpy(), default_logits=lg_def.numpy(),
         default_pred=lg_def.argmax(1).numpy(), **keep)


# ---------------------------------------------------------------- G7: EEGNet_Encoder eval + train
def g7():
    out = {}
    for tag, (C, T, B) in {"z6": (6, 250, 5), "c128": (128, 96, 3)}.items():
        torch.manual_seed(0)
        m = EEGNet_Encoder(C, 32, dropout=0.0)
        # non-trivial running stats / affine params
        with torch.no_grad():
            for mod in m.modules():
                if isinstance(mod, torch.nn.BatchNorm2d):
                    mod.running_mean.uniform_(-0.2, 0.2)
                    mod.running_var.uniform_(0.5, 1.5)
                    mod.weight.uniform_(0.5, 1.5)
                    mod.bias.uniform_(-0.3, 0.3)
        out.update(sd_np(m, f"{tag}.sd."))
        x = torch.randn(B, C, T, requires_grad=True)
        m.eval()
        with torch.no_grad():
            out[f"{tag}.y_eval"] = m(x).numpy()
        m.train()
        y = m(x)
        y.square().sum().backward()
        out[f"{tag}.x"], out[f"{tag}.y_train"], out[f"{tag}.dx"] = x.detach().numpy(), y.detach().numpy(), x.grad.numpy()
        out.update(grads_np(m, f"{tag}.grad."))
        out.update(sd_np(m, f"{tag}.sd_after."))
    save("g7_eegnet.npz", **out)


def _bn_module_goldens(make, cases, fname):
    out = {}
    for tag, (C, T, B) in cases.items():
        torch.manual_seed(0)
        m = make(C)
        with torch.no_grad():
            for mod in m.modules():
                if isinstance(mod, torch.nn.BatchNorm2d):
                    mod.running_mean.uniform_(-0.2, 0.2)
                    mod.running_var.uniform_(0.5, 1.5)
                    mod.weight.uniform_(0.5, 1.5)
                    mod.bias.uniform_(-0.3, 0.3)
        out.update(sd_np(m, f"{tag}.sd."))
        x = torch.randn(B, C, T, requires_grad=True)
        m.eval()
        with torch.no_grad():
            out[f"{tag}.y_eval"] = m(x).numpy()
        m.train()
        y = m(x)
        y.square().sum().backward()
        out[f"{tag}.x"], out[f"{tag}.y_train"], out[f"{tag}.dx"] = x.detach().numpy(), y.detach().numpy(), x.grad.numpy()
        out.update(grads_np(m, f"{tag}.grad."))
        out.update(sd_np(m, f"{tag}.sd_after."))
    save(fname, **out)


# ---------------------------------------------------------------- G10: CVBlock eval + train (dropout 0)
def g10():
    _bn_module_goldens(lambda C: CVBlock(C, 32, dropout=0.0), {"z6": (6, 250, 5), "z15": (15, 250, 3)},
                       "g10_cvblock.npz")


# ---------------------------------------------------------------- G11: HeadConv_Paper_Version eval + train
def g11():
    _bn_module_goldens(lambda C: HeadConv_Paper_Version(C, 32), {"z6": (6, 250, 5), "z15": (15, 250, 3)},
                       "g11_paperhead.npz")


# ---------------------------------------------------------------- G8: cosine schedule
def g8():
    # src/fast/train/trainer.py is not importable (lightning absent): the 12-line function is
    # evaluated from its published formula (linspace warm-up + half-cosine), values as in SURVEY 8c.
    base, final, epochs, it, wu = 1, 0.1, 200, 5, 10
    warm = np.linspace(0, base, wu * it)
    iters = np.arange(epochs * it - wu * it)
    sched = np.concatenate((warm, final + 0.5 * (base - final) * (1 + np.cos(np.pi * iters / len(iters)))))
    save("g8_cosine.npz", schedule=sched, args=np.array([base, final, epochs, it, wu]))


def g12():
    # Row A12 (notebooks/svm_baseline.ipynb:238): MNE is absent, so this fixture is produced by SciPY ALONE, on a
    # path independent of oracle/fir.py: taps = firwin low-pass differences, application = overlap-add convolution
    # of the odd-reflected row (MNE's _overlap_add_filter does the same with its own FFT blocks).
    from scipy.signal import firwin, oaconvolve
    out = {}
    for tag, (sf, lo, hi, T) in {"n": (250.0, 4.0, 40.0, 795), "s": (256.0, 8.0, 30.0, 512)}.items():
        nyq = sf / 2
        lt, ht = min(max(0.25 * lo, 2.0), lo), min(max(0.25 * hi, 2.0), nyq - hi)
        n = int(round(3.3 * sf / min(lt, ht)))
        n += (n - 1) % 2
        h = np.zeros(n)
        for sign, f0, f1 in ((1.0, hi, hi + ht), (-1.0, lo - lt, lo)):
            m = int(round(3.3 * sf / (f1 - f0)))
            m += 1 - m % 2
            off = (n - m) // 2
            h[off:n - off] += sign * firwin(m, (f0 + f1) / 2, window="hamming", fs=sf)
        x = np.random.default_rng(12).standard_normal((2, 3, T))
        ne = min(n, T) - 1
        xp = np.concatenate([2 * x[..., :1] - x[..., ne:0:-1], x, 2 * x[..., -1:] - x[..., -2:-ne - 2:-1]], axis=-1)
        y = oaconvolve(xp, h.reshape(1, 1, -1), mode="full", axes=-1)[..., (n - 1) // 2 + ne:][..., :T]
        out.update({f"{tag}.args": np.array([sf, lo, hi]), f"{tag}.taps": h, f"{tag}.x": x, f"{tag}.y": y})
    save("g12_fir.npz", **out)


def _randomize_bn(m):
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.running_mean.uniform_(-0.2, 0.2)
                mod.running_var.uniform_(0.5, 1.5)
                mod.weight.uniform_(0.5, 1.5)
                mod.bias.uniform_(-0.3, 0.3)


# ---------------------------------------------------------------- G13: EEGNet_Encoder at the stress configuration's shapes
def g13():
    """BASELINE config 5 head shapes (src/fast/models/fast.py:122-167): EEGNet_Encoder(40 bands x 128 ch = 5120, 32) on
    the [B, 5120, 65] feature map and EEGNet_Encoder(128, 32) on raw [B, 128, 4096] trials.  Inputs are regenerated by
    the tests from the NumPy seed; the input gradient is stored for every `dx_stride`-th channel."""
    out = {}
    for tag, (C, T, B, stride) in {"f5120": (5120, 65, 2, 97), "r128": (128, 4096, 2, 16)}.items():
        torch.manual_seed(13)
        m = EEGNet_Encoder(C, 32, dropout=0.0)
        _randomize_bn(m)
        out.update(sd_np(m, f"{tag}.sd."))
        x = torch.from_numpy(np.random.default_rng(13).standard_normal((B, C, T)).astype(np.float32)).requires_grad_()
        m.eval()
        with torch.no_grad():
            out[f"{tag}.y_eval"] = m(x).numpy()
        m.train()
        y = m(x)
        y.square().sum().backward()
        out[f"{tag}.cfg"] = np.array([C, T, B, stride])
        out[f"{tag}.y_train"], out[f"{tag}.dx_sub"] = y.detach().numpy(), x.grad.numpy()[:, ::stride].copy()
        out.update(grads_np(m, f"{tag}.grad."))
        out.update({k: v for k, v in sd_np(m, f"{tag}.sd_after.").items() if "running" in k or "num_batches" in k})
    save("g13_eegnet_cfg5.npz", **out)


# ---------------------------------------------------------------- G14: stress configuration composed end to end
def g14():
    """BASELINE config 5 composed: trials [2, 128, 4096] @ 1024 Hz -> spec S with the 40-band set, nperseg 1024 /
    noverlap 960 (scipy butter / sosfilt / stft; J = 65) -> the reference EEGNet_Encoder(5120, 32) in train mode
    (dropout 0) -> Linear(32, 5) -> CrossEntropyLoss -> every gradient."""
    fs, T, C, B = 1024.0, 4096, 128, 2
    x = np.random.default_rng(14).standard_normal((B, C, T)).astype(np.float32)
    feats = []
    for _, lo, hi in odsp.BANDS_40:
        sos = ss.butter(4, (lo, hi), "bandpass", fs=fs, output="sos")
        yb = ss.sosfilt(sos, x.astype(np.float64), axis=-1)
        f, _, Z = ss.stft(yb, fs=fs, nperseg=1024, noverlap=960)
        idx = np.where((f >= lo) & (f <= hi))[0]
        feats.append(np.log((np.abs(Z[..., idx, :]) ** 2).mean(axis=-2) + 1e-10))
    feat = np.stack(feats, axis=1).astype(np.float32)                 # [B, 40, C, 65]
    torch.manual_seed(14)
    enc = EEGNet_Encoder(40 * C, 32, dropout=0.0)
    fc = torch.nn.Linear(32, 5)
    _randomize_bn(enc)
    enc.train()
    labels = torch.tensor([3, 1])
    logits = fc(enc(torch.from_numpy(feat).reshape(B, 40 * C, -1)))
    loss = torch.nn.CrossEntropyLoss()(logits, labels)
    loss.backward()
    out = {"cfg": np.array([B, C, T, fs, 1024, 960, 40]), "labels": labels.numpy().astype(np.uint8),
           "feat_sub": feat[:, ::3, ::16].copy(),                      # [2, 14, 8, 65] of the feature map
           "feat_sum": feat.astype(np.float64).sum(axis=(2, 3)),       # per (trial, band) checksum of the rest
           "logits": logits.detach().numpy(), "loss": loss.detach().numpy()}
    out.update(sd_np(enc, "enc.sd."))
    out.update(sd_np(fc, "fc.sd."))
    out.update(grads_np(enc, "enc.grad."))
    out.update(grads_np(fc, "fc.grad."))
    save("g14_cfg5_composed.npz", **out)


# ---------------------------------------------------------------- G15: bf16-mixed (autocast) classifier at 576 channels
def g15_inputs(m, fc):
    """Parameters and inputs of G15 from NumPy's seeded generator (the tests rebuild them the same way, so the 2.4 MB
    cnn2.weight is not stored): U(-1/sqrt(fan_in), 1/sqrt(fan_in)) like the torch default initialisation."""
    rng = np.random.default_rng(15)
    with torch.no_grad():
        for mod, fan in ((m.cnn1, 5), (m.cnn2, 32 * 576), (m.cnn3, 160), (m.cnn4, 160), (fc, 32)):
            for prm in mod.parameters():
                prm.copy_(torch.from_numpy(rng.uniform(-1, 1, tuple(prm.shape)).astype(np.float32) / np.sqrt(fan)))
    x = torch.from_numpy((rng.standard_normal((8, 576, 17)) * 2 - 5).astype(np.float32))      # log-power-like values
    y = torch.from_numpy(rng.integers(0, 5, 8))
    return x, y


def g15():
    """BASELINE config 3: the reference trains with precision='bf16-mixed' (scripts/train_fast.py:277), i.e.
    torch.autocast(bfloat16) around the module.  The reference's Conv4Layers(576, 32) (fast.py:103-119) + Linear(32, 5)
    on a spec-S shaped feature map [8, 576, 17], fp32 master weights: logits, loss and every gradient under CPU
    autocast, and the same in fp32 (what the tolerance of the bf16 path is stated against)."""
    m = Conv4Layers(576, 32)
    fc = torch.nn.Linear(32, 5)
    x, y = g15_inputs(m, fc)
    out = {"labels": y.numpy().astype(np.uint8)}
    for tag, ctx in (("fp32", torch.autocast("cpu", enabled=False)), ("bf16", torch.autocast("cpu", dtype=torch.bfloat16))):
        m.zero_grad(); fc.zero_grad()
        with ctx:
            logits = fc(m(x))
            loss = torch.nn.CrossEntropyLoss()(logits.float(), y)
        loss.backward()
        out[f"{tag}.logits"], out[f"{tag}.loss"] = logits.detach().float().numpy(), loss.detach().numpy()
        for k, v in {**grads_np(m, f"{tag}.cnn.grad."), **grads_np(fc, f"{tag}.fc.grad.")}.items():
            out[k] = v[:, :, ::9].copy() if k.endswith("cnn2.weight") else v      # every 9th of the 576 channels
    save("g15_bf16_autocast.npz", **out)


def g16():
    """BASELINE config 3 for the reference-native modes: G5's FAST(small_config) (same seed, same parameters -- they are
    in g5_fast_small.npz --, same x and labels) under torch.autocast(bfloat16), the precision scripts/train_fast.py:277
    trains with: forward_head features, and logits / loss / every gradient of 'train_head' and 'default'."""
    cfg = small_config()
    torch.manual_seed(0)
    m = FAST(cfg)
    m.train()
    x = torch.randn(2, 8, 500)
    labels = torch.tensor([0, 2])
    out = {}
    with torch.autocast("cpu", dtype=torch.bfloat16):
        out["features"] = m.forward_head(x).detach().float().numpy()
    for mode in ("train_head", "default"):
        m.zero_grad()
        with torch.autocast("cpu", dtype=torch.bfloat16):
            logits = m(x, forward_mode=mode)
            loss = torch.nn.CrossEntropyLoss()(logits.float(), labels)
        loss.backward()
        out[f"{mode}.logits"], out[f"{mode}.loss"] = logits.detach().float().numpy(), loss.detach().numpy()
        for k, p in m.named_parameters():
            if p.grad is not None:
                out[f"{mode}.grad.{k}"] = p.grad.detach().float().numpy().copy()
    save("g16_fast_small_autocast.npz", **out)


if __name__ == "__main__":
    only = sys.argv[1:]
    for fn in (g1, g2, g3, g4, g5_g9, g6, g7, g8, g10, g11, g12, g13, g14, g15, g16):
        if not only or fn.__name__ in only:
            fn()
