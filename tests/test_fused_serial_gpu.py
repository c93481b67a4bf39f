"""GPU: the one-row-per-lane fused extractor (csrc/fb.hip fused_serial_kernel, round 4) against the sixteen-lanes-per-row
kernel it replaces for rows of whole 32-sample chunks, against the materialising two-kernel path and against the fp64
oracle -- same inputs, both kernels selected through ISD_FUSED_SERIAL, the launched family read back from the library
(isd_features_fused_last_path), so a silent fall-back to the old kernel cannot pass for the new one.

The cascade arithmetic of the two kernels is the same operation for operation (only the fp64 summation order of the
carried chunk-end states differs), so their feature maps may differ by the band-power stage alone: the windowed direct
DFT of the old kernel against unwindowed symmetric-pair sums + the Hann window in the frequency domain.
"""
import os

import numpy as np
import pytest
import torch

from oracle import dsp as odsp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def isd():
    import isd_amd
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return isd_amd


def _run(fx, x, serial, **kw):
    from isd_amd import _lib
    old = os.environ.get("ISD_FUSED_SERIAL")
    os.environ["ISD_FUSED_SERIAL"] = "1" if serial else "0"
    try:
        out = fx(x, fused=True, **kw)
        torch.cuda.synchronize()
        path = int(_lib.lib().isd_features_fused_last_path())
    finally:
        if old is None:
            del os.environ["ISD_FUSED_SERIAL"]
        else:
            os.environ["ISD_FUSED_SERIAL"] = old
    return out, path


@pytest.mark.parametrize("B,C,T,fs,bands", [
    (8, 64, 512, 256.0, odsp.BANDS_9),            # the headline shape: 3 waves x 3 bands
    (3, 5, 512, 256.0, odsp.BANDS_9),             # 15 rows: a ragged workgroup
    (2, 64, 800, 250.0, odsp.BANDS_9),            # the reference-native trial (25 chunks; one-bin bands at 250 Hz)
    (2, 7, 1024, 256.0, odsp.BANDS_9[:4]),        # 2 waves x 2 bands
    (2, 3, 480, 256.0, odsp.BANDS_9[2:7]),        # 2 waves, 3 + 2 bands
    (1, 4, 32, 256.0, odsp.BANDS_9[:1]),          # one chunk, one band, one wave
    (2, 4, 64, 256.0, odsp.BANDS_9[:2]),
    (3, 9, 256, 256.0, tuple(odsp.BANDS_9) + (("x1", 10.0, 14.0), ("x2", 18.0, 22.0), ("x3", 26.0, 30.0))),   # 12 bands: 4 waves
])
def test_serial_kernel_equals_lane_scan_kernel_and_oracle(isd, B, C, T, fs, bands):
    X, _ = odsp.synth_trials(B, C, T, fs, seed=T + C)
    x = torch.from_numpy(X).cuda()
    fx = isd.FeatureExtractor(T, fs, bands)
    assert fx.fb.precision == "f32"
    a, pa = _run(fx, x, True)
    b, pb = _run(fx, x, False)
    assert (pa, pb) == (2, 1), (pa, pb)                   # the new kernel ran, and the old one ran
    two = fx(x, fused=False)
    # scipy's stft shortens its window for rows below 64 samples; the NumPy restatement keeps the 64-point frames
    oracle = odsp.extract_features_scipy if T >= 64 else odsp.extract_features
    ref = oracle(X, fs=fs, bands=bands).astype(np.float64)
    got = a.cpu().numpy().astype(np.float64)
    # north-star gate everywhere, the plain log-domain 1e-4 on frames that are not near-silent
    assert (np.abs(got - ref) <= 1e-4 * np.maximum(1.0, np.abs(ref))).all()
    loud = ref > np.median(ref, axis=-1, keepdims=True) - np.log(1e4)
    assert np.abs(got - ref)[loud].max() < 1e-4
    # the kernels share their cascade arithmetic: what is left is the fp32 noise of two ways to take a 64-point DFT
    d_old = (a - b).abs().cpu().numpy()
    assert d_old[loud].max() < 1e-5, d_old[loud].max()
    assert float((a - b).abs().max()) < 5e-5 and float((a - two).abs().max()) < 5e-5


def test_serial_kernel_full_batch_against_both_paths(isd):
    """BASELINE config 2's batch: every value of the new kernel against the old fused kernel and the two-kernel path."""
    torch.manual_seed(4)
    x = torch.randn(4096, 64, 512, device="cuda")
    fx = isd.FeatureExtractor(512, 256.0, odsp.BANDS_9)
    a, pa = _run(fx, x, True)
    b, pb = _run(fx, x, False)
    assert (pa, pb) == (2, 1)
    assert bool(torch.isfinite(a).all())
    assert float((a - b).abs().max()) < 1e-5
    two = fx(x, fused=False)
    assert float((a - two).abs().max()) < 5e-5
    assert torch.equal(_run(fx, x[1234:1235].contiguous(), True)[0][0], a[1234])      # batch-size independent


def test_serial_kernel_bf16_map_and_other_modes(isd):
    X, _ = odsp.synth_trials(4, 64, 512, 256.0, seed=9)
    x = torch.from_numpy(X).cuda()
    fx = isd.FeatureExtractor(512, 256.0, odsp.BANDS_9)
    a, pa = _run(fx, x, True)
    h, ph = _run(fx, x, True, out_dtype=torch.bfloat16)
    assert pa == 2 and ph == 2
    assert torch.equal(h, a.to(torch.bfloat16))           # the bf16 map is the fp32 map rounded to nearest even
    for mode in ("power", "magnitude"):
        fm = isd.FeatureExtractor(512, 256.0, odsp.BANDS_9, mode=mode)
        s, ps = _run(fm, x, True)
        o, po = _run(fm, x, False)
        assert (ps, po) == (2, 1)
        assert float(((s - o).abs() / o.abs().clamp_min(1e-30)).max()) < 2e-5, mode


def test_shapes_the_serial_kernel_does_not_take_fall_back(isd):
    """Rows that are not whole chunks, bands of more than two bins and mixed-precision plans keep the old kernels."""
    for T, bands in ((250, odsp.BANDS_9), (500, odsp.BANDS_9), (512, odsp.BANDS_5[1:])):
        x = torch.randn(2, 3, T, device="cuda")
        fx = isd.FeatureExtractor(T, 256.0, bands)
        _, p = _run(fx, x, True)
        assert p == 1, (T, p)


def test_serial_kernel_random_shapes_against_the_lane_scan_kernel(isd):
    """Seeded sweep over batch, channel count (row groups that start and end inside a trial, one-channel trials, a
    single row), row length (1 .. 32 chunks) and band subsets (1 .. 12 bands: every wave / bands-per-wave split the
    launcher can choose): the two extractors agree to 5e-5 everywhere and both stay inside the north-star gate."""
    rng = np.random.default_rng(2024)
    extra = (("x1", 10.0, 14.0), ("x2", 18.0, 22.0), ("x3", 26.0, 30.0))
    pool = tuple(odsp.BANDS_9) + extra
    for trial in range(14):
        B = int(rng.integers(1, 6))
        C = int(rng.choice([1, 2, 3, 7, 16, 33, 64, 65]))
        T = 32 * int(rng.integers(1, 33))
        nb = int(rng.integers(1, 13))
        idx = np.sort(rng.choice(len(pool), nb, replace=False))
        bands = tuple(pool[i] for i in idx)
        X = rng.standard_normal((B, C, T)).astype(np.float32)
        x = torch.from_numpy(X).cuda()
        fx = isd.FeatureExtractor(T, 256.0, bands)
        a, pa = _run(fx, x, True)
        b, pb = _run(fx, x, False)
        assert (pa, pb) == (2, 1), (trial, pa, pb)
        assert bool(torch.isfinite(a).all())
        assert float((a - b).abs().max()) < 5e-5, (trial, B, C, T, nb, float((a - b).abs().max()))
        if T >= 64 and trial % 3 == 0:
            ref = odsp.extract_features_scipy(X, fs=256.0, bands=bands).astype(np.float64)
            got = a.cpu().numpy().astype(np.float64)
            assert (np.abs(got - ref) <= 1e-4 * np.maximum(1.0, np.abs(ref))).all(), (trial, B, C, T, nb)


def test_serial_kernel_empty_batch_and_bands_per_wave_overrides(isd):
    fx = isd.FeatureExtractor(512, 256.0, odsp.BANDS_9)
    out, _ = _run(fx, torch.empty(0, 64, 512, device="cuda"), True)
    assert out.shape == (0, 9, 64, 17)
    X, _ = odsp.synth_trials(3, 64, 512, 256.0, seed=5)
    x = torch.from_numpy(X).cuda()
    base, p = _run(fx, x, True)
    assert p == 2
    old = {k: os.environ.get(k) for k in ("ISD_SERIAL_BPW", "ISD_SERIAL_GROUPS")}
    try:
        for bpw, groups in (("1", "1"), ("2", "2"), ("3", "1"), ("3", "4")):      # every launch geometry: the same values
            os.environ["ISD_SERIAL_BPW"], os.environ["ISD_SERIAL_GROUPS"] = bpw, groups
            alt, p = _run(fx, x, True)
            assert p == 2 and torch.equal(alt, base), (bpw, groups)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
