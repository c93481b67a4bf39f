"""GPU parity: HIP filterbank / STFT / fused extractor (through the C ABI) vs the oracle and goldens."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from oracle import dsp as odsp

pytestmark = pytest.mark.gpu

TOL_FILT = 1e-4     # north star: features within 1e-4 rel (fp32); filtered signal relative to its peak
TOL_FEAT = 1e-4     # log-domain absolute tolerance (SURVEY 8d parity gates)


@pytest.fixture(scope="module")
def isd():
    import isd_amd
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return isd_amd


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).cuda()


# ---------------------------------------------------------------- filterbank
@pytest.mark.parametrize("tag,bands", [("b5", odsp.BANDS_5), ("b9", odsp.BANDS_9), ("b40", odsp.BANDS_40)])
def test_filterbank_matches_scipy_golden(isd, tag, bands):
    g = load_golden("g2_sos.npz")
    x, sel, fs = g[f"{tag}_x"], g[f"{tag}_sel"], float(g[f"{tag}_fs"])
    fb = isd.Filterbank(bands, fs, order=4, precision="auto")
    y = fb.forward(dev(x)).cpu().numpy()
    assert y.shape == (x.shape[0], len(bands), x.shape[1], x.shape[2])
    for j, b in enumerate(sel):
        assert rel_err(y[:, b], g[f"{tag}_y"][:, j]) < TOL_FILT, (tag, b, fb.precision)


@pytest.mark.parametrize("T", [512, 800, 4096, 250, 1000, 795, 33, 2048, 2100])
@pytest.mark.parametrize("precision", ["f32", "f64"])
def test_filterbank_vs_oracle_shapes(isd, T, precision):
    rng = np.random.default_rng(T)
    B, Cc = 3, 5                                   # 15 rows: not a multiple of the 4 rows per wave
    x = rng.standard_normal((B, Cc, T)).astype(np.float32)
    bands = odsp.BANDS_9[:3] + odsp.BANDS_5[3:]
    fb = isd.Filterbank(bands, 256.0, precision=precision)
    y = fb.forward(dev(x)).cpu().numpy()
    for bi, (lo, hi) in enumerate(odsp.band_edges(bands)):
        ref = odsp.sosfilt(odsp.butter_bandpass_sos(4, lo, hi, 256.0), x)
        tol = 2e-6 if precision == "f64" else TOL_FILT
        assert rel_err(y[:, bi], ref) < tol, (T, precision, bi)


def test_filterbank_auto_precision_policy(isd):
    assert isd.Filterbank(odsp.BANDS_9, 256.0).precision == "f32"
    assert isd.Filterbank(odsp.BANDS_5, 256.0).precision == "mixed"    # 0.5-4 Hz needs the fp64 recursion, the rest fp32
    assert isd.Filterbank(odsp.BANDS_40, 1024.0).precision == "mixed"  # the five lowest 2-Hz bands (4 - 14 Hz)
    assert isd.Filterbank(odsp.BANDS_5, 256.0, precision="f64").precision == "f64"
    assert isd.Filterbank(odsp.BANDS_5[:1], 256.0).precision == "f64"


def test_filterbank_empty_batch_and_errors(isd):
    fb = isd.Filterbank(odsp.BANDS_9, 256.0)
    y = fb.forward(torch.empty((0, 64, 512), device="cuda"))
    assert y.shape == (0, 9, 64, 512)
    with pytest.raises(TypeError):
        fb.forward(torch.zeros(2, 4, 512))                               # CPU tensor: no CPU path
    with pytest.raises(ValueError):
        isd.Filterbank([(10.0, 200.0)], 256.0)                           # above Nyquist


def test_filterbank_linearity_at_full_size(isd):
    # BASELINE config 2 size: 4096 x 64 x 512, 9 bands -> property check (oracle too slow here)
    torch.manual_seed(0)
    fb = isd.Filterbank(odsp.BANDS_9, 256.0)
    a = torch.randn(4096, 64, 512, device="cuda")
    b = torch.randn(4096, 64, 512, device="cuda")
    ya, yb = fb.forward(a), fb.forward(b)
    yab = fb.forward(0.5 * a - 2.0 * b)
    err = (yab - (0.5 * ya - 2.0 * yb)).abs().max() / yab.abs().max()
    assert float(err) < 2e-5
    # spot-check 3 rows against the oracle
    idx = [(0, 0), (2047, 31), (4095, 63)]
    xs = np.stack([a[i, c].cpu().numpy() for i, c in idx])
    for bi, (lo, hi) in enumerate(odsp.band_edges(odsp.BANDS_9)):
        ref = odsp.sosfilt(odsp.butter_bandpass_sos(4, lo, hi, 256.0), xs)
        got = np.stack([ya[i, bi, c].cpu().numpy() for i, c in idx])
        assert rel_err(got, ref) < TOL_FILT


# ---------------------------------------------------------------- STFT
@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_stft_matches_scipy_golden(isd, tag):
    g = load_golden("g1_stft.npz")
    T, fs, nperseg = [int(v) for v in g[f"{tag}_cfg"]]
    st = isd.Stft(T, nperseg, nperseg // 2)
    Z = st.forward(dev(g[f"{tag}_x"])).cpu().numpy()
    ref = g[f"{tag}_Z"]
    assert Z.shape == ref.shape
    assert np.abs(Z - ref).max() < 1e-5 * np.abs(ref).max()
    # reference use: one trace, five bands, mean magnitude (global_shap_analysis.py:151-156)
    bins = isd.band_bins(fs, nperseg, odsp.BANDS_5)
    x = dev(g[f"{tag}_x"])                                   # [2, 3, T] -> B=2, one shared signal, C=3
    bm = st.bandpower(x[:, None].contiguous(), bins, mode="magnitude", shared_signal=True).cpu().numpy()
    ref_bm = np.moveaxis(g[f"{tag}_band5"], -2, 1)           # [2, 3, 5, J] -> [2, 5, 3, J]
    np.testing.assert_allclose(bm, ref_bm, rtol=2e-4, atol=1e-6)


@pytest.mark.parametrize("T,nperseg,noverlap", [(4096, 1024, 960), (4000, 1024, 960), (1000, 1024, 960),
                                                 (2048, 256, 224), (130, 512, 448), (4096, 256, 192)])
@pytest.mark.parametrize("mode", ["power", "magnitude", "logpower"])
def test_overlapped_frames_bandpower_vs_oracle(isd, T, nperseg, noverlap, mode):
    """Heavily overlapped frames (stress configuration): the block-sum kernel against the oracle's STFT and band means."""
    rng = np.random.default_rng(T + nperseg)
    bins = [(1, 1), (4, 6), (nperseg // 2 - 6, nperseg // 2 - 1), (9, 10)]      # 1, 3, 6 and 2 bins per band
    y = rng.standard_normal((2, len(bins), 3, T)).astype(np.float32)
    st = isd.Stft(T, nperseg, noverlap)
    got = st.bandpower(dev(y), bins, mode=mode).cpu().numpy()
    _, _, Z = odsp.stft(y.astype(np.float64), 256.0, nperseg, noverlap)          # [2, nb, 3, nfreq, J]
    ref = np.empty(got.shape)
    for b, (lo, hi) in enumerate(bins):
        P = np.abs(Z[:, b, :, lo:hi + 1, :]) ** 2
        ref[:, b] = (np.sqrt(P) if mode == "magnitude" else P).mean(axis=-2)
    if mode == "logpower":
        assert np.abs(got - np.log(ref + 1e-10)).max() < TOL_FEAT
    else:
        assert np.abs(got - ref).max() < 2e-5 * ref.max()


@pytest.mark.parametrize("T,nperseg,noverlap", [(512, 64, 32), (500, 64, 32), (800, 64, 48), (100, 16, 4),
                                                 (4096, 1024, 960), (37, 8, 0), (300, 256, 128)])
def test_stft_vs_oracle_param_sweep(isd, T, nperseg, noverlap):
    x = np.random.default_rng(T + nperseg).standard_normal((7, T)).astype(np.float32)
    st = isd.Stft(T, nperseg, noverlap)
    _, _, ref = odsp.stft(x, 256.0, nperseg, noverlap)
    assert st.n_frames == ref.shape[-1] and st.n_bins == ref.shape[-2]
    Z = st.forward(dev(x)).cpu().numpy()
    assert np.abs(Z - ref).max() < 1e-5 * np.abs(ref).max()


def test_stft_rejects_bad_plans(isd):
    from isd_amd._lib import IsdError
    for args in [(512, 60, 30), (512, 64, 64), (512, 4, 0), (0, 64, 32)]:
        with pytest.raises(IsdError):
            isd.Stft(*args)


# ---------------------------------------------------------------- spec-S features
@pytest.mark.parametrize("tag,bands", [("c1", odsp.BANDS_5), ("c2", odsp.BANDS_9), ("c5", odsp.BANDS_40[:6]),
                                       ("c800", odsp.BANDS_9)])
@pytest.mark.parametrize("fused", [False, True])
def test_extract_features_matches_scipy_golden(isd, tag, bands, fused):
    g = load_golden("g3_features.npz")
    B, Cc, T, fs, nperseg, nov, nb = g[f"{tag}_cfg"]
    B, Cc, T, nperseg, nov = int(B), int(Cc), int(T), int(nperseg), int(nov)
    if fused and not ((nperseg == 64 and T <= 1024) or (nperseg - nov == 64 and T <= 4096)):
        pytest.skip("fused kernels cover nperseg 64 / hop 32 / T<=1024 and hop 64 / T<=4096")
    x = g[f"{tag}_x"] if f"{tag}_x" in g.files else \
        np.random.default_rng(3).standard_normal((B, Cc, T)).astype(np.float32)
    feat = isd.extract_features(dev(x), fs=float(fs), bands=bands, nperseg=nperseg, noverlap=nov, fused=fused)
    assert feat.shape == g[f"{tag}_feat"].shape and feat.dtype == torch.float32
    np.testing.assert_allclose(feat.cpu().numpy(), g[f"{tag}_feat"], rtol=0, atol=TOL_FEAT)


@pytest.mark.parametrize("T,fs", [(250, 250.0), (512, 256.0)])
def test_features_relative_tolerance_on_near_silent_frames(isd, T, fs):
    """North-star gate: features within 1e-4 relative.  Frames that hold almost no in-band power (the last frames of
    a short trial: P six orders below the row's typical power) are where the fp32 cascade's absolute noise shows in
    the log domain; the relative bound must hold there too, and the plain 1e-4 log-domain bound wherever the power
    is within four orders of typical."""
    X, _ = odsp.synth_trials(6, 64, T, fs, seed=3)
    ref = odsp.extract_features_scipy(X, fs=fs, bands=odsp.BANDS_9).astype(np.float64)
    got = isd.extract_features(dev(X), fs=fs, bands=odsp.BANDS_9).cpu().numpy().astype(np.float64)
    err = np.abs(got - ref)
    assert (err <= 1e-4 * np.maximum(1.0, np.abs(ref))).all()
    typical = np.median(ref, axis=-1, keepdims=True)
    loud = ref > typical - np.log(1e4)
    assert err[loud].max() < TOL_FEAT
    got64 = isd.extract_features(dev(X), fs=fs, bands=odsp.BANDS_9, precision="f64").cpu().numpy()
    assert np.abs(got64 - ref).max() < 3e-5                      # fp64 cascade: fp32 DFT noise only


def test_extract_features_numpy_in_numpy_out(isd):
    x = np.random.default_rng(5).standard_normal((2, 4, 512)).astype(np.float32)
    f = isd.extract_features(x, fs=256.0, bands=odsp.BANDS_9)
    assert isinstance(f, np.ndarray) and f.shape == (2, 9, 4, 17)
    ref = odsp.extract_features(x, fs=256.0, bands=odsp.BANDS_9)
    np.testing.assert_allclose(f, ref, rtol=0, atol=TOL_FEAT)


@pytest.mark.parametrize("T,fs,nperseg,noverlap,bands", [
    (4096, 1024.0, 1024, 960, odsp.BANDS_40[:7]),       # stress shape; bands on both sides of the fp32 / fp64 split
    (4096, 1024.0, 1024, 960, odsp.BANDS_40[20:24]),    # fp32 bands only
    (3000, 1024.0, 1024, 960, odsp.BANDS_40[:3]),       # ragged row: last pass and last block partial
    (1500, 512.0, 256, 192, [("a", 6.0, 10.0), ("b", 20.0, 24.0)]),   # one pass, 4 blocks per frame
    # whole 512-sample passes: four rows per wave (fused_rows4_kernel / fb_rows4_kernel), 3 and 4 passes, rows shorter
    # than 64 blocks, 16 and 4 blocks per frame; 15 rows = a ragged last quad
    (1536, 1024.0, 1024, 960, odsp.BANDS_40[:5]),
    (2048, 512.0, 256, 192, [("a", 6.0, 10.0), ("b", 20.0, 24.0), ("c", 30.0, 38.0)]),
])
def test_fused_long_rows_vs_oracle_and_two_kernel_path(isd, T, fs, nperseg, noverlap, bands):
    X, _ = odsp.synth_trials(3, 5, T, fs, seed=T)
    fx = isd.FeatureExtractor(T, fs, bands, nperseg=nperseg, noverlap=noverlap)
    assert fx.can_fuse
    a = fx(dev(X), fused=True).cpu().numpy().astype(np.float64)
    b = fx(dev(X), fused=False).cpu().numpy().astype(np.float64)
    ref = odsp.extract_features_scipy(X, fs=fs, bands=bands, nperseg=nperseg, noverlap=noverlap).astype(np.float64)
    assert a.shape == ref.shape
    for got in (a, b):
        assert (np.abs(got - ref) <= 1e-4 * np.maximum(1.0, np.abs(ref))).all()     # north-star gate: 1e-4 relative
    loud = ref > np.median(ref, axis=-1, keepdims=True) - np.log(1e4)
    assert np.abs(a - ref)[loud].max() < TOL_FEAT and np.abs(a - b)[loud].max() < TOL_FEAT


@pytest.mark.parametrize("T", [512, 480, 250, 33, 800, 1024, 1000, 513, 795])
def test_fused_equals_two_kernel_path(isd, T):
    x = torch.randn(5, 3, T, device="cuda")
    fx = isd.FeatureExtractor(T, 256.0, odsp.BANDS_5[1:] + odsp.BANDS_9[5:])
    assert fx.can_fuse
    a, b = fx(x, fused=True), fx(x, fused=False)
    assert a.shape == b.shape == (5, fx.n_bands, 3, fx.n_frames)
    assert float((a - b).abs().max()) < 5e-5
    ref = odsp.extract_features(x.cpu().numpy(), fs=256.0, bands=odsp.BANDS_5[1:] + odsp.BANDS_9[5:]).astype(np.float64)
    got = a.cpu().numpy().astype(np.float64)
    # north-star gate (1e-4 relative) everywhere; the plain log-domain 1e-4 wherever the frame is not near-silent
    # (the last frame of a ragged row holds a handful of samples: P five orders below the row's typical power)
    assert (np.abs(got - ref) <= 1e-4 * np.maximum(1.0, np.abs(ref))).all()
    loud = ref > np.median(ref, axis=-1, keepdims=True) - np.log(1e4)
    assert np.abs(got - ref)[loud].max() < TOL_FEAT


def test_features_full_size_consistency(isd):
    # BASELINE config 2 size; fused vs materialising path must agree everywhere
    torch.manual_seed(1)
    x = torch.randn(4096, 64, 512, device="cuda")
    fx = isd.FeatureExtractor(512, 256.0, odsp.BANDS_9)
    a, b = fx(x, fused=True), fx(x, fused=False)
    assert a.shape == (4096, 9, 64, 17)
    assert torch.isfinite(a).all()
    assert float((a - b).abs().max()) < 5e-5
    ref = odsp.extract_features(x[:1].cpu().numpy(), fs=256.0, bands=odsp.BANDS_9)
    np.testing.assert_allclose(a[:1].cpu().numpy(), ref, rtol=0, atol=TOL_FEAT)


def test_reference_native_trials_full_batch(isd):
    """4096 trials x 64 ch x 800 samples @ 250 Hz (the reference's own trial, preprocess.py:62) through the fused
    extractor (two 16-lane groups per row): sampled rows against the oracle, every value finite, and a batch of one
    equal to the same trial inside the full batch."""
    torch.manual_seed(2)
    x = torch.randn(4096, 64, 800, device="cuda")
    fx = isd.FeatureExtractor(800, 250.0, odsp.BANDS_9)
    assert fx.can_fuse and fx.n_frames == 26
    a = fx(x)
    assert a.shape == (4096, 9, 64, 26) and bool(torch.isfinite(a).all())
    pick = [0, 2047, 4095]
    ref = odsp.extract_features_scipy(x[pick][:, ::21].cpu().numpy(), fs=250.0, bands=odsp.BANDS_9).astype(np.float64)
    got = a[pick][:, :, ::21].cpu().numpy().astype(np.float64)
    assert (np.abs(got - ref) <= 1e-4 * np.maximum(1.0, np.abs(ref))).all()
    loud = ref > np.median(ref, axis=-1, keepdims=True) - np.log(1e4)
    assert np.abs(got - ref)[loud].max() < TOL_FEAT
    assert torch.equal(fx(x[2047:2048].contiguous())[0], a[2047])
