"""Oracle (test infrastructure): filterbank + STFT + log band power on the CPU.

Restates, in NumPy float64, the scipy.signal 1.15.3 algorithms that define
"spec S" (SURVEY.md 8d).  Anchors in the reference:

* STFT call:   scripts/global_shap_analysis.py:132
               ``scipy.signal.stft(sig, fs=sfreq, nperseg=64, noverlap=32)``
* magnitude:   scripts/global_shap_analysis.py:135
* band dict:   scripts/global_shap_analysis.py:138-144
* band masks:  scripts/global_shap_analysis.py:151-156  (``f >= lo & f <= hi``,
               both ends inclusive, mean over the selected bins)
* band-pass:   notebooks/svm_baseline.ipynb:238 (MNE, 4-40 Hz; unpinned) ->
               spec S uses a Butterworth SOS cascade instead.

Third-party algorithms restated here (scipy 1.15.3):
  scipy/signal/_signaltools.py  ``sosfilt``   (DF2T biquad cascade, float64)
  scipy/signal/_spectral_py.py  ``stft`` -> ``_spectral_helper`` with the legacy
      defaults window='hann' (periodic), boundary='zeros', padded=True,
      detrend=False, return_onesided=True, scaling='spectrum'.
Filter *design* (``butter(..., output='sos')``) is taken from scipy directly;
the product carries its own design code which the tests compare against it.

Pinned by tests/golden/g1_stft.npz, g2_sos.npz, g3_features.npz (scipy output).
"""
import numpy as np

# scripts/global_shap_analysis.py:138-144 (insertion order is the band order)
BANDS_5 = (("Delta", 0.5, 4.0), ("Theta", 4.0, 8.0), ("Alpha", 8.0, 13.0),
           ("Beta", 13.0, 30.0), ("Gamma", 30.0, 100.0))
# tiles the notebook's 4-40 Hz pass band (svm_baseline.ipynb:238) in 4 Hz steps
BANDS_9 = tuple((f"B{i}", 4.0 + 4.0 * i, 8.0 + 4.0 * i) for i in range(9))
# stress config: 2-Hz bands from 4 to 84 Hz
BANDS_40 = tuple((f"N{i}", 4.0 + 2.0 * i, 6.0 + 2.0 * i) for i in range(40))


def band_edges(bands):
    return [(float(b[-2]), float(b[-1])) for b in bands]


def butter_bandpass_sos(order, lo, hi, fs):
    """float64 [order, 6] SOS table (scipy.signal.butter, third-party)."""
    import scipy.signal as ss
    return ss.butter(order, (lo, hi), "bandpass", fs=fs, output="sos")


def sosfilt(sos, x):
    """Causal zero-state biquad cascade along the last axis, float64.

    Restates scipy ``sosfilt`` (direct-form II transposed per section):
        y  = b0*x + s1 ; s1 = b1*x - a1*y + s2 ; s2 = b2*x - a2*y
    """
    sos = np.asarray(sos, dtype=np.float64)
    y = np.array(x, dtype=np.float64, copy=True)
    flat = y.reshape(-1, y.shape[-1])
    for b0, b1, b2, a0, a1, a2 in sos:
        b0, b1, b2, a1, a2 = b0 / a0, b1 / a0, b2 / a0, a1 / a0, a2 / a0
        s1 = np.zeros(flat.shape[0])
        s2 = np.zeros(flat.shape[0])
        for n in range(flat.shape[1]):
            xn = flat[:, n].copy()
            yn = b0 * xn + s1
            s1 = b1 * xn - a1 * yn + s2
            s2 = b2 * xn - a2 * yn
            flat[:, n] = yn
    return y


def hann_periodic(n):
    """scipy ``get_window('hann', n)`` (fftbins=True -> periodic)."""
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def stft_frames(T, nperseg, noverlap):
    """Number of frames J and total padded length for the legacy scipy stft."""
    hop = nperseg - noverlap
    L = T + 2 * (nperseg // 2)                 # boundary='zeros'
    pad = (-(L - nperseg) % hop) % nperseg      # padded=True
    L += pad
    return (L - nperseg) // hop + 1, L


def stft(x, fs, nperseg=64, noverlap=None):
    """Restatement of ``scipy.signal.stft`` legacy defaults (see module doc).

    x[..., T] -> (f[nfreq], t[J], Z[..., nfreq, J]); complex64 for float32 in.
    """
    x = np.asarray(x)
    if noverlap is None:
        noverlap = nperseg // 2
    hop = nperseg - noverlap
    T = x.shape[-1]
    J, L = stft_frames(T, nperseg, noverlap)
    half = nperseg // 2
    xp = np.zeros(x.shape[:-1] + (L,), dtype=x.dtype)
    xp[..., half:half + T] = x
    win = hann_periodic(nperseg)
    scale = 1.0 / win.sum()                     # scaling='spectrum' -> sqrt(1/sum(w)^2)
    if x.dtype == np.float32:
        win = win.astype(np.float32)
    idx = np.arange(J)[:, None] * hop + np.arange(nperseg)[None, :]
    frames = xp[..., idx] * win                 # [..., J, nperseg]
    Z = np.fft.rfft(frames, axis=-1) * scale    # [..., J, nfreq]
    if x.dtype == np.float32:
        Z = Z.astype(np.complex64)
    Z = np.moveaxis(Z, -1, -2)
    f = np.fft.rfftfreq(nperseg, 1.0 / fs)
    t = np.arange(half, L - half + 1, hop) / float(fs) - half / float(fs)
    return f, t, Z


def band_bins(fs, nperseg, bands):
    """Inclusive [klo, khi] rfft-bin range of each band, or (1, 0) if empty.

    global_shap_analysis.py:153: ``np.where((f >= low) & (f <= high))``.
    """
    f = np.fft.rfftfreq(nperseg, 1.0 / fs)
    out = []
    for lo, hi in band_edges(bands):
        k = np.where((f >= lo) & (f <= hi))[0]
        out.append((int(k[0]), int(k[-1])) if len(k) else (1, 0))
    return out


def band_magnitude(Z, fs, nperseg, bands):
    """global_shap_analysis.py:151-156: mean |Z| over in-band bins -> [..., nb, J]."""
    S = np.abs(Z)
    rows = []
    for klo, khi in band_bins(fs, nperseg, bands):
        if khi >= klo:
            rows.append(S[..., klo:khi + 1, :].mean(axis=-2))
        else:
            rows.append(np.zeros(S.shape[:-2] + S.shape[-1:], S.dtype))
    return np.stack(rows, axis=-2)


def extract_features(trials, *, fs, bands, order=4, nperseg=64, noverlap=None,
                     eps=1e-10, return_filtered=False):
    """Spec S: trials f32 [B,C,T] -> f32 [B, nb, C, J].

    (1) sos_b = butter(order, band_b, 'bandpass', fs, 'sos')
    (2) y_b   = sosfilt(sos_b, trials)              (float64, zero state)
    (3) Z_b   = stft(y_b) with scipy legacy defaults
    (4) P_b   = mean_{k in band_b} |Z_b[k]|^2       (inclusive bin mask)
    (5) log(P + eps)
    """
    x = np.asarray(trials, dtype=np.float64)
    B, C, T = x.shape
    if noverlap is None:
        noverlap = nperseg // 2
    J, _ = stft_frames(T, nperseg, noverlap)
    bins = band_bins(fs, nperseg, bands)
    out = np.zeros((B, len(bins), C, J), dtype=np.float64)
    filt = []
    for bi, ((lo, hi), (klo, khi)) in enumerate(zip(band_edges(bands), bins)):
        y = sosfilt(butter_bandpass_sos(order, lo, hi, fs), x)
        if return_filtered:
            filt.append(y)
        _, _, Z = stft(y, fs, nperseg, noverlap)
        if khi >= klo:
            P = (np.abs(Z[..., klo:khi + 1, :]) ** 2).mean(axis=-2)
        else:
            P = np.zeros((B, C, J))
        out[:, bi] = np.log(P + eps)
    out = out.astype(np.float32)
    if return_filtered:
        return out, np.stack(filt, axis=1)       # [B, nb, C, T] float64
    return out


def extract_features_scipy(trials, *, fs, bands, order=4, nperseg=64, noverlap=None, eps=1e-10):
    """Spec S through scipy's own C routines (``butter`` / ``sosfilt`` / ``stft``): the reference-equivalent CPU
    path of BASELINE.md section 2, used by bench.py's ``cpu_baseline`` (the NumPy restatement above is the checker;
    tests pin the two against each other)."""
    import scipy.signal as ss
    x = np.asarray(trials, dtype=np.float64)
    if noverlap is None:
        noverlap = nperseg // 2
    outs = []
    for (lo, hi), (klo, khi) in zip(band_edges(bands), band_bins(fs, nperseg, bands)):
        y = ss.sosfilt(ss.butter(order, (lo, hi), "bandpass", fs=fs, output="sos"), x, axis=-1)
        _, _, Z = ss.stft(y, fs=fs, nperseg=nperseg, noverlap=noverlap)
        P = (np.abs(Z[..., klo:khi + 1, :]) ** 2).mean(axis=-2) if khi >= klo else np.zeros(Z.shape[:-2] + Z.shape[-1:])
        outs.append(np.log(P + eps))
    return np.stack(outs, axis=1).astype(np.float32)


def synth_trials(B, C=64, T=512, fs=256.0, seed=0, zones=None):
    """Synthetic EEG of SURVEY.md 8d: white noise + class-dependent tone.

    Returns (X f32 [B,C,T], y uint8 [B]); label order = CLASSES.
    """
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((B, C, T)).astype(np.float32)
    y = rng.integers(0, 5, B).astype(np.uint8)
    tone = np.array([6.0, 10.0, 18.0, 26.0, 34.0])
    phase = rng.uniform(0.0, 2.0 * np.pi, B)
    t = np.arange(T) / fs
    if zones is None:
        from .cnn import zone_index_lists
        zones = zone_index_lists()
    for i in range(B):
        ch = [c for c in zones[int(y[i]) % len(zones)] if c < C]
        X[i, ch] += (0.5 * np.sin(2.0 * np.pi * tone[y[i]] * t + phase[i])).astype(np.float32)
    return X, y
