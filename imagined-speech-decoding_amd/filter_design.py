"""Butterworth band-pass design on the host (float64), no scipy dependency.

Spec S step 1: ``butter(order, (lo, hi), 'bandpass', fs=fs, output='sos')``.
The standard construction is restated here: analog low-pass prototype ->
low-pass-to-band-pass transform (pre-warped edges) -> bilinear transform.
Two factorizations of the same transfer function are produced:

* ``butter_bandpass_resonators``: what the HIP filterbank consumes -- every
  section is (1 - z^-2) / (1 + a1 z^-1 + a2 z^-2) plus one gain per band;
* ``butter_bandpass_sos``: scipy's [order, 6] table with its 'nearest' pole/zero
  pairing and section order, for callers that want the familiar layout.
"""
import numpy as np


def _bandpass_zpk(order, lo, hi, fs):
    if not (0.0 < lo < hi < fs / 2.0):
        raise ValueError(f"band edges must satisfy 0 < lo < hi < fs/2, got ({lo}, {hi}) at fs={fs}")
    if order < 1:
        raise ValueError("order must be >= 1")
    fs2 = 2.0 * fs
    w1 = fs2 * np.tan(np.pi * lo / fs)             # pre-warped analog edges
    w2 = fs2 * np.tan(np.pi * hi / fs)
    bw, wo = w2 - w1, np.sqrt(w1 * w2)
    m = np.arange(-order + 1, order, 2)
    p_lp = -np.exp(1j * np.pi * m / (2 * order))   # Butterworth prototype poles (left half plane)
    p_s = p_lp * bw / 2.0
    root = np.sqrt(p_s ** 2 - wo ** 2)
    p_bp = np.concatenate((p_s + root, p_s - root))
    k_bp = bw ** order                             # prototype gain 1; zeros: `order` at s = 0
    p_z = (fs2 + p_bp) / (fs2 - p_bp)              # bilinear
    z_z = np.concatenate((np.ones(order), -np.ones(order)))
    k_z = k_bp * np.real(fs2 ** order / np.prod(fs2 - p_bp))
    return z_z, p_z, float(k_z)


def butter_bandpass_resonators(order, lo, hi, fs):
    """-> (a12 float64 [order, 2], gain float).  Sections sorted by pole radius (least resonant first)."""
    _, p, k = _bandpass_zpk(order, lo, hi, fs)
    tol = 1e-12
    pu = p[np.imag(p) > tol]
    pr = np.sort(np.real(p[np.abs(np.imag(p)) <= tol]))       # odd order + wide band: two real poles
    rows = [(-2.0 * q.real, abs(q) ** 2, abs(q)) for q in pu]
    rows += [(-(pr[i] + pr[i + 1]), pr[i] * pr[i + 1], max(abs(pr[i]), abs(pr[i + 1]))) for i in range(0, len(pr), 2)]
    if len(rows) != order:
        raise ValueError("band-pass poles could not be grouped into second-order sections")
    rows.sort(key=lambda r: r[2])
    return np.array([(r[0], r[1]) for r in rows]), k


def butter_bandpass_sos(order, lo, hi, fs):
    """scipy-layout SOS [order, 6] ('nearest' pairing: worst pole last, paired with its nearest zeros)."""
    z, p, k = _bandpass_zpk(order, lo, hi, fs)
    pu = list(p[np.imag(p) > 1e-12])
    if len(pu) != order:
        raise ValueError("real poles (odd order, very wide band): only the resonator form is provided")
    zs = list(np.sort(z))
    sos = np.zeros((order, 6))
    for si in range(order - 1, -1, -1):
        i = int(np.argmin([abs(1.0 - abs(q)) for q in pu]))
        p1 = pu.pop(i)
        z1 = zs.pop(int(np.argmin([abs(q - p1) for q in zs])))
        z2 = zs.pop(int(np.argmin([abs(q - p1) for q in zs])))
        sos[si, :3] = (1.0, -(z1 + z2), z1 * z2)
        sos[si, 3:] = (1.0, -2.0 * p1.real, abs(p1) ** 2)
    sos[0, :3] *= k
    return sos


def filterbank_tables(bands, fs, order=4):
    """Stack the resonator tables of several (lo, hi) bands -> (a12 [nb, order, 2], gain [nb])."""
    a, g = zip(*(butter_bandpass_resonators(order, lo, hi, fs) for lo, hi in bands))
    return np.stack(a), np.array(g)


# ----------------------------------------------------------------------------- zero-phase FIR (row A12)
# MNE's documented defaults for ``mne.filter.filter_data(X, sfreq, l_freq, h_freq)`` (the reference's only call:
# notebooks/svm_baseline.ipynb:238-239) restated: method 'fir', fir_design 'firwin', fir_window 'hamming',
# phase 'zero', transition bands 'auto', filter_length 'auto'.  MNE itself is not vendored in the reference and not
# installed here, so the design is pinned against scipy.signal.firwin only (tests/test_oracle.py).
FIR_LENGTH_FACTORS = {"hann": 3.1, "hamming": 3.3, "blackman": 5.0}


def _fir_window(name, n):
    if n == 1:
        return np.ones(1)
    t = 2.0 * np.pi * np.arange(n) / (n - 1)
    if name == "hamming":
        return 0.54 - 0.46 * np.cos(t)
    if name == "hann":
        return 0.5 - 0.5 * np.cos(t)
    if name == "blackman":
        return 0.42 - 0.5 * np.cos(t) + 0.08 * np.cos(2.0 * t)
    raise ValueError(f"fir_window must be one of {sorted(FIR_LENGTH_FACTORS)}, got {name!r}")


def firwin_lowpass(numtaps, cutoff, window="hamming"):
    """Windowed-sinc low-pass with unit DC gain; ``cutoff`` as a fraction of Nyquist
    (== scipy.signal.firwin(numtaps, cutoff, window=window, pass_zero=True, fs=2))."""
    m = np.arange(numtaps) - 0.5 * (numtaps - 1)
    h = cutoff * np.sinc(cutoff * m) * _fir_window(window, numtaps)
    return h / h.sum()


def fir_transition_bands(sfreq, l_freq, h_freq, l_trans_bandwidth="auto", h_trans_bandwidth="auto"):
    """'auto' transition widths: min(max(0.25 f, 2), f) below, min(max(0.25 f, 2), nyquist - f) above."""
    lt = ht = None
    if l_freq is not None:
        lt = min(max(0.25 * l_freq, 2.0), l_freq) if isinstance(l_trans_bandwidth, str) else float(l_trans_bandwidth)
        if lt <= 0 or l_freq - lt < 0:
            raise ValueError(f"l_trans_bandwidth {lt} puts the lower stop edge below 0 Hz")
    if h_freq is not None:
        ht = (min(max(0.25 * h_freq, 2.0), sfreq / 2.0 - h_freq) if isinstance(h_trans_bandwidth, str)
              else float(h_trans_bandwidth))
        if ht <= 0 or h_freq + ht > sfreq / 2.0:
            raise ValueError(f"h_trans_bandwidth {ht} puts the upper stop edge above Nyquist")
    return lt, ht


def fir_design(sfreq, l_freq, h_freq, filter_length="auto", l_trans_bandwidth="auto", h_trans_bandwidth="auto",
               fir_window="hamming"):
    """Odd-length symmetric taps (float64) of the zero-phase low-, high- or band-pass.

    Built the 'firwin' way: one windowed-sinc low-pass per transition band (its own length from the band's
    width), centred in the full-length filter and added or subtracted.  At (250, 4, 40) this gives 413 taps =
    lowpass(83 taps, 45 Hz) - lowpass(413 taps, 3 Hz)."""
    sfreq = float(sfreq)
    if l_freq is None and h_freq is None:
        raise ValueError("l_freq and h_freq cannot both be None")
    if l_freq is not None and h_freq is not None and not l_freq < h_freq:
        raise NotImplementedError("band-stop (l_freq >= h_freq) is not provided")
    nyq = sfreq / 2.0
    for f in (l_freq, h_freq):
        if f is not None and not 0 < f < nyq:
            raise ValueError(f"cut-off {f} Hz must lie in (0, {nyq}) Hz")
    factor = FIR_LENGTH_FACTORS.get(fir_window)
    if factor is None:
        raise ValueError(f"fir_window must be one of {sorted(FIR_LENGTH_FACTORS)}, got {fir_window!r}")
    lt, ht = fir_transition_bands(sfreq, l_freq, h_freq, l_trans_bandwidth, h_trans_bandwidth)
    if isinstance(filter_length, str):
        if filter_length != "auto":
            raise NotImplementedError("filter_length must be 'auto' or a number of samples")
        n = max(int(round(factor * sfreq / min(w for w in (lt, ht) if w is not None))), 1)
    else:
        n = int(filter_length)
        if n < 1:
            raise ValueError("filter_length must be positive")
    n += (n - 1) % 2                                                       # odd: linear phase type I
    # frequency / gain break points as fractions of Nyquist, DC first
    if l_freq is None:
        freq, gain = [0.0, h_freq, h_freq + ht], [1, 1, 0]
    elif h_freq is None:
        freq, gain = [l_freq - lt, l_freq, nyq], [0, 1, 1]
    else:
        freq, gain = [l_freq - lt, l_freq, h_freq, h_freq + ht], [0, 1, 1, 0]
    if freq[-1] != nyq:
        freq, gain = freq + [nyq], gain + [0]
    if freq[0] != 0:
        freq, gain = [0.0] + freq, [0] + gain
    freq = np.asarray(freq, dtype=np.float64) / nyq
    h = np.zeros(n)
    prev_f, prev_g = freq[-1], gain[-1]
    if prev_g == 1:
        h[n // 2] = 1.0                                                    # start from "pass everything"
    for f, g in zip(freq[::-1][1:], gain[::-1][1:]):
        if g != prev_g:
            width = (prev_f - f) / 2.0
            m = int(round(factor / width))
            m += 1 - m % 2
            if m > n:
                raise ValueError(f"filter_length {n} is too short for a transition of {width * sfreq:.3g} Hz "
                                 f"(needs {m} taps)")
            lp = firwin_lowpass(m, (prev_f + f) / 2.0, fir_window)
            off = (n - m) // 2
            if g == 0:
                h[off:n - off] -= lp
            else:
                h[off:n - off] += lp
        prev_f, prev_g = f, g
    return 0.5 * (h + h[::-1])                                            # exactly symmetric
