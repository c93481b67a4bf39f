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
