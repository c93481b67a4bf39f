"""Oracle (test infrastructure): zero-phase FIR band-pass on the CPU, float64.

Reference anchor: ``filter_data(X_tr, sfreq, l_freq=4, h_freq=40, verbose=False)`` with ``sfreq = 250``
(notebooks/svm_baseline.ipynb:237-239, again at :968-969), ``from mne.filter import filter_data``.

PARITY UNPINNED against MNE: MNE is a third-party dependency of the notebook (the notebook output shows
mne 1.11.0; it is not in pyproject.toml / requirements.txt), its source is not under /root/reference and it is
not installed here.  What is restated below is MNE's published FIR algorithm at its documented defaults:

* design (``mne.filter.create_filter`` with method='fir', fir_design='firwin', fir_window='hamming',
  phase='zero', transition bandwidths and length 'auto'): transition widths
  ``min(max(0.25 l_freq, 2), l_freq)`` and ``min(max(0.25 h_freq, 2), sfreq/2 - h_freq)``; length
  ``round(3.3 * sfreq / min(widths))`` made odd; the filter is a sum/difference of ``scipy.signal.firwin``
  low-passes, one per transition band, each with its own odd length ``round(3.3 / (width / sfreq))`` and
  cut-off at the middle of its transition band, centred in the full length;
* application (``_overlap_add_filter`` with phase='zero', pad='reflect_limited'): the row is extended on both
  sides by ``min(n_taps, T) - 1`` samples of odd reflection (``2 x[0] - x[d]``, ``2 x[-1] - x[-1-d]``),
  convolved with the taps, shifted by ``(n_taps - 1) // 2`` and cropped to the original length.

What IS pinned: the taps by scipy.signal.firwin (called directly below), the convolution by np.convolve.
"""
import numpy as np
from scipy.signal import firwin

LENGTH_FACTORS = {"hann": 3.1, "hamming": 3.3, "blackman": 5.0}


def design(sfreq, l_freq, h_freq, fir_window="hamming"):
    """Taps (float64, odd length) of the 'auto' low-/high-/band-pass, built from scipy.signal.firwin."""
    nyq = sfreq / 2.0
    fac = LENGTH_FACTORS[fir_window]
    widths, edges = [], []                    # (stop edge, pass edge) per transition band, as Hz
    if l_freq is not None:
        lt = min(max(0.25 * l_freq, 2.0), l_freq)
        widths.append(lt)
        edges.append(("hp", l_freq - lt, l_freq))
    if h_freq is not None:
        ht = min(max(0.25 * h_freq, 2.0), nyq - h_freq)
        widths.append(ht)
        edges.append(("lp", h_freq, h_freq + ht))
    n = max(int(round(fac * sfreq / min(widths))), 1)
    n += (n - 1) % 2
    h = np.zeros(n)
    if h_freq is None:
        h[n // 2] = 1.0                       # high-pass: everything passes, then the low end is removed
    for kind, f0, f1 in edges[::-1]:
        m = int(round(fac / ((f1 - f0) / 2.0 / nyq)))
        m += 1 - m % 2
        lp = firwin(m, (f0 + f1) / 2.0, window=fir_window, pass_zero=True, fs=sfreq)
        off = (n - m) // 2
        if kind == "lp":
            h[off:n - off] += lp
        else:
            h[off:n - off] -= lp
    return h


def smart_pad(x, n_pad):
    """1-D odd reflection limited to len(x) - 1 samples, zeros beyond ('reflect_limited')."""
    if n_pad == 0:
        return x
    z = np.zeros(max(n_pad - len(x) + 1, 0), dtype=x.dtype)
    return np.concatenate([z, 2 * x[0] - x[n_pad:0:-1], x, 2 * x[-1] - x[-2:-n_pad - 2:-1], z])


def zero_phase(x, h):
    """Apply symmetric taps ``h`` along the last axis of ``x`` with the delay compensated (float64)."""
    x = np.asarray(x, dtype=np.float64)
    T = x.shape[-1]
    n_edge = max(min(len(h), T) - 1, 0)
    shift = (len(h) - 1) // 2 + n_edge
    out = np.empty_like(x)
    flat, oflat = x.reshape(-1, T), out.reshape(-1, T)
    for i in range(flat.shape[0]):
        full = np.convolve(smart_pad(flat[i], n_edge), h)        # 'full'
        oflat[i] = full[shift:shift + T]
    return out


def filter_data(data, sfreq, l_freq, h_freq, fir_window="hamming"):
    return zero_phase(data, design(sfreq, l_freq, h_freq, fir_window))
