"""CPU: zero-phase FIR band-pass (row A12) -- the oracle against the scipy golden, the product's tap design
against scipy.signal.firwin, and the facts the reference's call site fixes (250 Hz, 4-40 Hz)."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import fir as ofir


@pytest.mark.parametrize("tag", ["n", "s"])
def test_oracle_matches_scipy_golden(tag):
    g = load_golden("g12_fir.npz")
    sf, lo, hi = g[f"{tag}.args"]
    h = ofir.design(sf, lo, hi)
    np.testing.assert_allclose(h, g[f"{tag}.taps"], rtol=0, atol=1e-16)
    np.testing.assert_allclose(ofir.zero_phase(g[f"{tag}.x"], h), g[f"{tag}.y"], rtol=0, atol=1e-13)


def test_notebook_filter_facts():
    # filter_data(X, 250, l_freq=4, h_freq=40): transitions 2 Hz / 10 Hz, 413 taps, -6 dB at 3 Hz and 45 Hz
    from isd_amd.filter_design import fir_design, fir_transition_bands
    assert fir_transition_bands(250.0, 4.0, 40.0) == (2.0, 10.0)
    h = fir_design(250, 4, 40)
    assert len(h) == 413 and np.array_equal(h, h[::-1])
    f = np.fft.rfftfreq(1 << 16, 1 / 250.0)
    H = np.abs(np.fft.rfft(h, 1 << 16))
    at = lambda q: H[np.argmin(np.abs(f - q))]
    assert abs(at(3.0) - 0.5) < 5e-3 and abs(at(45.0) - 0.5) < 5e-3
    assert at(0.0) < 1e-12 and at(1.0) < 2e-3 and at(60.0) < 2e-3
    assert np.all(np.abs(H[(f >= 5) & (f <= 38)] - 1.0) < 3e-3)


@pytest.mark.parametrize("sf,lo,hi,win", [(250.0, 4.0, 40.0, "hamming"), (256.0, 8.0, 30.0, "hamming"),
                                          (1024.0, 4.0, 84.0, "hamming"), (250.0, None, 40.0, "hamming"),
                                          (250.0, 1.0, None, "hamming"), (250.0, 0.5, 100.0, "hann"),
                                          (500.0, 13.0, 30.0, "blackman")])
def test_product_design_matches_scipy_firwin_construction(sf, lo, hi, win):
    from isd_amd.filter_design import fir_design
    h = fir_design(sf, lo, hi, fir_window=win)
    ref = ofir.design(sf, lo, hi, win)
    assert len(h) == len(ref) and len(h) % 2 == 1
    np.testing.assert_allclose(h, ref, rtol=0, atol=2e-16)


def test_design_argument_errors():
    from isd_amd.filter_design import fir_design
    with pytest.raises(ValueError):
        fir_design(250, None, None)
    with pytest.raises(NotImplementedError):
        fir_design(250, 40, 4)                       # band-stop
    with pytest.raises(ValueError):
        fir_design(250, 4, 130)                      # above Nyquist
    with pytest.raises(ValueError):
        fir_design(250, 4, 40, filter_length=101)    # too short for the 2 Hz transition
    with pytest.raises(ValueError):
        fir_design(250, 4, 40, fir_window="kaiser")
    assert len(fir_design(250, 4, 40, filter_length=500)) == 501


def test_oracle_short_rows_zero_beyond_the_reflection():
    # T < n_taps: the reflection covers T - 1 samples, zeros beyond (mne _smart_pad 'reflect_limited')
    h = ofir.design(250, 4, 40)
    x = np.random.default_rng(3).standard_normal(100)
    y = ofir.zero_phase(x, h)
    T, half = 100, 206

    def xe(m):
        if 0 <= m < T:
            return x[m]
        d = -m if m < 0 else m - (T - 1)
        if d > T - 1:
            return 0.0
        return 2 * x[0] - x[d] if m < 0 else 2 * x[T - 1] - x[T - 1 - d]
    for n in (0, 1, 50, 99):
        assert abs(y[n] - sum(h[k] * xe(n - half + k) for k in range(413))) < 1e-13


def test_filter_data_rejects_what_it_does_not_provide():
    import isd_amd
    x = np.zeros((2, 3, 100))
    for kw in ({"method": "iir"}, {"phase": "minimum"}, {"fir_design": "firwin2"}, {"pad": "edge"}, {"picks": [0]}):
        with pytest.raises(NotImplementedError):
            isd_amd.filter_data(x, 250, 4, 40, **kw)
