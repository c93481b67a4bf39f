"""CPU-side checks of the product: host logic, C ABI surface, filter design (no GPU compute)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, load_golden
from oracle import dsp as odsp


def test_import_does_not_need_gpu_or_library():
    import isd_amd
    assert callable(isd_amd.extract_features)
    assert isd_amd.CLASSES == ["hello", "help-me", "stop", "thank-you", "yes"]
    assert sum(len(z) for z in isd_amd.zone_index_lists()) == 64
    assert sorted(i for z in isd_amd.zone_index_lists() for i in z) == list(range(64))


def test_library_builds_loads_and_exports_every_declared_symbol():
    import __graft_entry__
    __graft_entry__.build()
    from isd_amd import _lib
    h = _lib.lib()
    header = open(os.path.join(ROOT, "include", "isd_hip.h")).read()
    declared = set(re.findall(r"\b(isd_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(h, name), name
    assert h.isd_abi_version() == 1
    assert h.isd_device_count() >= 0


def test_c_abi_argument_errors_without_gpu():
    from isd_amd import _lib
    h = _lib.lib()
    p = ctypes.c_void_p()
    a12 = _lib.double_array([0.0, 1.5])                       # |a2| >= 1: unstable section
    assert h.isd_fb_plan_create(ctypes.byref(p), 1, 1, a12, _lib.double_array([1.0]), _lib.FB_AUTO) == -1
    assert b"not stable" in h.isd_last_error()
    assert h.isd_stft_plan_create(ctypes.byref(p), 512, 60, 30) == -1
    assert b"power of two" in h.isd_last_error()
    with pytest.raises(_lib.IsdError):
        _lib.check(h.isd_fb_forward(None, None, None, 1, 1, 1, None))


def test_product_path_fails_loudly_without_library(monkeypatch, tmp_path):
    from isd_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "missing.so"))
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.lib()


def test_no_product_module_imports_the_oracle():
    pkg = os.path.join(ROOT, "imagined-speech-decoding_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f


@pytest.mark.parametrize("tag,bands", [("b5", odsp.BANDS_5), ("b9", odsp.BANDS_9), ("b40", odsp.BANDS_40)])
def test_filter_design_matches_scipy_golden_tables(tag, bands):
    from isd_amd import butter_bandpass_resonators, butter_bandpass_sos
    g = load_golden("g2_sos.npz")
    fs = float(g[f"{tag}_fs"])
    for b, (_, lo, hi) in enumerate(bands):
        np.testing.assert_allclose(butter_bandpass_sos(4, lo, hi, fs), g[f"{tag}_sos"][b], rtol=1e-9, atol=1e-300)
        a12, gain = butter_bandpass_resonators(4, lo, hi, fs)
        # same poles as the scipy table
        want = np.sort_complex(np.concatenate([np.roots(s[3:]) for s in g[f"{tag}_sos"][b]]))
        got = np.sort_complex(np.concatenate([np.roots([1.0, a1, a2]) for a1, a2 in a12]))
        np.testing.assert_allclose(got, want, rtol=1e-10)
        np.testing.assert_allclose(gain, np.prod(g[f"{tag}_sos"][b][:, 0]), rtol=1e-9)


def test_band_bins_inclusive_edges():
    from isd_amd import band_bins
    assert band_bins(256.0, 64, odsp.BANDS_9)[0] == (1, 2)           # 4 and 8 Hz bins, both edges included
    assert band_bins(250.0, 64, odsp.BANDS_5) == odsp.band_bins(250.0, 64, odsp.BANDS_5)
    assert band_bins(256.0, 64, [("x", 1.0, 3.0)]) == [(1, 0)]       # empty band
