"""GPU parity: zero-phase FIR band-pass (csrc/fir.hip, row A12) through the C ABI vs oracle/fir.py and the scipy
golden.  fp64 kernel: 1e-12 relative to the row's scale; fp32 kernel: 1e-5 (north star: within 1e-4 rel fp32)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import fir as ofir

pytestmark = pytest.mark.gpu

TOL64, TOL32 = 1e-12, 1e-5


@pytest.fixture(scope="module")
def isd():
    import isd_amd
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return isd_amd


def _err(got, ref):
    return float(np.abs(np.asarray(got, dtype=np.float64) - ref).max() / max(np.abs(ref).max(), 1e-30))


@pytest.mark.parametrize("tag", ["n", "s"])
def test_matches_scipy_golden(isd, tag):
    g = load_golden("g12_fir.npz")
    sf, lo, hi = g[f"{tag}.args"]
    y = isd.filter_data(g[f"{tag}.x"], sf, lo, hi)                       # ndarray in -> fp64 on the GPU -> ndarray
    assert y.dtype == np.float64 and y.shape == g[f"{tag}.x"].shape
    assert _err(y, g[f"{tag}.y"]) < TOL64
    y32 = isd.filter_data(torch.as_tensor(g[f"{tag}.x"], dtype=torch.float32).cuda(), sf, lo, hi)
    assert y32.dtype == torch.float32 and _err(y32.cpu().numpy(), g[f"{tag}.y"]) < TOL32


@pytest.mark.parametrize("rows,T", [(1, 795), (2, 800), (3, 512), (5, 513), (4, 100), (3, 1), (2, 7), (7, 1500),
                                    (1, 4096), (130, 250)])
def test_shapes_vs_oracle(isd, rows, T):
    # odd row counts (the fp32 kernel pairs rows), ragged last tiles, rows shorter than the filter (zero fill
    # beyond the limited reflection), a single sample
    flt = isd.FirFilter(250, 4, 40)
    x = np.random.default_rng(rows * 10007 + T).standard_normal((rows, T))
    ref = ofir.zero_phase(x, ofir.design(250, 4, 40))
    xd = torch.as_tensor(x).cuda()
    assert _err(flt(xd).cpu().numpy(), ref) < TOL64
    assert _err(flt(xd.float()).cpu().numpy(), ref) < TOL32


@pytest.mark.parametrize("sf,lo,hi", [(250.0, None, 40.0), (250.0, 1.0, None), (1024.0, 4.0, 84.0), (256.0, 8.0, 12.0)])
def test_other_designs_vs_oracle(isd, sf, lo, hi):
    x = np.random.default_rng(5).standard_normal((3, 2, 1024))
    ref = ofir.filter_data(x, sf, lo, hi)
    assert _err(isd.filter_data(x, sf, lo, hi), ref) < TOL64
    y32 = isd.filter_data(torch.as_tensor(x, dtype=torch.float32).cuda(), sf, lo, hi)
    assert y32.shape == (3, 2, 1024) and _err(y32.cpu().numpy(), ref) < TOL32


@pytest.mark.parametrize("n_taps", [1, 3, 15, 17, 33])
def test_explicit_taps(isd, n_taps):
    rng = np.random.default_rng(n_taps)
    h = rng.standard_normal(n_taps)
    h = 0.5 * (h + h[::-1])
    x = rng.standard_normal((4, 300))
    flt = isd.FirFilter(250, None, None, taps=h)
    assert _err(flt(torch.as_tensor(x).cuda()).cpu().numpy(), ofir.zero_phase(x, h)) < TOL64


def test_plan_errors(isd):
    from isd_amd._lib import IsdError
    with pytest.raises(IsdError):
        isd.FirFilter(250, None, None, taps=np.ones(4))                  # even length
    with pytest.raises(IsdError):
        isd.FirFilter(250, None, None, taps=np.array([1.0, 2.0, 3.0]))   # not symmetric
    flt = isd.FirFilter(250, 4, 40)
    x = torch.zeros(2, 64, device="cuda")
    with pytest.raises(ValueError):
        flt(x, out=x)
    with pytest.raises(TypeError):
        flt(torch.zeros(2, 64))
    assert flt(torch.zeros(0, 64, device="cuda")).shape == (0, 64)


def test_full_size_properties(isd):
    # BASELINE cfg2 batch [4096, 64, 512] fp32: linearity, a pass-band tone survives, DC and a stop-band tone vanish
    flt = isd.FirFilter(256, 4, 40)
    g = torch.Generator(device="cuda").manual_seed(0)
    a = torch.randn(4096, 64, 512, device="cuda", generator=g)
    b = torch.randn(4096, 64, 512, device="cuda", generator=g)
    ya, yb = flt(a), flt(b)
    yab = flt(2.0 * a - 3.0 * b)
    assert float((yab - (2.0 * ya - 3.0 * yb)).abs().max()) < 2e-5 * float(yab.abs().max())
    rows = torch.randint(0, 4096 * 64, (16,), generator=torch.Generator().manual_seed(1))
    ref = ofir.zero_phase(a.view(-1, 512)[rows.cuda()].double().cpu().numpy(), flt.taps)
    assert _err(ya.view(-1, 512)[rows.cuda()].cpu().numpy(), ref) < TOL32
    t = torch.arange(2048, device="cuda", dtype=torch.float64) / 256.0
    tones = torch.stack([torch.sin(2 * np.pi * 20.0 * t), torch.ones_like(t), torch.sin(2 * np.pi * 90.0 * t)])
    y = flt(tones)
    mid = slice(600, 1400)
    assert float((y[0, mid] - tones[0, mid]).abs().max()) < 5e-3
    assert float(y[1, mid].abs().max()) < 1e-9 and float(y[2, mid].abs().max()) < 2e-3
