"""GPU: the order-independent accumulator behind the deterministic BatchNorm heads (csrc/exact.h) in isolation.
isd_exact_sum adds n fp32 values with 64-bit integer atomics on their fixed-point image and rounds once: the result
must equal Python's exact math.fsum to the last bits for every magnitude mix, and be the same bits on every call."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _exact_sum(x):
    import isd_amd._lib as L
    lib = L.lib()
    xd = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda()
    out = torch.zeros(1, dtype=torch.float64, device="cuda")
    ws = torch.zeros(int(lib.isd_exact_sum_workspace_bytes()) // 8, dtype=torch.int64, device="cuda")
    L.check(lib.isd_exact_sum(xd.data_ptr(), xd.numel(), out.data_ptr(), ws.data_ptr(), 0))
    torch.cuda.synchronize()
    return float(out[0])


@pytest.mark.parametrize("case", ["normal", "wide", "cancel", "subnormal", "negative", "large_n", "single", "empty"])
def test_exact_sum_equals_fsum(case):
    rng = np.random.default_rng(3)
    if case == "normal":
        x = rng.standard_normal(100_000).astype(np.float32)
    elif case == "wide":                                             # magnitudes from 1e-30 to 1e30: every digit pair in use
        x = (rng.standard_normal(200_000) * 10.0 ** rng.uniform(-30, 30, 200_000)).astype(np.float32)
    elif case == "cancel":                                           # huge terms that cancel exactly, a tiny remainder
        big = (rng.standard_normal(50_000) * 1e20).astype(np.float32)
        x = np.concatenate([big, -big, np.full(7, 1e-20, np.float32)])
        rng.shuffle(x)
    elif case == "subnormal":
        x = (rng.integers(-2 ** 22, 2 ** 22, 10_000).astype(np.float64) * 2.0 ** -149).astype(np.float32)
        assert np.any((x != 0) & (np.abs(x) < np.finfo(np.float32).tiny))
    elif case == "negative":
        x = -np.abs(rng.standard_normal(65_537)).astype(np.float32) * 3e5
    elif case == "large_n":                                          # 2^24 equal values: carries ripple through a digit
        x = np.full(1 << 24, 0.1, dtype=np.float32)
    elif case == "single":
        x = np.array([-1.5e-7], dtype=np.float32)
    else:
        x = np.zeros(0, dtype=np.float32)
    want = math.fsum(float(v) for v in x) if len(x) <= 300_000 else float(np.float64(x[0]) * len(x))
    got = _exact_sum(x)
    assert got == _exact_sum(x[::-1].copy())                          # any order, any grid: the same bits
    if want == 0.0:
        assert got == 0.0
    else:
        assert abs(got - want) <= 2.0 ** -49 * abs(want), (case, got, want)


def test_exact_sum_flags_non_finite_values():
    assert math.isnan(_exact_sum(np.array([1.0, np.inf, 2.0], dtype=np.float32)))
    assert math.isnan(_exact_sum(np.array([1.0, np.nan], dtype=np.float32)))
