"""Zero-phase FIR band-pass in front of the classical baseline.

``filter_data(data, sfreq, l_freq, h_freq)`` has the positional surface of the one call the reference makes,
``mne.filter.filter_data(X_tr, 250, l_freq=4, h_freq=40)`` (notebooks/svm_baseline.ipynb:238-239, :968-969),
with MNE's documented defaults (windowed-sinc 'firwin' design, hamming window, zero phase by delay
compensation, 'reflect_limited' edge padding).  The taps are designed on the host
(``filter_design.fir_design``); the convolution runs in libisd_hip.so (csrc/fir.hip).  MNE is not vendored in
the reference and is not installed here: parity is pinned against scipy / NumPy (oracle/fir.py), not against MNE.
There is no CPU fallback.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .filter_design import fir_design


class FirFilter:
    """Plan for one zero-phase FIR filter: host taps (float64) + the device tables of csrc/fir.hip."""

    def __init__(self, sfreq, l_freq, h_freq, filter_length="auto", l_trans_bandwidth="auto",
                 h_trans_bandwidth="auto", fir_window="hamming", taps=None):
        self.sfreq, self.l_freq, self.h_freq = float(sfreq), l_freq, h_freq
        self.taps = (np.ascontiguousarray(taps, dtype=np.float64) if taps is not None else
                     fir_design(sfreq, l_freq, h_freq, filter_length, l_trans_bandwidth, h_trans_bandwidth, fir_window))
        self._h = C.c_void_p()
        _lib.check(_lib.lib().isd_fir_plan_create(C.byref(self._h), len(self.taps), _lib.double_array(self.taps)))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _lib.lib().isd_fir_plan_destroy(h)
            except Exception:
                pass

    def __call__(self, x, out=None):
        """x CUDA tensor [..., T], float32 (fp32 arithmetic) or float64 (fp64 arithmetic) -> same shape / dtype."""
        if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype in (torch.float32, torch.float64)):
            raise TypeError("x must be a float32 or float64 CUDA tensor")
        x = x.contiguous()
        T = x.shape[-1]
        rows = x.numel() // T if T else 0
        if out is None:
            out = torch.empty_like(x)
        elif out.shape != x.shape or out.dtype != x.dtype or not out.is_cuda or not out.is_contiguous():
            raise ValueError("out must be a contiguous CUDA tensor of x's shape and dtype")
        if x.numel() and out.data_ptr() == x.data_ptr():
            raise ValueError("in-place filtering is not supported")
        if rows == 0:
            return out
        fn = _lib.lib().isd_fir_zero_phase_f32 if x.dtype == torch.float32 else _lib.lib().isd_fir_zero_phase_f64
        with torch.cuda.device(x.device):
            _lib.check(fn(self._h, x.data_ptr(), out.data_ptr(), rows, T,
                          C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return out


def filter_data(data, sfreq, l_freq, h_freq, picks=None, filter_length="auto", l_trans_bandwidth="auto",
                h_trans_bandwidth="auto", n_jobs=None, method="fir", iir_params=None, copy=True, phase="zero",
                fir_window="hamming", fir_design="firwin", pad="reflect_limited", *, verbose=None, device=None):
    """Band-pass (or low-/high-pass) the last axis of ``data`` -- argument order and defaults of
    ``mne.filter.filter_data``.  ndarray in -> float64 ndarray out (computed in fp64 on the GPU, as MNE computes
    in float64); CUDA tensor in -> CUDA tensor of the same dtype out.  Options of MNE this path does not provide
    (IIR, minimum phase, firwin2, other paddings, picks) raise NotImplementedError."""
    for name, val, ok in (("method", method, "fir"), ("phase", phase, "zero"), ("fir_design", fir_design, "firwin"),
                          ("pad", pad, "reflect_limited")):
        if val != ok:
            raise NotImplementedError(f"{name}={val!r}: only {ok!r} is provided")
    if picks is not None or iir_params is not None:
        raise NotImplementedError("picks / iir_params are not provided")
    flt = FirFilter(sfreq, l_freq, h_freq, filter_length, l_trans_bandwidth, h_trans_bandwidth, fir_window)
    if isinstance(data, torch.Tensor):
        if not data.is_cuda:
            raise TypeError("tensor input must live on the GPU (there is no CPU fallback)")
        return flt(data)
    if not torch.cuda.is_available():
        raise RuntimeError("isd_amd.filter_data needs an MI355X GPU: there is no CPU fallback")
    arr = np.asarray(data)
    if arr.dtype.kind != "f":
        raise TypeError("data must be floating point")
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    x = torch.as_tensor(np.ascontiguousarray(arr, dtype=np.float64)).to(dev)
    return flt(x).cpu().numpy()
