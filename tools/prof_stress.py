"""Launch the stress-configuration feature kernel and the FIR band-pass a few times (target of rocprofv3 --pmc runs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import isd_amd

B, C, T = 128, 128, 4096
x = torch.randn(B, C, T, device="cuda")
fx = isd_amd.FeatureExtractor(T, 1024.0, isd_amd.BANDS_40, nperseg=1024, noverlap=960)
out = torch.empty(B, 40, C, fx.n_frames, device="cuda")
flt = isd_amd.FirFilter(256, 4, 40)
xf = torch.randn(4096, 64, 512, device="cuda")
yf = torch.empty_like(xf)
for _ in range(3):
    fx(x, fused=True, out=out)
    flt(xf, out=yf)
torch.cuda.synchronize()
