"""Launch the feature kernels a few times (target of rocprofv3 --pmc / --kernel-trace runs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import isd_amd

B, C, T = 4096, 64, 512
x = torch.randn(B, C, T, device="cuda")
fx = isd_amd.FeatureExtractor(T, 256.0, isd_amd.BANDS_9)
y = torch.empty(B, 9, C, T, device="cuda")
out = torch.empty(B, 9, C, fx.n_frames, device="cuda")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for _ in range(n):
    fx.fb.forward(x, out=y)
    fx(x, fused=True, out=out)
torch.cuda.synchronize()
