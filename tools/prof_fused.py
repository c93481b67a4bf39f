"""Launch the fused extractor a few times at the headline shape (target of rocprofv3 runs; ISD_FUSED_SERIAL picks the kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import isd_amd

B, C, T = 4096, 64, 512
x = torch.randn(B, C, T, device="cuda")
fx = isd_amd.FeatureExtractor(T, 256.0, isd_amd.BANDS_9)
out = torch.empty(B, 9, C, fx.n_frames, device="cuda")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for _ in range(n):
    fx(x, fused=True, out=out)
torch.cuda.synchronize()
