"""rocprofv3 target: the stress configuration's feature kernels at B = 128 (materialising filterbank, block-sum band
power, fused extractor) and, with `cfg2`, the headline configuration's fused extractor and filterbank at B = 4096."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import isd_amd

which = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
if which == "cfg5":
    B, C, T = 128, 128, 4096
    fx = isd_amd.FeatureExtractor(T, 1024.0, isd_amd.BANDS_40, nperseg=1024, noverlap=960)
else:
    B, C, T = 4096, 64, 512
    fx = isd_amd.FeatureExtractor(T, 256.0, isd_amd.BANDS_9)
x = torch.randn(B, C, T, device="cuda")
y = torch.empty(B, fx.n_bands, C, T, device="cuda")
out = torch.empty(B, fx.n_bands, C, fx.n_frames, device="cuda")
for _ in range(3):
    fx.fb.forward(x, out=y)
    fx.stft.bandpower(y, fx.bins, out=out)
    fx(x, fused=True, out=out)
torch.cuda.synchronize()
