"""rocprofv3 target: the stress configuration's feature kernels at B = 128 (materialising filterbank, block-sum band
power, fused extractor) and, with `cfg2`, the headline configuration's fused extractor and filterbank at B = 4096."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import isd_amd

which = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
if which == "tail":                     # FAST's transformer tail, one launch per direction, B = 4096 (512 waves)
    import isd_amd.nn as inn
    torch.manual_seed(0)
    net = inn.FAST(inn.fast_config()).cuda().train()
    feat = torch.randn(4096, 5, 8, 32, device="cuda")
    lab = torch.randint(0, 5, (4096,), device="cuda")
    for _ in range(3):
        net.zero_grad(set_to_none=True)
        inn.token_mean_cross_entropy(net.forward_transformer(feat).unsqueeze(1), lab).backward()
    torch.cuda.synchronize()
    sys.exit(0)
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
