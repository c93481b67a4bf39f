"""Timing of the fused extractor per kernel family (ISD_FUSED_SERIAL) and bands per wave (ISD_SERIAL_BPW), headline shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import isd_amd

B, C, T = 4096, 64, 512
x = torch.randn(B, C, T, device="cuda")
def t(fx, out, n=20):
    for _ in range(5): fx(x, fused=True, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fx(x, fused=True, out=out)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
nbs = [int(a) for a in sys.argv[1:]] or [9]
for nb in nbs:
    bands = isd_amd.BANDS_9[:nb]
    fx = isd_amd.FeatureExtractor(T, 256.0, bands)
    out = torch.empty(B, nb, C, fx.n_frames, device="cuda")
    os.environ["ISD_FUSED_SERIAL"] = "0"
    r = [f"lane-scan {t(fx, out):.4f}"]
    os.environ["ISD_FUSED_SERIAL"] = "1"
    for bpw in (1, 2, 3):
        os.environ["ISD_SERIAL_BPW"] = str(bpw)
        r.append(f"serial bpw{bpw} {t(fx, out):.4f}")
    os.environ["ISD_FUSED_SERIAL"] = "0"
    r.append(f"lane-scan {t(fx, out):.4f}")
    print(f"bands {nb}: " + "  ".join(r), flush=True)
