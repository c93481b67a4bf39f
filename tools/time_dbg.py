import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import isd_amd
B, C, T = 4096, 64, 512
x = torch.randn(B, C, T, device="cuda")
fx = isd_amd.FeatureExtractor(T, 256.0, isd_amd.BANDS_9)
out = torch.empty(B, 9, C, fx.n_frames, device="cuda")
def t(n=20):
    for _ in range(5): fx(x, fused=True, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fx(x, fused=True, out=out)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for bpw in ("3", "2"):
    os.environ["ISD_SERIAL_BPW"] = bpw
    for dbg in (0, 1, 2, 4, 3, 7, 0):
        os.environ["ISD_SER_DBG"] = str(dbg)
        print(f"bpw {bpw} dbg {dbg} (1 no barrier, 2 no stores, 4 no dma wait): {t():.4f} ms", flush=True)
