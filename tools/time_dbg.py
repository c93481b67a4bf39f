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
t(); t()
for bpw, groups in (("3", "1"), ("3", "2"), ("3", "4"), ("2", "1"), ("2", "2"), ("1", "1"), ("3", "4")):
    os.environ["ISD_SERIAL_BPW"] = bpw; os.environ["ISD_SERIAL_GROUPS"] = groups
    r = []
    for dbg in (0, 1, 2, 3):
        os.environ["ISD_SER_DBG"] = str(dbg)
        r.append(f"dbg{dbg} {t():.4f}")
    print(f"bpw {bpw} groups {groups}: " + "  ".join(r) + "   (dbg: 1 no barrier, 2 no stores)", flush=True)
os.environ["ISD_SER_DBG"] = "0"; os.environ.pop("ISD_SERIAL_GROUPS"); os.environ.pop("ISD_SERIAL_BPW")
os.environ["ISD_FUSED_SERIAL"] = "0"; a = t(); os.environ["ISD_FUSED_SERIAL"] = "1"; b = t()
print(f"defaults: lane-scan {a:.4f}  serial {b:.4f}")
