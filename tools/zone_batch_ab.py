"""A/B check of the zone-batched launches (isd_zone_batch_*) against the per-zone calls for the three BatchNorm heads:
outputs, parameter gradients and running statistics of one training step (ISD_ZONE_BATCH_OFF=1 selects the per-zone path).
The test suite holds the same comparison: tests/test_bnheads_gpu.py::test_zone_batched_launches_equal_per_zone_calls."""
import os, sys, time
sys.path.insert(0, '/root/repo')
import torch
import isd_amd.nn as inn
from isd_amd import constants as K
torch.manual_seed(0)
def run(head, off):
    if off: os.environ["ISD_ZONE_BATCH_OFF"] = "1"
    else: os.environ.pop("ISD_ZONE_BATCH_OFF", None)
    torch.manual_seed(1)
    h = inn.Head(head, K.ELECTRODES, K.ZONES, 32).cuda().train()
    for e in h.encoders.values():
        if hasattr(e, "p"): e.p = 0.0
    x = torch.randn(40, 64, 250, device="cuda")
    w = torch.randn(40, 8, 32, device="cuda")
    f = h(x)
    (f * w).sum().backward()
    g = torch.cat([p.grad.reshape(-1) for p in h.parameters()])
    bufs = torch.cat([b.reshape(-1).float() for b in h.buffers()])
    return f.detach(), g, bufs
for head in ("EEGNet_Encoder", "CVBlock", "HeadConv_Paper_Version"):
    f0, g0, b0 = run(head, True)
    f1, g1, b1 = run(head, False)
    print(head, "out", float((f0 - f1).abs().max()), "grad", float((g0 - g1).abs().max() / g0.abs().max()), "bufs", float((b0 - b1).abs().max()), torch.equal(f0, f1), torch.equal(g0, g1))
    d = (f0 - f1).abs()
    print("  per-zone max diff:", [round(float(d[:, z].max()), 4) for z in range(8)], " rows differing:", int((d.amax(dim=(1, 2)) > 1e-6).sum()), "cols:", int((d.amax(dim=(0, 1)) > 1e-6).sum()))
