"""How the classifier-tail kernels' time splits into a fixed part (fragment staging, cross-wave combine, slab write)
and a per-item part: one classifier step on spec-S features at batch sys.argv[1] (rocprofv3 --kernel-trace --stats around it)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import isd_amd
from isd_amd.classifier import _FeatureModel
B = int(sys.argv[1]); bf16 = len(sys.argv) > 2 and sys.argv[2] == "bf16"
torch.manual_seed(0)
m = _FeatureModel(576, 32, 5, 4, "bf16" if bf16 else "f32").cuda()
tr = isd_amd.Trainer(m)
x = torch.randn(B, 576, 17, device="cuda")
if bf16:
    x = x.to(torch.bfloat16)
y = torch.randint(0, 5, (B,), device="cuda")
for _ in range(12):
    tr.step(x, y)
torch.cuda.synchronize()
