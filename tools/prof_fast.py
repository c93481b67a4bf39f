"""One FAST train_head step at B=1024, T=512 (target of rocprofv3 --pmc runs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import isd_amd
from isd_amd.classifier import _FastModel
from isd_amd.nn import fast_config

torch.manual_seed(0)
m = _FastModel(fast_config(seq_len=512, act_dtype=os.environ.get("ISD_PROF_ACT", "f32"))).cuda()
tr = isd_amd.Trainer(m)
B = int(os.environ.get("ISD_PROF_B", "1024"))
x = torch.randn(B, 64, 512, device="cuda")
y = torch.randint(0, 5, (B,), device="cuda")
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    tr.step(x, y)
torch.cuda.synchronize()
