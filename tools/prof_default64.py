"""rocprofv3 target: FAST 'default'-mode optimisation steps at the reference's batch (64 trials x 64 ch x 800 samples),
eager launches (the graph replay runs the same kernels)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import isd_amd.nn as inn

torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
net = inn.FAST(inn.fast_config(head=os.environ.get("ISD_PROF_HEAD", "Conv4Layers"))).cuda().train()
opt = torch.optim.AdamW(net.parameters(), lr=5e-4, fused=True)
x = torch.randn(B, 64, 800, device="cuda")
y = torch.randint(0, 5, (B,), device="cuda")
for _ in range(10):
    opt.zero_grad(set_to_none=True)
    inn.token_mean_cross_entropy(net(x, forward_mode="default"), y).backward()
    opt.step()
torch.cuda.synchronize()
