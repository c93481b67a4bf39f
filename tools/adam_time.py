"""AdamW over FAST's 94 parameter tensors (0.19 M elements) inside a HIP graph, 20 dependent steps back to back:
isd_amd.FusedAdamW (one launch; device-side rate and step count) against torch's fused capturable AdamW."""
import sys, time, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import isd_amd, isd_amd.nn as inn
torch.manual_seed(0)
net = inn.FAST(inn.fast_config(seq_len=800)).cuda()
ps = [p for p in net.parameters()]
print('tensors', len(ps), 'elements', sum(p.numel() for p in ps), 'max', max(p.numel() for p in ps))
for p in ps: p.grad = torch.randn_like(p)
def bench(opt, name):
    for _ in range(5): opt.step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        opt.step()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        for _ in range(20): opt.step()
    g.replay(); torch.cuda.synchronize()
    e0.record(); 
    for _ in range(10): g.replay()
    e1.record(); torch.cuda.synchronize()
    print(name, 'us per step (graph of 20 back-to-back steps):', e0.elapsed_time(e1) / 200 * 1e3)
bench(isd_amd.FusedAdamW(ps, lr=torch.tensor(1e-3, device='cuda'), capturable=True), 'FusedAdamW capturable')
bench(torch.optim.AdamW(ps, lr=torch.tensor(1e-3, device='cuda'), capturable=True, fused=True), 'torch fused capturable')
