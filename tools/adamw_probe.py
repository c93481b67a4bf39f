import torch, time
n = 111365
for nch in (1, 8, 16, 24, 32, 48, 64, 128):
    flat = torch.randn(n, device="cuda"); g = torch.randn(n, device="cuda")
    step = -(-n // nch)
    ps = []
    for lo in range(0, n, step):
        p = torch.nn.Parameter(flat[lo:lo+step]); p.grad = g[lo:lo+step]; ps.append(p)
    opt = torch.optim.AdamW(ps, lr=5e-4, weight_decay=1e-2, fused=True)
    for _ in range(5): opt.step()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): opt.step()
    e1.record(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): opt.step()
    torch.cuda.synchronize()
    print(f"chunks {len(ps):4d}: gpu {e0.elapsed_time(e1)/50*1e3:7.1f} us/step, wall {(time.perf_counter()-t0)/50*1e6:7.1f} us/step")
