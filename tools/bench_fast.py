"""Throughput of the reference-native paths on raw EEG (fwd+bwd+AdamW, device-resident trials):
FAST 'train_head' through the autograd-free Trainer, and FAST 'default' -- the mode the reference trains
(src/fast/train/trainer.py:58: zone CNN -> Linear+GELU -> cls/pos embedding -> 4 pre-LN transformer blocks ->
classifier) -- through the autograd modules of isd_amd.nn with isd_amd.FusedAdamW (--torch-adamw: torch's fused AdamW)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import isd_amd
import isd_amd.nn as inn
from isd_amd.classifier import _FastModel
from isd_amd.nn import fast_config


def make_opt(net, capturable):
    """isd_amd.FusedAdamW (what isd_amd.experiment trains with); --torch-adamw: torch's fused multi-tensor AdamW"""
    if "--torch-adamw" in sys.argv:
        if capturable:
            return torch.optim.AdamW(net.parameters(), lr=torch.tensor(5e-4, device="cuda"), capturable=True, fused=True)
        return torch.optim.AdamW(net.parameters(), lr=5e-4, fused=True)
    if capturable:
        return isd_amd.FusedAdamW(net.parameters(), lr=torch.tensor(5e-4, device="cuda"), capturable=True)
    return isd_amd.FusedAdamW(net.parameters(), lr=5e-4)


def timed(step, n):
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, out


def tail_only():
    """forward_transformer forward + backward on resident zone features: one launch per direction vs per operator."""
    for B in (64, 4096):
        torch.manual_seed(0)
        net = inn.FAST(fast_config()).cuda().train()
        feat = torch.randn(B, 5, 8, 32, device="cuda")
        y = torch.randint(0, 5, (B,), device="cuda")
        for fused in (True, False):
            net.fuse_tail = fused

            def step():
                net.zero_grad(set_to_none=True)
                loss = inn.token_mean_cross_entropy(net.forward_transformer(feat).unsqueeze(1), y)
                loss.backward()
                return loss
            dt, _ = timed(step, 50)
            with torch.no_grad():
                di, _ = timed(lambda: net.forward_transformer(feat), 50)
            print(f"FAST tail only  B={B} {'one launch per direction' if fused else 'per-operator launches'}: "
                  f"fwd+bwd {dt*1e3:.3f} ms, inference {di*1e3:.3f} ms")


def heads():
    """FAST 'default' step at the reference's batch with each registry head (fast.py:203), eager and graph replay."""
    from isd_amd.graph import GraphedTrainStep, graph_safe
    B, T = 64, 800
    x = torch.randn(B, 64, T, device="cuda")
    y = torch.randint(0, 5, (B,), device="cuda")
    for head in ("Conv4Layers", "EEGNet_Encoder", "CVBlock", "HeadConv_Paper_Version"):
        for graph in (False, True):
            torch.manual_seed(0)
            net = inn.FAST(fast_config(seq_len=T, head=head)).cuda().train()
            if graph and not graph_safe(net):
                print(f"FAST default head={head:<22} B={B}: not graph-safe")
                continue
            if graph:
                opt = make_opt(net, True)
                gs = GraphedTrainStep(net, opt, x, y, B)
                idx = torch.arange(B, device="cuda")

                def step():
                    gs.step(idx, 5e-4)
            else:
                opt = make_opt(net, False)

                def step():
                    opt.zero_grad(set_to_none=True)
                    inn.token_mean_cross_entropy(net(x, forward_mode="default"), y).backward()
                    opt.step()
            dt, _ = timed(step, 20)
            print(f"FAST default head={head:<22} B={B} {'graph replay' if graph else 'eager       '}: {dt*1e3:.3f} ms/step, "
                  f"{B/dt:.0f} trials/s")


def replay_only(act):
    """--replay f32|bf16 [--batch B]: 200 graph replays of the FAST 'default' step (default: the reference's batch of
    64), for rocprofv3."""
    from isd_amd.graph import GraphedTrainStep
    B, T = (int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 64), 800
    x = torch.randn(B, 64, T, device="cuda")
    y = torch.randint(0, 5, (B,), device="cuda")
    torch.manual_seed(0)
    net = inn.FAST(fast_config(seq_len=T, act_dtype=act)).cuda().train()
    opt = make_opt(net, True)
    gs = GraphedTrainStep(net, opt, x, y, B)
    idx = torch.arange(B, device="cuda")
    dt, _ = timed(lambda: gs.step(idx, 5e-4), 200)
    print(f"FAST default {act} B={B} T={T} graph replay: {dt*1e3:.3f} ms/step")


def main():
    if "--replay" in sys.argv:
        return replay_only(sys.argv[sys.argv.index("--replay") + 1])
    if "--tail" in sys.argv:
        return tail_only()
    if "--heads" in sys.argv:
        return heads()
    for B, T, act in ((64, 800, "f32"), (1024, 512, "f32"), (4096, 512, "f32"), (1024, 800, "f32"), (64, 800, "bf16"),
                      (4096, 512, "bf16"), (1024, 800, "bf16")):
        torch.manual_seed(0)
        m = _FastModel(fast_config(seq_len=T, act_dtype=act)).cuda()
        tr = isd_amd.Trainer(m)
        x = torch.randn(B, 64, T, device="cuda")
        y = torch.randint(0, 5, (B,), device="cuda")
        dt, out = timed(lambda: tr.step(x, y), 20 if B <= 1024 else 5)
        print(f"FAST train_head {act} B={B} T={T}: {dt*1e3:.3f} ms/step, {B/dt:.0f} trials/s, loss {float(out['loss']):.4f}, "
              f"mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
    from isd_amd.graph import GraphedTrainStep
    for B, T, act in ((64, 800, "f32"), (4096, 800, "f32"), (4096, 512, "f32"), (64, 800, "bf16"), (4096, 800, "bf16")):
        x = torch.randn(B, 64, T, device="cuda")
        y = torch.randint(0, 5, (B,), device="cuda")
        for graph in (False, True):
            torch.manual_seed(0)
            net = inn.FAST(fast_config(seq_len=T, act_dtype=act)).cuda().train()
            if graph:                           # the whole step as one HIP graph replay (isd_amd.graph)
                opt = make_opt(net, True)
                gs = GraphedTrainStep(net, opt, x, y, B)
                idx = torch.arange(B, device="cuda")

                def step():
                    gs.step(idx, 5e-4)
                    return gs.loss_sum
            else:
                opt = make_opt(net, False)

                def step():
                    opt.zero_grad(set_to_none=True)
                    loss = inn.token_mean_cross_entropy(net(x, forward_mode="default"), y)
                    loss.backward()
                    opt.step()
                    return loss
            dt, loss = timed(step, 20 if B <= 1024 else 5)
            net.eval()
            with torch.no_grad():
                di, _ = timed(lambda: net(x, forward_mode="default"), 20 if B <= 1024 else 5)
            print(f"FAST default {act:>4} B={B} T={T} {'graph replay' if graph else 'eager       '}: {dt*1e3:.3f} ms/step, "
                  f"{B/dt:.0f} trials/s | inference {di*1e3:.3f} ms, {B/di:.0f} trials/s")
    tail_only()


if __name__ == "__main__":
    main()
