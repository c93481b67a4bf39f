"""Reference-native training shape (B=64, 64 ch, T=800): wall time per step vs GPU time (launch-bound regime)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import isd_amd
from isd_amd.classifier import _FastModel, _FeatureModel
from isd_amd.nn import fast_config


def run(name, tr, x, y, n=50):
    for _ in range(5):
        tr.step(x, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        tr.step(x, y)
    e1.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n * 1e3
    print(f"{name}: wall {wall:.3f} ms/step ({x.shape[0] / wall * 1e3:.0f} trials/s), GPU span {e0.elapsed_time(e1) / n:.3f} ms/step")


def main():
    torch.manual_seed(0)
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    m = _FastModel(fast_config(seq_len=800)).cuda()
    run(f"FAST train_head B={B} T=800", isd_amd.Trainer(m), torch.randn(B, 64, 800, device="cuda"),
        torch.randint(0, 5, (B,), device="cuda"))
    fm = _FeatureModel(576, 32, 5, 4).cuda()
    run(f"feature CNN B={B} [576 x 17]", isd_amd.Trainer(fm), torch.randn(B, 576, 17, device="cuda"),
        torch.randint(0, 5, (B,), device="cuda"))


if __name__ == "__main__":
    main()
