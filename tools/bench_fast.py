"""Throughput of the reference-native path: FAST 'train_head' on raw EEG (fwd+bwd+AdamW), device-resident."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import isd_amd
from isd_amd.classifier import _FastModel
from isd_amd.nn import fast_config

def main():
    for B, T in ((1024, 512), (4096, 512), (1024, 800)):
        torch.manual_seed(0)
        m = _FastModel(fast_config(seq_len=T)).cuda()
        tr = isd_amd.Trainer(m)
        x = torch.randn(B, 64, T, device="cuda")
        y = torch.randint(0, 5, (B,), device="cuda")
        for _ in range(2):
            tr.step(x, y)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            out = tr.step(x, y)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"FAST train_head B={B} T={T}: {dt*1e3:.2f} ms/step, {B/dt:.0f} trials/s, loss {float(out['loss']):.4f}, "
              f"mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")

if __name__ == "__main__":
    main()
