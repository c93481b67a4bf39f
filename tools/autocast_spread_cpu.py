#!/usr/bin/env python3
"""CPU only: the accuracy gate's CPU reference under torch.autocast(bfloat16) on inputs scaled by 1 + 1e-6 k -- the same
perturbation protocol tools/bf16_spread.py applies to the HIP bf16 path -- so that the two distributions can be compared
run for run (does the reference's own bf16-mixed training have non-converging outliers too?).

    python tools/autocast_spread_cpu.py [n_runs]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_accuracy_gate_gpu as gate  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    Xtr, ytr = gate._task(gate.N_TRAIN, 10)
    Xte, yte = gate._task(gate.N_TEST, 11)
    fte = torch.from_numpy(gate._oracle_features(Xte, workers=8))
    res = []
    for k in range(n):
        ftr = torch.from_numpy(gate._oracle_features(Xtr * np.float32(1.0 + 1e-6 * k), workers=8))
        p, last = gate._oracle_fit(ftr, ytr, min(torch.get_num_threads(), 16), autocast=True)
        acc = float((gate._oracle_predict(fte, p, autocast=True) == yte).mean())
        res.append((acc, last))
        print(f"k={k}: cpu autocast accuracy {acc:.4f} last-epoch loss {last:.3f}", flush=True)
    a = np.array([r[0] for r in res])
    print(f"cpu reference under bf16 autocast, inputs scaled by 1 + 1e-6 k, k < {n}: min {a.min():.4f} median "
          f"{np.median(a):.4f} max {a.max():.4f}; runs below 0.97: {(a < 0.97).sum()}")


if __name__ == "__main__":
    main()
