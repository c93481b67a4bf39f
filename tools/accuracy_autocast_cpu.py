#!/usr/bin/env python3
"""CPU only: the accuracy gate's CPU reference trained in fp32 and under torch.autocast(bfloat16) (the reference's
precision='bf16-mixed', scripts/train_fast.py:277) on the gate's own task -- what accuracy / last-epoch loss does the
REFERENCE's arithmetic reach in bf16?  (tests/test_accuracy_gate_gpu.py runs the same legs beside the HIP ones.)

    python tools/accuracy_autocast_cpu.py [cache.npz]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_accuracy_gate_gpu as gate  # noqa: E402


def main():
    cache = sys.argv[1] if len(sys.argv) > 1 else None
    Xtr, ytr = gate._task(gate.N_TRAIN, 10)
    Xte, yte = gate._task(gate.N_TEST, 11)
    if cache and os.path.exists(cache):
        f_all = np.load(cache)["f"]
    else:
        f_all = gate._oracle_features(np.concatenate([Xtr, Xte]), workers=8)
        if cache:
            np.savez(cache, f=f_all)
    f_all = torch.from_numpy(f_all)
    ftr, fte = f_all[:gate.N_TRAIN], f_all[gate.N_TRAIN:]
    for tag, kw in (("fp32", {}), ("autocast bf16", {"autocast": True})):
        p, last = gate._oracle_fit(ftr, ytr, min(torch.get_num_threads(), 16), **kw)
        pr = gate._oracle_predict(fte, p, **kw)
        print(f"{tag}: held-out accuracy {float((pr == yte).mean()):.4f}, last-epoch training loss {last:.4f}", flush=True)


if __name__ == "__main__":
    main()
