"""Is a HIP bf16 training of the accuracy gate's task repeatable run to run (identical inputs)?  And how do the fp32-map /
bf16-map variants spread under 1e-6 input perturbations?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import isd_amd
import test_accuracy_gate_gpu as gate

Xtr, ytr = gate._task(gate.N_TRAIN, 10)
Xte, yte = gate._task(gate.N_TEST, 11)
def fit(X, prec, **kw):
    clf = isd_amd.FilterbankCNNClassifier(max_epochs=gate.EPOCHS, batch_size=gate.BS, warmup_epochs=2, seed=1, shuffle=False,
                                          precision=prec, **kw)
    clf.fit(X, ytr)
    flat = clf.model_.flat_params().detach().cpu().numpy().copy()
    return float((clf.predict(Xte) == yte).mean()), clf.history_[-1], flat
for prec in ("bf16", "fp32"):
    runs = [fit(Xtr, prec) for _ in range(3)]
    same = all(np.array_equal(runs[0][2], r[2]) for r in runs[1:])
    print(f"{prec}: three identical-input runs: " + "  ".join(f"{a:.4f}/{l:.3f}" for a, l, _ in runs) + f"   parameters bitwise equal: {same}", flush=True)
for kw in ({}, {"fused": False}):
    res = [fit(Xtr * np.float32(1.0 + 1e-6 * k), "bf16", **kw)[:2] for k in range(6)]
    print(f"bf16 {kw}: " + "  ".join(f"{a:.4f}/{l:.3f}" for a, l in res), flush=True)
