"""How reproducible is a bf16-mixed training of the accuracy gate's task?  HIP bf16 / HIP fp32 trainings on inputs that
differ by 1e-6-relative perturbations (and through either fused extractor), held-out accuracy and last-epoch loss of each.
(The CPU reference under autocast is in tests/test_accuracy_gate_gpu.py; tools/accuracy_autocast_cpu.py.)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import isd_amd
import test_accuracy_gate_gpu as gate

Xtr, ytr = gate._task(gate.N_TRAIN, 10)
Xte, yte = gate._task(gate.N_TEST, 11)
for prec in ("bf16", "fp32"):
    for serial in ("0", "1"):
        os.environ["ISD_FUSED_SERIAL"] = serial
        res = []
        for k in range(5):
            Xp = Xtr * np.float32(1.0 + 1e-6 * k)
            clf = isd_amd.FilterbankCNNClassifier(max_epochs=gate.EPOCHS, batch_size=gate.BS, warmup_epochs=2, seed=1,
                                                  shuffle=False, precision=prec)
            clf.fit(Xp, ytr)
            res.append((float((clf.predict(Xte) == yte).mean()), clf.history_[-1]))
        print(f"hip {prec} extractor {'serial' if serial == '1' else 'lane-scan'}: " +
              "  ".join(f"{a:.4f}/{l:.3f}" for a, l in res) + f"   mean acc {np.mean([a for a, _ in res]):.4f}", flush=True)
