import sys, time, tempfile
sys.path.insert(0, '.')
import numpy as np, torch
from isd_amd import experiment as E
rng = np.random.default_rng(0)
tv = {sid: (rng.standard_normal((40, 64, 800)).astype(np.float32), rng.integers(0, 5, 40).astype(np.uint8)) for sid in ("01", "02")}
te = {"01": (rng.standard_normal((10, 64, 800)).astype(np.float32), rng.integers(0, 5, 10).astype(np.uint8))}
d = tempfile.mkdtemp()
t0 = time.perf_counter()
rows = E.finetune_per_subject_cv(tv, te, d, None, n_folds=2, max_epochs=3, batch_size=16, seed=1)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
steps = 2 * 2 * 3 * 2      # subjects x folds x epochs x iterations (20 train trials / 16 -> 2)
print(rows)
print(E.process_results(d)[1])
print(f"{dt:.2f} s total, ~{dt / steps * 1e3:.1f} ms per optimisation step incl. validation (production FAST, default mode, B=16)")
