"""GPU (slow): the north star's accuracy gate -- "classifier accuracy within +-0.1 % of CPU reference" -- at a size
that can resolve 0.1 %.

Trainings of the BASELINE config-2 classifier on the same 1 024 synthetic trials (SURVEY.md 8d task with extra white
noise), same initial parameters, same batches, same schedule:
  * the CPU reference path: scipy butter / sosfilt / stft features (oracle.dsp) + the functional torch restatement of
    Conv4Layers + Linear + CE (oracle.cnn) + torch AdamW on the host -- run TWICE, with all host threads and with one;
  * the HIP path in fp32;
  * the HIP path with bf16 activations / gradients (BASELINE config 3);
  * the CPU reference under ``torch.autocast("cpu", bfloat16)`` -- the reference's own precision='bf16-mixed'
    (scripts/train_fast.py:277) -- again with all host threads and with one.
Each is evaluated on 4 096 held-out trials (another seed): one trial is 0.024 % of the set, so the 0.1 % gate is four
trials wide.

Why two CPU runs: AdamW training is a chaotic map of its rounding errors.  Calibrating this test on the CPU reference
alone (30 epochs, three runs that differ only in the host thread count or in a 1e-6 relative perturbation of the
initial parameters) gave held-out accuracies of 90.7 / 99.1 / 96.1 % with one extra unit of noise at half amplitude
-- the reference does not define its own accuracy to 0.1 % there, so no implementation can be "within 0.1 %" of it --
and 99.90 / 99.88 / 99.90 % (0 - 1 of 4 096 predictions differ) at the noise level used here.  The gate is therefore
evaluated where the reference itself is reproducible to 0.1 %, the test checks that precondition on the box it runs
on, and it also checks the part that does not depend on the trajectory: the HIP-trained parameters evaluated through
the CPU pipeline give the HIP pipeline's predictions.

bf16: neither the reference's own autocast training nor the HIP one defines its accuracy to 0.1 % at this setting --
bf16-mixed training of this task is chaotic at the PERCENT level.  Measured (profiles/r04_accuracy_autocast_cpu.txt,
profiles/r04_training_spread.txt): the CPU reference under autocast, same task and initial parameters, only the host
thread count (= the summation order inside the bf16 convolutions) changed: 99.88 / 99.78 / 99.19 % in the build
container (8 / 4 / 1 threads), 99.54 / 100.00 % on a GPU box (16 / 1); the HIP bf16 path on inputs scaled by
1 + 1e-6 k, k = 0..4: 99.29 / 99.27 / 99.58 / 100.00 / 92.77 % -- against 99.80 ... 99.98 % for the HIP fp32 path under
the same perturbations and 99.90 % for the CPU fp32 reference at every thread count.  One run of each says nothing, so
(the CPU reference under the same 1 + 1e-6 k protocol, ten runs: nine between 99.78 and 99.90 %, one at 95.70 %:
both implementations lose about one run in six to ten to a training that has not converged after 30 epochs) the bf16 leg
compares MEDIANS: five HIP trainings (perturbed as above) against the CPU autocast reference at three
thread counts, medians within 1 %, everything printed.  (A single-step comparison, where there is no chaos, is G15 /
G16 in tests/test_cnn_gpu.py: the HIP bf16 step is closer to the fp32 reference than the reference's own autocast is.)
"""
import concurrent.futures as cf
import multiprocessing as mp

import numpy as np
import pytest
import torch

from oracle import cnn as ocnn, dsp as odsp

pytestmark = [pytest.mark.gpu, pytest.mark.slow]

N_TRAIN, N_TEST, EPOCHS, BS, NOISE = 1024, 4096, 30, 64, 0.75


def _task(n, seed):
    """SURVEY 8d trials with 0.75 units of white noise on top."""
    X, y = odsp.synth_trials(n, 64, 512, 256.0, seed=seed)
    X += NOISE * np.random.default_rng(seed + 1000).standard_normal(X.shape, dtype=np.float32)
    return X, y


def _scipy_features(X):
    return odsp.extract_features_scipy(X, fs=256.0, bands=odsp.BANDS_9)


def _oracle_features(X, workers=8):
    """oracle.dsp on host processes that never touch the GPU (scipy is single-threaded: ~45 trials/s per core)."""
    chunks = np.array_split(X, workers * 2)
    with cf.ProcessPoolExecutor(workers, mp_context=mp.get_context("spawn")) as ex:
        return np.concatenate(list(ex.map(_scipy_features, chunks)))


def _oracle_predict(fte, p, autocast=False):
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
        return torch.cat([ocnn.predict(ocnn.feature_cnn_logits(fte[i:i + 512], p).float())
                          for i in range(0, len(fte), 512)]).numpy()


def _oracle_fit(ftr, ytr, threads, autocast=False):
    """The CPU reference training: same initial parameters (seed 1), same schedule, same batch order as the estimator.
    ``autocast``: forward under torch.autocast(bfloat16) with fp32 master parameters and an fp32 loss -- what the
    reference's precision='bf16-mixed' (scripts/train_fast.py:277) does."""
    from isd_amd.classifier import _FeatureModel
    old = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        torch.manual_seed(1)
        ref_model = _FeatureModel(9 * 64, 32, 5, 4)
        p = {k[len("net."):]: v.detach().clone().requires_grad_() for k, v in ref_model.state_dict().items()}
        opt = torch.optim.AdamW(list(p.values()), lr=5e-4)
        iters = N_TRAIN // BS
        table = ocnn.cosine_scheduler(1, 0.1, EPOCHS, iters, warmup_epochs=2)
        yt = torch.from_numpy(ytr)
        step, last = 0, 0.0
        for ep in range(EPOCHS):
            tot = 0.0
            for i in range(iters):
                for gr in opt.param_groups:
                    gr["lr"] = 5e-4 * ocnn.lr_multiplier(table, step)
                opt.zero_grad()
                sl = slice(i * BS, (i + 1) * BS)
                with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
                    logits = ocnn.feature_cnn_logits(ftr[sl], p)
                ls = ocnn.cross_entropy(logits.float(), yt[sl])
                ls.backward()
                opt.step()
                tot += float(ls.detach()) * BS
                step += 1
            last = tot / N_TRAIN
        return p, last
    finally:
        torch.set_num_threads(old)


def test_held_out_accuracy_within_a_tenth_of_a_percent_of_the_cpu_reference():
    import isd_amd
    Xtr, ytr = _task(N_TRAIN, 10)
    Xte, yte = _task(N_TEST, 11)
    acc, loss, clfs, preds = {}, {}, {}, {}
    for prec in ("fp32", "bf16"):
        clf = isd_amd.FilterbankCNNClassifier(max_epochs=EPOCHS, batch_size=BS, warmup_epochs=2, seed=1, shuffle=False,
                                              precision=prec)
        clf.fit(Xtr, ytr)
        preds[prec] = clf.predict(Xte)
        acc[prec] = float((preds[prec] == yte).mean())
        loss[prec] = clf.history_[-1]
        clfs[prec] = clf
    hip_bf16 = [(acc["bf16"], loss["bf16"])]                      # + four trainings on inputs scaled by 1 + 1e-6 k
    for k in range(1, 5):
        clf = isd_amd.FilterbankCNNClassifier(max_epochs=EPOCHS, batch_size=BS, warmup_epochs=2, seed=1, shuffle=False,
                                              precision="bf16")
        clf.fit(Xtr * np.float32(1.0 + 1e-6 * k), ytr)
        hip_bf16.append((float((clf.predict(Xte) == yte).mean()), clf.history_[-1]))
    f_all = torch.from_numpy(_oracle_features(np.concatenate([Xtr, Xte])))
    ftr, fte = f_all[:N_TRAIN], f_all[N_TRAIN:]
    # (1) trajectory-independent: the HIP-trained parameters through the CPU pipeline = the HIP pipeline's predictions
    p_hip = {k[len("net."):]: v.detach().cpu() for k, v in clfs["fp32"].model_.state_dict().items()}
    flips = int((_oracle_predict(fte, p_hip) != preds["fp32"]).sum())
    # (2) the CPU reference, twice: every host thread, and one
    n_thr = min(torch.get_num_threads(), 16)
    p_a, last_a = _oracle_fit(ftr, ytr, n_thr)
    p_b, last_b = _oracle_fit(ftr, ytr, 1)
    acc["cpu"] = float((_oracle_predict(fte, p_a) == yte).mean())
    acc["cpu_1thread"] = float((_oracle_predict(fte, p_b) == yte).mean())
    # (3) the CPU reference under bf16 autocast (the reference's bf16-mixed) at three thread counts
    cpu_bf16 = []
    for thr in sorted({n_thr, max(n_thr // 4, 2), 1}, reverse=True):
        p_c, last_c = _oracle_fit(ftr, ytr, thr, autocast=True)
        cpu_bf16.append((float((_oracle_predict(fte, p_c, autocast=True) == yte).mean()), last_c, thr))
    fmt = lambda runs: " ".join(f"{r[0]:.4f}/{r[1]:.3f}" for r in runs)                                  # noqa: E731
    print(f"held-out accuracy on {N_TEST} trials: cpu reference {acc['cpu']:.4f} ({n_thr} threads) / "
          f"{acc['cpu_1thread']:.4f} (1 thread), hip fp32 {acc['fp32']:.4f}; last-epoch training loss cpu {last_a:.4f} / "
          f"{last_b:.4f}, hip fp32 {loss['fp32']:.4f}; HIP-trained parameters: {flips} of {N_TEST} predictions differ "
          f"between the HIP and the CPU pipeline.  bf16-mixed (accuracy/last-epoch loss per run): cpu reference under "
          f"autocast at {[r[2] for r in cpu_bf16]} threads: {fmt(cpu_bf16)}; hip bf16 on inputs scaled by 1 + 1e-6 k: "
          f"{fmt(hip_bf16)}")
    assert flips <= 2, flips                               # inference parity at scale (ties at the 1e-6 level only)
    assert acc["cpu"] > 0.9, acc                           # the task is learnt
    # the precondition of the gate: the reference defines its own accuracy to 0.1 % at this setting
    assert abs(acc["cpu"] - acc["cpu_1thread"]) <= 0.001 + 1e-9, acc
    assert abs(acc["fp32"] - acc["cpu"]) <= 0.001 + 1e-9, acc
    # bf16-mixed: medians (module docstring)
    med_hip, med_cpu = float(np.median([r[0] for r in hip_bf16])), float(np.median([r[0] for r in cpu_bf16]))
    assert abs(med_hip - med_cpu) <= 0.01, (hip_bf16, cpu_bf16)
    assert med_hip > 0.95, hip_bf16
