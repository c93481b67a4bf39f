"""GPU (slow): the north star's accuracy gate -- "classifier accuracy within +-0.1 % of CPU reference" -- at a size
that can resolve 0.1 %.

Three trainings of the BASELINE config-2 classifier on the same 1 024 synthetic trials (SURVEY.md 8d task, made
harder with extra noise so that the held-out accuracy sits below 100 %), same initial parameters, same batches,
same schedule:
  * the CPU reference path: scipy butter / sosfilt / stft features (oracle.dsp) + the functional torch restatement of
    Conv4Layers + Linear + CE (oracle.cnn) + torch AdamW on the host;
  * the HIP path in fp32;
  * the HIP path with bf16 activations / gradients (BASELINE config 3).
Each is evaluated on 4 096 held-out trials (another seed): one trial is 0.024 % of the set, so the 0.1 % gate is four
trials wide.
"""
import concurrent.futures as cf
import multiprocessing as mp

import numpy as np
import pytest
import torch

from oracle import cnn as ocnn, dsp as odsp

pytestmark = [pytest.mark.gpu, pytest.mark.slow]

N_TRAIN, N_TEST, EPOCHS, BS = 1024, 4096, 30, 64


def _task(n, seed):
    """SURVEY 8d trials with a second unit of white noise on top (noise power x 2): calibrated on the CPU reference
    so that it learns the task without saturating it (held-out accuracy ~97 %, ~0.3 % of the held-out trials within
    1e-2 of a decision boundary)."""
    X, y = odsp.synth_trials(n, 64, 512, 256.0, seed=seed)
    X += np.random.default_rng(seed + 1000).standard_normal(X.shape, dtype=np.float32)
    return X, y


def _scipy_features(X):
    return odsp.extract_features_scipy(X, fs=256.0, bands=odsp.BANDS_9)


def _oracle_features(X, workers=8):
    """oracle.dsp on host processes that never touch the GPU (scipy is single-threaded: ~45 trials/s per core)."""
    chunks = np.array_split(X, workers * 2)
    with cf.ProcessPoolExecutor(workers, mp_context=mp.get_context("spawn")) as ex:
        return np.concatenate(list(ex.map(_scipy_features, chunks)))


def test_held_out_accuracy_within_a_tenth_of_a_percent_of_the_cpu_reference():
    import isd_amd
    from isd_amd.classifier import _FeatureModel
    Xtr, ytr = _task(N_TRAIN, 10)
    Xte, yte = _task(N_TEST, 11)
    acc, loss = {}, {}
    for prec in ("fp32", "bf16"):
        clf = isd_amd.FilterbankCNNClassifier(max_epochs=EPOCHS, batch_size=BS, warmup_epochs=2, seed=1, shuffle=False,
                                              precision=prec)
        clf.fit(Xtr, ytr)
        acc[prec] = float((clf.predict(Xte) == yte).mean())
        loss[prec] = clf.history_[-1]
    # the CPU reference: same initial parameters (seed 1), same schedule, same batch order
    f_all = torch.from_numpy(_oracle_features(np.concatenate([Xtr, Xte])))
    ftr, fte = f_all[:N_TRAIN], f_all[N_TRAIN:]
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
            ls = ocnn.cross_entropy(ocnn.feature_cnn_logits(ftr[sl], p), yt[sl])
            ls.backward()
            opt.step()
            tot += float(ls) * BS
            step += 1
        last = tot / N_TRAIN
    with torch.no_grad():
        pred = torch.cat([ocnn.predict(ocnn.feature_cnn_logits(fte[i:i + 512], p)) for i in range(0, N_TEST, 512)])
    acc["cpu"] = float((pred.numpy() == yte).mean())
    print(f"held-out accuracy on {N_TEST} trials: cpu reference {acc['cpu']:.4f}, hip fp32 {acc['fp32']:.4f}, "
          f"hip bf16 {acc['bf16']:.4f}; last-epoch training loss cpu {last:.4f}, fp32 {loss['fp32']:.4f}, "
          f"bf16 {loss['bf16']:.4f}")
    assert 0.5 < acc["cpu"] < 0.995, acc                  # the task is learnt and is not saturated: the gate can bite
    assert abs(acc["fp32"] - acc["cpu"]) <= 0.001 + 1e-9, acc
    assert abs(loss["fp32"] - last) < 2e-3 * max(1.0, last), (loss, last)
    # bf16 activations: stated, not gated at 0.1 % (its logits differ at the 1e-2 level by construction)
    assert abs(acc["bf16"] - acc["cpu"]) <= 0.01, acc
