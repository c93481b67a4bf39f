"""Experiment driver and report aggregation with the reference's outputs (scripts/train_fast.py:68-265,
scripts/benchmark.py:35-102): per-subject K-fold cross-validated fine-tuning of FAST, best-fold selection on
validation accuracy, evaluation on a held-out test split, and the same CSV artefacts:

    <out>/FAST/sub-XX/fold_metrics.csv          Fold,Best_Val_Acc
    <out>/FAST/sub-XX/best_subject.pth          state_dict (reference key names)
    <out>/FAST/sub-XX/test_predictions.csv      '# Predicted,True'
    <out>/FAST/summary_per_subject.csv          Subject,Best_Val_Acc,Test_Acc,Test_F1
    <out>/FAST/global_test_predictions.csv

The model is ``isd_amd.nn.FAST`` in the mode the reference trains (``forward_mode='default'``,
trainer.py:58) with AdamW(5e-4) and the per-step cosine multiplier; all tensor work runs on the HIP kernels.
"""
import argparse
import csv
import os

import numpy as np
import torch

from .classifier import cosine_scheduler, lr_multiplier
from .data import DeviceStager
from .graph import GraphedTrainStep, graph_safe
from .optim import FusedAdamW
from .nn import FAST, fast_config, reset_dropout_streams, token_mean_cross_entropy, unit_grad


def accuracy(y_true, y_pred):
    return float((np.asarray(y_true) == np.asarray(y_pred)).mean()) if len(y_true) else float("nan")


def macro_prf(y_true, y_pred, n_classes=None):
    """Macro precision / recall / F1 (sklearn's ``average='macro'`` with zero_division=0)."""
    y_true, y_pred = np.asarray(y_true), np.asarray(y_pred)
    labels = np.unique(np.concatenate((y_true, y_pred))) if n_classes is None else np.arange(n_classes)
    P, R, F = [], [], []
    for c in labels:
        tp = float(((y_pred == c) & (y_true == c)).sum())
        fp = float(((y_pred == c) & (y_true != c)).sum())
        fn = float(((y_pred != c) & (y_true == c)).sum())
        p = tp / (tp + fp) if tp + fp else 0.0
        r = tp / (tp + fn) if tp + fn else 0.0
        P.append(p)
        R.append(r)
        F.append(2 * p * r / (p + r) if p + r else 0.0)
    return float(np.mean(P)), float(np.mean(R)), float(np.mean(F))


def kfold_indices(n, n_folds, seed):
    """Shuffled K-fold split (KFold(n_splits, shuffle=True, random_state=seed) semantics: contiguous folds of a
    seeded permutation; fold sizes differ by at most one)."""
    perm = np.random.RandomState(seed).permutation(n)
    sizes = np.full(n_folds, n // n_folds)
    sizes[: n % n_folds] += 1
    out, start = [], 0
    for s in sizes:
        val = perm[start:start + s]
        out.append((np.setdiff1d(perm, val, assume_unique=True), val))
        start += s
    return out


def predict(model, X, batch_size=256, forward_mode="default"):
    """``inference_on_loader`` (trainer.py:82-93): argmax class indices, int64.  ``X`` may already be on the device;
    host arrays are uploaded batch by batch through pinned buffers, the copy of batch k+1 under the kernels of batch k."""
    model.eval()
    outs = []
    n = len(X)
    on_dev = isinstance(X, torch.Tensor) and X.is_cuda
    with torch.no_grad():
        if on_dev:
            for i in range(0, n, batch_size):
                outs.append(model(X[i:i + batch_size].contiguous(), forward_mode=forward_mode).argmax(dim=1))
        elif n:
            st = DeviceStager()
            st.put(np.asarray(X[:batch_size], dtype=np.float32))
            for i in range(0, n, batch_size):
                if i + batch_size < n:
                    st.put(np.asarray(X[i + batch_size:i + 2 * batch_size], dtype=np.float32))
                outs.append(model(st.get(), forward_mode=forward_mode).argmax(dim=1))
    return torch.cat(outs).cpu().numpy() if outs else np.zeros((0,), np.int64)


def train_one_fold(config, Xtr, ytr, Xva, yva, max_epochs, batch_size, seed, forward_mode="default", lr=5e-4,
                   warmup_epochs=10, graph=None):
    """One fine-tuning run; returns (best_val_acc, best_state_dict, history).  The fold's trials go to the device
    once, asynchronously from pinned memory (the validation split uploads under the first training steps).
    ``graph`` (default: whenever the model allows it, isd_amd.graph.graph_safe): the optimisation step is captured
    once as a HIP graph and replayed -- at the reference's batch of 64 a step is launch-bound on the host."""
    torch.manual_seed(seed)
    reset_dropout_streams()                              # the fold's masks do not depend on what ran before it
    model = FAST(config).cuda()
    n = len(Xtr)
    bs = min(batch_size, n)
    iters = (n + bs - 1) // bs
    table = cosine_scheduler(1, 0.1, max_epochs, iters, warmup_epochs=min(warmup_epochs, max(max_epochs - 1, 0)))
    st = DeviceStager()
    st.put(np.asarray(Xtr, dtype=np.float32), np.asarray(ytr))
    st.put(np.asarray(Xva, dtype=np.float32))
    Xd, yd = st.get()
    Xva = st.get()
    use_graph = graph_safe(model) if graph is None else bool(graph)
    if use_graph:
        opt = FusedAdamW(model.parameters(), lr=torch.tensor(float(lr), device=Xd.device), capturable=True)
        model.train()
        try:
            gstep = GraphedTrainStep(model, opt, Xd, yd, bs, forward_mode)
        except RuntimeError as e:                        # a capture the runtime refuses: train eagerly, say so
            if graph:
                raise
            import warnings
            warnings.warn(f"HIP-graph capture of the training step failed ({e}); continuing with eager launches")
            torch.cuda.synchronize()
            use_graph = False
            torch.manual_seed(seed)
            reset_dropout_streams()
            model = FAST(config).cuda()
    if not use_graph:
        opt = FusedAdamW(model.parameters(), lr=lr)
    gen = torch.Generator().manual_seed(seed)
    best, best_sd, hist, step = -1.0, None, [], 0
    for ep in range(max_epochs):
        model.train()
        order = torch.randperm(n, generator=gen).cuda()
        tot = torch.zeros((), device=Xd.device)
        if use_graph:
            gstep.loss_sum.zero_()
        for i in range(iters):
            idx = order[i * bs:(i + 1) * bs]
            if use_graph:
                gstep.step(idx, lr * lr_multiplier(table, step))
            else:
                for g in opt.param_groups:
                    g["lr"] = lr * lr_multiplier(table, step)
                opt.zero_grad(set_to_none=True)
                logits = model(Xd[idx].contiguous(), forward_mode=forward_mode)
                loss = token_mean_cross_entropy(logits, yd[idx].contiguous())
                loss.backward(unit_grad(loss.device))
                opt.step()
                tot += loss.detach() * len(idx)          # summed on the device: one host read per epoch
            step += 1
        tot = float(gstep.loss_sum if use_graph else tot)
        val_acc = accuracy(yva, predict(model, Xva, batch_size, forward_mode))
        hist.append({"loss": tot / n, "val_acc": val_acc})
        if val_acc > best:                                   # ModelCheckpoint(monitor='val_acc', mode='max', top-1)
            best = val_acc
            best_sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    return best, best_sd, hist


def _save_predictions(path, y_pred, y_true):
    np.savetxt(path, np.array([y_pred, y_true]).T, delimiter=",", fmt="%d", header="Predicted,True")


def _fold_worker(device, tasks, results):
    """Worker process of the fold packing: bound to one GPU, trains (subject, fold) tasks until the sentinel."""
    torch.cuda.set_device(device)
    while True:
        task = tasks.get()
        if task is None:
            return
        key = task[0]
        try:
            _, cfg, Xtr, ytr, Xva, yva, max_epochs, batch_size, seed, mode = task
            acc, sd, _ = train_one_fold(cfg, Xtr, ytr, Xva, yva, max_epochs, batch_size, seed, mode)
            results.put((key, acc, {k: v.cpu().numpy() for k, v in sd.items()}))
        except Exception as e:                           # the parent re-raises; a silent worker would hang it
            import traceback
            results.put((key, None, f"{type(e).__name__}: {e}\n{traceback.format_exc()}"))


def run_folds(fold_tasks, workers=0, devices=None):
    """Train independent (subject, fold) fine-tunings.  ``fold_tasks``: {key: (cfg, Xtr, ytr, Xva, yva, max_epochs,
    batch_size, seed, mode)}.  ``workers == 0``: one after another in this process.  Otherwise the trainings -- 75 of
    them in the reference's protocol (scripts/train_fast.py:86-111), each launch-bound at batch 64 -- are packed over
    ``workers`` processes bound round-robin to ``devices`` (default: every visible GPU): several small trainings share
    a GPU, and the GPUs of a node work on different folds.  Returns {key: (best_val_acc, state_dict)}."""
    if workers <= 0:
        return {k: train_one_fold(*t)[:2] for k, t in fold_tasks.items()}
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    devices = list(range(torch.cuda.device_count())) if devices is None else list(devices)
    tasks, results = ctx.Queue(), ctx.Queue()
    procs = [ctx.Process(target=_fold_worker, args=(devices[w % len(devices)], tasks, results)) for w in range(workers)]
    for p in procs:
        p.start()
    for k, t in fold_tasks.items():
        tasks.put((k,) + tuple(t))
    for _ in procs:
        tasks.put(None)
    import queue as _queue
    out = {}

    def _abort(msg):
        for p in procs:
            if p.is_alive():
                p.terminate()
        for p in procs:
            p.join()
        raise RuntimeError(msg)
    while len(out) < len(fold_tasks):
        try:
            k, acc, sd = results.get(timeout=1.0)
        except _queue.Empty:
            dead = [p for p in procs if p.exitcode not in (None, 0)]
            if dead:                                     # killed (OOM, signal): it will never post its result
                _abort(f"fold worker exited with code {dead[0].exitcode}")
            if not any(p.is_alive() for p in procs) and results.empty():
                _abort("fold workers ended without returning every fold")
            continue
        if acc is None:
            _abort(f"fold {k} failed in its worker:\n{sd}")
        out[k] = (acc, {n: torch.from_numpy(v) for n, v in sd.items()})
    for p in procs:
        p.join()
        if p.exitcode != 0:
            raise RuntimeError(f"fold worker exited with code {p.exitcode}")
    return out


def finetune_per_subject_cv(train_val, test, out_dir, config=None, n_folds=5, max_epochs=200, batch_size=64, seed=42,
                            forward_mode="default", workers=0, devices=None, precision="32"):
    """``train_val`` / ``test``: {SID: (X [n,C,T], y)}.  Mirrors scripts/train_fast.py:68-265; returns the summary rows.
    ``workers`` > 0 packs the subject x fold trainings over that many processes / the GPUs in ``devices``
    (``run_folds``); results do not depend on the packing.  ``precision``: '32', or 'bf16-mixed' (the reference
    script's default, scripts/train_fast.py:277): the zone CNN keeps bf16 activations / gradients and runs on the bf16
    matrix cores, parameters and the transformer tail stay fp32.  A ``config`` that sets ``act_dtype`` wins."""

    def with_precision(cfg):
        if hasattr(cfg, "act_dtype"):
            return cfg
        import copy
        cfg = copy.copy(cfg)
        cfg.act_dtype = "bf16" if str(precision).startswith("bf16") else "f32"
        return cfg
    save_dir = os.path.join(out_dir, "FAST")
    os.makedirs(save_dir, exist_ok=True)
    rows, gp, gt = [], [], []
    fold_tasks = {}
    for sid, (X, y) in train_val.items():
        cfg = with_precision(config or fast_config(seq_len=int(X.shape[-1])))
        for fi, (tr, va) in enumerate(kfold_indices(len(X), n_folds, seed)):
            fold_tasks[(sid, fi)] = (cfg, X[tr], y[tr], X[va], y[va], max_epochs, batch_size, seed, forward_mode)
    done = run_folds(fold_tasks, workers, devices)
    for sid, (X, y) in train_val.items():
        cfg = with_precision(config or fast_config(seq_len=int(X.shape[-1])))
        sub_dir = os.path.join(save_dir, f"sub-{sid}")
        os.makedirs(sub_dir, exist_ok=True)
        fold_rows, best_acc, best_sd = [], -1.0, None
        for fi in range(n_folds):
            acc, sd = done[(sid, fi)]
            fold_rows.append([fi, acc])
            if acc > best_acc:
                best_acc, best_sd = acc, sd
        with open(os.path.join(sub_dir, "fold_metrics.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Fold", "Best_Val_Acc"])
            w.writerows(fold_rows)
        torch.save({k: v.cpu() for k, v in best_sd.items()}, os.path.join(sub_dir, "best_subject.pth"))
        test_acc = test_f1 = float("nan")
        if sid in test:
            model = FAST(cfg).cuda()
            model.load_state_dict(best_sd)
            Xt, yt = test[sid]
            yp = predict(model, Xt, batch_size, forward_mode)
            test_acc, test_f1 = accuracy(yt, yp), macro_prf(yt, yp)[2]
            _save_predictions(os.path.join(sub_dir, "test_predictions.csv"), yp, np.asarray(yt))
            gp.append(yp)
            gt.append(np.asarray(yt))
        rows.append([sid, best_acc, test_acc, test_f1])
    with open(os.path.join(save_dir, "summary_per_subject.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Subject", "Best_Val_Acc", "Test_Acc", "Test_F1"])
        w.writerows(rows)
    if gp:
        _save_predictions(os.path.join(save_dir, "global_test_predictions.csv"), np.concatenate(gp), np.concatenate(gt))
    return rows


def _read_predictions(path):
    a = np.loadtxt(path, delimiter=",", comments="#", ndmin=2)
    return a[:, 0].astype(int), a[:, 1].astype(int)


def process_results(results_dir, model_name="FAST"):
    """scripts/benchmark.py:35-102: per-subject and global accuracy / macro-F1 / precision / recall from the CSVs."""
    folder = os.path.join(results_dir, model_name)
    if not os.path.isdir(folder):
        return None, None
    per = []
    for item in sorted(os.listdir(folder)):
        p = os.path.join(folder, item, "test_predictions.csv")
        if item.startswith("sub-") and os.path.exists(p):
            yp, yt = _read_predictions(p)
            pr, rc, f1 = macro_prf(yt, yp)
            per.append({"Subject": int(item[4:]), "Accuracy": accuracy(yt, yp), "F1": f1, "Precision": pr,
                        "Recall": rc, "N_samples": len(yt)})
    if not per:
        return None, None
    accs = np.array([r["Accuracy"] for r in per])
    f1s = np.array([r["F1"] for r in per])
    gpath = os.path.join(folder, "global_test_predictions.csv")
    if os.path.exists(gpath):
        yp, yt = _read_predictions(gpath)
        pr, rc, f1 = macro_prf(yt, yp)
        ga, gf, gpr, grc = accuracy(yt, yp), f1, pr, rc
    else:
        ga, gf = float(accs.mean()), float(f1s.mean())
        gpr, grc = float(np.mean([r["Precision"] for r in per])), float(np.mean([r["Recall"] for r in per]))
    std = lambda v: float(np.std(v, ddof=1)) if len(v) > 1 else float("nan")     # pandas .std()
    summary = {"Model": model_name, "Acc_Mean": ga, "Acc_Std": std(accs), "F1_Mean": gf, "F1_Std": std(f1s),
               "Precision_Mean": gpr, "Recall_Mean": grc, "N_subjects": len(per)}
    return per, summary


def main(argv=None):
    ap = argparse.ArgumentParser(description="Per-subject cross-validated FAST fine-tuning on MI355X")
    ap.add_argument("--data", required=True, help="standardized cache (.npz / .h5) with {SID}/X, {SID}/Y")
    ap.add_argument("--test", default=None, help="standardized cache of the held-out test split")
    ap.add_argument("--output_dir", default="results/finetune_official")
    ap.add_argument("--gpu", type=int, default=0)
    ap.add_argument("--epochs", type=int, default=200)
    ap.add_argument("--batch_size", type=int, default=64)
    ap.add_argument("--n_folds", type=int, default=5)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--forward_mode", default="default", choices=["default", "train_head"])
    ap.add_argument("--precision", default="bf16-mixed", choices=["bf16-mixed", "32"],
                    help="training precision (scripts/train_fast.py:277 defaults to bf16-mixed)")
    ap.add_argument("--workers", type=int, default=0, help="pack the subject x fold trainings over this many processes")
    ap.add_argument("--devices", type=int, nargs="*", default=None, help="GPUs the workers are bound to (default: all)")
    args = ap.parse_args(argv)
    from .data import load_standardized
    torch.cuda.set_device(args.gpu)
    tv = load_standardized(args.data)
    te = load_standardized(args.test) if args.test else {}
    rows = finetune_per_subject_cv(tv, te, args.output_dir, None, args.n_folds, args.epochs, args.batch_size, args.seed,
                                   args.forward_mode, args.workers, args.devices, args.precision)
    per, summary = process_results(args.output_dir)
    print(summary if summary else rows)


if __name__ == "__main__":
    main()
