"""GPU: estimators (fit/predict), the autograd-free hot path vs the autograd modules, optimizer wiring."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from oracle import cnn as ocnn, dsp as odsp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def isd():
    import isd_amd
    assert torch.cuda.is_available()
    return isd_amd


def test_hot_path_equals_autograd_modules(isd):
    from isd_amd.classifier import _FeatureModel
    import isd_amd.nn as inn
    torch.manual_seed(0)
    m = _FeatureModel(9 * 8, 32, 5, 4).cuda()
    feats = torch.randn(10, 72, 17, device="cuda")
    y = torch.randint(0, 5, (10,), device="cuda")
    out = isd.HotPath(m).forward(feats, y, want_grad=True)
    g_manual = m.flat_grads().clone()
    m.zero_grad(set_to_none=True)
    loss = inn.token_mean_cross_entropy(m.net.token_logits(feats.view(10, 9, 8, 17)), y)
    loss.backward()
    g_auto = torch.cat([p.grad.reshape(-1) for p in m._ordered_params()])
    assert abs(float(out["loss"]) - float(loss)) < 1e-6
    assert rel_err(g_manual.cpu(), g_auto.cpu()) < 1e-6
    assert np.array_equal(out["pred"].cpu().numpy(), out["logits"].argmax(1).cpu().numpy())


def test_fast_adamw_trajectory_matches_reference_golden(isd):
    """G9: three AdamW steps of FAST(small_config) train_head through Trainer == the reference's losses/params."""
    from isd_amd.classifier import _FastModel
    import isd_amd.nn as inn
    g5, g9 = load_golden("g5_fast_small.npz"), load_golden("g9_adamw.npz")
    electrodes = ["Fp1", "Fp2", "F3", "F4", "C3", "C4", "O1", "O2"]
    zones = {"Frontal": ["Fp1", "Fp2", "F3", "F4"], "Central": ["C3", "C4"], "Occipital": ["O1", "O2"]}
    cfg = inn.fast_config(electrodes, zones, dim_cnn=16, dim_token=16, seq_len=500, n_classes=3, num_layers=1,
                          num_heads=4, dropout=0.0)
    m = _FastModel(cfg).cuda()
    sd = {k[3:]: torch.from_numpy(g5[k]) for k in g5.files if k.startswith("sd.")}
    m.net.load_state_dict(sd)
    tr = isd.Trainer(m, lr=5e-4, weight_decay=1e-2, schedule=None)
    x = torch.from_numpy(g5["x"]).cuda()
    y = torch.from_numpy(g5["labels"]).cuda()
    losses = [float(tr.step(x, y)["loss"]) for _ in range(3)]
    np.testing.assert_allclose(losses, g9["losses"], rtol=2e-5)
    for k, p in m.net.named_parameters():
        if "final." + k in g9.files:                       # the parameters train_head optimises
            assert rel_err(p.detach().cpu(), g9["final." + k]) < 2e-5, k


def test_fast_head_classifier_reference_checkpoint_predicts_bitexact(isd):
    g = load_golden("g6_fast_prod.npz")
    clf = isd.FASTHeadClassifier()
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd.")}
    clf.load_reference_state_dict(sd)
    X = np.random.default_rng(6).standard_normal((4, 64, 800)).astype(np.float32)
    pred = clf.predict(X)
    assert pred.dtype == np.int64 and np.array_equal(pred, g["train_head_pred"])
    assert rel_err(clf.decision_function(X), g["train_head_logits"]) < 1e-4


def test_filterbank_cnn_classifier_learns_and_matches_cpu_path_accuracy(isd):
    """Synthetic task of SURVEY 8d (class-dependent tone): the HIP pipeline learns it, and its accuracy equals
    the oracle pipeline's (same init, same batches) within 0.1 %."""
    X, y = odsp.synth_trials(256, 64, 512, 256.0, seed=0)
    clf = isd.FilterbankCNNClassifier(max_epochs=40, batch_size=64, warmup_epochs=2, seed=1, shuffle=False)
    clf.fit(X, y)
    acc = clf.score(X, y)
    assert acc > 0.9, acc
    assert clf.predict(X[:7]).shape == (7,)
    # oracle-side replay: same initial parameters, same schedule, same batch order
    torch.manual_seed(1)
    from isd_amd.classifier import _FeatureModel
    ref_model = _FeatureModel(9 * 64, 32, 5, 4)
    p = {k[len("net."):]: v.detach().clone().requires_grad_() for k, v in ref_model.state_dict().items()}
    feats = torch.from_numpy(odsp.extract_features(X, fs=256.0, bands=odsp.BANDS_9))
    opt = torch.optim.AdamW(list(p.values()), lr=5e-4)
    table = ocnn.cosine_scheduler(1, 0.1, 40, 4, warmup_epochs=2)
    step = 0
    yt = torch.from_numpy(y)
    for ep in range(40):
        for i in range(4):
            for gr in opt.param_groups:
                gr["lr"] = 5e-4 * ocnn.lr_multiplier(table, step)
            opt.zero_grad()
            sl = slice(i * 64, (i + 1) * 64)
            ocnn.cross_entropy(ocnn.feature_cnn_logits(feats[sl], p), yt[sl]).backward()
            opt.step()
            step += 1
    with torch.no_grad():
        acc_ref = float((ocnn.predict(ocnn.feature_cnn_logits(feats, p)).numpy() == y).mean())
        # trajectory-independent: the HIP-trained parameters through the CPU pipeline give the HIP pipeline's predictions
        p_hip = {k[len("net."):]: v.detach().cpu() for k, v in clf.model_.state_dict().items()}
        flips = int((ocnn.predict(ocnn.feature_cnn_logits(feats, p_hip)).numpy() != clf.predict(X)).sum())
    assert flips <= 1, flips
    # The two TRAININGS are chaotic maps of their rounding errors: HIP trainings of this classifier on inputs scaled by
    # 1 + 1e-6 k end between 99.80 % and 99.98 % on 4 096 held-out trials (tools/bf16_spread.py, profiles/
    # r04_training_spread.txt), so on these 256 trials an accuracy is defined to one trial, not to a tenth of one: the
    # round-3 form of this assertion (|acc - acc_ref| <= 0.1 % = identical predictions) held for one rounding pattern of
    # the feature kernels and broke for another whose features differ from it by 1e-5.  The 0.1 % gate of the north star
    # is tests/test_accuracy_gate_gpu.py, at a size that resolves it.
    assert abs(acc - acc_ref) <= 1.0 / len(y) + 1e-9, (acc, acc_ref)


def test_estimator_accepts_device_tensors_and_uint8_labels(isd):
    x = torch.randn(16, 64, 512, device="cuda")
    y = torch.randint(0, 5, (16,), dtype=torch.uint8)
    clf = isd.FilterbankCNNClassifier(max_epochs=1, batch_size=8, n_layers=2, bands=isd.BANDS_5)
    assert clf.fit(x, y) is clf
    assert clf.predict(x).shape == (16,)
    assert clf.extractor_.fb.precision in ("f64", "mixed")


def test_experiment_driver_writes_reference_artifacts(isd, tmp_path):
    """Per-subject K-fold fine-tuning in the reference's trained mode ('default') on tiny synthetic data."""
    from isd_amd import experiment as E
    import isd_amd.nn as inn
    rng = np.random.default_rng(0)
    electrodes = ["Fp1", "Fp2", "F3", "F4", "C3", "C4", "O1", "O2"]
    zones = {"Frontal": ["Fp1", "Fp2", "F3", "F4"], "Central": ["C3", "C4"], "Occipital": ["O1", "O2"]}
    cfg = inn.fast_config(electrodes, zones, dim_cnn=16, dim_token=16, seq_len=500, n_classes=3, num_layers=1,
                          num_heads=4, dropout=0.1)
    tv = {sid: (rng.standard_normal((12, 8, 500)).astype(np.float32), rng.integers(0, 3, 12).astype(np.uint8))
          for sid in ("01", "02")}
    te = {"01": (rng.standard_normal((5, 8, 500)).astype(np.float32), rng.integers(0, 3, 5).astype(np.uint8))}
    rows = E.finetune_per_subject_cv(tv, te, str(tmp_path), cfg, n_folds=2, max_epochs=2, batch_size=4, seed=1)
    assert [r[0] for r in rows] == ["01", "02"] and all(0.0 <= r[1] <= 1.0 for r in rows)
    base = tmp_path / "FAST"
    for f in ("sub-01/fold_metrics.csv", "sub-01/best_subject.pth", "sub-01/test_predictions.csv",
              "summary_per_subject.csv", "global_test_predictions.csv", "sub-02/best_subject.pth"):
        assert (base / f).exists(), f
    assert open(base / "sub-01" / "test_predictions.csv").readline().strip() == "# Predicted,True"
    sd = torch.load(base / "sub-01" / "best_subject.pth")
    assert "head.encoders.Frontal.cnn2.weight" in sd and "transformer.0.attn.in_proj_weight" in sd
    per, summary = E.process_results(str(tmp_path))
    assert summary["N_subjects"] == 1 and per[0]["N_samples"] == 5


def test_fused_classifier_step_modes_match_layerwise_path(isd):
    """isd_featcnn_step (one call) against the layer-wise call sequence: inference, loss-only and training modes,
    uint8 and int64 labels, a batch that does not fill the last wave."""
    from isd_amd.classifier import _FeatureModel
    import isd_amd._lib as L
    torch.manual_seed(5)
    m = _FeatureModel(9 * 8, 32, 5, 4).cuda()
    hp = isd.HotPath(m)
    feats = torch.randn(37, 72, 17, device="cuda")
    y = torch.randint(0, 5, (37,), device="cuda")
    assert L.lib().isd_featcnn_supported(m.conv_plan(feats)._h, 37, 17, 5) == 1
    lw = m.net.token_logits(feats.view(37, 9, 8, 17)).squeeze(1)            # autograd modules = layer-wise kernels
    inf = hp.forward(feats)
    assert "loss" not in inf and rel_err(inf["logits"].cpu(), lw.detach().cpu()) < 1e-6
    assert np.array_equal(inf["pred"].cpu().numpy(), lw.argmax(1).cpu().numpy())
    ev = hp.forward(feats, y.to(torch.uint8), global_batch=74)
    ref_loss = torch.nn.functional.cross_entropy(lw, y, reduction="sum") / 74
    assert abs(float(ev["loss"]) - float(ref_loss)) < 1e-6
    m.zero_grad(set_to_none=True)
    tr = hp.forward(feats, y, global_batch=74, want_grad=True)
    g_fused = m.flat_grads().clone()
    m.zero_grad(set_to_none=True)
    ref_loss.backward()
    g_ref = torch.cat([p.grad.reshape(-1) for p in m._ordered_params()])
    assert abs(float(tr["loss"]) - float(ref_loss)) < 1e-6 and rel_err(g_fused.cpu(), g_ref.cpu()) < 1e-5


def test_second_fit_starts_from_scratch_and_is_reproducible(isd):
    """fit() is not a warm start -- two fits with one seed give identical parameters (the kernels of this
    path are deterministic: partial slabs, no float atomics); warm_start=True continues instead."""
    X, y = odsp.synth_trials(32, 64, 512, 256.0, seed=4)
    clf = isd.FilterbankCNNClassifier(max_epochs=3, batch_size=16, warmup_epochs=1, seed=11)
    clf.fit(X, y)
    first = clf.model_.flat_params().clone()
    n_hist = len(clf.history_)
    clf.fit(X, y)
    assert torch.equal(first, clf.model_.flat_params()) and len(clf.history_) == n_hist
    clf.set_params(warm_start=True)
    clf.fit(X, y)
    assert not torch.equal(first, clf.model_.flat_params()) and len(clf.history_) == 2 * n_hist
    from isd_amd.classifier import NotFittedError
    with pytest.raises(NotFittedError):
        isd.FilterbankCNNClassifier().predict(X)
    with pytest.raises(ValueError, match="expected 576 channels"):
        clf.trainer_.path.forward(torch.zeros(4, 64, 17, device="cuda"))             # raw trials instead of features


def test_zone_encoders_of_one_head_draw_different_dropout_masks(isd):
    """encoders called in lockstep on identically shaped inputs must not share masks."""
    import isd_amd.nn as inn
    torch.manual_seed(0)
    a = inn.EEGNet_Encoder(6, 16, dropout=0.5).cuda().train()
    b = inn.EEGNet_Encoder(6, 16, dropout=0.5).cuda().train()
    b.load_state_dict(a.state_dict())
    x = torch.randn(8, 6, 250, device="cuda")
    ya, yb = a(x).detach(), b(x).detach()
    assert a._calls == b._calls == 1 and not torch.allclose(ya, yb)


def test_device_stager_uploads_asynchronously_and_in_order(isd):
    """Pinned double-buffered H2D staging: items come back in order, on the device, equal to the host arrays -- also
    when a pinned buffer is reused while an earlier item is still being consumed."""
    from isd_amd.data import DeviceStager
    rng = np.random.default_rng(0)
    items = [(rng.standard_normal((17, 8, 500)).astype(np.float32), rng.integers(0, 5, 17).astype(np.uint8))
             for _ in range(5)]
    st = DeviceStager()
    st.put(*items[0])
    got = []
    for k in range(5):
        if k + 1 < 5:
            st.put(*items[k + 1])
        xd, yd = st.get()
        assert xd.is_cuda and yd.is_cuda and yd.dtype == torch.uint8
        got.append((xd.square().sum(), xd, yd))                 # work queued on the current stream behind the copy
    torch.cuda.synchronize()
    for (s, xd, yd), (xh, yh) in zip(got, items):
        assert np.array_equal(xd.cpu().numpy(), xh) and np.array_equal(yd.cpu().numpy(), yh)
        assert abs(float(s) - float((xh.astype(np.float64) ** 2).sum())) < 1e-2 * float((xh ** 2).sum())


def test_device_stager_never_overwrites_a_pinned_buffer_in_flight(isd):
    """in the put(k+1) / get(k) loop the queue length is 1 at every put, so the pinned slot must come from a
    put counter (not the queue length) and a slot is refilled only after the DMA out of it has completed.  Large items
    (64 MB: the copy is still in flight when the next put arrives) over six batches."""
    from isd_amd.data import DeviceStager
    n_items, shape = 6, (64, 64, 4096)
    items = [np.full(shape, float(k + 1), dtype=np.float32) for k in range(n_items)]
    for it in items:
        it[:, :, ::97] += np.arange(it[:, :, ::97].shape[-1], dtype=np.float32)
    st = DeviceStager()
    st.put(items[0])
    slots, sums = [], []
    for k in range(n_items):
        if k + 1 < n_items:
            st.put(items[k + 1])
        xd = st.get()
        sums.append((xd.double().sum(), xd[:, :, 1].clone()))
    torch.cuda.synchronize()
    assert st._puts == n_items and len({key[0][0] for key in st._pinned}) == 2       # both slots in use, alternating
    for (s, col), it in zip(sums, items):
        assert float(s) == float(it.astype(np.float64).sum())
        assert np.array_equal(col.cpu().numpy(), it[:, :, 1])


def test_fold_packing_over_worker_processes_equals_serial_run(isd, tmp_path):
    """The subject x fold trainings packed over two worker processes (scripts/train_fast.py:86-111 runs them one after
    another) give the results of the serial run: same fold accuracies, same best checkpoints."""
    from isd_amd import experiment as E
    import isd_amd.nn as inn
    rng = np.random.default_rng(1)
    electrodes = ["Fp1", "Fp2", "F3", "F4", "C3", "C4", "O1", "O2"]
    zones = {"Frontal": ["Fp1", "Fp2", "F3", "F4"], "Central": ["C3", "C4"], "Occipital": ["O1", "O2"]}
    cfg = inn.fast_config(electrodes, zones, dim_cnn=16, dim_token=16, seq_len=500, n_classes=3, num_layers=1,
                          num_heads=4, dropout=0.1)
    tv = {sid: (rng.standard_normal((12, 8, 500)).astype(np.float32), rng.integers(0, 3, 12).astype(np.uint8))
          for sid in ("01", "02")}
    a = E.finetune_per_subject_cv(tv, {}, str(tmp_path / "serial"), cfg, n_folds=2, max_epochs=2, batch_size=4, seed=1)
    b = E.finetune_per_subject_cv(tv, {}, str(tmp_path / "packed"), cfg, n_folds=2, max_epochs=2, batch_size=4, seed=1,
                                  workers=2)
    assert [r[:2] for r in a] == [r[:2] for r in b]
    for sid in ("01", "02"):
        sa = torch.load(tmp_path / "serial" / "FAST" / f"sub-{sid}" / "best_subject.pth")
        sb = torch.load(tmp_path / "packed" / "FAST" / f"sub-{sid}" / "best_subject.pth")
        assert sa.keys() == sb.keys()
        for k in sa:
            assert rel_err(sb[k].float(), sa[k].float()) < 1e-5, k


@pytest.mark.timeout(300)
def test_fold_worker_failure_reaches_the_parent_instead_of_hanging_it(isd):
    """a fold that raises inside its worker process (here: a window longer than the trials) must surface
    as an exception in run_folds, with the worker's traceback, and the other workers must be shut down."""
    from isd_amd import experiment as E
    import isd_amd.nn as inn
    rng = np.random.default_rng(2)
    electrodes = ["Fp1", "Fp2", "F3", "F4", "C3", "C4", "O1", "O2"]
    zones = {"Frontal": ["Fp1", "Fp2", "F3", "F4"], "Central": ["C3", "C4"], "Occipital": ["O1", "O2"]}
    good = inn.fast_config(electrodes, zones, dim_cnn=16, dim_token=16, seq_len=500, n_classes=3, num_layers=1,
                           num_heads=4, dropout=0.0)
    bad = inn.fast_config(electrodes, zones, dim_cnn=16, dim_token=16, seq_len=100, window_len=250, n_classes=3,
                          num_layers=1, num_heads=4, dropout=0.0)            # 250-sample windows of 100-sample trials
    X = rng.standard_normal((8, 8, 500)).astype(np.float32)
    y = rng.integers(0, 3, 8).astype(np.uint8)
    tasks = {("01", 0): (good, X[:6], y[:6], X[6:], y[6:], 1, 4, 1, "default"),
             ("01", 1): (bad, X[:6, :, :100], y[:6], X[6:, :, :100], y[6:], 1, 4, 1, "default")}
    with pytest.raises(RuntimeError, match="failed in its worker"):
        E.run_folds(tasks, workers=2)


def test_graphed_training_step_matches_eager_and_advances_dropout(isd):
    """The optimisation step captured as a HIP graph (isd_amd.graph): with dropout off, a fold trained through graph
    replays follows the eager run (same batches, same schedule; AdamW capturable vs foreach arithmetic apart); with
    dropout on, every replay draws new masks through the device-resident step counter."""
    from isd_amd import experiment as E
    from isd_amd.graph import GraphedTrainStep, graph_safe
    import isd_amd.nn as inn
    rng = np.random.default_rng(3)
    X = rng.standard_normal((40, 64, 800)).astype(np.float32)
    y = rng.integers(0, 5, 40).astype(np.uint8)
    cfg = inn.fast_config(dropout=0.0)
    runs = {}
    for graph in (False, True):
        acc, sd, hist = E.train_one_fold(cfg, X[:32], y[:32], X[32:], y[32:], 3, 12, seed=5, graph=graph)
        runs[graph] = (hist, sd)
    for he, hg in zip(runs[False][0], runs[True][0]):
        assert abs(he["loss"] - hg["loss"]) < 2e-3 * max(1.0, abs(he["loss"])), (he, hg)
    for k, v in runs[False][1].items():
        assert float((v - runs[True][1][k]).abs().max()) < 5e-3 * max(1.0, float(v.abs().max())), k
    # bf16-mixed (the reference script's default precision): the zone CNN on the bf16 matrix cores, inside the graph
    acc16, sd16, h16 = E.train_one_fold(inn.fast_config(dropout=0.0, act_dtype="bf16"), X[:32], y[:32], X[32:], y[32:], 3,
                                        12, seed=5)
    for he, hb in zip(runs[False][0], h16):
        assert abs(he["loss"] - hb["loss"]) < 5e-2, (he, hb)
    # dropout: replaying the same batch at learning rate 0 gives a different loss each time, eval mode does not
    torch.manual_seed(0)
    m = inn.FAST(inn.fast_config(dropout=0.3)).cuda().train()
    assert graph_safe(m)
    Xd, yd = torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()
    opt = torch.optim.AdamW(m.parameters(), lr=torch.tensor(0.0, device="cuda"), capturable=True)
    g = GraphedTrainStep(m, opt, Xd, yd, 16)
    before = [p.detach().clone() for p in m.parameters()]
    idx = torch.arange(16, device="cuda")
    losses = []
    for _ in range(3):
        g.loss_sum.zero_()
        g.step(idx, 0.0)
        losses.append(float(g.loss_sum) / 16)
    assert len(set(losses)) == 3 and all(np.isfinite(losses))
    assert int(m.seed_dev) == 3
    assert all(torch.equal(a, b.detach()) for a, b in zip(before, m.parameters()))       # lr 0: nothing moved
    g.step(idx[:5], 0.0)                                                                 # ragged batch: eager, same code
    assert int(m.seed_dev) == 4
    m.fuse_tail = False
    assert not graph_safe(m)                                  # per-operator blocks take their dropout seeds by value


@pytest.mark.parametrize("head", ["EEGNet_Encoder", "HeadConv_Paper_Version"])
def test_graphed_step_with_batchnorm_heads(isd, head):
    """The BatchNorm heads inside the captured step: the zones run on forked HIP streams (parallel branches of the
    graph), running statistics and num_batches_tracked update on the device, and EEGNet's dropout masks advance through
    the plan's device-resident counter.  Dropout off: the replayed fold follows the eager one."""
    from isd_amd import experiment as E
    from isd_amd.graph import GraphedTrainStep, graph_safe
    import isd_amd.nn as inn
    rng = np.random.default_rng(4)
    X = rng.standard_normal((30, 64, 800)).astype(np.float32)
    y = rng.integers(0, 5, 30).astype(np.uint8)
    cfg = inn.fast_config(dropout=0.0, head=head)
    hist, sds, accs = {}, {}, {}
    for graph in (False, True):
        accs[graph], sds[graph], hist[graph] = E.train_one_fold(_no_head_dropout(cfg), X[:24], y[:24], X[24:], y[24:], 2,
                                                                12, seed=7, graph=graph)
    for he, hg in zip(hist[False], hist[True]):
        assert abs(he["loss"] - hg["loss"]) < 5e-3 * max(1.0, abs(he["loss"])), (head, he, hg)
    # the capture's warm-up steps must not leak into the BatchNorm buffers -- the graphed fold ends with
    # the eager fold's running statistics, step count and validation accuracy
    n_buf = 0
    for k, v in sds[False].items():
        if "num_batches_tracked" in k:
            assert int(v) == int(sds[True][k]), k
            n_buf += 1
        elif "running_" in k:
            assert float((v - sds[True][k]).abs().max()) < 5e-3 * max(1.0, float(v.abs().max())), k
            n_buf += 1
    assert n_buf > 0 and accs[False] == accs[True]
    # masks: EEGNet zones draw dropout (0.25) -- replays at learning rate 0 differ, the counter advances
    if head == "EEGNet_Encoder":
        torch.manual_seed(0)
        m = inn.FAST(inn.fast_config(dropout=0.0, head=head)).cuda().train()
        assert graph_safe(m)
        Xd, yd = torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()
        opt = torch.optim.AdamW(m.parameters(), lr=torch.tensor(0.0, device="cuda"), capturable=True, fused=True)
        g = GraphedTrainStep(m, opt, Xd, yd, 16)
        idx = torch.arange(16, device="cuda")
        losses = []
        for _ in range(3):
            g.loss_sum.zero_()
            g.step(idx, 0.0)
            losses.append(float(g.loss_sum) / 16)
        assert len(set(losses)) == 3 and all(np.isfinite(losses)) and int(m.seed_dev) == 3
        nbt = next(iter(m.head.encoders.values())).temporal_conv[1].num_batches_tracked
        assert int(nbt) == 3                # BatchNorm bookkeeping advanced inside the replays, and only there


def _no_head_dropout(cfg):
    """A config whose BatchNorm zone encoders are built without dropout (the registry constructors take none)."""
    import copy
    import isd_amd.nn as inn
    cfg = copy.copy(cfg)

    class _Quiet(inn.EEGNet_Encoder):
        def __init__(self, c, f):
            super().__init__(c, f, dropout=0.0)
    if cfg.head == "EEGNet_Encoder":
        inn.register_head("EEGNet_Encoder_nodrop", _Quiet)
        cfg.head = "EEGNet_Encoder_nodrop"
    return cfg
