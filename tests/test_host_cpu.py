"""CPU-side checks of the product: host logic, C ABI surface, filter design (no GPU compute)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, load_golden
from oracle import dsp as odsp


def test_import_does_not_need_gpu_or_library():
    import isd_amd
    assert callable(isd_amd.extract_features)
    assert isd_amd.CLASSES == ["hello", "help-me", "stop", "thank-you", "yes"]
    assert sum(len(z) for z in isd_amd.zone_index_lists()) == 64
    assert sorted(i for z in isd_amd.zone_index_lists() for i in z) == list(range(64))


def test_library_builds_loads_and_exports_every_declared_symbol():
    import __graft_entry__
    __graft_entry__.build()
    from isd_amd import _lib
    h = _lib.lib()
    header = open(os.path.join(ROOT, "include", "isd_hip.h")).read()
    declared = set(re.findall(r"\b(isd_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(h, name), name
    assert h.isd_abi_version() == 1
    assert h.isd_device_count() >= 0


def test_c_abi_argument_errors_without_gpu():
    from isd_amd import _lib
    h = _lib.lib()
    p = ctypes.c_void_p()
    a12 = _lib.double_array([0.0, 1.5])                       # |a2| >= 1: unstable section
    assert h.isd_fb_plan_create(ctypes.byref(p), 1, 1, a12, _lib.double_array([1.0]), _lib.FB_AUTO) == -1
    assert b"not stable" in h.isd_last_error()
    assert h.isd_stft_plan_create(ctypes.byref(p), 512, 60, 30) == -1
    assert b"power of two" in h.isd_last_error()
    with pytest.raises(_lib.IsdError):
        _lib.check(h.isd_fb_forward(None, None, None, 1, 1, 1, None))


def test_product_path_fails_loudly_without_library(monkeypatch, tmp_path):
    from isd_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "missing.so"))
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.lib()


def test_no_product_module_imports_the_oracle():
    pkg = os.path.join(ROOT, "imagined-speech-decoding_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f


@pytest.mark.parametrize("tag,bands", [("b5", odsp.BANDS_5), ("b9", odsp.BANDS_9), ("b40", odsp.BANDS_40)])
def test_filter_design_matches_scipy_golden_tables(tag, bands):
    from isd_amd import butter_bandpass_resonators, butter_bandpass_sos
    g = load_golden("g2_sos.npz")
    fs = float(g[f"{tag}_fs"])
    for b, (_, lo, hi) in enumerate(bands):
        np.testing.assert_allclose(butter_bandpass_sos(4, lo, hi, fs), g[f"{tag}_sos"][b], rtol=1e-9, atol=1e-300)
        a12, gain = butter_bandpass_resonators(4, lo, hi, fs)
        # same poles as the scipy table
        want = np.sort_complex(np.concatenate([np.roots(s[3:]) for s in g[f"{tag}_sos"][b]]))
        got = np.sort_complex(np.concatenate([np.roots([1.0, a1, a2]) for a1, a2 in a12]))
        np.testing.assert_allclose(got, want, rtol=1e-10)
        np.testing.assert_allclose(gain, np.prod(g[f"{tag}_sos"][b][:, 0]), rtol=1e-9)


def test_band_bins_inclusive_edges():
    from isd_amd import band_bins
    assert band_bins(256.0, 64, odsp.BANDS_9)[0] == (1, 2)           # 4 and 8 Hz bins, both edges included
    assert band_bins(250.0, 64, odsp.BANDS_5) == odsp.band_bins(250.0, 64, odsp.BANDS_5)
    assert band_bins(256.0, 64, [("x", 1.0, 3.0)]) == [(1, 0)]       # empty band


def test_mat_ingestion_and_standardized_cache_roundtrip(tmp_path):
    import scipy.io
    from isd_amd import data as D
    rng = np.random.default_rng(0)
    for sub, key in (("Training set", "epo_train"), ("Validation set", "epo_validation")):
        os.makedirs(tmp_path / sub)
        for sid in ("01", "02"):
            n = 7 if key == "epo_train" else 3
            x = rng.standard_normal((795, 64, n)).astype(np.float32)             # [T, C, n] as in the .mat files
            lab = rng.integers(0, 5, n)
            onehot = np.eye(5)[lab].T                                             # [5, n]
            scipy.io.savemat(tmp_path / sub / f"Data_Sample{sid}.mat", {key: {"x": x, "y": onehot}})
    X, Y = D.load_subject_train_val(str(tmp_path), "01")
    assert X.shape == (10, 64, 800) and X.dtype == np.float32 and Y.dtype == np.uint8 and Y.shape == (10,)
    assert np.array_equal(X[:, :, 795:], np.repeat(X[:, :, 794:795], 5, axis=2))  # edge padding 795 -> 800
    Xa, Ya = D.load_training_set(str(tmp_path), ["01", "02"])
    assert Xa.shape == (14, 64, 800)
    path = D.save_standardized(str(tmp_path / "cache.npz"), {"01": (X, Y)})
    back = D.load_standardized(path)
    assert np.array_equal(back["01"][0], X) and np.array_equal(back["01"][1], Y)
    ds = D.BasicDataset(X.reshape(2, 5, 64, 800), Y)
    assert len(ds) == 10 and ds[3][0].shape == (64, 800) and ds.labels.dtype == np.uint8


def test_report_aggregation_matches_known_metrics(tmp_path):
    from isd_amd import experiment as E
    folder = tmp_path / "FAST"
    truth = np.array([0, 1, 2, 3, 4] * 10)
    for sid, flip in ((1, 0), (2, 10)):
        os.makedirs(folder / f"sub-{sid:02d}")
        pred = truth.copy()
        pred[:flip] = (pred[:flip] + 1) % 5
        E._save_predictions(str(folder / f"sub-{sid:02d}" / "test_predictions.csv"), pred, truth)
    per, summary = E.process_results(str(tmp_path))
    assert [r["Subject"] for r in per] == [1, 2]
    assert per[0]["Accuracy"] == 1.0 and abs(per[1]["Accuracy"] - 0.8) < 1e-12
    from sklearn.metrics import f1_score, precision_score
    pred2 = truth.copy()
    pred2[:10] = (pred2[:10] + 1) % 5
    assert abs(per[1]["F1"] - f1_score(truth, pred2, average="macro")) < 1e-12
    assert abs(per[1]["Precision"] - precision_score(truth, pred2, average="macro")) < 1e-12
    assert summary["N_subjects"] == 2 and abs(summary["Acc_Mean"] - 0.9) < 1e-12
    folds = E.kfold_indices(23, 5, seed=42)
    assert sorted(np.concatenate([v for _, v in folds]).tolist()) == list(range(23))
    from sklearn.model_selection import KFold
    ref = [v for _, v in KFold(5, shuffle=True, random_state=42).split(np.arange(23))]
    assert all(sorted(a.tolist()) == sorted(b.tolist()) for (_, a), b in zip(folds, ref))


def test_hot_path_rejects_operands_the_kernels_cannot_take():
    """HotPath hands raw pointers to the kernels, so dtype / layout / channel count / label checks are host
    side and raise before any launch (all reachable without a GPU)."""
    import torch
    from isd_amd.classifier import HotPath, _FeatureModel
    hp = HotPath(_FeatureModel(9 * 8, 32, 5, 4))
    good = torch.zeros(4, 72, 17)
    with pytest.raises(TypeError, match="float32"):
        hp.forward(good.double())
    with pytest.raises(ValueError, match="contiguous"):
        hp.forward(torch.zeros(4, 17, 72).transpose(1, 2))
    with pytest.raises(ValueError, match="expected 72 channels"):
        hp.forward(torch.zeros(4, 64, 17))
    with pytest.raises(ValueError, match=r"\[batch, channels, time\]"):
        hp.forward(torch.zeros(4, 72))
    with pytest.raises(TypeError, match="uint8 or int64"):
        hp.forward(good, torch.zeros(4, dtype=torch.int32))
    with pytest.raises(ValueError, match="contiguous \\[batch\\] vector"):
        hp.forward(good, torch.zeros(5, dtype=torch.int64))
    with pytest.raises(TypeError, match="no CPU path"):
        hp.forward(good, torch.zeros(4, dtype=torch.int64))


def test_estimators_follow_the_sklearn_protocol_on_the_host_side():
    """get_params / set_params / sklearn.clone, NotFittedError before fit, label range checked before any launch."""
    import isd_amd
    from isd_amd.classifier import NotFittedError
    from sklearn.base import clone
    clf = isd_amd.FilterbankCNNClassifier(max_epochs=3, batch_size=16, bands=isd_amd.BANDS_5, n_layers=2, seed=7)
    params = clf.get_params()
    assert params["max_epochs"] == 3 and params["bands"] is isd_amd.BANDS_5 and params["warm_start"] is False
    twin = clone(clf)
    assert twin is not clf and twin.get_params() == params and twin.model_ is None
    assert clf.set_params(lr=1e-3) is clf and clf.lr == 1e-3
    with pytest.raises(ValueError, match="invalid parameter"):
        clf.set_params(nonsense=1)
    X = np.zeros((6, 64, 512), np.float32)
    for est in (clf, isd_amd.FASTHeadClassifier()):
        with pytest.raises(NotFittedError):
            est.predict(X)
        with pytest.raises(NotFittedError):
            est.decision_function(X)
        with pytest.raises(ValueError, match=r"labels must lie in \[0, 5\)"):
            est.fit(X, np.array([1, 2, 3, 4, 5, 1]))          # the answer sheet's 1-based labels
        with pytest.raises(ValueError, match=r"labels must lie in"):
            est.fit(X, np.array([0, 1, 2, -100, 0, 1]))
        with pytest.raises(ValueError, match="disagree"):
            est.fit(X, np.zeros(5, np.int64))
        with pytest.raises(TypeError, match="integer class indices"):
            est.fit(X, np.zeros(6, np.float32))
    assert isinstance(NotFittedError("x"), (ValueError, AttributeError))


def test_dropout_streams_differ_per_module_instance_and_call():
    """the zone encoders of one Head run in lockstep; their masks must not coincide."""
    import torch
    import isd_amd.nn as inn
    torch.manual_seed(0)
    a, b = inn.EEGNet_Encoder(6, 32), inn.EEGNet_Encoder(6, 32)
    assert a._stream_id != b._stream_id
    seeds = {inn._dropout_seed(sid, call) for sid in (a._stream_id, b._stream_id) for call in (1, 2, 3)}
    assert len(seeds) == 6 and all(0 <= s < 2 ** 63 for s in seeds)
    head = inn.Head("EEGNet_Encoder", ["a", "b", "c", "d"], {"z0": ["a", "b"], "z1": ["c", "d"]}, 16)
    ids = [enc._stream_id for enc in head.encoders.values()]
    assert len(set(ids)) == len(ids)


def test_zone_batch_recorder_states_without_a_gpu():
    """isd_zone_batch_* (include/isd_hip.h): the recorder's own state machine needs no device -- a launch with nothing
    open fails, a second begin fails, abort closes, an empty batch launches nothing."""
    import isd_amd._lib as L
    lib = L.lib()
    assert lib.isd_zone_batch_launch(None) != 0
    assert b"no zone batch is open" in lib.isd_last_error()
    assert lib.isd_zone_batch_begin() == 0
    assert lib.isd_zone_batch_begin() != 0
    assert lib.isd_zone_batch_abort() == 0
    assert lib.isd_zone_batch_next() != 0
    assert lib.isd_zone_batch_begin() == 0
    assert lib.isd_zone_batch_next() == 0
    assert lib.isd_zone_batch_launch(None) == 0


def test_optimizers_reject_what_the_hip_kernels_cannot_take():
    """isd_amd.FusedAdamW / Trainer are HIP-only (no silent torch fallback): host tensors and inconsistent options are
    refused with a message, before any launch."""
    import torch
    import isd_amd
    p = torch.nn.Parameter(torch.zeros(8))
    with pytest.raises(TypeError, match="HIP device"):
        isd_amd.FusedAdamW([p], lr=1e-3)
    with pytest.raises(ValueError, match="capturable"):
        isd_amd.FusedAdamW([p], lr=1e-3, capturable=True)                 # needs lr as a device tensor
    with pytest.raises(TypeError, match="tensor lr"):
        isd_amd.FusedAdamW([p], lr=torch.zeros(1))                        # a host tensor is not a device-side rate
    with pytest.raises(ValueError):
        isd_amd.FusedAdamW([], lr=1e-3)
    assert issubclass(isd_amd.FusedAdamW, torch.optim.Optimizer)
    from isd_amd.classifier import _FeatureModel
    with pytest.raises((RuntimeError, TypeError, isd_amd._lib.IsdError if hasattr(isd_amd, "_lib") else RuntimeError)):
        isd_amd.Trainer(_FeatureModel(8, 32, 5, 4))                        # parameters on the host
