"""Data ingestion for the BCI Competition 2020 Track 3 imagined-speech set, in the reference's formats
(src/fast/data/preprocess.py, src/fast/data/loaders.py, scripts/preprocess.py).  Host-side I/O only.

* ``.mat`` v5 splits (``epo_train`` / ``epo_validation``: ``x`` [T, C, n], one-hot ``y`` [5, n]) through
  ``scipy.io.loadmat``  -> X float32 [n, 64, 800] (edge-padded 795 -> 800, preprocess.py:58-62), Y uint8.
* ``.mat`` v7.3 test split + Excel answer sheet (preprocess.py:118-121) need ``h5py`` / ``openpyxl``; they are
  imported lazily and a clear error is raised when the image lacks them.
* standardized cache: the reference writes HDF5 ``{SID}/X``, ``{SID}/Y`` (preprocess.py:220-223).  The same
  hierarchy is written as HDF5 when ``h5py`` is importable and as ``.npz`` (keys ``"{SID}/X"``) otherwise;
  ``load_standardized`` reads either.
"""
import os

import numpy as np

NAME = "BCIC2020Track3"
SUBJECTS = [f"{i:02d}" for i in range(1, 16)]
TARGET_TIMEPOINTS = 800


def pad_time(x, target=TARGET_TIMEPOINTS):
    """Edge-pad the last axis up to ``target`` samples (preprocess.py:62)."""
    if x.shape[-1] >= target:
        return x
    return np.pad(x, ((0, 0), (0, 0), (0, target - x.shape[-1])), "edge")


def load_mat_split(path, key):
    """One v5 ``.mat`` split -> (X float32 [n, C, 800], Y uint8 [n]); ``key`` is 'epo_train' or 'epo_validation'."""
    import scipy.io
    data = scipy.io.loadmat(path)
    x = np.asarray(data[key]["x"])[0][0]
    y = np.asarray(data[key]["y"])[0][0].argmax(0)
    x = np.transpose(x, (2, 1, 0)).astype(np.float32)
    return pad_time(x), y.astype(np.uint8)


def _collect(base_folder, sub, key, subjects):
    X, Y = [], []
    for sid in subjects:
        path = os.path.join(base_folder, sub, f"Data_Sample{sid}.mat")
        if os.path.exists(path):
            x, y = load_mat_split(path, key)
            X.append(x)
            Y.append(y)
    if not X:
        raise FileNotFoundError(f"no Data_Sample*.mat under {os.path.join(base_folder, sub)}")
    return np.concatenate(X, axis=0), np.concatenate(Y, axis=0)


def load_training_set(base_folder, subjects=SUBJECTS):
    return _collect(base_folder, "Training set", "epo_train", subjects)


def load_validation_set(base_folder, subjects=SUBJECTS):
    return _collect(base_folder, "Validation set", "epo_validation", subjects)


def load_subject_train_val(base_folder, sid):
    """Train + validation trials of one subject, concatenated (preprocess.py:164-190)."""
    parts = []
    for sub, key in (("Training set", "epo_train"), ("Validation set", "epo_validation")):
        path = os.path.join(base_folder, sub, f"Data_Sample{sid}.mat")
        if os.path.exists(path):
            parts.append(load_mat_split(path, key))
    if not parts:
        raise FileNotFoundError(f"no data for subject {sid} under {base_folder}")
    return np.concatenate([p[0] for p in parts], axis=0), np.concatenate([p[1] for p in parts], axis=0)


def load_test_set_per_subject(base_folder, excel_path, subjects=SUBJECTS):
    """{SID: (X, Y)} from the v7.3 test files and the Excel answer sheet (columns 2(i+1), rows 3:53, 1-based labels)."""
    try:
        import h5py
        import pandas as pd
    except ImportError as e:                                   # not in this image
        raise ImportError("the official test split needs h5py (MATLAB v7.3) and pandas+openpyxl (answer sheet)") from e
    labels = pd.read_excel(excel_path, header=None)
    out = {}
    for i, sid in enumerate(subjects):
        path = os.path.join(base_folder, "Test set", f"Data_Sample{sid}.mat")
        if not os.path.exists(path):
            continue
        with h5py.File(path, "r") as f:
            if "epo_test" not in f:
                continue
            x = pad_time(np.array(f["epo_test"]["x"]).astype(np.float32))
        raw = pd.to_numeric(labels.iloc[3:53, 2 * (i + 1)], errors="coerce").values
        out[sid] = (x, (raw - 1).astype(np.uint8))
    return out


def save_standardized(path, per_subject):
    """Write {SID: (X, Y)} as the reference's ``{SID}/X``, ``{SID}/Y`` hierarchy (HDF5 if possible, else .npz)."""
    if path.endswith((".h5", ".hdf5")):
        import h5py                                           # explicit request for HDF5: let the ImportError through
        with h5py.File(path, "w") as f:
            for sid, (x, y) in per_subject.items():
                f.create_dataset(f"{sid}/X", data=np.asarray(x, np.float32))
                f.create_dataset(f"{sid}/Y", data=np.asarray(y, np.uint8))
        return path
    arrays = {}
    for sid, (x, y) in per_subject.items():
        arrays[f"{sid}/X"] = np.asarray(x, np.float32)
        arrays[f"{sid}/Y"] = np.asarray(y, np.uint8)
    np.savez(path, **arrays)
    return path if path.endswith(".npz") else path + ".npz"


def load_standardized(path, subjects=None):
    """Read a standardized cache -> {SID: (X float32, Y uint8)} (loaders.py:27-45 for the HDF5 flavour)."""
    out = {}
    if path.endswith((".h5", ".hdf5")):
        import h5py
        with h5py.File(path, "r") as f:
            for sid in (subjects or list(f.keys())):
                out[sid] = (np.array(f[f"{sid}/X"], dtype=np.float32), np.array(f[f"{sid}/Y"], dtype=np.uint8))
        return out
    with np.load(path) as z:
        sids = subjects or sorted({k.split("/")[0] for k in z.files})
        for sid in sids:
            out[sid] = (z[f"{sid}/X"].astype(np.float32), z[f"{sid}/Y"].astype(np.uint8))
    return out


class BasicDataset:
    """``BasicDataset(data, label)`` of loaders.py:11-24: 4-D inputs are flattened to [n, C, T]; labels stay uint8."""

    def __init__(self, data, label):
        data = np.asarray(data, dtype=np.float32)
        if data.ndim == 4:
            data = data.reshape(-1, data.shape[-2], data.shape[-1])
        label = np.asarray(label).reshape(-1).astype(np.uint8)
        if len(data) != len(label):
            raise ValueError("data and label disagree on the number of trials")
        self.data, self.labels = data, label

    def __len__(self):
        return len(self.data)

    def __getitem__(self, idx):
        return self.data[idx], self.labels[idx]
