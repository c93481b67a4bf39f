"""Data ingestion for the BCI Competition 2020 Track 3 imagined-speech set, in the reference's formats
(src/fast/data/preprocess.py, src/fast/data/loaders.py, scripts/preprocess.py).  Host-side I/O only.

* ``.mat`` v5 splits (``epo_train`` / ``epo_validation``: ``x`` [T, C, n], one-hot ``y`` [5, n]) through
  ``scipy.io.loadmat``  -> X float32 [n, 64, 800] (edge-padded 795 -> 800, preprocess.py:58-62), Y uint8.
* ``.mat`` v7.3 test split (HDF5) + Excel answer sheet (preprocess.py:104-121): through ``h5py`` and
  pandas + ``openpyxl`` when they are installed, otherwise through ``isd_amd.h5lite`` (ctypes over the HDF5 C
  library) and ``isd_amd.xlsx`` (zip + XML from the standard library).
* standardized cache: the reference writes HDF5 ``{SID}/X``, ``{SID}/Y`` (preprocess.py:220-223) and a gzip
  flavour with file attributes (scripts/preprocess.py:83-99).  ``save_standardized`` / ``save_splits`` write the
  same hierarchies (HDF5 for ``.h5`` / ``.hdf5`` paths, ``.npz`` with keys ``"{SID}/X"`` otherwise);
  ``load_standardized`` / ``load_splits`` read them.
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


def _h5():
    """``h5py`` when installed, else the ctypes binding of the HDF5 C library (same call surface for this module)."""
    try:
        import h5py
        return h5py
    except ImportError:
        from . import h5lite
        if not h5lite.available():
            raise ImportError("HDF5 files need h5py or a loadable libhdf5 (set ISD_HDF5_LIB)") from None
        return h5lite


def read_answer_sheet(excel_path):
    """``col -> float[50]``: the numeric cells ``iloc[3:53, col]`` of the competition's answer sheet."""
    try:
        import openpyxl  # noqa: F401
        import pandas as pd
        frame = pd.read_excel(excel_path, header=None)
        return lambda col: pd.to_numeric(frame.iloc[3:53, col], errors="coerce").values
    except ImportError:
        from .xlsx import numeric_column, read_sheet
        grid = read_sheet(excel_path)
        return lambda col: numeric_column(grid, col, 3, 53)


def load_test_set_per_subject(base_folder, excel_path, subjects=SUBJECTS):
    """{SID: (X, Y)} from the v7.3 test files and the Excel answer sheet (columns 2(i+1), rows 3:53, 1-based labels)."""
    h5py = _h5()
    column = read_answer_sheet(excel_path)
    out = {}
    for i, sid in enumerate(subjects):
        path = os.path.join(base_folder, "Test set", f"Data_Sample{sid}.mat")
        if not os.path.exists(path):
            continue
        with h5py.File(path, "r") as f:
            if "epo_test" not in f:
                continue
            x = pad_time(np.array(f["epo_test"]["x"]).astype(np.float32))
        raw = column(2 * (i + 1))
        out[sid] = (x, (raw - 1).astype(np.uint8))
    return out


def load_test_set(base_folder, excel_path, subjects=SUBJECTS):
    """All test subjects concatenated -> (X, Y) (preprocess.py:96-129)."""
    per = load_test_set_per_subject(base_folder, excel_path, subjects)
    if not per:
        raise FileNotFoundError(f"no test data under {os.path.join(base_folder, 'Test set')}")
    return (np.concatenate([per[s][0] for s in per], axis=0), np.concatenate([per[s][1] for s in per], axis=0))


SPLIT_ATTRS = ("n_subjects", "n_classes", "classes", "electrodes", "sfreq")


def save_splits(path, splits, attrs=None):
    """scripts/preprocess.py:83-99: ``X_train``/``Y_train``/``X_valid``/... gzip datasets + file attributes."""
    from .constants import CLASSES, ELECTRODES
    meta = {"n_subjects": len(SUBJECTS), "n_classes": len(CLASSES), "classes": str(list(CLASSES)),
            "electrodes": str(list(ELECTRODES)), "sfreq": 250}
    meta.update(attrs or {})
    with _h5().File(path, "w") as f:
        for name, (x, y) in splits.items():
            f.create_dataset(f"X_{name}", data=np.asarray(x, np.float32), compression="gzip")
            f.create_dataset(f"Y_{name}", data=np.asarray(y, np.uint8), compression="gzip")
        for k, v in meta.items():
            f.attrs[k] = v
    return path


def load_splits(path):
    """-> ({'train': (X, Y), ...}, attrs dict) from a ``save_splits`` / scripts/preprocess.py file."""
    out, meta = {}, {}
    with _h5().File(path, "r") as f:
        for key in f.keys():
            if key.startswith("X_") and f"Y_{key[2:]}" in f:
                out[key[2:]] = (np.array(f[key], dtype=np.float32), np.array(f[f"Y_{key[2:]}"], dtype=np.uint8))
        for k in f.attrs.keys():
            meta[k] = f.attrs[k]
    return out, meta


def save_standardized(path, per_subject):
    """Write {SID: (X, Y)} as the reference's ``{SID}/X``, ``{SID}/Y`` hierarchy (HDF5 if possible, else .npz)."""
    if path.endswith((".h5", ".hdf5")):
        with _h5().File(path, "w") as f:
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
        with _h5().File(path, "r") as f:
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


class DeviceStager:
    """Asynchronous host -> device staging (the ``pin_memory=True`` DataLoader + Lightning batch transfer of
    scripts/train_fast.py:104-111,127-140): arrays are copied into pinned host buffers and uploaded with
    ``non_blocking`` copies on a dedicated HIP stream, so the upload of the NEXT fold / subject / batch runs under the
    kernels of the current one.  ``put`` returns at once; ``get`` makes the CURRENT stream wait (stream-side, no host
    sync) for the copy that was started ``depth`` puts ago.

        st = DeviceStager()
        st.put(X0, y0)
        for k in range(n):
            if k + 1 < n: st.put(X[k + 1], y[k + 1])      # upload k+1 ...
            xd, yd = st.get()                             # ... while batch k computes
    """

    def __init__(self, device=None):
        import torch
        self.torch = torch
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.stream = torch.cuda.Stream(self.device)
        self._queue = []
        self._pinned = {}                                  # (slot, shape, dtype) -> reusable pinned buffer
        self._puts = 0                                     # monotonic: the slot alternates whatever the queue holds
        self._slot_done = [None, None]                     # the event behind the last upload out of each slot

    def _pin(self, slot, arr):
        torch = self.torch
        t = torch.as_tensor(np.ascontiguousarray(arr))
        key = (slot, tuple(t.shape), t.dtype)
        buf = self._pinned.get(key)
        if buf is None:
            buf = self._pinned[key] = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        buf.copy_(t)
        return buf

    def put(self, *arrays):
        """Start the upload of one item (any number of arrays: trials, labels)."""
        torch = self.torch
        slot = self._puts % 2                              # two pinned buffers per array position: double buffering
        self._puts += 1
        if self._slot_done[slot] is not None:
            self._slot_done[slot].synchronize()            # the DMA out of this slot's pinned buffers has finished
        outs = []
        with torch.cuda.stream(self.stream):
            for i, a in enumerate(arrays):
                host = self._pin((slot, i), a)
                outs.append(host.to(self.device, non_blocking=True))
            done = torch.cuda.Event()
            done.record(self.stream)
        self._slot_done[slot] = done
        self._queue.append((outs, done))

    def get(self):
        """Device tensors of the oldest pending item; the current stream is made to wait for its copy."""
        outs, done = self._queue.pop(0)
        self.torch.cuda.current_stream(self.device).wait_event(done)
        for t in outs:
            t.record_stream(self.torch.cuda.current_stream(self.device))
        return outs if len(outs) > 1 else outs[0]
