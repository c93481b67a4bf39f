"""Minimal HDF5 access over the C library (``libhdf5`` through ctypes) for images without ``h5py``.

The reference keeps its data in HDF5: the standardized cache ``{SID}/X`` f32 [n, 64, 800], ``{SID}/Y`` uint8
(src/fast/data/preprocess.py:220-223, read back by src/fast/data/loaders.py:27-45), the gzip flavour with file
attributes (scripts/preprocess.py:83-99) and the MATLAB v7.3 test split (src/fast/data/preprocess.py:108-112).
This module offers the small part of the ``h5py`` surface those call sites use -- ``File(path, mode)`` as a
context manager, ``in``, ``keys()``, ``f["a/b"]``, ``np.array(dataset)``, ``create_dataset(name, data=...,
compression="gzip")`` and ``attrs`` -- on top of the HDF5 C API.  ``isd_amd.data`` prefers ``h5py`` and falls
back to this when the import fails.  Host-side I/O only; numeric (integer / float) datasets and int, float and
str attributes.
"""
import ctypes as C
import ctypes.util
import os

import numpy as np

_hid, _herr, _hsize = C.c_int64, C.c_int, C.c_uint64
_LIB = None
_CANDIDATES = ("libhdf5.so", "libhdf5.so.103", "libhdf5.so.200", "libhdf5.so.310", "libhdf5_serial.so",
               "/opt/conda/lib/libhdf5.so", "/opt/conda/lib/libhdf5.so.103")

H5F_ACC_RDONLY, H5F_ACC_RDWR, H5F_ACC_TRUNC = 0, 1, 2
H5T_INTEGER, H5T_FLOAT, H5T_STRING = 0, 1, 3
H5S_SCALAR = 0
H5T_VARIABLE = C.c_size_t(-1).value
_NATIVE = {"int8": "INT8", "uint8": "UINT8", "int16": "INT16", "uint16": "UINT16", "int32": "INT32",
           "uint32": "UINT32", "int64": "INT64", "uint64": "UINT64", "float32": "FLOAT", "float64": "DOUBLE"}


class _G(C.Structure):          # H5G_info_t
    _fields_ = [("storage_type", C.c_int), ("nlinks", _hsize), ("max_corder", C.c_int64), ("mounted", C.c_uint)]


def available():
    try:
        _lib()
        return True
    except OSError:
        return False


def _lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    names = [os.environ["ISD_HDF5_LIB"]] if os.environ.get("ISD_HDF5_LIB") else []
    found = ctypes.util.find_library("hdf5")
    names += ([found] if found else []) + list(_CANDIDATES)
    err = None
    for n in names:
        try:
            lib = C.CDLL(n)
            break
        except OSError as e:
            err = e
    else:
        raise OSError(f"libhdf5 not found (set ISD_HDF5_LIB to its path, or install h5py): {err}")
    sig = {
        "H5open": (_herr, []), "H5Eset_auto2": (_herr, [_hid, C.c_void_p, C.c_void_p]),
        "H5Fopen": (_hid, [C.c_char_p, C.c_uint, _hid]), "H5Fcreate": (_hid, [C.c_char_p, C.c_uint, _hid, _hid]),
        "H5Fclose": (_herr, [_hid]),
        "H5Gopen2": (_hid, [_hid, C.c_char_p, _hid]), "H5Gclose": (_herr, [_hid]),
        "H5Gcreate2": (_hid, [_hid, C.c_char_p, _hid, _hid, _hid]),
        "H5Gget_info": (_herr, [_hid, C.POINTER(_G)]),
        "H5Lexists": (C.c_int, [_hid, C.c_char_p, _hid]),
        "H5Lget_name_by_idx": (C.c_ssize_t, [_hid, C.c_char_p, C.c_int, C.c_int, _hsize, C.c_char_p, C.c_size_t, _hid]),
        "H5Dopen2": (_hid, [_hid, C.c_char_p, _hid]), "H5Dclose": (_herr, [_hid]),
        "H5Dget_space": (_hid, [_hid]), "H5Dget_type": (_hid, [_hid]),
        "H5Dread": (_herr, [_hid, _hid, _hid, _hid, _hid, C.c_void_p]),
        "H5Dwrite": (_herr, [_hid, _hid, _hid, _hid, _hid, C.c_void_p]),
        "H5Dcreate2": (_hid, [_hid, C.c_char_p, _hid, _hid, _hid, _hid, _hid]),
        "H5Sget_simple_extent_ndims": (C.c_int, [_hid]),
        "H5Sget_simple_extent_dims": (C.c_int, [_hid, C.POINTER(_hsize), C.POINTER(_hsize)]),
        "H5Screate_simple": (_hid, [C.c_int, C.POINTER(_hsize), C.POINTER(_hsize)]),
        "H5Screate": (_hid, [C.c_int]), "H5Sclose": (_herr, [_hid]),
        "H5Tget_class": (C.c_int, [_hid]), "H5Tget_size": (C.c_size_t, [_hid]), "H5Tget_sign": (C.c_int, [_hid]),
        "H5Tis_variable_str": (C.c_int, [_hid]), "H5Tcopy": (_hid, [_hid]), "H5Tset_size": (_herr, [_hid, C.c_size_t]),
        "H5Tset_cset": (_herr, [_hid, C.c_int]), "H5Tclose": (_herr, [_hid]),
        "H5Pcreate": (_hid, [_hid]), "H5Pclose": (_herr, [_hid]),
        "H5Pset_chunk": (_herr, [_hid, C.c_int, C.POINTER(_hsize)]), "H5Pset_deflate": (_herr, [_hid, C.c_uint]),
        "H5Pset_shuffle": (_herr, [_hid]),
        "H5Pset_create_intermediate_group": (_herr, [_hid, C.c_uint]),
        "H5Aexists": (C.c_int, [_hid, C.c_char_p]), "H5Aopen": (_hid, [_hid, C.c_char_p, _hid]),
        "H5Acreate2": (_hid, [_hid, C.c_char_p, _hid, _hid, _hid, _hid]), "H5Aclose": (_herr, [_hid]),
        "H5Aget_type": (_hid, [_hid]), "H5Aget_space": (_hid, [_hid]),
        "H5Aread": (_herr, [_hid, _hid, C.c_void_p]), "H5Awrite": (_herr, [_hid, _hid, C.c_void_p]),
        "H5Adelete": (_herr, [_hid, C.c_char_p]),
        "H5Aget_num_attrs": (C.c_int, [_hid]),
        "H5Aget_name_by_idx": (C.c_ssize_t, [_hid, C.c_char_p, C.c_int, C.c_int, _hsize, C.c_char_p, C.c_size_t, _hid]),
        "H5free_memory": (_herr, [C.c_void_p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    lib.H5open()
    lib.H5Eset_auto2(0, None, None)                     # errors are reported through return codes -> exceptions
    _LIB = lib
    return lib


def _glob(name):
    return _hid.in_dll(_lib(), name).value


def _native(dtype):
    key = np.dtype(dtype).name
    if key not in _NATIVE:
        raise TypeError(f"h5lite handles integer / float data, not {dtype}")
    return _glob(f"H5T_NATIVE_{_NATIVE[key]}_g")


def _check(rc, what):
    if rc < 0:
        raise OSError(f"HDF5: {what} failed")
    return rc


def _file_dtype(tid):
    L = _lib()
    cls, size = L.H5Tget_class(tid), L.H5Tget_size(tid)
    if cls == H5T_FLOAT and size in (4, 8):
        return np.dtype(f"f{size}")
    if cls == H5T_INTEGER and size in (1, 2, 4, 8):
        return np.dtype(("i" if L.H5Tget_sign(tid) else "u") + str(size))
    raise TypeError(f"unsupported HDF5 datatype (class {cls}, {size} bytes)")


class Attrs:
    def __init__(self, oid):
        self._oid = oid

    def __contains__(self, name):
        return _lib().H5Aexists(self._oid, name.encode()) > 0

    def keys(self):
        L, out = _lib(), []
        for i in range(max(L.H5Aget_num_attrs(self._oid), 0)):
            n = L.H5Aget_name_by_idx(self._oid, b".", 0, 0, i, None, 0, 0)
            buf = C.create_string_buffer(n + 1)
            L.H5Aget_name_by_idx(self._oid, b".", 0, 0, i, buf, n + 1, 0)
            out.append(buf.value.decode())
        return out

    def __getitem__(self, name):
        L = _lib()
        aid = L.H5Aopen(self._oid, name.encode(), 0)
        if aid < 0:
            raise KeyError(name)
        tid, sid = L.H5Aget_type(aid), L.H5Aget_space(aid)
        try:
            if L.H5Tget_class(tid) == H5T_STRING:
                if L.H5Tis_variable_str(tid) > 0:
                    p = C.c_char_p()
                    _check(L.H5Aread(aid, tid, C.byref(p)), "H5Aread")
                    val = (p.value or b"").decode("utf-8")
                    L.H5free_memory(p)
                    return val
                buf = C.create_string_buffer(L.H5Tget_size(tid) + 1)
                _check(L.H5Aread(aid, tid, buf), "H5Aread")
                return buf.value.decode("utf-8")
            dt = _file_dtype(tid)
            nd = L.H5Sget_simple_extent_ndims(sid)
            dims = (_hsize * max(nd, 1))()
            if nd > 0:
                L.H5Sget_simple_extent_dims(sid, dims, None)
            out = np.empty(tuple(dims[:nd]), dtype=dt)
            _check(L.H5Aread(aid, _native(dt), out.ctypes.data_as(C.c_void_p)), "H5Aread")
            return out[()] if nd == 0 else out
        finally:
            L.H5Tclose(tid); L.H5Sclose(sid); L.H5Aclose(aid)

    def __setitem__(self, name, value):
        L = _lib()
        if L.H5Aexists(self._oid, name.encode()) > 0:
            L.H5Adelete(self._oid, name.encode())
        if isinstance(value, (str, bytes)):
            raw = value.encode("utf-8") if isinstance(value, str) else value
            tid = L.H5Tcopy(_glob("H5T_C_S1_g"))
            L.H5Tset_size(tid, H5T_VARIABLE)               # variable-length UTF-8, what h5py writes for a str
            L.H5Tset_cset(tid, 1)
            sid = L.H5Screate(H5S_SCALAR)
            aid = _check(L.H5Acreate2(self._oid, name.encode(), tid, sid, 0, 0), "H5Acreate2")
            p = C.c_char_p(raw)
            _check(L.H5Awrite(aid, tid, C.byref(p)), "H5Awrite")
            L.H5Aclose(aid); L.H5Sclose(sid); L.H5Tclose(tid)
            return
        arr = np.asarray(value)
        if arr.ndim and not arr.flags.c_contiguous:
            arr = np.ascontiguousarray(arr)
        if arr.dtype == np.bool_:
            arr = arr.astype(np.uint8)
        if arr.dtype.kind == "i" and arr.dtype.itemsize != 8 and not isinstance(value, np.generic):
            arr = arr.astype(np.int64)
        tid = _native(arr.dtype)
        if arr.ndim == 0:
            sid = L.H5Screate(H5S_SCALAR)
        else:
            sid = L.H5Screate_simple(arr.ndim, (_hsize * arr.ndim)(*arr.shape), None)
        aid = _check(L.H5Acreate2(self._oid, name.encode(), tid, sid, 0, 0), "H5Acreate2")
        _check(L.H5Awrite(aid, tid, arr.ctypes.data_as(C.c_void_p)), "H5Awrite")
        L.H5Aclose(aid); L.H5Sclose(sid)


class Dataset:
    def __init__(self, did, name):
        self._did, self.name = did, name
        L = _lib()
        sid, tid = L.H5Dget_space(did), L.H5Dget_type(did)
        try:
            nd = L.H5Sget_simple_extent_ndims(sid)
            dims = (_hsize * max(nd, 1))()
            if nd > 0:
                L.H5Sget_simple_extent_dims(sid, dims, None)
            self.shape = tuple(int(d) for d in dims[:nd])
            self.dtype = _file_dtype(tid)
        finally:
            L.H5Sclose(sid); L.H5Tclose(tid)
        self.attrs = Attrs(did)

    @property
    def ndim(self):
        return len(self.shape)

    def read(self):
        out = np.empty(self.shape, dtype=self.dtype)
        if out.size:
            _check(_lib().H5Dread(self._did, _native(self.dtype), 0, 0, 0, out.ctypes.data_as(C.c_void_p)), "H5Dread")
        return out

    def __array__(self, dtype=None, copy=None):
        a = self.read()
        return a if dtype is None else a.astype(dtype, copy=False)

    def __getitem__(self, key):
        return self.read()[key]

    def __len__(self):
        return self.shape[0]

    def _close(self):
        if self._did:
            _lib().H5Dclose(self._did)
            self._did = 0


class Group:
    def __init__(self, gid, name, root):
        self._gid, self.name, self._root = gid, name, root
        self.attrs = Attrs(gid)

    def __contains__(self, name):
        L, loc = _lib(), self._gid
        parts = [p for p in name.split("/") if p]
        path = ""
        for p in parts:                                    # H5Lexists wants every intermediate link to exist
            path = f"{path}/{p}" if path else p
            if L.H5Lexists(loc, path.encode(), 0) <= 0:
                return False
        return bool(parts)

    def keys(self):
        L, info = _lib(), _G()
        _check(L.H5Gget_info(self._gid, C.byref(info)), "H5Gget_info")
        out = []
        for i in range(info.nlinks):
            n = L.H5Lget_name_by_idx(self._gid, b".", 0, 0, i, None, 0, 0)
            buf = C.create_string_buffer(n + 1)
            L.H5Lget_name_by_idx(self._gid, b".", 0, 0, i, buf, n + 1, 0)
            out.append(buf.value.decode())
        return out

    def __iter__(self):
        return iter(self.keys())

    def __getitem__(self, name):
        if name not in self:
            raise KeyError(f"{name!r} is not in {self.name!r}")
        L = _lib()
        did = L.H5Dopen2(self._gid, name.encode(), 0)
        full = f"{self.name.rstrip('/')}/{name}"
        if did >= 0:
            ds = Dataset(did, full)
            self._root._open.append(ds)
            return ds
        gid = L.H5Gopen2(self._gid, name.encode(), 0)
        if gid < 0:
            raise KeyError(f"{name!r}: neither a dataset nor a group")
        g = Group(gid, full, self._root)
        self._root._open.append(g)
        return g

    def create_group(self, name):
        L = _lib()
        lcpl = L.H5Pcreate(_glob("H5P_CLS_LINK_CREATE_ID_g"))
        L.H5Pset_create_intermediate_group(lcpl, 1)
        gid = _check(L.H5Gcreate2(self._gid, name.encode(), lcpl, 0, 0), f"create_group({name!r})")
        L.H5Pclose(lcpl)
        g = Group(gid, f"{self.name.rstrip('/')}/{name}", self._root)
        self._root._open.append(g)
        return g

    def create_dataset(self, name, data, compression=None, compression_opts=None, shuffle=False):
        L = _lib()
        arr = np.asarray(data)
        if arr.ndim and not arr.flags.c_contiguous:
            arr = np.ascontiguousarray(arr)
        tid = _native(arr.dtype)
        if compression not in (None, "gzip"):
            raise ValueError("compression must be None or 'gzip'")
        if arr.ndim == 0:
            sid = L.H5Screate(H5S_SCALAR)
        else:
            sid = L.H5Screate_simple(arr.ndim, (_hsize * arr.ndim)(*arr.shape), None)
        lcpl = L.H5Pcreate(_glob("H5P_CLS_LINK_CREATE_ID_g"))
        L.H5Pset_create_intermediate_group(lcpl, 1)
        dcpl = L.H5Pcreate(_glob("H5P_CLS_DATASET_CREATE_ID_g"))
        if compression == "gzip" and arr.ndim and arr.size:
            chunk = list(arr.shape)                         # ~1 MiB chunks, split along the leading axes
            i = 0
            while int(np.prod(chunk)) * arr.itemsize > (1 << 20) and i < arr.ndim:
                per = int(np.prod(chunk[i + 1:])) * arr.itemsize
                chunk[i] = max(1, (1 << 20) // max(per, 1)) if per <= (1 << 20) else 1
                i += 1
            L.H5Pset_chunk(dcpl, arr.ndim, (_hsize * arr.ndim)(*chunk))
            if shuffle:
                L.H5Pset_shuffle(dcpl)
            L.H5Pset_deflate(dcpl, 4 if compression_opts is None else int(compression_opts))
        did = L.H5Dcreate2(self._gid, name.encode(), tid, sid, lcpl, dcpl, 0)
        L.H5Pclose(lcpl); L.H5Pclose(dcpl)
        if did < 0:
            L.H5Sclose(sid)
            raise OSError(f"HDF5: create_dataset({name!r}) failed (does it exist already?)")
        if arr.size:
            rc = L.H5Dwrite(did, tid, 0, 0, 0, arr.ctypes.data_as(C.c_void_p))
            if rc < 0:
                L.H5Dclose(did); L.H5Sclose(sid)
                raise OSError(f"HDF5: writing {name!r} failed")
        L.H5Sclose(sid)
        ds = Dataset(did, f"{self.name.rstrip('/')}/{name}")
        self._root._open.append(ds)
        return ds

    def _close(self):
        if self._gid:
            _lib().H5Gclose(self._gid)
            self._gid = 0


class File(Group):
    """``File(path, "r" | "w" | "r+")``; use as a context manager."""

    def __init__(self, path, mode="r"):
        L = _lib()
        p = os.fsencode(path)
        if mode == "r":
            fid = L.H5Fopen(p, H5F_ACC_RDONLY, 0)
        elif mode == "r+":
            fid = L.H5Fopen(p, H5F_ACC_RDWR, 0)
        elif mode == "w":
            fid = L.H5Fcreate(p, H5F_ACC_TRUNC, 0, 0)
        else:
            raise ValueError("mode must be 'r', 'r+' or 'w'")
        if fid < 0:
            raise OSError(f"HDF5: cannot open {path!r} (mode {mode!r})")
        self._fid, self._open = fid, []
        gid = _check(L.H5Gopen2(fid, b"/", 0), "open root group")
        super().__init__(gid, "/", self)

    def close(self):
        for o in reversed(self._open):
            o._close()
        self._open = []
        self._close()
        if self._fid:
            _lib().H5Fclose(self._fid)
            self._fid = 0

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
