"""A few calls of the HDF5 C library through ctypes: enough to write and read back the openPMD files of
`hipace_amd/openpmd_writer.py` (groups, contiguous datasets of doubles / 64-bit unsigned integers, attributes of the types
openPMD-api's HDF5 backend uses).  The image has the HDF5 C library (libhdf5.so 1.10, no h5py, no openPMD-api); where the
library cannot be loaded `available()` is False and the writer keeps to its npz / JSON containers.

Datatypes on disk follow openPMD-api (IO/HDF5/HDF5IOHandler.cpp, writeAttribute): a string is a fixed-length H5T_C_S1 of
its own length, a list of strings a 1-d array of fixed-length strings (longest + 1), a bool the h5py enum {FALSE = 0,
TRUE = 1} over int8, floating-point values doubles unless the attribute's openPMD type says float, integers uint32 /
uint64 as the standard's tables name them, lists 1-d simple dataspaces.
"""
import ctypes as C
import ctypes.util
import os

import numpy as np

hid_t = C.c_int64
hsize_t = C.c_uint64
herr_t = C.c_int
_L = None
_G = {}

H5F_ACC_RDONLY, H5F_ACC_TRUNC = 0, 2
H5P_DEFAULT = 0
H5S_ALL = 0
H5S_SCALAR = 0
H5T_INTEGER, H5T_FLOAT, H5T_STRING, H5T_ENUM = 0, 1, 3, 8
H5_INDEX_NAME, H5_ITER_INC = 0, 0


def _load():
    global _L
    if _L is not None:
        return _L
    names = [os.environ.get("HPS_HDF5_LIB"), ctypes.util.find_library("hdf5"), "libhdf5.so", "/opt/conda/lib/libhdf5.so",
             "/opt/conda/lib/libhdf5.so.103", "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so"]
    for n in names:
        if not n:
            continue
        try:
            L = C.CDLL(n)
            if L.H5open() < 0:
                continue
        except (OSError, AttributeError):
            continue
        maj, mnr, rel = C.c_uint(), C.c_uint(), C.c_uint()
        L.H5get_libversion(C.byref(maj), C.byref(mnr), C.byref(rel))
        if (maj.value, mnr.value) < (1, 10):      # hid_t is 64 bits wide from 1.10 on
            continue
        _L = L
        _declare(L)
        return L
    _L = False
    return False


def available():
    return bool(_load())


def _declare(L):
    sig = {
        "H5Fcreate": (hid_t, [C.c_char_p, C.c_uint, hid_t, hid_t]), "H5Fopen": (hid_t, [C.c_char_p, C.c_uint, hid_t]),
        "H5Fclose": (herr_t, [hid_t]),
        "H5Gcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t]), "H5Gopen2": (hid_t, [hid_t, C.c_char_p, hid_t]),
        "H5Gclose": (herr_t, [hid_t]),
        "H5Oopen": (hid_t, [hid_t, C.c_char_p, hid_t]), "H5Oclose": (herr_t, [hid_t]),
        "H5Pcreate": (hid_t, [hid_t]), "H5Pclose": (herr_t, [hid_t]), "H5Pset_create_intermediate_group": (herr_t, [hid_t, C.c_uint]),
        "H5Screate": (hid_t, [C.c_int]), "H5Screate_simple": (hid_t, [C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
        "H5Sclose": (herr_t, [hid_t]), "H5Sget_simple_extent_ndims": (C.c_int, [hid_t]),
        "H5Sget_simple_extent_dims": (C.c_int, [hid_t, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
        "H5Dcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]), "H5Dopen2": (hid_t, [hid_t, C.c_char_p, hid_t]),
        "H5Dwrite": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]), "H5Dread": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
        "H5Dget_space": (hid_t, [hid_t]), "H5Dget_type": (hid_t, [hid_t]), "H5Dclose": (herr_t, [hid_t]),
        "H5Acreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t]), "H5Awrite": (herr_t, [hid_t, hid_t, C.c_void_p]),
        "H5Aread": (herr_t, [hid_t, hid_t, C.c_void_p]), "H5Aclose": (herr_t, [hid_t]),
        "H5Aopen_by_idx": (hid_t, [hid_t, C.c_char_p, C.c_int, C.c_int, hsize_t, hid_t, hid_t]),
        "H5Aget_name": (C.c_ssize_t, [hid_t, C.c_size_t, C.c_char_p]), "H5Aget_type": (hid_t, [hid_t]), "H5Aget_space": (hid_t, [hid_t]),
        "H5Aget_num_attrs": (C.c_int, [hid_t]),
        "H5Tcopy": (hid_t, [hid_t]), "H5Tset_size": (herr_t, [hid_t, C.c_size_t]), "H5Tclose": (herr_t, [hid_t]),
        "H5Tget_class": (C.c_int, [hid_t]), "H5Tget_size": (C.c_size_t, [hid_t]), "H5Tget_sign": (C.c_int, [hid_t]),
        "H5Tenum_create": (hid_t, [hid_t]), "H5Tenum_insert": (herr_t, [hid_t, C.c_char_p, C.c_void_p]),
        "H5Gget_info": (herr_t, [hid_t, C.c_void_p]),
        "H5Lget_name_by_idx": (C.c_ssize_t, [hid_t, C.c_char_p, C.c_int, C.c_int, hsize_t, C.c_char_p, C.c_size_t, hid_t]),
        "H5Iget_type": (C.c_int, [hid_t]),
        "H5Eset_auto2": (herr_t, [hid_t, C.c_void_p, C.c_void_p]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype, f.argtypes = res, args
    for g in ("H5T_NATIVE_DOUBLE_g", "H5T_NATIVE_FLOAT_g", "H5T_NATIVE_UINT64_g", "H5T_NATIVE_UINT32_g", "H5T_NATIVE_INT32_g",
              "H5T_NATIVE_INT64_g", "H5T_NATIVE_INT8_g", "H5T_C_S1_g", "H5P_CLS_LINK_CREATE_ID_g"):
        _G[g] = hid_t.in_dll(L, g).value
    L.H5Eset_auto2(0, None, None)       # errors come back as negative ids: raised below, not printed by the library


class _GInfo(C.Structure):      # H5G_info_t
    _fields_ = [("storage_type", C.c_int), ("nlinks", hsize_t), ("max_corder", C.c_int64), ("mounted", C.c_uint)]


def _chk(v, what):
    if v < 0:
        raise IOError(f"HDF5: {what} failed")
    return v


def _dims(shape):
    return (hsize_t * len(shape))(*shape)


class _Typed:
    """value -> (file / memory datatype id, dataspace id, buffer, [ids to close])"""

    @staticmethod
    def of(L, name, v, kind=None):
        close = []
        if isinstance(v, (bool, np.bool_)):
            t = _chk(L.H5Tenum_create(_G["H5T_NATIVE_INT8_g"]), "H5Tenum_create")
            for nm, val in ((b"FALSE", 0), (b"TRUE", 1)):
                b = C.c_int8(val)
                _chk(L.H5Tenum_insert(t, nm, C.byref(b)), "H5Tenum_insert")
            close.append(t)
            return t, _chk(L.H5Screate(H5S_SCALAR), "H5Screate"), C.c_int8(1 if v else 0), close
        if isinstance(v, str):
            raw = v.encode()
            t = _chk(L.H5Tcopy(_G["H5T_C_S1_g"]), "H5Tcopy")
            _chk(L.H5Tset_size(t, max(len(raw), 1)), "H5Tset_size")
            close.append(t)
            return t, _chk(L.H5Screate(H5S_SCALAR), "H5Screate"), C.create_string_buffer(raw, max(len(raw), 1)), close
        if isinstance(v, (list, tuple)) and v and isinstance(v[0], str):
            n = max(len(s.encode()) for s in v) + 1
            t = _chk(L.H5Tcopy(_G["H5T_C_S1_g"]), "H5Tcopy")
            _chk(L.H5Tset_size(t, n), "H5Tset_size")
            close.append(t)
            buf = C.create_string_buffer(n * len(v))
            for i, s in enumerate(v):
                raw = s.encode()
                buf[i * n:i * n + len(raw)] = raw
            return t, _chk(L.H5Screate_simple(1, _dims((len(v),)), None), "H5Screate_simple"), buf, close
        tmap = {"UINT": ("H5T_NATIVE_UINT32_g", np.uint32), "ULONG": ("H5T_NATIVE_UINT64_g", np.uint64), "FLOAT": ("H5T_NATIVE_FLOAT_g", np.float32),
                "DOUBLE": ("H5T_NATIVE_DOUBLE_g", np.float64), "INT": ("H5T_NATIVE_INT32_g", np.int32), "LONG": ("H5T_NATIVE_INT64_g", np.int64)}
        if isinstance(v, (list, tuple, np.ndarray)):
            k = kind or ("DOUBLE" if any(isinstance(x, (float, np.floating)) for x in v) or len(v) == 0 else "LONG")
            g, dt = tmap[k]
            a = np.ascontiguousarray(np.asarray(v, dtype=dt))
            return _G[g], _chk(L.H5Screate_simple(1, _dims((a.size,)), None), "H5Screate_simple"), a, close
        k = kind or ("DOUBLE" if isinstance(v, (float, np.floating)) else "INT")
        g, dt = tmap[k]
        a = np.asarray([v], dtype=dt)
        return _G[g], _chk(L.H5Screate(H5S_SCALAR), "H5Screate"), a, close


def _ptr(buf):
    return buf.ctypes.data_as(C.c_void_p) if isinstance(buf, np.ndarray) else C.cast(C.byref(buf) if not isinstance(buf, C.Array) else buf, C.c_void_p)


class File:
    """with File(name, "w") as f: f.group("/data/0/fields"); f.dataset("/data/0/fields/Ez", array); f.attr(path, name, value)"""

    def __init__(self, name, mode="r"):
        L = _load()
        if not L:
            raise IOError("no HDF5 library")
        self.L = L
        if mode == "w":
            self.id = _chk(L.H5Fcreate(name.encode(), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT), f"H5Fcreate({name})")
        else:
            self.id = _chk(L.H5Fopen(name.encode(), H5F_ACC_RDONLY, H5P_DEFAULT), f"H5Fopen({name})")
        self.lcpl = _chk(L.H5Pcreate(_G["H5P_CLS_LINK_CREATE_ID_g"]), "H5Pcreate")
        _chk(L.H5Pset_create_intermediate_group(self.lcpl, 1), "H5Pset_create_intermediate_group")
        self._groups = {"/"}

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def close(self):
        if self.id is not None:
            self.L.H5Pclose(self.lcpl)
            _chk(self.L.H5Fclose(self.id), "H5Fclose")
            self.id = None

    # ---- writing ----
    def group(self, path):
        path = "/" + path.strip("/")
        if path in self._groups:
            return
        o = self.L.H5Oopen(self.id, path.encode(), H5P_DEFAULT)      # (an intermediate group of a dataset created earlier)
        if o >= 0:
            self.L.H5Oclose(o)
            self._groups.add(path)
            return
        g = _chk(self.L.H5Gcreate2(self.id, path.encode(), self.lcpl, H5P_DEFAULT, H5P_DEFAULT), f"H5Gcreate2({path})")
        self.L.H5Gclose(g)
        parts = path.strip("/").split("/")
        for i in range(1, len(parts) + 1):
            self._groups.add("/" + "/".join(parts[:i]))

    def dataset(self, path, array):
        a = np.ascontiguousarray(array)
        t = {np.dtype(np.float64): "H5T_NATIVE_DOUBLE_g", np.dtype(np.uint64): "H5T_NATIVE_UINT64_g", np.dtype(np.float32): "H5T_NATIVE_FLOAT_g",
             np.dtype(np.int32): "H5T_NATIVE_INT32_g", np.dtype(np.int64): "H5T_NATIVE_INT64_g", np.dtype(np.uint32): "H5T_NATIVE_UINT32_g"}[a.dtype]
        L = self.L
        sp = _chk(L.H5Screate_simple(a.ndim, _dims(a.shape), None), "H5Screate_simple")
        d = _chk(L.H5Dcreate2(self.id, path.encode(), _G[t], sp, self.lcpl, H5P_DEFAULT, H5P_DEFAULT), f"H5Dcreate2({path})")
        if a.size:
            _chk(L.H5Dwrite(d, _G[t], H5S_ALL, H5S_ALL, H5P_DEFAULT, a.ctypes.data_as(C.c_void_p)), f"H5Dwrite({path})")
        L.H5Dclose(d)
        L.H5Sclose(sp)

    def attr(self, path, name, value, kind=None):
        L = self.L
        o = _chk(L.H5Oopen(self.id, path.encode(), H5P_DEFAULT), f"H5Oopen({path})")
        t, sp, buf, close = _Typed.of(L, name, value, kind)
        a = _chk(L.H5Acreate2(o, name.encode(), t, sp, H5P_DEFAULT, H5P_DEFAULT), f"H5Acreate2({path}@{name})")
        _chk(L.H5Awrite(a, t, _ptr(buf)), f"H5Awrite({path}@{name})")
        L.H5Aclose(a)
        L.H5Sclose(sp)
        for c in close:
            L.H5Tclose(c)
        L.H5Oclose(o)

    # ---- reading ----
    def children(self, path):
        L = self.L
        g = _chk(L.H5Gopen2(self.id, path.encode(), H5P_DEFAULT), f"H5Gopen2({path})")
        info = _GInfo()
        _chk(L.H5Gget_info(g, C.byref(info)), "H5Gget_info")
        out = []
        for i in range(info.nlinks):
            ln = L.H5Lget_name_by_idx(g, b".", H5_INDEX_NAME, H5_ITER_INC, i, None, 0, H5P_DEFAULT)
            buf = C.create_string_buffer(ln + 1)
            L.H5Lget_name_by_idx(g, b".", H5_INDEX_NAME, H5_ITER_INC, i, buf, ln + 1, H5P_DEFAULT)
            out.append(buf.value.decode())
        L.H5Gclose(g)
        return out

    def is_dataset(self, path):
        o = self.L.H5Oopen(self.id, path.encode(), H5P_DEFAULT)
        if o < 0:
            return False
        t = self.L.H5Iget_type(o)          # H5I_GROUP = 2, H5I_DATASET = 5
        self.L.H5Oclose(o)
        return t == 5

    def _np_type(self, t):
        L = self.L
        cls, size = L.H5Tget_class(t), L.H5Tget_size(t)
        if cls == H5T_FLOAT:
            return {4: np.float32, 8: np.float64}[size], {4: "H5T_NATIVE_FLOAT_g", 8: "H5T_NATIVE_DOUBLE_g"}[size]
        if cls == H5T_INTEGER:
            signed = L.H5Tget_sign(t) != 0
            return ({(4, True): np.int32, (8, True): np.int64, (4, False): np.uint32, (8, False): np.uint64, (1, True): np.int8}[(size, signed)],
                    {(4, True): "H5T_NATIVE_INT32_g", (8, True): "H5T_NATIVE_INT64_g", (4, False): "H5T_NATIVE_UINT32_g",
                     (8, False): "H5T_NATIVE_UINT64_g", (1, True): "H5T_NATIVE_INT8_g"}[(size, signed)])
        raise IOError("HDF5: unsupported datatype class %d" % cls)

    def _shape(self, sp):
        nd = self.L.H5Sget_simple_extent_ndims(sp)
        if nd <= 0:
            return ()
        d = (hsize_t * nd)()
        self.L.H5Sget_simple_extent_dims(sp, d, None)
        return tuple(int(x) for x in d)

    def read(self, path):
        L = self.L
        d = _chk(L.H5Dopen2(self.id, path.encode(), H5P_DEFAULT), f"H5Dopen2({path})")
        t, sp = L.H5Dget_type(d), L.H5Dget_space(d)
        dt, g = self._np_type(t)
        a = np.empty(self._shape(sp), dtype=dt)
        if a.size:
            _chk(L.H5Dread(d, _G[g], H5S_ALL, H5S_ALL, H5P_DEFAULT, a.ctypes.data_as(C.c_void_p)), f"H5Dread({path})")
        L.H5Tclose(t); L.H5Sclose(sp); L.H5Dclose(d)
        return a

    def attrs(self, path):
        L = self.L
        o = _chk(L.H5Oopen(self.id, path.encode(), H5P_DEFAULT), f"H5Oopen({path})")
        out = {}
        for i in range(L.H5Aget_num_attrs(o)):
            a = _chk(L.H5Aopen_by_idx(o, b".", H5_INDEX_NAME, H5_ITER_INC, i, H5P_DEFAULT, H5P_DEFAULT), "H5Aopen_by_idx")
            ln = L.H5Aget_name(a, 0, None)
            nb = C.create_string_buffer(ln + 1)
            L.H5Aget_name(a, ln + 1, nb)
            t, sp = L.H5Aget_type(a), L.H5Aget_space(a)
            shape, cls, size = self._shape(sp), L.H5Tget_class(t), L.H5Tget_size(t)
            n = int(np.prod(shape)) if shape else 1
            if cls == H5T_STRING:
                buf = C.create_string_buffer(size * n)
                _chk(L.H5Aread(a, t, buf), "H5Aread")
                vals = [buf.raw[k * size:(k + 1) * size].split(b"\0")[0].decode() for k in range(n)]
                v = vals if shape else vals[0]
            elif cls == H5T_ENUM:
                b = (C.c_int8 * n)()
                _chk(L.H5Aread(a, t, b), "H5Aread")
                v = bool(b[0])
            else:
                dt, g = self._np_type(t)
                arr = np.empty(n, dtype=dt)
                _chk(L.H5Aread(a, _G[g], arr.ctypes.data_as(C.c_void_p)), "H5Aread")
                v = arr.tolist() if shape else arr[0].item()
            out[nb.value.decode()] = v
            L.H5Tclose(t); L.H5Sclose(sp); L.H5Aclose(a)
        L.H5Oclose(o)
        return out

    def walk(self, path="/"):
        """every object below `path`: yields (path, is_dataset)"""
        for c in self.children(path):
            p = path.rstrip("/") + "/" + c
            ds = self.is_dataset(p)
            yield p, ds
            if not ds:
                yield from self.walk(p)
