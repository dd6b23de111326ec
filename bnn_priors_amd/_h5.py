"""Minimal ctypes binding of the HDF5 C library (libhdf5 1.10+), just enough for the
sample / metrics store (storage.py): extendible chunked datasets with Fletcher-32
checksums and fill values, hyperslab writes and reads, SWMR writer / reader modes.

h5py -- what the reference uses (bnn_priors/exp_utils.py:409-551) -- is not part of
this image, the C library it wraps is (``/opt/conda/lib/libhdf5.so``), so the files
written here are ordinary HDF5 that h5py / h5dump / the reference's ``eval_bnn.py``
read.  Set ``BNN_PRIORS_HDF5_LIB`` to point at another copy of the library.
"""
import ctypes as C
import ctypes.util
import os

import numpy as np

hid_t = C.c_int64
hsize_t = C.c_uint64
herr_t = C.c_int

ACC_RDONLY, ACC_RDWR, ACC_TRUNC, ACC_EXCL = 0x0, 0x1, 0x2, 0x4
ACC_SWMR_WRITE, ACC_SWMR_READ = 0x20, 0x40
UNLIMITED = 2 ** 64 - 1
_LIBVER_LATEST = 2          # H5F_LIBVER_V110: what h5py's libver="latest" selects on HDF5 1.10
_SCOPE_GLOBAL = 1
_SELECT_SET = 0
_CLASS_INTEGER, _CLASS_FLOAT = 0, 1

_lib = None


class HDF5Error(OSError):
    pass


def _candidates():
    env = os.environ.get("BNN_PRIORS_HDF5_LIB")
    if env:
        yield env
    found = ctypes.util.find_library("hdf5")
    if found:
        yield found
    for d in ("/opt/conda/lib", "/usr/lib/x86_64-linux-gnu", "/usr/lib/x86_64-linux-gnu/hdf5/serial",
              "/usr/local/lib", "/usr/lib64"):
        for n in ("libhdf5.so", "libhdf5_serial.so"):
            yield os.path.join(d, n)


def available():
    try:
        lib()
        return True
    except HDF5Error:
        return False


def lib():
    "the loaded library, with argument types declared (raises HDF5Error if there is none)"
    global _lib
    if _lib is not None:
        return _lib
    last = None
    for path in _candidates():
        try:
            L = C.CDLL(path)
            break
        except OSError as e:
            last = e
    else:
        raise HDF5Error(f"libhdf5 not found (set BNN_PRIORS_HDF5_LIB): {last}")

    def sig(name, res, *args):
        f = getattr(L, name)
        f.restype, f.argtypes = res, list(args)

    P = C.POINTER
    sig("H5open", herr_t)
    sig("H5get_libversion", herr_t, P(C.c_uint), P(C.c_uint), P(C.c_uint))
    sig("H5Eset_auto2", herr_t, hid_t, C.c_void_p, C.c_void_p)
    sig("H5Fcreate", hid_t, C.c_char_p, C.c_uint, hid_t, hid_t)
    sig("H5Fopen", hid_t, C.c_char_p, C.c_uint, hid_t)
    sig("H5Fclose", herr_t, hid_t)
    sig("H5Fflush", herr_t, hid_t, C.c_int)
    sig("H5Fstart_swmr_write", herr_t, hid_t)
    sig("H5Pcreate", hid_t, hid_t)
    sig("H5Pclose", herr_t, hid_t)
    sig("H5Pset_libver_bounds", herr_t, hid_t, C.c_int, C.c_int)
    sig("H5Pset_cache", herr_t, hid_t, C.c_int, C.c_size_t, C.c_size_t, C.c_double)
    sig("H5Pset_chunk", herr_t, hid_t, C.c_int, P(hsize_t))
    sig("H5Pget_chunk", C.c_int, hid_t, C.c_int, P(hsize_t))
    sig("H5Pset_fletcher32", herr_t, hid_t)
    sig("H5Pget_nfilters", C.c_int, hid_t)
    sig("H5Pget_filter2", C.c_int, hid_t, C.c_uint, P(C.c_uint), P(C.c_size_t), P(C.c_uint), C.c_size_t,
        C.c_char_p, P(C.c_uint))
    sig("H5Pset_fill_value", herr_t, hid_t, hid_t, C.c_void_p)
    sig("H5Pset_create_intermediate_group", herr_t, hid_t, C.c_uint)
    sig("H5Screate_simple", hid_t, C.c_int, P(hsize_t), P(hsize_t))
    sig("H5Sclose", herr_t, hid_t)
    sig("H5Sselect_hyperslab", herr_t, hid_t, C.c_int, P(hsize_t), P(hsize_t), P(hsize_t), P(hsize_t))
    sig("H5Sget_simple_extent_ndims", C.c_int, hid_t)
    sig("H5Sget_simple_extent_dims", C.c_int, hid_t, P(hsize_t), P(hsize_t))
    sig("H5Dcreate2", hid_t, hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t)
    sig("H5Dopen2", hid_t, hid_t, C.c_char_p, hid_t)
    sig("H5Dclose", herr_t, hid_t)
    sig("H5Dset_extent", herr_t, hid_t, P(hsize_t))
    sig("H5Dget_space", hid_t, hid_t)
    sig("H5Dget_type", hid_t, hid_t)
    sig("H5Dget_create_plist", hid_t, hid_t)
    sig("H5Dwrite", herr_t, hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p)
    sig("H5Dread", herr_t, hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p)
    sig("H5Drefresh", herr_t, hid_t)
    sig("H5Dflush", herr_t, hid_t)
    sig("H5Tget_class", C.c_int, hid_t)
    sig("H5Tget_size", C.c_size_t, hid_t)
    sig("H5Tclose", herr_t, hid_t)
    sig("H5Gopen2", hid_t, hid_t, C.c_char_p, hid_t)
    sig("H5Gclose", herr_t, hid_t)
    sig("H5Gget_info", herr_t, hid_t, C.c_void_p)
    sig("H5Lget_name_by_idx", C.c_ssize_t, hid_t, C.c_char_p, C.c_int, C.c_int, hsize_t, C.c_char_p,
        C.c_size_t, hid_t)
    if L.H5open() < 0:
        raise HDF5Error("H5open failed")
    L.H5Eset_auto2(0, None, None)       # errors are reported through return codes -> HDF5Error
    maj, mnr, rel = C.c_uint(), C.c_uint(), C.c_uint()
    L.H5get_libversion(maj, mnr, rel)
    if (maj.value, mnr.value) < (1, 10):
        raise HDF5Error(f"libhdf5 {maj.value}.{mnr.value}.{rel.value} is older than 1.10 (no SWMR)")
    L.version = (maj.value, mnr.value, rel.value)
    g = lambda name: hid_t.in_dll(L, name).value
    L.types = {np.dtype(np.float32): (g("H5T_IEEE_F32LE_g"), g("H5T_NATIVE_FLOAT_g")),
               np.dtype(np.float64): (g("H5T_IEEE_F64LE_g"), g("H5T_NATIVE_DOUBLE_g")),
               np.dtype(np.int64): (g("H5T_STD_I64LE_g"), g("H5T_NATIVE_INT64_g"))}
    L.FAPL, L.DCPL, L.LCPL = (g("H5P_CLS_FILE_ACCESS_ID_g"), g("H5P_CLS_DATASET_CREATE_ID_g"),
                              g("H5P_CLS_LINK_CREATE_ID_g"))
    _lib = L
    return L


def _chk(code, what):
    if code < 0:
        raise HDF5Error(f"HDF5: {what} failed")
    return code


def _dims(seq):
    return (hsize_t * len(seq))(*seq)


def fill_value(dtype):
    "what a never-written row reads as: NaN, or the int64 NaN stand-in -2**63 (exp_utils.py:465)"
    dtype = np.dtype(dtype)
    return np.array(np.nan if dtype.kind == "f" else -2 ** 63, dtype=dtype)


class Dataset:
    def __init__(self, file, name, handle):
        self.file, self.name, self.h = file, name, handle
        L = lib()
        t = _chk(L.H5Dget_type(handle), "H5Dget_type")
        cls, size = L.H5Tget_class(t), L.H5Tget_size(t)
        L.H5Tclose(t)
        try:
            self.dtype = {(_CLASS_FLOAT, 4): np.dtype(np.float32), (_CLASS_FLOAT, 8): np.dtype(np.float64),
                          (_CLASS_INTEGER, 8): np.dtype(np.int64)}[(cls, size)]
        except KeyError:
            raise HDF5Error(f"dataset {name!r}: unsupported element type (class {cls}, {size} bytes)")

    # A file this process writes changes its extents through ``resize`` alone: the shape is queried once and kept (a stored
    # sample is ~170 datasets, each asked for its shape three times per row: a quarter of the store's time).  A SWMR
    # reader refreshes and asks every time.
    _shape = None

    @property
    def shape(self):
        if self._shape is not None and not self.file.swmr_read:
            return self._shape
        L = lib()
        if self.file.swmr_read:
            L.H5Drefresh(self.h)
        s = _chk(L.H5Dget_space(self.h), "H5Dget_space")
        n = L.H5Sget_simple_extent_ndims(s)
        d = (hsize_t * max(n, 1))()
        L.H5Sget_simple_extent_dims(s, d, None)
        L.H5Sclose(s)
        shape = tuple(int(d[i]) for i in range(n))
        if not self.file.swmr_read:
            self._shape = shape
        return shape

    def __len__(self):
        return self.shape[0]

    def resize(self, n):
        shp = self.shape
        _chk(lib().H5Dset_extent(self.h, _dims((n,) + shp[1:])), f"H5Dset_extent({self.name})")
        if self._shape is not None:
            self._shape = (n,) + shp[1:]

    def _slab(self, start, count, shape):
        L = lib()
        fs = _chk(L.H5Dget_space(self.h), "H5Dget_space")
        _chk(L.H5Sselect_hyperslab(fs, _SELECT_SET, _dims((start,) + (0,) * (len(shape) - 1)), None,
                                   _dims((count,) + shape[1:]), None), "H5Sselect_hyperslab")
        ms = _chk(L.H5Screate_simple(len(shape), _dims((count,) + shape[1:]), None), "H5Screate_simple")
        return fs, ms

    def write_rows(self, start, values):
        "rows [start, start+len(values)) <- values (the dataset must already be long enough)"
        L = lib()
        values = np.ascontiguousarray(values, dtype=self.dtype)
        shape = self.shape
        if values.shape[1:] != shape[1:]:
            raise ValueError(f"{self.name}: rows of shape {values.shape[1:]}, dataset has {shape[1:]}")
        if len(values) == 0:
            return
        fs, ms = self._slab(start, len(values), shape)
        try:
            _chk(L.H5Dwrite(self.h, L.types[self.dtype][1], ms, fs, 0, values.ctypes.data),
                 f"H5Dwrite({self.name})")
        finally:
            L.H5Sclose(fs); L.H5Sclose(ms)

    def read_rows(self, start, count):
        L = lib()
        shape = self.shape
        out = np.empty((count,) + shape[1:], dtype=self.dtype)
        if count == 0:
            return out
        fs, ms = self._slab(start, count, shape)
        try:
            _chk(L.H5Dread(self.h, L.types[self.dtype][1], ms, fs, 0, out.ctypes.data), f"H5Dread({self.name})")
        finally:
            L.H5Sclose(fs); L.H5Sclose(ms)
        return out

    def __getitem__(self, idx):
        "numpy-style indexing of the leading axis (whole rows are read, then indexed)"
        n = len(self)
        if isinstance(idx, slice):
            start, stop, step = idx.indices(n)
            if step == 1:
                return self.read_rows(start, max(0, stop - start))
        elif isinstance(idx, (int, np.integer)):
            i = int(idx) + (n if idx < 0 else 0)
            if not 0 <= i < n:
                raise IndexError(idx)
            return self.read_rows(i, 1)[0]
        return self.read_rows(0, n)[idx]

    def creation_properties(self):
        "(chunk shape, filter ids) -- for tests"
        L = lib()
        p = _chk(L.H5Dget_create_plist(self.h), "H5Dget_create_plist")
        nd = len(self.shape)
        ch = (hsize_t * nd)()
        L.H5Pget_chunk(p, nd, ch)
        filters = []
        for i in range(L.H5Pget_nfilters(p)):
            flags, ncd, cfg = C.c_uint(), C.c_size_t(0), C.c_uint()
            filters.append(L.H5Pget_filter2(p, i, flags, ncd, None, 0, None, cfg))
        L.H5Pclose(p)
        return tuple(int(c) for c in ch), filters

    def close(self):
        if self.h is not None:
            lib().H5Dclose(self.h)
            self.h = None


class File:
    """``File(path, "w")`` truncates/creates with the latest file-format bounds (SWMR capable) and
    no raw-data chunk cache (exp_utils.py:419-420); ``"a"``/``"r+"`` reopen for writing; ``"r"``
    reads, ``swmr=True`` for a file some writer still holds open."""

    def __init__(self, path, mode="r", swmr=False):
        L = lib()
        self.path, self.mode = os.fspath(path), mode
        self.swmr_read = bool(swmr) and mode == "r"
        self.swmr_mode = False
        self._dsets = {}
        fapl = _chk(L.H5Pcreate(L.FAPL), "H5Pcreate")
        try:
            _chk(L.H5Pset_libver_bounds(fapl, _LIBVER_LATEST, _LIBVER_LATEST), "H5Pset_libver_bounds")
            _chk(L.H5Pset_cache(fapl, 0, 521, 0, 0.75), "H5Pset_cache")      # rdcc_nbytes = 0
            b = self.path.encode()
            if mode == "w":
                h = L.H5Fcreate(b, ACC_TRUNC, 0, fapl)
            elif mode in ("w-", "x"):
                h = L.H5Fcreate(b, ACC_EXCL, 0, fapl)
            elif mode == "r":
                h = L.H5Fopen(b, ACC_RDONLY | (ACC_SWMR_READ if swmr else 0), fapl)
            elif mode in ("r+", "a"):
                if mode == "a" and not os.path.exists(self.path):
                    h = L.H5Fcreate(b, ACC_EXCL, 0, fapl)
                else:
                    h = L.H5Fopen(b, ACC_RDWR, fapl)
            else:
                raise ValueError(f"mode {mode!r}")
        finally:
            L.H5Pclose(fapl)
        if h < 0:
            raise HDF5Error(f"cannot open {self.path!r} (mode {mode!r}): not an HDF5 file, missing, or locked")
        self.h = h

    # ---- structure
    def create_dataset(self, name, row_shape, dtype, chunk_rows=1, fletcher32=True):
        "extendible dataset of shape (0, *row_shape), chunks (chunk_rows, *row_shape), NaN fill"
        L = lib()
        if self.swmr_mode:
            raise HDF5Error(f"cannot create dataset {name!r}: the file is already in SWMR write mode")
        dtype = np.dtype(dtype)
        if dtype not in L.types:
            raise TypeError(f"{name}: dtype {dtype} cannot hold the NaN fill (float32/float64/int64 only)")
        row_shape = tuple(int(s) for s in row_shape)
        space = _chk(L.H5Screate_simple(1 + len(row_shape), _dims((0,) + row_shape),
                                        _dims((UNLIMITED,) + row_shape)), "H5Screate_simple")
        dcpl = _chk(L.H5Pcreate(L.DCPL), "H5Pcreate")
        lcpl = _chk(L.H5Pcreate(L.LCPL), "H5Pcreate")
        try:
            # a zero-sized trailing dimension cannot be chunked; h5py bumps such chunk dims to 1 too
            _chk(L.H5Pset_chunk(dcpl, 1 + len(row_shape),
                                _dims((max(1, chunk_rows),) + tuple(max(1, s) for s in row_shape))),
                 "H5Pset_chunk")
            if fletcher32:
                _chk(L.H5Pset_fletcher32(dcpl), "H5Pset_fletcher32")
            fv = fill_value(dtype)
            _chk(L.H5Pset_fill_value(dcpl, L.types[dtype][1], fv.ctypes.data), "H5Pset_fill_value")
            _chk(L.H5Pset_create_intermediate_group(lcpl, 1), "H5Pset_create_intermediate_group")
            h = _chk(L.H5Dcreate2(self.h, name.encode(), L.types[dtype][0], space, lcpl, dcpl, 0),
                     f"H5Dcreate2({name})")
        finally:
            L.H5Pclose(dcpl); L.H5Pclose(lcpl); L.H5Sclose(space)
        d = self._dsets[name] = Dataset(self, name, h)
        return d

    def __getitem__(self, name):
        d = self._dsets.get(name)
        if d is None:
            h = lib().H5Dopen2(self.h, name.encode(), 0)
            if h < 0:
                raise KeyError(name)
            d = self._dsets[name] = Dataset(self, name, h)
        return d

    def __contains__(self, name):
        try:
            self[name]
            return True
        except KeyError:
            return False

    def _links(self, group_path):
        L = lib()
        g = _chk(L.H5Gopen2(self.h, group_path.encode(), 0), f"H5Gopen2({group_path})")
        try:
            info = (C.c_uint64 * 4)()       # H5G_info_t {int storage; hsize_t nlinks; int64 max_corder; hbool}
            _chk(L.H5Gget_info(g, info), "H5Gget_info")
            names = []
            for i in range(int(info[1])):
                n = L.H5Lget_name_by_idx(g, b".", 0, 0, i, None, 0, 0)
                buf = C.create_string_buffer(n + 1)
                L.H5Lget_name_by_idx(g, b".", 0, 0, i, buf, n + 1, 0)
                names.append(buf.value.decode())
            return names
        finally:
            L.H5Gclose(g)

    def keys(self):
        "top-level link names, alphabetical (what ``h5py.File.keys()`` gives)"
        return self._links("/")

    def dataset_names(self, group="/"):
        "every dataset, nested groups included, as slash-joined paths"
        out = []
        for n in self._links(group):
            path = (group.rstrip("/") + "/" + n).lstrip("/")
            if path in self:
                out.append(path)
            else:
                out.extend(self.dataset_names("/" + path))
        return out

    # ---- modes
    def start_swmr_write(self):
        "single-writer/multiple-reader mode: readers may open the file now; no new datasets after this"
        if not self.swmr_mode:
            # open dataset handles must be closed around the switch on 1.10 only if they hold
            # old-format indexes; ours are created under the latest bounds, so they may stay
            _chk(lib().H5Fstart_swmr_write(self.h), "H5Fstart_swmr_write")
            self.swmr_mode = True

    def flush(self):
        _chk(lib().H5Fflush(self.h, _SCOPE_GLOBAL), "H5Fflush")

    def close(self):
        if self.h is not None:
            for d in self._dsets.values():
                d.close()
            self._dsets.clear()
            lib().H5Fclose(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
