"""Metric / sample sinks with the interface the runners use
(``add_scalar(name, value, step)``, ``flush(every_s)``, ``add_state_dict(sd, step)``,
``load_samples(idx, keep_steps)``; reference: bnn_priors/exp_utils.py:409-551).

``HDF5ModelSaver`` / ``HDF5Metrics`` write the reference's file layout -- one extendible
dataset per key, shape ``(n, *shape)`` in chunks of ``(chunk, *shape)``, Fletcher-32
checksums, NaN fill (-2**63 in int64 columns), ``steps`` int64 and ``timestamps`` float64,
SWMR so that a run can be read while it is being written -- through the HDF5 C library
(``_h5.py``; h5py itself is not in this image).  ``Memory*`` / ``Npz*`` are the
dependency-free sinks the tests and the benchmark use.
"""
import time
from collections import OrderedDict

import numpy as np
import torch

__all__ = ("MemoryMetrics", "MemoryModelSaver", "NpzModelSaver", "HDF5ModelSaver", "HDF5Metrics",
           "load_samples", "reject_samples_")


class MemoryMetrics:
    "in-memory HDF5Metrics stand-in: rows keyed by step, NaN where a key was not logged"

    def __init__(self):
        self.rows = OrderedDict()   # step -> {name: value}
        self._step = -2 ** 63

    def add_scalar(self, name, value, step, dtype=None):
        if step < self._step:
            raise ValueError(f"step went backwards ({self._step} -> {step})")
        if step > self._step:
            self._step = step
            self.rows[step] = {"timestamps": time.time()}
        self.rows[step][name] = value

    def flush(self, every_s=0):
        pass

    def column(self, name):
        "(steps, values) with NaN fill, like reading the HDF5 dataset"
        steps = np.array(list(self.rows), dtype=np.int64)
        vals = np.array([float(r.get(name, np.nan)) for r in self.rows.values()], dtype=np.float64)
        return steps, vals

    def names(self):
        out = OrderedDict()
        for r in self.rows.values():
            for k in r:
                out[k] = True
        return [k for k in out if k != "timestamps"]


def state_dict_to_host(state_dict):
    """{k: CPU tensor} of a state dict whose tensors live on a GPU, with ONE device-to-host copy per (device, dtype)
    instead of one per tensor: the tensors of a group are concatenated on the device (one launch), copied, and handed
    out as views of the host buffer.  A googleresnet sample is 65 parameters + 63 BatchNorm buffers: 128 synchronous
    copies of a few KB each took 4-5 ms of every stored sample (exp_utils.py:489-509 copies tensor by tensor on the
    host it runs on).  Same values; tensors already on the host are passed through detached."""
    out, groups = {}, OrderedDict()
    for k, v in state_dict.items():
        v = v.detach()
        if v.is_cuda:
            groups.setdefault((v.device, v.dtype), []).append((k, v))
        else:
            out[k] = v
    for items in groups.values():
        if len(items) == 1:
            out[items[0][0]] = items[0][1].cpu()
            continue
        flat = torch.cat([v.reshape(-1) for _, v in items]).cpu()
        at = 0
        for k, v in items:
            out[k] = flat[at:at + v.numel()].view(v.shape)
            at += v.numel()
    return {k: out[k] for k in state_dict}       # (the caller's key order: the first write creates the datasets in it)


class MemoryModelSaver:
    def __init__(self):
        self.samples, self.steps, self.timestamps = [], [], []

    def add_state_dict(self, state_dict, step):
        self.samples.append({k: v.clone() for k, v in state_dict_to_host(state_dict).items()})
        self.steps.append(step)
        self.timestamps.append(time.time())

    def flush(self):
        pass

    def load_samples(self, idx=slice(None), keep_steps=True):
        if not self.samples:
            return {}
        out = {k: torch.stack([s[k] for s in self.samples])[idx] for k in self.samples[0]}
        if keep_steps:
            out["steps"] = torch.tensor(self.steps, dtype=torch.int64)[idx]
            out["timestamps"] = torch.tensor(self.timestamps, dtype=torch.float64)[idx]
        return out


class NpzModelSaver(MemoryModelSaver):
    "keeps samples in memory and rewrites ``path`` on flush (small nets: ~1 MB per sample)"

    def __init__(self, path):
        super().__init__()
        self.path = path

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.flush()

    def flush(self):
        if self.samples:
            np.savez(self.path, **{k: v.numpy() for k, v in self.load_samples().items()})


# ------------------------------------------------------------------------- HDF5 sinks
_INT64_NAN = -2 ** 63       # what a NaN becomes in an int64 column (exp_utils.py:465, test_exp_utils.py:63-68)


def _column_dtype(value):
    "the column type a first logged value selects: python/numpy ints (and bools) -> int64, else float64"
    if isinstance(value, (bool, np.bool_, int, np.integer)):
        return np.dtype(np.int64)
    if isinstance(value, np.floating) and np.dtype(type(value)) == np.float32:
        return np.dtype(np.float32)
    return np.dtype(np.float64)


def _nan_of(dtype):
    return np.nan if np.dtype(dtype).kind == "f" else _INT64_NAN


class HDF5ModelSaver:
    """Appends one row per ``add_state_dict`` to datasets named after the ``state_dict`` keys
    (exp_utils.py:409-489).  Datasets are created from the first row; the file then enters SWMR
    write mode, so later rows must carry the same keys."""
    chunk_size = 1

    def __init__(self, path, mode):
        self.path, self.mode = path, mode
        self._i = 0                    # next row to write
        self._created = False
        self.f = None

    def __enter__(self):
        from . import _h5
        self.f = _h5.File(self.path, self.mode)
        return self

    def __exit__(self, exc_type, exc_value, traceback):
        try:
            self.flush()               # anything still cached goes to disk before closing
        finally:
            self.f.close()

    def add_state_dict(self, state_dict, step):
        rows = {k: v.unsqueeze(0).numpy() for k, v in state_dict_to_host(state_dict).items()}
        rows["steps"] = np.array([step], dtype=np.int64)
        rows["timestamps"] = np.array([time.time()], dtype=np.float64)
        self._i += self._write_at_cursor(rows)

    def _write_at_cursor(self, rows):
        "rows[k][j] -> dataset k, row self._i + j; returns the number of rows (cursor NOT advanced)"
        if not self._created:
            for k, v in rows.items():
                self.f.create_dataset(k, v.shape[1:], v.dtype, chunk_rows=self.chunk_size)
            self.f.start_swmr_write()  # readable while running; no new keys from here on
            self._created = True
        n = None
        for k, v in rows.items():
            if n is None:
                n = len(v)
            elif n != len(v):
                raise AssertionError("lengths unequal")
            try:
                d = self.f[k]
            except KeyError:
                raise KeyError(f"{k!r} was not among the keys of the first write; datasets cannot be "
                               "added once the file is in SWMR mode") from None
            if self._i + n > len(d):
                d.resize(self._i + n)
            d.write_rows(self._i, v)
        assert n is not None
        return n

    def flush(self):
        if self.f is not None and self.f.h is not None:
            self.f.flush()

    def load_samples(self, idx=slice(None), keep_steps=True):
        try:
            self.flush()
        except OSError:
            pass
        return load_samples(self.path, idx=idx, keep_steps=keep_steps)


class HDF5Metrics(HDF5ModelSaver):
    """Scalar streams keyed by name, one row per distinct ``step`` (exp_utils.py:492-535): rows are
    staged in a ``chunk_size``-row cache (NaN where a key was not logged at that step) and written
    a chunk at a time; ``flush`` writes the partly filled chunk in place."""

    def __init__(self, path, mode, chunk_size=8 * 1024):
        super().__init__(path, mode)
        self.chunk_size = chunk_size
        self._step = -2 ** 63
        self._cache = {}
        self._row = -1                 # row of the cache the current step occupies
        self.last_flush = time.time()

    def add_scalar(self, name, value, step, dtype=None):
        if step > self._step:
            self._row += 1
            if self._row >= self.chunk_size:            # cache full: it becomes a chunk of the file
                self._i += self._write_at_cursor(self._cache)
                self._row = 0
                for col in self._cache.values():
                    col[:] = _nan_of(col.dtype)
            self._step = step
            self._put("steps", step, np.dtype(np.int64))
            self._put("timestamps", time.time(), np.dtype(np.float64))
        elif step < self._step:
            raise ValueError(f"step went backwards ({self._step} -> {step})")
        self._put(name, value, np.dtype(dtype) if dtype is not None else None)

    def _put(self, name, value, dtype):
        col = self._cache.get(name)
        if col is None:
            dtype = dtype or _column_dtype(value)
            col = self._cache[name] = np.full(self.chunk_size, _nan_of(dtype), dtype=dtype)
        if col.dtype.kind == "i" and isinstance(value, float) and value != value:
            value = _INT64_NAN
        col[self._row] = value

    def flush(self, every_s=0):
        "write the staged rows now, or only if the last flush is more than ``every_s`` seconds old"
        if self._row < 0:
            return
        now = time.time()
        if every_s <= 0 or now - self.last_flush > every_s:
            self.last_flush = now
            self._write_at_cursor({k: v[:self._row + 1] for k, v in self._cache.items()})
            super().flush()


def load_samples(path, idx=slice(None), keep_steps=True):
    """{key: tensor[idx]} of the top-level datasets of an HDF5 sample file (a file still being
    written is fine: SWMR read); falls back to ``torch.load`` for ``.pt`` sample files
    (exp_utils.py:538-551)."""
    from . import _h5
    try:
        f = _h5.File(path, "r", swmr=True)
    except _h5.HDF5Error:
        if not _h5.available():
            raise
        samples = torch.load(path)
        return {k: v[idx] for k, v in samples.items()}
    with f:
        skip = () if keep_steps else ("steps", "timestamps")
        return {k: torch.from_numpy(np.asarray(f[k][idx])) for k in f.keys()
                if k not in skip and k in f}


def reject_samples_(samples, metrics_file):
    """Replace every rejected sample by its predecessor, in place, using the
    ``acceptance/rejected`` stream of the run's metrics file (exp_utils.py:566-582).  Files
    without that stream are returned untouched."""
    try:
        rejected_col = metrics_file["acceptance/rejected"][:]
    except KeyError:
        return samples
    is_sample = metrics_file["acceptance/is_sample"][:] == 1
    rejected_col = rejected_col[is_sample]
    assert np.all((rejected_col == 0) | (rejected_col == 1))
    steps = metrics_file["steps"][:][is_sample]
    rejected = {int(s): bool(r) for s, r in zip(steps, rejected_col)}
    n = len(next(iter(samples.values())))
    for i in range(n):
        if rejected[int(samples["steps"][i])]:
            for k in samples:
                samples[k][i] = samples[k][i - 1]
    return samples
