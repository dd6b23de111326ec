"""Metric / sample sinks with the interface the runners use
(``add_scalar(name, value, step)``, ``flush(every_s)``, ``add_state_dict(sd, step)``,
``load_samples(keep_steps)``; reference: bnn_priors/exp_utils.py:409-536).

h5py is not part of this image, so the HDF5 container format itself is the
"next" row f2; key names and step numbering are already identical,
and ``NpzModelSaver`` writes the same arrays (``<state_dict key>``, ``steps``,
``timestamps``) to an ``.npz``.
"""
import time
from collections import OrderedDict

import numpy as np
import torch

__all__ = ("MemoryMetrics", "MemoryModelSaver", "NpzModelSaver")


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


class MemoryModelSaver:
    def __init__(self):
        self.samples, self.steps, self.timestamps = [], [], []

    def add_state_dict(self, state_dict, step):
        self.samples.append({k: v.detach().cpu().clone() for k, v in state_dict.items()})
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
