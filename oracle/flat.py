"""ctypes wrappers of the flat-arena transitions in oracle/csrc/sgmcmc_oracle.c
(TEST INFRASTRUCTURE).  Arrays are numpy, packed with 4-element alignment per
segment (``noise.packed_offsets``), so arena index == noise index."""
import ctypes

import numpy as np

from .noise import lib, packed_offsets

FLAG_INITIAL, FLAG_FINAL, FLAG_SAVE = 1, 2, 4


class StepParams(ctypes.Structure):
    _fields_ = [("grad_v", ctypes.c_double), ("bhn", ctypes.c_double), ("bh", ctypes.c_double),
                ("mom_decay", ctypes.c_double), ("noise_std", ctypes.c_double),
                ("alpha", ctypes.c_double), ("seed", ctypes.c_uint64), ("draw", ctypes.c_uint64),
                ("stream", ctypes.c_uint32), ("flags", ctypes.c_uint32)]


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _suffix(dtype):
    return {np.dtype(np.float32): "f32", np.dtype(np.float64): "f64"}[np.dtype(dtype)]


class FlatArena:
    """theta/g/m/v (+prev copies) of a list of segments in one packed array each."""

    def __init__(self, numels, dtype):
        self.numels = list(numels)
        self.off, self.total = packed_offsets(self.numels)
        self.dtype = np.dtype(dtype)
        z = lambda: np.zeros(self.total, dtype=self.dtype)  # noqa: E731
        self.theta, self.g, self.m, self.v = z(), z(), z(), z()
        self.prev_theta, self.prev_g, self.prev_m = z(), z(), z()
        self.M = np.ones(len(self.numels), dtype=np.float64)
        self.sums = np.zeros((len(self.numels), 6), dtype=np.float64)
        self._off = np.asarray(self.off, dtype=np.int64)
        self._numel = np.asarray(self.numels, dtype=np.int64)

    def seg(self, arr, s):
        return arr[self.off[s]:self.off[s] + self.numels[s]]

    def _params(self, **kw):
        return StepParams(**kw)

    def step(self, kind, *, grad_v, bhn, bh, mom_decay, noise_std, alpha, seed, draw, stream=0,
             flags=0, save_momentum=True):
        P = StepParams(grad_v=grad_v, bhn=bhn, bh=bh, mom_decay=mom_decay, noise_std=noise_std,
                       alpha=alpha, seed=seed, draw=draw, stream=stream, flags=flags)
        L, sfx, n = lib(), _suffix(self.dtype), len(self.numels)
        if kind == "sgld":
            fn = getattr(L, f"oracle_sgld_step_{sfx}")
            fn.restype = None
            fn(_ptr(self.theta), _ptr(self.g), _ptr(self.m), _ptr(self.v), ctypes.c_int(n),
               _ptr(self._off), _ptr(self._numel), _ptr(self.M), ctypes.byref(P), _ptr(self.sums))
        else:
            fn = getattr(L, f"oracle_{kind}_step_{sfx}")
            fn.restype = None
            fn(_ptr(self.theta), _ptr(self.g), _ptr(self.m), _ptr(self.v), _ptr(self.prev_theta),
               _ptr(self.prev_g), _ptr(self.prev_m if save_momentum else None), ctypes.c_int(n),
               _ptr(self._off), _ptr(self._numel), _ptr(self.M), ctypes.byref(P), _ptr(self.sums))
        return self.sums

    def sample_momentum(self, std, keep, seed, draw, stream=0):
        fn = getattr(lib(), f"oracle_sample_momentum_{_suffix(self.dtype)}")
        fn.restype = None
        fn(_ptr(self.m), ctypes.c_int(len(self.numels)), _ptr(self._off), _ptr(self._numel),
           ctypes.c_double(std), ctypes.c_double(keep), ctypes.c_uint64(seed),
           ctypes.c_uint32(stream), ctypes.c_uint64(draw))
