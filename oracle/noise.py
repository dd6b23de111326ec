"""Noise specification, oracle side (TEST INFRASTRUCTURE).

Two independent statements of the spec in DESIGN.md "Noise":

* ``philox4x32_10_np`` / ``uniform_np`` -- numpy integer arithmetic (the
  counter-based generator and the exact u32 -> uniform map);
* the C library ``oracle/_build/libsgmcmc_oracle.so`` -- the same generator
  plus the fmaf-polynomial Box-Muller, which numpy cannot state bit-exactly
  (no fused multiply-add).

The reference draws ``torch.randn_like(p)`` per tensor (mcmc/verlet_sgld.py:163,
mcmc/sgld.py:66-69,142) and ``torch.rand(())`` for Metropolis-Hastings
(mcmc/verlet_sgld.py:61); ``NoiseSource`` hands out the Philox-spec replacement
for exactly those draws, in the same consumption order.
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libsgmcmc_oracle.so")
_lib = None

PURPOSE_STEP, PURPOSE_MOMENTUM, PURPOSE_MH = 0, 1, 2


def build(force=False):
    """Compile the C restatement with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "csrc", "sgmcmc_oracle.c")
    if (force or not os.path.exists(_LIB_PATH)
            or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)):
        subprocess.check_call(["make", "-C", _HERE, "-B"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        L.oracle_normals_f32.argtypes = [ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint64,
                                         ctypes.c_uint32, ctypes.c_int64, ctypes.c_int64,
                                         ctypes.c_void_p]
        L.oracle_normals_f32.restype = None
        L.oracle_mh_uniform.argtypes = [ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint64]
        L.oracle_mh_uniform.restype = ctypes.c_double
        L.oracle_philox4x32_10.argtypes = [ctypes.c_void_p] * 3
        L.oracle_philox4x32_10.restype = None
        _lib = L
    return _lib


# ----------------------------------------------------------------- numpy spec
_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)


def philox4x32_10_np(ctr, key):
    """Vectorised Philox4x32-10.  ctr: uint32[...,4], key: uint32[2] -> uint32[...,4]."""
    c = [np.asarray(ctr[..., i], dtype=np.uint32) for i in range(4)]
    k0, k1 = np.uint32(key[0]), np.uint32(key[1])
    mask = np.uint64(0xFFFFFFFF)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = _M0 * c[0].astype(np.uint64)
            p1 = _M1 * c[2].astype(np.uint64)
            n0 = (p1 >> np.uint64(32)).astype(np.uint32) ^ c[1] ^ k0
            n1 = (p1 & mask).astype(np.uint32)
            n2 = (p0 >> np.uint64(32)).astype(np.uint32) ^ c[3] ^ k1
            n3 = (p0 & mask).astype(np.uint32)
            c = [n0, n1, n2, n3]
            k0 = np.uint32(k0 + _W0)
            k1 = np.uint32(k1 + _W1)
    return np.stack(c, axis=-1)


def noise_counter_np(quad, draw, stream, purpose):
    quad = np.asarray(quad, dtype=np.uint64)
    ctr = np.empty(quad.shape + (4,), dtype=np.uint32)
    ctr[..., 0] = (quad & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    ctr[..., 1] = (quad >> np.uint64(32)).astype(np.uint32)
    ctr[..., 2] = np.uint32(draw & 0xFFFFFFFF)
    ctr[..., 3] = np.uint32((purpose << 28) | ((stream & 0xFFF) << 16) | ((draw >> 32) & 0xFFFF))
    return ctr


def uniform_np(x):
    """u = (2*(x>>9)+1) * 2^-24, exactly representable in fp32."""
    x = np.asarray(x, dtype=np.uint32)
    return ((x >> np.uint32(9)).astype(np.float64) * 2.0 + 1.0) * 2.0 ** -24


def normals_np_f64(seed, stream, draw, purpose, start, n):
    """Box-Muller of the spec's uniforms with float64 libm -- NOT bit-identical to
    the spec's fmaf polynomials; used to bound their error in tests."""
    idx = np.arange(start, start + n, dtype=np.uint64)
    quad = idx >> np.uint64(2)
    uq, inv = np.unique(quad, return_inverse=True)
    x = philox4x32_10_np(noise_counter_np(uq, draw, stream, purpose),
                         (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))
    u = uniform_np(x)
    z = np.empty_like(u)
    for h in range(2):
        rad = np.sqrt(-2.0 * np.log(u[:, 2 * h]))
        ang = 2.0 * np.pi * u[:, 2 * h + 1]
        z[:, 2 * h] = rad * np.cos(ang)
        z[:, 2 * h + 1] = rad * np.sin(ang)
    return z[inv, (idx & np.uint64(3)).astype(np.int64)]


# -------------------------------------------------------------- C-backed spec
def normals(seed, stream, draw, purpose, start, n):
    out = np.empty(n, dtype=np.float32)
    lib().oracle_normals_f32(seed, stream, draw, purpose, start, n, out.ctypes.data)
    return out


def mh_uniform(seed, stream, draw):
    return lib().oracle_mh_uniform(seed, stream, draw)


def packed_offsets(numels):
    """Noise index base of each segment: n_0 = 0, n_{s+1} = n_s + 4*ceil(numel_s/4)."""
    offs, acc = [], 0
    for n in numels:
        offs.append(acc)
        acc += (n + 3) // 4 * 4
    return offs, acc


class NoiseSource:
    """Hands out the spec's draws in the reference's consumption order.

    One *sweep* (a step-fn pass over all tensors, a ``sample_momentum`` call, or
    one M-H uniform) consumes one value of the monotone ``draw`` counter.
    """

    def __init__(self, seed, numels, stream=0):
        self.seed, self.stream = int(seed), int(stream)
        self.offsets, self.total = packed_offsets(list(numels))
        self.numels = list(numels)
        self.draw = 0

    def begin_sweep(self):
        d = self.draw
        self.draw += 1
        return d

    def tensor_normals(self, draw, purpose, index, like):
        z = normals(self.seed, self.stream, draw, purpose, self.offsets[index], self.numels[index])
        return torch.from_numpy(z).to(like.dtype).reshape(like.shape)

    def uniform(self):
        return mh_uniform(self.seed, self.stream, self.begin_sweep())
