"""Shared test plumbing: hooks for scenarios.drive and golden comparison."""
import os

import numpy as np
import torch

import scenarios as S

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_cache = {}


def golden(dtype_name):
    if dtype_name not in _cache:
        _cache[dtype_name] = dict(np.load(os.path.join(GOLDEN, f"samplers_{dtype_name}.npz")))
    return _cache[dtype_name]


class PlainHooks:
    "implementation draws its own (spec) noise; nothing to redirect"

    def __init__(self, opt):
        self.opt = opt

    def state(self, p):
        return self.opt.state[p]

    @staticmethod
    def flat(tensors):
        return torch.cat([t.detach().reshape(-1) for t in tensors]).double().cpu().numpy().copy()

    def call(self, purpose, fn, *a, **kw):
        return fn(*a, **kw)


class default_dtype:
    def __init__(self, dtype):
        self.dtype = dtype

    def __enter__(self):
        self.old = torch.get_default_dtype()
        torch.set_default_dtype(self.dtype)

    def __exit__(self, *exc):
        torch.set_default_dtype(self.old)


def compare(rec, gold, name, rtol, atol, traj_rtol=None, traj_atol=None, u_eps=0.0):
    """accept/reject flags and step indices bit-exact; floats within tolerance."""
    g = {k.split("/", 1)[1]: v for k, v in gold.items() if k.startswith(name + "/")}
    assert np.array_equal(rec["mh_rejected"], g["mh_rejected"]), \
        (name, rec["mh_rejected"], g["mh_rejected"], rec["mh_delta_energy"], g["mh_delta_energy"])
    assert np.array_equal(rec["mh_step"], g["mh_step"])
    assert np.array_equal(rec["rec_steps"], g["rec_steps"])
    np.testing.assert_array_equal(rec["lr"], g["lr"])
    traj_rtol = rtol if traj_rtol is None else traj_rtol
    traj_atol = atol if traj_atol is None else traj_atol
    for k in ("theta", "mom", "final_theta"):
        np.testing.assert_allclose(rec[k], g[k], rtol=traj_rtol, atol=traj_atol, err_msg=f"{name}:{k}")
    # delta_energy contains (U - U_prev) * N with U an fp32 `.item()` in the reference
    # (verlet_sgld.py:40-41): one ulp of U moves dE by N * ulp(U).  That noise is inherent to
    # the reference's formula; allow a few ulps of it on top of the general tolerance.
    extra = 0.0
    if u_eps:
        extra = 4 * S.SCENARIOS[name]["N"] * u_eps * float(np.abs(g["loss"]).max())
    for k in ("delta_energy", "prev_delta", "est_temp", "est_cfg", "mh_delta_energy",
              "mh_log_acc", "loss", "final_precond"):
        a, b = np.asarray(rec[k], dtype=np.float64), np.asarray(g[k], dtype=np.float64)
        assert np.array_equal(np.isnan(a), np.isnan(b)), (name, k)
        assert np.array_equal(np.isinf(a), np.isinf(b)), (name, k)
        m = np.isfinite(b)
        at = atol + (extra if k in ("mh_delta_energy", "mh_log_acc") else 0.0)
        np.testing.assert_allclose(a[m], b[m], rtol=rtol, atol=at, err_msg=f"{name}:{k}")
    # accept/reject margins: reject <=> log(u) > log_acc.  The decision is only as exact as dE;
    # report the smallest |log u - log_acc| and require it to exceed the tolerance applied to dE
    margins = []
    for u, la, de in zip(rec.get("mh_u", []), g["mh_log_acc"], g["mh_delta_energy"]):
        if np.isfinite(u) and np.isfinite(la):
            margins.append(abs(np.log(u) - la))
    if margins:
        T = max(S.SCENARIOS[name]["T"], 1e-12)
        floor = (atol + extra + rtol * float(np.abs(g["mh_delta_energy"][np.isfinite(g["mh_delta_energy"])]).max())) / T
        MARGINS[name] = (min(margins), floor)
        assert min(margins) > floor, f"{name}: accept margin {min(margins):.3g} <= dE tolerance {floor:.3g}"
    return g


MARGINS = {}
