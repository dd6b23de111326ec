"""Generate the golden vectors under tests/golden/ by IMPORTING THE REFERENCE.

Runs only in the build container (needs /root/reference); the outputs are
committed, this script documents how they were made:

    python tests/golden/make_goldens.py

For every scenario of tests/scenarios.py and both dtypes the reference's own
``bnn_priors.mcmc.{SGLD,VerletSGLD,HMC}`` are driven on the reference's own
models, with ``torch.randn_like`` / ``torch.rand`` redirected to the build's
noise specification (oracle/noise.py) so that the same trajectory can be
reproduced by the oracle and by the HIP kernels from (seed, stream, draw).
Nothing from the reference's source text is stored -- only inputs and outputs.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import ref_stubs  # noqa: E402

ref_stubs.install()

import scenarios as S  # noqa: E402
from oracle.noise import NoiseSource, PURPOSE_MOMENTUM, PURPOSE_STEP  # noqa: E402


class ReferenceHooks:
    """Redirects the reference's torch RNG draws to the Philox spec."""

    def __init__(self, opt, noise):
        self.opt, self.noise = opt, noise

    def state(self, p):
        return self.opt.state[p]

    @staticmethod
    def flat(tensors):
        return torch.cat([t.detach().reshape(-1) for t in tensors]).double().numpy().copy()

    def call(self, purpose, fn, *a, **kw):
        if purpose == "mh":
            if self.opt.param_groups[0]["temperature"] == 0.0:
                return fn(*a, **kw)
            u = self.noise.uniform()
            real = torch.rand
            torch.rand = lambda *s, **k: torch.tensor(u, dtype=torch.float64)
            try:
                return fn(*a, **kw)
            finally:
                torch.rand = real
        draw = self.noise.begin_sweep()
        code = PURPOSE_MOMENTUM if purpose == "momentum" else PURPOSE_STEP
        counter = [0]

        def fake_randn_like(t, **k):
            i = counter[0]
            counter[0] += 1
            return self.noise.tensor_normals(draw, code, i, t)
        real = torch.randn_like
        torch.randn_like = fake_randn_like
        try:
            return fn(*a, **kw)
        finally:
            torch.randn_like = real


def run_reference(name, cfg, dtype_name):
    from bnn_priors import mcmc as ref_mcmc
    from bnn_priors import models as ref_models
    dtype = getattr(torch, dtype_name)
    old = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        torch.manual_seed(0)
        model, closure = S.make_model(cfg["model"], ref_models, dtype)
        classes = dict(sgld=ref_mcmc.SGLD, verlet=ref_mcmc.VerletSGLD, hmc=ref_mcmc.HMC)
        opt = S.build_optimizer(classes, model.parameters(), cfg)
        S.preset(model, opt, cfg, dtype, lambda p: opt.state[p])
        noise = NoiseSource(S.SEED, [p.numel() for p in model.parameters()], stream=0)
        return S.drive(opt, model, closure, cfg, ReferenceHooks(opt, noise))
    finally:
        torch.set_default_dtype(old)


def main():
    for dtype_name in S.DTYPES:
        out = {}
        for name, cfg in S.SCENARIOS.items():
            rec = run_reference(name, cfg, dtype_name)
            for k, v in rec.items():
                if dtype_name == "float32" and k in ("theta", "mom", "final_theta"):
                    v = v.astype(np.float32)
                out[f"{name}/{k}"] = v
            n_rej = int(rec["mh_rejected"].sum())
            print(f"{dtype_name:8s} {name:28s} rejections {n_rej}/{len(rec['mh_rejected'])} "
                  f"dE {np.round(rec['mh_delta_energy'], 4).tolist()}")
        np.savez_compressed(os.path.join(HERE, f"samplers_{dtype_name}.npz"), **out)
    meta = dict(scenarios=S.SCENARIOS, n_steps=S.N_STEPS, mh_every=S.MH_EVERY, lr_decay=S.LR_DECAY,
                seed=S.SEED, torch=torch.__version__)
    with open(os.path.join(HERE, "samplers_meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
