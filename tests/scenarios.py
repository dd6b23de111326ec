"""Sampler-level parity scenarios, shared by the golden generator and the tests.

A scenario fixes a target model, a sampler family, hyper-parameters, a Philox
seed and a schedule of calls (``sample_momentum`` / ``initial_step`` / ``step`` /
``final_step`` / ``delta_energy`` / ``maybe_reject`` / ``update_preconditioner``).
``drive`` executes that schedule against *any* implementation of the sampler API
-- the imported reference (tests/golden/make_goldens.py, with the spec's noise
injected), the CPU oracle, or the HIP product -- and records the trajectory.

The call pattern is the one the reference's own tests use
(testing/test_verlet_sgld.py:95-118, testing/test_hmc.py:85-105): an M-H test
every ``mh_every`` steps, momentum refresh before each initial step for HMC.
"""
import math

import numpy as np
import torch

# name -> config.  lr decays by 3 % per optimizer call to exercise per-step scalars.
SCENARIOS = {
    # ---- VerletSGLD on a product of Gaussians (testing/test_verlet_sgld.py:58) ----
    "verlet_gauss_a0.9_T0.75": dict(kind="verlet", model="gauss", a=0.9, T=0.75, lr=1 / 32, N=3),
    "verlet_gauss_a0_T0.75": dict(kind="verlet", model="gauss", a=0.0, T=0.75, lr=1 / 32, N=3),
    "verlet_gauss_a0.994_T1": dict(kind="verlet", model="gauss", a=0.994, T=1.0, lr=1 / 16, N=3),
    "verlet_gauss_a1_T1": dict(kind="verlet", model="gauss", a=1.0, T=1.0, lr=1 / 32, N=3),
    "verlet_gauss_a0.9_T0": dict(kind="verlet", model="gauss", a=0.9, T=0.0, lr=1 / 32, N=3),
    "verlet_gauss_biglr": dict(kind="verlet", model="gauss", a=0.5, T=1.0, lr=40.0, N=1),
    # ---- the closed-form energy test target (testing/test_verlet_sgld.py:148) ----
    "verlet_funnel": dict(kind="verlet", model="funnel", a=127 / 128, T=0.75, lr=1 / 32, N=1),
    # ---- SGLD (testing/test_sgld.py:13,61) ----
    "sgld_gauss_a0.9_T0.75": dict(kind="sgld", model="gauss", a=0.9, T=0.75, lr=1 / 512, N=3),
    "sgld_gauss_a0_T0.75": dict(kind="sgld", model="gauss", a=0.0, T=0.75, lr=1 / 512, N=3),
    "sgld_gauss_a0.9_T0": dict(kind="sgld", model="gauss", a=0.9, T=0.0, lr=1 / 512, N=3),
    # ---- HMC (testing/test_hmc.py:68) ----
    "hmc_gauss": dict(kind="hmc", model="gauss", a=1.0, T=1.0, lr=1 / 32, N=5),
    "hmc_gauss_biglr": dict(kind="hmc", model="gauss", a=1.0, T=1.0, lr=120.0, N=1),
    # ---- a real network gradient: tiny ClassificationDenseNet on 32 fixed inputs ----
    "verlet_dense": dict(kind="verlet", model="dense", a=0.9, T=1.0, lr=0.01, N=32),
    "hmc_dense": dict(kind="hmc", model="dense", a=1.0, T=1.0, lr=0.005, N=32),
}
# ---- BASELINE configs[4]: HMC trajectories of L = 50 leapfrog steps at T in {1, 0.1, 0.01} (T != 1 extends the
# reference, whose HMC asserts T == 1: no reference goldens exist for these; HIP is compared with the oracle)
TEMPERED = {
    f"hmc_dense_L50_T{T:g}": dict(kind="hmc", model="dense", a=1.0, T=T, lr=0.002 * T, N=32, n_steps=100,
                                  mh_every=50, lr_decay=1.0, tempered=True)
    for T in (1.0, 0.1, 0.01)
}
DTYPES = ("float32", "float64")
N_STEPS, MH_EVERY, LR_DECAY, SEED = 24, 4, 0.97, 20240607


def make_model(name, models, dtype, device="cpu"):
    """Build the target with ``models`` = the module providing the model classes
    (the reference's ``bnn_priors.models`` or ``bnn_priors_amd.models``).
    Returns (model, closure); deterministic given the torch seed set by the caller."""
    if name == "gauss":
        model = models.GaussianModel(N=3, D=50, mean=1., std=2.).to(device)
        return model, model.potential_avg_closure
    if name == "funnel":
        model = models.NealFunnelT().to(device)
        return model, model.potential_avg_closure
    if name == "dense":
        g = torch.Generator().manual_seed(77)
        x = torch.rand(32, 20, generator=g, dtype=torch.float64).to(dtype).to(device)
        y = torch.randint(0, 10, (32,), generator=g).to(device)
        model = models.ClassificationDenseNet(20, 10, 8, 3).to(device)

        def closure():
            model.zero_grad()
            loss = model.potential_avg(x, y, 32.)
            loss.backward()
            return loss
        closure.data = (x, y)
        return model, closure
    raise KeyError(name)


def drive(opt, model, closure, cfg, hooks, record_every=4):
    """Run the scenario's call schedule on ``opt``.

    ``hooks`` adapts the implementation: ``hooks.call(purpose, fn, *a, **kw)``
    wraps every noise-consuming optimizer call (so the reference's torch RNG
    calls can be redirected to the spec), ``hooks.state(p)`` returns the state
    mapping of parameter ``p`` and ``hooks.flat(tensors)`` a float64 numpy copy.
    """
    kind = cfg["kind"]
    # the schedule: module defaults unless the scenario overrides them (TEMPERED)
    N_STEPS, MH_EVERY, LR_DECAY = (cfg.get("n_steps", globals()["N_STEPS"]), cfg.get("mh_every", globals()["MH_EVERY"]),
                                   cfg.get("lr_decay", globals()["LR_DECAY"]))
    params = list(model.parameters())
    rec = dict(theta=[], mom=[], rec_steps=[], delta_energy=[], prev_delta=[], est_temp=[],
               est_cfg=[], mh_delta_energy=[], mh_log_acc=[], mh_rejected=[], mh_step=[],
               lr=[], loss=[], mh_u=[])
    # mirror of the spec's sweep counter (one value per sample_momentum / step sweep / M-H test)
    # so that the M-H uniforms -- and with them the accept margins -- can be reported
    from oracle.noise import mh_uniform
    counter = [0]
    real_call = hooks.call

    def counted_call(purpose, fn, *a, **kw):
        if purpose == "mh":
            if cfg["T"] != 0:
                rec["mh_u"].append(mh_uniform(SEED, 0, counter[0]))
                counter[0] += 1
            else:
                rec["mh_u"].append(float("nan"))
        else:
            counter[0] += 1
        return real_call(purpose, fn, *a, **kw)

    def snapshot(step):
        rec["rec_steps"].append(step)
        rec["theta"].append(hooks.flat(params))
        rec["mom"].append(hooks.flat([hooks.state(p)["momentum_buffer"] for p in params])
                          if "momentum_buffer" in hooks.state(params[0]) else
                          np.zeros(sum(p.numel() for p in params)))

    def scalars(loss):
        st = [hooks.state(p) for p in params]
        rec["delta_energy"].append([float(s.get("delta_energy", math.nan)) for s in st])
        rec["prev_delta"].append([float(s.get("prev_new_momentum_delta", math.nan)) for s in st])
        rec["est_temp"].append([float(s.get("est_temperature", math.nan)) for s in st])
        rec["est_cfg"].append([float(s.get("est_config_temp", math.nan)) for s in st])
        rec["lr"].append(opt.param_groups[0]["lr"])
        rec["loss"].append(float(loss))

    def decay():
        for g in opt.param_groups:
            g["lr"] *= LR_DECAY

    call = counted_call
    call("momentum", opt.sample_momentum)
    prev_loss = None
    for step in range(N_STEPS + 1):
        if step % MH_EVERY == 0:
            if step != 0:
                # SGLD with a=0 cannot compute metrics on a final step (reference bug,
                # mcmc/sgld.py:132-137)
                cm = not (kind == "sgld" and cfg["a"] == 0)
                loss = call("step", opt.final_step, closure, calc_metrics=cm).item()
                scalars(loss)
                de = opt.delta_energy(prev_loss, loss)
                if kind == "sgld":
                    rejected, log_acc = False, 0.0
                else:
                    rejected, log_acc = call("mh", opt.maybe_reject, de)
                rec["mh_delta_energy"].append(de)
                rec["mh_log_acc"].append(log_acc)
                rec["mh_rejected"].append(bool(rejected))
                rec["mh_step"].append(step)
                snapshot(step)
                if step == N_STEPS:
                    break
            if kind == "hmc":
                call("momentum", opt.sample_momentum)
            if kind == "sgld":
                prev_loss = call("step", opt.step, closure).item()
            else:
                prev_loss = call("step", opt.initial_step, closure, save_state=True).item()
            scalars(prev_loss)
        else:
            loss = call("step", opt.step, closure).item()
            scalars(loss)
        decay()
        if step % record_every == 1:
            snapshot(step)
    opt.update_preconditioner()
    rec["final_precond"] = [float(hooks.state(p)["preconditioner"]) for p in params]
    rec["final_theta"] = hooks.flat(params)
    return {k: np.asarray(v) for k, v in rec.items()}


def preset(model, opt, cfg, dtype, state_of):
    """Overwrite the model's parameters and the preconditioners with values that
    depend only on (scenario, dtype) so every implementation starts identically."""
    rng = np.random.RandomState(SEED % (2 ** 31))
    with torch.no_grad():
        for p in model.parameters():
            if cfg["model"] == "gauss":
                v = 1.0 + 2.0 * math.sqrt(max(cfg["T"], 0.25)) * rng.standard_normal(p.shape)
            elif cfg["model"] == "funnel":
                v = 0.5 * rng.standard_normal(p.shape) * np.linspace(0.01, 1, 100)
            else:
                v = 0.3 * rng.standard_normal(p.shape)
            p.copy_(torch.from_numpy(np.asarray(v)).to(p.dtype))
    for p in model.parameters():
        state_of(p)["preconditioner"] = float((rng.uniform() + 0.2) / 2.0)


def build_optimizer(classes, params, cfg, **extra):
    "classes = dict(sgld=..., verlet=..., hmc=...)"
    if cfg["kind"] == "hmc":
        if cfg.get("tempered"):
            extra = dict(extra, temperature=cfg["T"])
        return classes["hmc"](params, lr=cfg["lr"], num_data=cfg["N"], **extra)
    return classes[cfg["kind"]](params, lr=cfg["lr"], num_data=cfg["N"], momentum=cfg["a"],
                                temperature=cfg["T"], **extra)
