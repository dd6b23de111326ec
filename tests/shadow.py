"""Lock-step oracle shadow of a runner on the HIP path (TEST INFRASTRUCTURE).

A model-scale trajectory cannot be compared with a CPU run of the same net: a ResNet with
BatchNorm amplifies one-ulp differences of the parameters into per-cent differences of the
gradient within a few steps, so after a 50-step trajectory two correct
implementations disagree in every digit.  What CAN be pinned at model scale is every single
transition: ``shadowed(RunnerClass)`` runs the product runner unchanged (captured-graph steps
included) and drives ``oracle.samplers.Ref*`` beside it --

* before a transition the shadow is given the device's (theta, m, v) and afterwards the gradient
  the device used (``p.grad`` -- "gradients from the same device autograd, fed to the oracle"),
  performs the reference's per-tensor update with the same Philox key / draw counter and must
  land on the device's (theta', m', v') to fp32 rounding;
* the per-tensor energy bookkeeping (``delta_energy``, ``prev_new_momentum_delta``; kinetic
  energies for HMC) accumulates in the shadow ON ITS OWN over the whole trajectory, from fp32
  ``.item()`` dots as the reference forms them, and is compared with the device's fp64 reductions
  at every Metropolis-Hastings point, where the shadow also takes its own accept / reject
  decision from the same uniform;
* momentum refreshes are compared bit for bit, roll-backs bit for bit, preconditioners to 1e-6.

Everything is recorded in ``runner.shadow.log`` for the test to assert on.
"""
import numpy as np
import torch

from oracle.noise import NoiseSource
from oracle.samplers import RefHMC, RefSGLD, RefVerletSGLD

ULP = 2.0 ** -23


def _cpu(t):
    return t.detach().to("cpu", copy=True)


def _cpu_all(tensors):
    "host copies of a list of device tensors through ONE transfer (65 tensors x 7 arrays per step otherwise)"
    tensors = [t.detach() for t in tensors]
    flat = torch.cat([t.reshape(-1) for t in tensors]).cpu()
    out, i = [], 0
    for t in tensors:
        out.append(flat[i:i + t.numel()].clone().reshape(t.shape))
        i += t.numel()
    return out


class Shadow:
    def __init__(self, opt, params, names, seed, chain_id, kind, temperature, momentum, lr, num_data):
        self.opt, self.dev_params, self.names = opt, list(params), list(names)
        self.params = [torch.nn.Parameter(_cpu(p)) for p in self.dev_params]
        noise = NoiseSource(seed, [p.numel() for p in self.params], stream=chain_id)
        if kind == "hmc":
            self.ref = RefHMC(self.params, lr=lr, num_data=num_data, noise=noise, temperature=temperature)
        else:
            cls = RefSGLD if kind == "sgld" else RefVerletSGLD
            self.ref = cls(self.params, lr=lr, num_data=num_data, momentum=momentum, temperature=temperature,
                           noise=noise)
        self.log = dict(steps=0, max_err=dict(theta=0.0, mom=0.0, v=0.0), mh=[], refresh=0, precond=0, restores=0)
        self._lr = lr

    # ------------------------------------------------------------------ device <-> shadow
    def _dev_state(self, p):
        return self.opt.state[p]

    def pull(self, grads=False):
        "shadow <- device: parameters, momentum, square_avg (and the gradient)"
        states = [self._dev_state(p) for p in self.dev_params]
        has_m = "momentum_buffer" in states[0]
        theta = _cpu_all(self.dev_params)
        mom = _cpu_all([st["momentum_buffer"] for st in states]) if has_m else None
        v = _cpu_all([st["square_avg"] for st in states])
        g = _cpu_all([p.grad for p in self.dev_params]) if grads else None
        for i, (st, q) in enumerate(zip(states, self.params)):
            sq = self.ref.state[q]
            q.data.copy_(theta[i])
            if has_m:
                sq["momentum_buffer"] = mom[i].reshape(q.shape)
            sq["square_avg"] = v[i].reshape(q.shape)
            sq["preconditioner"] = float(st["preconditioner"])
            if grads:
                q.grad = g[i]

    def _sync_scalars(self, lr=None):
        g, gd = self.ref.param_groups[0], self.opt.param_groups[0]
        g["lr"] = gd["lr"] if lr is None else lr
        g["temperature"] = gd["temperature"]

    def _check(self, what=("theta", "mom", "v")):
        "device results of the transition just made vs the shadow's"
        worst = self.log["max_err"]
        states = [self._dev_state(p) for p in self.dev_params]
        dev = dict(theta=_cpu_all(self.dev_params) if "theta" in what else None,
                   mom=_cpu_all([st["momentum_buffer"] for st in states])
                   if "mom" in what and "momentum_buffer" in states[0] else None,
                   v=_cpu_all([st["square_avg"] for st in states]) if "v" in what else None)
        for i, (name, q) in enumerate(zip(self.names, self.params)):
            sq = self.ref.state[q]
            refs = dict(theta=q, mom=sq.get("momentum_buffer"), v=sq["square_avg"])
            for k in what:
                if dev[k] is None:
                    continue
                a, b = dev[k][i].double().numpy().ravel(), refs[k].detach().double().numpy().ravel()
                scale = max(float(np.abs(b).max()), 1e-30)
                err = float(np.abs(a - b).max()) / scale
                worst[k] = max(worst[k], err)
                # one fused multiply-add against torch's rounding of the same expression: a few ulp of the
                # tensor's largest element (cancelling elements carry the absolute, not the relative, error)
                assert err <= 8 * ULP, (name, k, err)

    def _check_draw(self):
        assert self.ref.noise.draw == self.opt.engine.draw, (self.ref.noise.draw, self.opt.engine.draw)

    # ------------------------------------------------------------------ mirrored calls
    def before_step(self):
        self.pull()
        self._lr = self.opt.param_groups[0]["lr"]

    def after_step(self):
        "an ordinary leapfrog step has run on the device (graph replay or eager)"
        for q, g in zip(self.params, _cpu_all([p.grad for p in self.dev_params])):
            q.grad = g
        self._sync_scalars(self._lr)
        self.ref.step(calc_metrics=True)
        self._check()
        self._check_draw()
        self.log["steps"] += 1

    def sample_momentum(self, dev_call, *a, **kw):
        out = dev_call(*a, **kw)
        self._sync_scalars()
        self.ref.sample_momentum(*a, **kw)
        for name, p, q in zip(self.names, self.dev_params, self.params):
            m_dev, m_ref = _cpu(self._dev_state(p)["momentum_buffer"]), self.ref.state[q]["momentum_buffer"]
            # the same Philox normals times the same fp32 standard deviation: identical bits
            assert torch.equal(m_dev.reshape(-1), m_ref.reshape(-1)), name
        self._check_draw()
        self.log["refresh"] += 1
        return out

    def edge_step(self, which, dev_call, *a, **kw):
        "initial_step / final_step (eager on the device): same inputs, then the same call on both sides"
        self.pull(grads=True)
        self._sync_scalars()
        out = dev_call(*a, **kw)
        getattr(self.ref, which)(*a, **kw)
        self._check(("mom",) if which == "final_step" else ("theta", "mom", "v"))
        self._check_draw()
        return out

    def energies(self):
        "(sum_p |initial energy| + |final energy|) of the trajectory that just ended, for the dE tolerance"
        tot = 0.0
        for q in self.params:
            st = self.ref.state[q]
            tot += abs(st.get("delta_energy", 0.0)) + abs(self.ref._point_energy(self.ref.param_groups[0], q, st))
        return tot

    def delta_energy(self, dev_value, prev_potential, potential):
        if isinstance(self.ref, RefSGLD):
            return dev_value
        ref = self.ref.delta_energy(prev_potential, float(potential))
        self._pending_mh = dict(delta_energy=float(dev_value), delta_energy_ref=float(ref), scale=self.energies(),
                                T=self.opt.param_groups[0]["temperature"])
        return dev_value

    def maybe_reject(self, dev_call, delta_energy):
        rejected, log_acc = dev_call(delta_energy)
        rec = self._pending_mh
        u_draw = self.ref.noise.draw
        rej_ref, log_acc_ref = self.ref.maybe_reject(rec["delta_energy_ref"])
        from oracle.noise import mh_uniform
        rec.update(rejected=bool(rejected), rejected_ref=bool(rej_ref), log_acc=float(log_acc),
                   log_acc_ref=float(log_acc_ref),
                   log_u=float(np.log(mh_uniform(self.ref.noise.seed, self.ref.noise.stream, u_draw)))
                   if rec["T"] > 0 else 0.0)
        self.log["mh"].append(rec)
        self._check_draw()
        if rejected and rej_ref:
            # both sides restored the copies saved at the trajectory's initial_step -- the same bits went into both
            for name, p, q in zip(self.names, self.dev_params, self.params):
                assert torch.equal(_cpu(p), q.detach()), name
                assert torch.equal(_cpu(p.grad), q.grad), name
                assert torch.equal(_cpu(self._dev_state(p)["momentum_buffer"]).reshape(-1),
                                   self.ref.state[q]["momentum_buffer"].reshape(-1)), name
            self.log["restores"] += 1
        return rejected, log_acc

    def update_preconditioner(self, dev_call):
        out = dev_call()
        for p, q in zip(self.dev_params, self.params):
            self.ref.state[q]["square_avg"] = _cpu(self._dev_state(p)["square_avg"])
        self.ref.update_preconditioner()
        for name, p, q in zip(self.names, self.dev_params, self.params):
            a, b = float(self._dev_state(p)["preconditioner"]), self.ref.state[q]["preconditioner"]
            assert abs(a - b) <= 1e-6 * abs(b), (name, a, b)
        self.log["precond"] += 1
        return out


def shadowed(base, kind):
    "``base`` with an oracle shadow attached to its optimizer; kind in {'verlet', 'hmc', 'sgld'}"

    class Shadowed(base):
        def _make_optimizer(self, params):
            opt = super()._make_optimizer(params)
            g = opt.param_groups[0]
            sh = self.shadow = Shadow(opt, params, self.param_names, self.seed, self.chain_id, kind,
                                      g["temperature"], g["momentum"], g["lr"], g["num_data"])
            for name in ("sample_momentum", "maybe_reject", "update_preconditioner"):
                dev = getattr(opt, name)
                setattr(opt, name, (lambda dev, fn: lambda *a, **kw: fn(dev, *a, **kw))(dev, getattr(sh, name)))
            for name in ("initial_step", "final_step"):
                dev = getattr(opt, name)
                setattr(opt, name, (lambda dev, name: lambda *a, **kw: sh.edge_step(name, dev, *a, **kw))(dev, name))
            return opt

        def leapfrog(self, step, x, y, last_of_epoch):
            self.shadow.before_step()
            acc = super().leapfrog(step, x, y, last_of_epoch)
            self.shadow.after_step()
            return acc

        def _delta_energy(self, potential):
            return self.shadow.delta_energy(super()._delta_energy(potential), self._initial_potential,
                                            potential.item() if isinstance(potential, torch.Tensor) else potential)
    return Shadowed


def check_mh_points(log, min_points):
    """every M-H point of the run: the device's energy difference against the shadow's within the rounding of the fp32
    dots the reference forms (4 ulp of the summed per-tensor energies), the SAME accept / reject decision, and a
    margin |log u - log_acc| beyond that tolerance so that the agreement of the flags is not luck"""
    assert len(log["mh"]) >= min_points, log["mh"]
    for rec in log["mh"]:
        tol = 4 * ULP * rec["scale"] + 1e-6 * abs(rec["delta_energy_ref"]) + 1e-9
        assert abs(rec["delta_energy"] - rec["delta_energy_ref"]) <= tol, rec
        assert rec["rejected"] == rec["rejected_ref"], rec
        if rec["T"] > 0:
            assert abs(rec["log_acc"] - rec["log_acc_ref"]) <= tol / rec["T"], rec
            rec["margin"] = abs(rec["log_u"] - rec["log_acc_ref"])
            assert rec["margin"] > tol / rec["T"], ("inconclusive: decision inside the dE tolerance", rec)
