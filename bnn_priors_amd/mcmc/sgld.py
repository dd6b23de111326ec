"""SGLD (SGHMC with symplectic-Euler integration) on the HIP engine.

Drop-in for ``bnn_priors.mcmc.SGLD`` (reference: bnn_priors/mcmc/sgld.py:14-179):
same constructor, ``sample_momentum`` / ``step`` / ``initial_step`` /
``final_step`` / ``delta_energy`` / ``update_preconditioner``, same
``param_groups`` keys and ``state[p]`` keys, same exceptions.  The per-tensor
Python loop with its 6 ATen launches and up to 2 ``.item()`` syncs per tensor is
replaced by ONE fused kernel launch per parameter group and no sync.

Extensions (keyword-only, defaults keep the reference behaviour):
``seed`` / ``chain_id`` key the in-kernel Philox noise (default seed: drawn from
torch's global CPU generator, so ``torch.manual_seed`` makes runs repeatable).
"""
import math

import torch

from .. import _hip
from .engine import Engine, SegState

__all__ = ("SGLD", "dot")


def dot(a, b):
    "(a*b).sum() as a Python float (reference: mcmc/sgld.py:9-11)"
    return (a.reshape(-1) @ b.reshape(-1)).item()


class SGLD(torch.optim.Optimizer):
    """SGLD with momentum, preconditioning and temperature diagnostics
    (Wenzel et al. 2020), fused HIP implementation.

    Args: as the reference (mcmc/sgld.py:17-34): params, lr, num_data,
    momentum=0, temperature=1., rmsprop_alpha=0.99, rmsprop_eps=1e-8,
    raise_on_no_grad=True, raise_on_nan=False.
    """
    _KIND = _hip.SGLD

    def __init__(self, params, lr, num_data, momentum=0, temperature=1.,
                 rmsprop_alpha=0.99, rmsprop_eps=1e-8, raise_on_no_grad=True,
                 raise_on_nan=False, *, seed=None, chain_id=0, grad_clamp=0.0, **engine_options):
        assert lr >= 0 and num_data >= 0 and momentum >= 0 and temperature >= 0
        defaults = dict(lr=lr, num_data=num_data, momentum=momentum, rmsprop_alpha=rmsprop_alpha,
                        rmsprop_eps=rmsprop_eps, temperature=temperature)
        super().__init__(params, defaults)
        self.raise_on_no_grad = raise_on_no_grad
        self.raise_on_nan = raise_on_nan
        self.grad_clamp = float(grad_clamp)
        self._engine = Engine(self.param_groups, seed=seed, chain_id=chain_id, **engine_options)
        for i, p in enumerate(self._engine.params):
            st = self.state[p] = SegState(self._engine, i)
            st['square_avg'] = self._engine.square_avg_view(i)
        self.update_preconditioner()
        self._step_count = 0  # keeps torch.optim.lr_scheduler happy (sgld.py:45)

    # ------------------------------------------------------------------ helpers
    @property
    def engine(self):
        return self._engine

    def _preconditioners(self):
        return [self.state[p].setdefault('preconditioner', 1.) for p in self._engine.params]

    def _install_momentum_views(self):
        views = self._engine.views("m")
        for i, p in enumerate(self._engine.params):
            st = self.state[p]
            if dict.get(st, 'momentum_buffer') is not views[i]:
                st['momentum_buffer'] = views[i]

    def _run_closure(self, closure):
        if closure is None:
            return None
        with torch.enable_grad():
            return closure()

    def _adopt_foreign_momentum(self):
        """tests and users may rebind state['momentum_buffer'] to their own tensor;
        copy such a tensor into the arena and restore the view."""
        eng = self._engine
        vm, vv = eng.views("m"), eng.views("v")
        for i, p in enumerate(eng.params):
            st = self.state[p]
            for key, view in (('momentum_buffer', vm[i]), ('square_avg', vv[i])):
                t = dict.get(st, key)
                if t is None or t is view:              # (the cached view itself: nothing was rebound)
                    continue
                if t.data_ptr() != view.data_ptr():     # e.g. after load_state_dict
                    view.copy_(t)
                dict.__setitem__(st, key, view)

    # set by the runners: the non-finite flag stays on the device and is tested at metric steps / epoch
    # ends (inference.SGLDRunner._check_finite) instead of with one host sync after every launch
    defer_nan_check = False

    def _check_nan(self):
        if self.raise_on_nan and not self.defer_nan_check and self._engine.nonfinite_seen():
            raise ValueError("Gradient is not finite")

    def _launch(self, kind, flags, group_scalars, needs_momentum=True):
        """refresh tables, then one fused launch per parameter group"""
        eng = self._engine
        eng.refresh(self._preconditioners(), raise_on_no_grad=self.raise_on_no_grad)
        if needs_momentum and not eng.momentum_ready:
            raise RuntimeError("No 'momentum_buffer' stored in state. "
                               "Perhaps you forgot to call `sample_momentum`?")
        # raise_on_no_grad=False: tensors whose grad is None carry a null gradient pointer in the segment
        # table and the kernels leave them untouched, as the reference's `continue` does (sgld.py:96-100)
        self._adopt_foreign_momentum()
        draw = eng.next_draw()
        for gi, group in enumerate(self.param_groups):
            sc = group_scalars(group)
            eng.step(gi, kind, flags, draw, grad_clamp=self.grad_clamp, **sc)
        self._check_nan()

    # ------------------------------------------------------------------ checkpointing
    # (The reference never checkpoints its optimizer; torch's own state_dict / load_state_dict would lose
    #  what lives on the device: the per-tensor running scalars, and the arena-backed SegState wrappers.)
    _TENSOR_KEYS = ("momentum_buffer", "square_avg", "prev_parameter", "prev_grad", "prev_momentum_buffer")

    def state_dict(self):
        """``torch.optim.Optimizer.state_dict`` plus the device-resident per-tensor scalars
        (``delta_energy``, ``prev_new_momentum_delta``, ``est_temperature``, ``est_config_temp``) as plain
        floats in each tensor's state, and the engine's Philox position under ``"sgmcmc_engine"``."""
        eng = self._engine
        eng.flush()
        lazy = {}
        for i, p in enumerate(eng.params):
            st = self.state[p]
            extra = {k: st[k] for k in SegState._LAZY if k in st}
            if extra:
                lazy[i] = extra
        for i, extra in lazy.items():               # materialise, snapshot, remove again
            for k, v in extra.items():
                dict.__setitem__(self.state[eng.params[i]], k, v)
        try:
            sd = super().state_dict()
            sd["state"] = {k: {kk: (vv.clone() if isinstance(vv, torch.Tensor) else vv) for kk, vv in v.items()}
                           for k, v in sd["state"].items()}
        finally:
            for i, extra in lazy.items():
                for k in extra:
                    dict.pop(self.state[eng.params[i]], k, None)
        sd["sgmcmc_engine"] = dict(seed=eng.seed, chain_id=eng.chain_id, draw=eng.draw,
                                   momentum_ready=eng.momentum_ready, energy_ready=eng.energy_ready,
                                   metrics_ready=eng.metrics_ready)
        return sd

    def load_state_dict(self, state_dict):
        """Restores a ``state_dict()``: tensors are copied INTO the arenas (the views stay the state's
        tensors), the running scalars go back to the device array, the Philox counter continues."""
        import numpy as np
        state_dict = dict(state_dict)
        meta = state_dict.pop("sgmcmc_engine", None)
        super().load_state_dict(state_dict)
        eng = self._engine
        eng.flush()
        host = eng.state_dev.cpu().numpy().reshape(eng.n_seg, -1).copy()
        views = dict(momentum_buffer=eng.momentum_view, square_avg=eng.square_avg_view)
        for i, p in enumerate(eng.params):
            loaded = dict(self.state.get(p, {}))
            st = self.state[p] = SegState(eng, i)
            if any(k.startswith("prev_") for k in loaded):
                eng.ensure_prev()
            prev = dict(prev_parameter=lambda j: eng._view(eng.prev_theta, j),
                        prev_grad=lambda j: eng._view(eng.prev_g, j),
                        prev_momentum_buffer=lambda j: eng._view(eng.prev_m, j))
            for k, v in loaded.items():
                if k in self._TENSOR_KEYS:
                    view = (views.get(k) or prev[k])(i)
                    view.copy_(v)
                    dict.__setitem__(st, k, view)
                elif k in SegState._LAZY:
                    host[i, _hip.SEG_STATE_FIELDS.index(SegState._LAZY[k][0])] = float(v)
                else:
                    st[k] = v
            st.setdefault('square_avg', eng.square_avg_view(i))
        eng.state_dev.copy_(torch.from_numpy(np.ascontiguousarray(host).reshape(-1)))
        eng._touch()
        eng._precond_dirty = eng._seg_dirty = True
        any_loaded = [dict.__contains__(self.state[p], "momentum_buffer") for p in eng.params]
        eng.momentum_ready = all(any_loaded) and bool(any_loaded)
        if meta is not None:
            eng.seed, eng.chain_id, eng.draw = int(meta["seed"]), int(meta["chain_id"]), int(meta["draw"])
            eng.energy_ready, eng.metrics_ready = bool(meta["energy_ready"]), bool(meta["metrics_ready"])
            eng.momentum_ready = eng.momentum_ready and bool(meta["momentum_ready"])

    # ------------------------------------------------------------------ fused priors
    def fuse_priors(self, model):
        """Let the engine differentiate the element-wise priors of ``model`` in one kernel
        (``add_prior_gradient``) instead of autograd.  Returns the Prior modules that can NOT be
        fused (tensor-valued or learnable loc/scale, other families): their ``log_prob`` must stay
        in the autograd potential.  Parameters without a prior (e.g. BatchNorm) get none.
        Hierarchical priors (``prior/hierarchical.py``): the weight tensor's row is linked to the
        segment of its one-element scale hyper-parameter; that parameter only ever receives a gradient
        from the priors, so ``add_prior_gradient`` gives it a zeroed ``.grad`` to accumulate into."""
        from ..prior import named_priors
        eng = self._engine
        # (a mixture's components share its tensor and own no density: prior/mixture.py)
        by_param = {id(pr.p): pr for _, pr in named_priors(model) if not getattr(pr, "is_component", False)}
        priors = [by_param.pop(id(p), None) for p in eng.params]
        specs = [pr.fused_spec() if pr is not None else None for pr in priors]
        links = [None] * len(specs)
        claimed = {}
        for i, pr in enumerate(priors):
            hyper = pr.scale_link() if (pr is not None and specs[i] is not None) else None
            if hyper is None:
                continue
            h = eng.index.get(id(hyper.p))
            # the hyper segment must be this optimizer's, fused itself, and feed exactly one tensor
            if h is None or specs[h] is None or h in claimed:
                specs[i] = None
                continue
            links[i] = claimed[h] = h
        leftover = [pr for pr, sp in zip(priors, specs) if pr is not None and sp is None]
        leftover.extend(by_param.values())   # priors whose .p this optimizer does not own
        eng.set_priors(specs, links)
        self._hyper_params = [eng.params[h] for h in sorted(claimed)]
        self._hyper_grads = [torch.zeros_like(p) for p in self._hyper_params]
        self._fused_any = any(sp is not None for sp in specs)
        return leftover

    def _prepare_hyper_grads(self):
        "scale hyper-parameters take no likelihood gradient: a persistent zeroed buffer stands in for None"
        hyper = getattr(self, "_hyper_params", None)
        if not hyper:
            return
        fresh = [g for p, g in zip(hyper, self._hyper_grads) if p.grad is None]
        if fresh:
            torch._foreach_zero_(fresh)
        for p, g in zip(hyper, self._hyper_grads):
            if p.grad is None:
                p.grad = g

    @torch.no_grad()
    def add_prior_gradient(self, calc_log_prior=False):
        """p.grad += d/dtheta[-log p(theta)/N] for every fused prior (one launch); with
        ``calc_log_prior`` the summed log-density is left on the device (``fused_log_prior()``)."""
        eng = self._engine
        self._prepare_hyper_grads()
        eng.refresh(self._preconditioners(), raise_on_no_grad=True)
        nd = self.param_groups[0]['num_data']
        assert all(g['num_data'] == nd for g in self.param_groups), "unclear which `num_data` to use"
        eng.prior_grad(nd, calc_log_prior)

    def fused_log_prior(self):
        "0-d float64 device tensor: log-density of the fused priors at the last add_prior_gradient"
        return self._engine.log_prior_total()

    # ------------------------------------------------------------------ reference API
    def delta_energy(self, a, b) -> float:
        return math.inf  # sgld.py:54-55

    def delta_energy_from_total(self, energy_total, prev_potential, potential):
        "delta_energy given the fused launch's energy total (graphed.py); SGLD has none"
        return math.inf

    @torch.no_grad()
    def sample_momentum(self, keep=0.0):
        "m <- sqrt(keep) m + sqrt(T (1-keep)) xi   (sgld.py:57-69)"
        assert 0 <= keep and keep <= 1.
        if keep == 1.:
            return
        eng = self._engine
        temps = {g['temperature'] for g in self.param_groups}
        if len(temps) != 1:
            raise NotImplementedError("sample_momentum with per-group temperatures")
        if keep != 0.0:
            self._adopt_foreign_momentum()
        eng.sample_momentum(math.sqrt(temps.pop() * (1 - keep)), float(keep), eng.next_draw())
        self._install_momentum_views()

    def _sgld_scalars(self, g):
        # sgld.py:114-117
        g['hn'] = math.sqrt(g['lr'] * g['num_data'])
        g['h'] = math.sqrt(g['lr'] / g['num_data'])
        g['noise_std'] = math.sqrt(2 * (1 - g['momentum']) * g['temperature'])
        return dict(num_data=g['num_data'], b2h2=g['lr'] / g['num_data'], bh=g['h'], bhn=g['hn'],
                    mom_decay=g['momentum'], grad_v=1.0,
                    noise_std=g['noise_std'] if g['temperature'] > 0 else 0.0,
                    rmsprop_alpha=g['rmsprop_alpha'])

    def _plain_step_spec(self, calc_metrics):
        """(kind, flags, scalars) of an ordinary ``step`` with the CURRENT lr / temperature --
        used by graphed.py to parameterise a replay without launching anything."""
        g = self.param_groups[0]
        has_mom = g['momentum'] > 0
        flags = (_hip.CALC_METRICS if calc_metrics else 0) | (0 if has_mom else _hip.NO_MOMENTUM)
        return _hip.SGLD, flags, self._sgld_scalars(g)

    def _sgld_transition(self, closure, calc_metrics, is_final):
        loss = self._run_closure(closure)
        moms = {g['momentum'] > 0 for g in self.param_groups}
        if len(moms) != 1:
            raise NotImplementedError("mixing momentum == 0 and momentum > 0 groups")
        has_mom = moms.pop()
        if not has_mom and is_final and calc_metrics:
            # the reference hits an unbound local here (sgld.py:132-137)
            raise UnboundLocalError("SGLD(momentum=0).final_step(calc_metrics=True) is undefined "
                                    "in the reference (mcmc/sgld.py:132-137)")
        flags = ((_hip.FINAL if is_final else 0) | (_hip.CALC_METRICS if calc_metrics else 0)
                 | (0 if has_mom else _hip.NO_MOMENTUM))
        self._launch(_hip.SGLD, flags, self._sgld_scalars, needs_momentum=has_mom)
        if calc_metrics:
            self._engine.metrics_ready = True
        return loss

    @torch.no_grad()
    def step(self, closure=None, calc_metrics=True, save_state=False):
        assert save_state is False
        return self._sgld_transition(closure, calc_metrics, False)
    initial_step = step

    @torch.no_grad()
    def final_step(self, closure=None, calc_metrics=True, save_state=False):
        assert save_state is False
        return self._sgld_transition(closure, calc_metrics, True)

    @torch.no_grad()
    def update_preconditioner(self):
        """M_p = ((mean(v_p)+eps) / min_q(mean(v_q)+eps))^(-1/4)   (sgld.py:156-179).
        The means come from one fused reduction over the arena (fp64 accumulate)."""
        eng = self._engine
        eng.refresh(self._preconditioners(), need_grad=False)
        sums = eng.segment_sums(0)
        precond, smallest, i = [], math.inf, 0
        for group in self.param_groups:
            for p in group['params']:
                s = sums[i] / p.numel() + group['rmsprop_eps']
                precond.append(s)
                smallest = min(smallest, s)
                i += 1
        for p, s in zip(eng.params, precond):
            self.state[p]['preconditioner'] = (s / smallest) ** (-1 / 4)
