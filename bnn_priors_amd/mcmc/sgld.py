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
        for i, p in enumerate(self._engine.params):
            self.state[p]['momentum_buffer'] = self._engine.momentum_view(i)

    def _run_closure(self, closure):
        if closure is None:
            return None
        with torch.enable_grad():
            return closure()

    def _adopt_foreign_momentum(self):
        """tests and users may rebind state['momentum_buffer'] to their own tensor;
        copy such a tensor into the arena and restore the view."""
        eng = self._engine
        for i, p in enumerate(eng.params):
            st = self.state[p]
            for key, view_of in (('momentum_buffer', eng.momentum_view), ('square_avg', eng.square_avg_view)):
                t = dict.get(st, key)
                if t is None:
                    continue
                view = view_of(i)
                if t.data_ptr() != view.data_ptr():     # e.g. after load_state_dict
                    view.copy_(t)
                    dict.__setitem__(st, key, view)

    def _check_nan(self):
        if self.raise_on_nan and self._engine.nonfinite_seen():
            raise ValueError("Gradient is not finite")

    def _launch(self, kind, flags, group_scalars, needs_momentum=True):
        """refresh tables, then one fused launch per parameter group"""
        eng = self._engine
        eng.refresh(self._preconditioners(), raise_on_no_grad=self.raise_on_no_grad)
        if needs_momentum and not eng.momentum_ready:
            raise RuntimeError("No 'momentum_buffer' stored in state. "
                               "Perhaps you forgot to call `sample_momentum`?")
        if any(p.grad is None for p in eng.params):
            # raise_on_no_grad=False: the reference skips such tensors (sgld.py:96-100);
            # the fused sweep cannot, so give them a zero gradient for this launch
            for p in eng.params:
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
            eng.refresh(self._preconditioners())
        self._adopt_foreign_momentum()
        draw = eng.next_draw()
        for gi, group in enumerate(self.param_groups):
            sc = group_scalars(group)
            eng.step(gi, kind, flags, draw, grad_clamp=self.grad_clamp, **sc)
        self._check_nan()

    # ------------------------------------------------------------------ fused priors
    def fuse_priors(self, model):
        """Let the engine differentiate the element-wise priors of ``model`` in one kernel
        (``add_prior_gradient``) instead of autograd.  Returns the Prior modules that can NOT be
        fused (tensor-valued or learnable loc/scale, other families): their ``log_prob`` must stay
        in the autograd potential.  Parameters without a prior (e.g. BatchNorm) get none."""
        from ..prior import named_priors
        by_param = {id(pr.p): pr for _, pr in named_priors(model)}
        specs, leftover = [], []
        for p in self._engine.params:
            pr = by_param.pop(id(p), None)
            sp = pr.fused_spec() if pr is not None else None
            if pr is not None and sp is None:
                leftover.append(pr)
            specs.append(sp)
        leftover.extend(by_param.values())   # priors whose .p this optimizer does not own
        self._engine.set_priors(specs)
        self._fused_any = any(sp is not None for sp in specs)
        return leftover

    @torch.no_grad()
    def add_prior_gradient(self, calc_log_prior=False):
        """p.grad += d/dtheta[-log p(theta)/N] for every fused prior (one launch); with
        ``calc_log_prior`` the summed log-density is left on the device (``fused_log_prior()``)."""
        eng = self._engine
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
