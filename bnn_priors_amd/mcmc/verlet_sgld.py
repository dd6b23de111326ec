"""VerletSGLD (GGMC: OBABO-style merged integrator with M-H energy accounting)
on the HIP engine.  Drop-in for ``bnn_priors.mcmc.VerletSGLD``
(reference: bnn_priors/mcmc/verlet_sgld.py:7-197)."""
import math

import torch

from .. import _hip
from .sgld import SGLD, dot

__all__ = ("VerletSGLD",)


class VerletSGLD(SGLD):
    _KIND = _hip.VERLET

    # ------------------------------------------------------------------ scalars
    def _update_group_fn(self, g):
        # verlet_sgld.py:138-146 (intermediate transition)
        g['b^2h^2'] = g['lr'] / g['num_data']
        g['bh'] = math.sqrt(g['b^2h^2'])
        g['bhn'] = math.sqrt(g['lr'] * g['num_data'])
        a = g['momentum']
        g['mom_decay'] = a
        g['grad_v'] = 1 + a
        g['noise_std'] = math.sqrt((1 - a ** 2) * g['temperature'])

    def _edge_group_fn(self, g, is_initial):
        # verlet_sgld.py:96-101 (initial) and :129-134 (final)
        self._update_group_fn(g)
        a = g['momentum']
        g['mom_decay'] = math.sqrt(a)
        g['grad_v'] = 1. if is_initial else g['mom_decay']
        g['noise_std'] = math.sqrt((1 - a) * g['temperature'])

    @staticmethod
    def _args_of(g):
        return dict(num_data=g['num_data'], b2h2=g['b^2h^2'], bh=g['bh'], bhn=g['bhn'],
                    mom_decay=g['mom_decay'], grad_v=g['grad_v'], noise_std=g['noise_std'],
                    rmsprop_alpha=g['rmsprop_alpha'])

    def _plain_step_spec(self, calc_metrics):
        g = self.param_groups[0]
        self._update_group_fn(g)
        return self._KIND, (_hip.CALC_METRICS if calc_metrics else 0), self._args_of(g)

    def _transition(self, closure, flags, group_fn):
        loss = self._run_closure(closure)

        def scalars(g):
            group_fn(g)
            return self._args_of(g)
        self._launch(self._KIND, flags, scalars)
        eng = self._engine
        eng.energy_ready = True
        if flags & _hip.CALC_METRICS:
            eng.metrics_ready = True
        if flags & _hip.SAVE_STATE:
            pt, pg, pm = eng.views("prev_theta"), eng.views("prev_g"), eng.views("prev_m")
            for i, p in enumerate(eng.params):
                st = self.state[p]
                if dict.get(st, 'prev_parameter') is not pt[i]:
                    st['prev_parameter'], st['prev_grad'], st['prev_momentum_buffer'] = pt[i], pg[i], pm[i]
        return loss

    # ------------------------------------------------------------------ reference API
    @torch.no_grad()
    def initial_step(self, closure=None, save_state=True, calc_metrics=True):
        "theta(n), m(n) -> theta(n+1), u(n+1)   (verlet_sgld.py:85-104)"
        self._step_count = getattr(self, '_step_count', 0) + 1
        flags = (_hip.INITIAL | (_hip.SAVE_STATE if save_state else 0)
                 | (_hip.CALC_METRICS if calc_metrics else 0))
        return self._transition(closure, flags, lambda g: self._edge_group_fn(g, True))

    @torch.no_grad()
    def step(self, closure=None, calc_metrics=True):
        "theta(n), u(n) -> theta(n+1), u(n+1)   (verlet_sgld.py:106-116)"
        return self._transition(closure, _hip.CALC_METRICS if calc_metrics else 0,
                                self._update_group_fn)

    @torch.no_grad()
    def final_step(self, closure=None, calc_metrics=True):
        "theta(n), u(n) -> theta(n), m(n)   (verlet_sgld.py:118-136)"
        self._step_count = getattr(self, '_step_count', 0) + 1
        flags = _hip.FINAL | (_hip.CALC_METRICS if calc_metrics else 0)
        return self._transition(closure, flags, lambda g: self._edge_group_fn(g, False))

    def _point_energy(self, group, p, state):
        # verlet_sgld.py:44-47 (API used by testing/test_verlet_sgld.py:190-196)
        M = state.setdefault('preconditioner', 1.)
        return (M ** 2 * group['num_data'] ** 2 * group['b^2h^2'] / 8) * dot(p.grad, p.grad)

    def _energy_kernel_args(self):
        g = self.param_groups[0]
        return g['num_data'], g.get('b^2h^2', g['lr'] / g['num_data'])

    def _energy_total_per_group(self):
        """several parameter groups (different lr): per-segment reductions on the device, the few
        per-tensor products of verlet_sgld.py:32-47 on the host in double"""
        eng = self._engine
        dots = eng.segment_sums(1 if self._KIND == _hip.HMC else 2)
        st = eng.fetch_state()
        col = _hip.SEG_STATE_FIELDS.index("delta_energy")
        total, i = 0., 0
        for g in self.param_groups:
            b2h2 = g.get('b^2h^2', g['lr'] / g['num_data'])
            for p in g['params']:
                if self._KIND == _hip.HMC:
                    point = .5 * dots[i]
                else:
                    M = self.state[p].setdefault('preconditioner', 1.)
                    point = (M ** 2 * g['num_data'] ** 2 * b2h2 / 8) * dots[i]
                total += float(st[i, col]) + point
                i += 1
        return total

    @torch.no_grad()
    def delta_energy(self, prev_potential, potential) -> float:
        """Energy difference since the last ``initial_step`` (verlet_sgld.py:27-42): one
        reduction over the current gradient (or momentum, HMC), summed in tensor order."""
        num_data = self.param_groups[0]['num_data']
        assert all(g['num_data'] == num_data for g in self.param_groups), \
            "unclear which `num_data` to use"
        if not self._engine.energy_ready:
            raise KeyError('delta_energy')
        eng = self._engine
        eng.refresh(self._preconditioners(), raise_on_no_grad=True)
        self._adopt_foreign_momentum()
        if len(self.param_groups) > 1:
            total = self._energy_total_per_group()
        else:
            n, b2h2 = self._energy_kernel_args()
            total = eng.delta_energy_total(self._KIND, n, b2h2, self.grad_clamp)
        if isinstance(potential, torch.Tensor):
            potential = potential.item()
        return total + (potential - prev_potential) * num_data

    def delta_energy_from_total(self, energy_total, prev_potential, potential):
        return energy_total + (potential - prev_potential) * self.param_groups[0]['num_data']

    def delta_energy_of_last_transition(self, prev_potential, potential):
        """``delta_energy`` when ``p.grad`` (HMC: the momentum) has not changed since the last
        ``initial_step`` / ``step`` / ``final_step``: the fused launch already summed
        ``state['delta_energy'] + _point_energy`` over the tensors, so no reduction is launched.
        Falls back to ``delta_energy`` for big arenas.  ``potential`` may be device tensors; the
        result is then a 0-d float64 device tensor (no sync)."""
        eng = self._engine
        if not eng.small_finalize:
            return self.delta_energy(prev_potential, potential)
        if not eng.energy_ready:
            raise KeyError('delta_energy')
        num_data = self.param_groups[0]['num_data']
        total = eng.last_transition_energy()
        if isinstance(potential, torch.Tensor) or isinstance(prev_potential, torch.Tensor):
            return total + (potential - prev_potential) * num_data
        return total.item() + (potential - prev_potential) * num_data

    @torch.no_grad()
    def maybe_reject(self, delta_energy):
        """Metropolis-Hastings test (verlet_sgld.py:49-70).  The uniform is the Philox
        spec's M-H draw; ``log`` and the comparison run on the host in double."""
        temperature = self.param_groups[0]['temperature']
        assert all(g['temperature'] == temperature for g in self.param_groups), \
            "unclear which `temperature` to use"
        if temperature == 0.0:
            return False, 0.
        log_accept_prob = -delta_energy / temperature
        reject = math.log(self._engine.mh_uniform()) > log_accept_prob
        if reject:
            eng = self._engine
            if eng.prev_theta is None:
                raise KeyError('prev_parameter')
            eng.refresh(self._preconditioners(), raise_on_no_grad=True)
            eng.restore(restore_momentum=any(g['momentum'] > 0 for g in self.param_groups))
        return reject, log_accept_prob
