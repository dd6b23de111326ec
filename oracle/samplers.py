"""Per-tensor PyTorch-CPU restatement of the reference samplers (TEST INFRASTRUCTURE).

``RefSGLD`` / ``RefVerletSGLD`` / ``RefHMC`` restate, tensor by tensor and in the
reference's operation order, what ``bnn_priors/mcmc/sgld.py``,
``verlet_sgld.py`` and ``hmc.py`` compute (equations: SURVEY.md Appendix A).
They are the parity checker for the HIP path and, in ``bench.py``, the timed
"reference-equivalent CPU path" (``cpu_baseline.kind == "port"``): one eager
op sequence and 2-4 ``.item()`` dots per tensor per step, as the reference
issues them.

``noise=None`` draws from torch's global generator exactly where the reference
does (so a run under ``torch.manual_seed`` is comparable bit for bit with the
imported reference); ``noise=NoiseSource(...)`` consumes the Philox spec, which
the HIP kernels implement in-kernel.

Never imported by ``bnn_priors_amd``.
"""
import math

import torch

from .noise import PURPOSE_MOMENTUM, PURPOSE_STEP

INITIAL, MIDDLE, FINAL = "initial", "middle", "final"


def _dot(a, b):
    # reference helper: mcmc/sgld.py:9-11
    return torch.dot(a.reshape(-1), b.reshape(-1)).item()


class _RefBase(torch.optim.Optimizer):
    """Constructor / momentum refresh / preconditioner shared by the three samplers
    (reference: mcmc/sgld.py:31-69,156-179)."""

    def __init__(self, params, lr, num_data, momentum=0, temperature=1.,
                 rmsprop_alpha=0.99, rmsprop_eps=1e-8,
                 raise_on_no_grad=True, raise_on_nan=False, noise=None):
        assert lr >= 0 and num_data >= 0 and momentum >= 0 and temperature >= 0
        super().__init__(params, dict(lr=lr, num_data=num_data, momentum=momentum,
                                      rmsprop_alpha=rmsprop_alpha, rmsprop_eps=rmsprop_eps,
                                      temperature=temperature))
        self.raise_on_no_grad = raise_on_no_grad
        self.raise_on_nan = raise_on_nan
        self.noise = noise
        self.update_preconditioner()
        self._step_count = 0

    # -- noise plumbing ------------------------------------------------------
    def _all_params(self):
        return [p for g in self.param_groups for p in g['params']]

    def _index_of(self, p):
        try:
            table = self._param_index
        except AttributeError:
            table = self._param_index = {id(q): i for i, q in enumerate(self._all_params())}
        return table[id(p)]

    def _normal_like(self, p, draw, purpose):
        if self.noise is None:
            return torch.randn_like(p)
        return self.noise.tensor_normals(draw, purpose, self._index_of(p), p)

    def _begin_sweep(self):
        return None if self.noise is None else self.noise.begin_sweep()

    # -- reference API -------------------------------------------------------
    def _precond(self, state):
        return state.setdefault('preconditioner', 1.)

    def delta_energy(self, a, b):
        return math.inf  # mcmc/sgld.py:54-55

    @torch.no_grad()
    def sample_momentum(self, keep=0.0):
        # mcmc/sgld.py:57-69
        assert 0 <= keep <= 1.
        if keep == 1.:
            return
        draw = self._begin_sweep()
        for group in self.param_groups:
            std = math.sqrt(group['temperature'] * (1 - keep))
            for p in group['params']:
                xi = self._normal_like(p, draw, PURPOSE_MOMENTUM)
                if keep == 0.0:
                    self.state[p]['momentum_buffer'] = xi.mul_(std)
                else:
                    self.state[p]['momentum_buffer'].mul_(math.sqrt(keep)).add_(xi, alpha=std)

    @torch.no_grad()
    def update_preconditioner(self):
        # mcmc/sgld.py:156-179
        means, smallest = [], math.inf
        for group in self.param_groups:
            for p in group['params']:
                st = self.state[p]
                if 'square_avg' not in st:
                    st['square_avg'] = torch.ones_like(p)
                s = st['square_avg'].mean().item() + group['rmsprop_eps']
                means.append((p, s))
                smallest = min(smallest, s)
        for p, s in means:
            self.state[p]['preconditioner'] = (s / smallest) ** (-1 / 4)

    def _sweep(self, group_scalars, tensor_update, closure, **kw):
        # driver loop: mcmc/sgld.py:88-112
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        draw = self._begin_sweep()
        for group in self.param_groups:
            group_scalars(group)
            for p in group['params']:
                if p.grad is None:
                    if self.raise_on_no_grad:
                        raise RuntimeError(f"No gradient for parameter with shape {p.shape}")
                    continue
                if self.raise_on_nan and not torch.isfinite(p.grad).all():
                    raise ValueError(f"Gradient of shape {p.shape} is not finite: {p.grad}")
                st = self.state[p]
                if kw.get('needs_momentum', True) and 'momentum_buffer' not in st:
                    raise RuntimeError("No 'momentum_buffer' stored in state. "
                                       "Perhaps you forgot to call `sample_momentum`?")
                tensor_update(group, p, st, draw)
        return loss

    def _rmsprop(self, group, p, st):
        a = group['rmsprop_alpha']
        st['square_avg'].mul_(a).addcmul_(p.grad, p.grad, value=1 - a)


class RefSGLD(_RefBase):
    """Symplectic-Euler SGHMC; mcmc/sgld.py:114-154."""

    def _scalars(self, g):
        g['hn'] = math.sqrt(g['lr'] * g['num_data'])
        g['h'] = math.sqrt(g['lr'] / g['num_data'])
        g['noise_std'] = math.sqrt(2 * (1 - g['momentum']) * g['temperature'])

    def _run(self, closure, calc_metrics, is_final):
        def update(group, p, st, draw):
            M, d, a = self._precond(st), p.numel(), group['momentum']
            if a > 0:
                mom = st['momentum_buffer']
                if calc_metrics:
                    st['est_temperature'] = _dot(mom, mom) / d
                if not is_final:
                    mom.mul_(a).add_(p.grad, alpha=-group['hn'] * M)
            else:
                mom = None
                if not is_final:
                    mom = p.grad.detach().mul(-group['hn'] * M)
                if calc_metrics:
                    # reference raises UnboundLocalError for a=0 & is_final (sgld.py:132-137)
                    st['est_temperature'] = _dot(mom, mom) / d
            if not is_final and group['temperature'] > 0:
                mom.add_(self._normal_like(p, draw, PURPOSE_STEP), alpha=group['noise_std'])
            if calc_metrics:
                st['est_config_temp'] = _dot(p, p.grad) * (group['num_data'] / d)
            if not is_final:
                p.add_(mom, alpha=group['h'] * M)
                self._rmsprop(group, p, st)
        needs_m = any(g['momentum'] > 0 for g in self.param_groups)
        return self._sweep(self._scalars, update, closure, needs_momentum=needs_m)

    @torch.no_grad()
    def step(self, closure=None, calc_metrics=True, save_state=False):
        assert save_state is False
        return self._run(closure, calc_metrics, False)
    initial_step = step

    @torch.no_grad()
    def final_step(self, closure=None, calc_metrics=True, save_state=False):
        assert save_state is False
        return self._run(closure, calc_metrics, True)


class RefVerletSGLD(_RefBase):
    """GGMC / "Verlet" SGLD; mcmc/verlet_sgld.py:27-197."""

    def _scalars(self, g, kind):
        # mcmc/verlet_sgld.py:138-146 with the per-kind overrides of :96-101,:129-134
        a, T = g['momentum'], g['temperature']
        g['b^2h^2'] = g['lr'] / g['num_data']
        g['bh'] = math.sqrt(g['b^2h^2'])
        g['bhn'] = math.sqrt(g['lr'] * g['num_data'])
        if kind == MIDDLE:
            g['mom_decay'], g['grad_v'] = a, 1 + a
            g['noise_std'] = math.sqrt((1 - a ** 2) * T)
        else:
            g['mom_decay'] = math.sqrt(a)
            g['grad_v'] = 1. if kind == INITIAL else g['mom_decay']
            g['noise_std'] = math.sqrt((1 - a) * T)

    def _check_group(self, g):
        pass

    def _point_energy(self, group, p, state):
        # mcmc/verlet_sgld.py:44-47
        M = self._precond(state)
        return (M ** 2 * group['num_data'] ** 2 * group['b^2h^2'] / 8) * _dot(p.grad, p.grad)

    def delta_energy(self, prev_potential, potential):
        # mcmc/verlet_sgld.py:27-42
        N = self.param_groups[0]['num_data']
        assert all(g['num_data'] == N for g in self.param_groups), "unclear which `num_data` to use"
        total = 0.
        for group in self.param_groups:
            for p in group['params']:
                st = self.state[p]
                total += st['delta_energy'] + self._point_energy(group, p, st)
        if isinstance(potential, torch.Tensor):
            potential = potential.item()
        return total + (potential - prev_potential) * N

    def _uniform(self):
        return torch.rand(()).item() if self.noise is None else self.noise.uniform()

    @torch.no_grad()
    def maybe_reject(self, delta_energy):
        # mcmc/verlet_sgld.py:49-70
        T = self.param_groups[0]['temperature']
        assert all(g['temperature'] == T for g in self.param_groups), "unclear which `temperature` to use"
        if T == 0.0:
            return False, 0.
        log_accept = -delta_energy / T
        reject = math.log(self._uniform()) > log_accept
        if reject:
            for p, st in self.state.items():
                p.data.copy_(st['prev_parameter'])
                p.grad.copy_(st['prev_grad'])
                if 'momentum_buffer' in st and 'prev_momentum_buffer' in st:
                    st['momentum_buffer'].copy_(st['prev_momentum_buffer'])
        return reject, log_accept

    def _save(self, group, p, st):
        # mcmc/verlet_sgld.py:72-83 (saved copies stay on the tensor's device here)
        for key, src in (('prev_parameter', p), ('prev_grad', p.grad)):
            if key in st:
                st[key].copy_(src)
            else:
                st[key] = src.detach().clone()
        if group['momentum'] > 0:
            if 'prev_momentum_buffer' in st:
                st['prev_momentum_buffer'].copy_(st['momentum_buffer'])
            else:
                st['prev_momentum_buffer'] = st['momentum_buffer'].detach().clone()

    def _update(self, kind, save_state, calc_metrics):
        def update(group, p, st, draw):
            # mcmc/verlet_sgld.py:149-197
            if save_state:
                self._save(group, p, st)
            M = self._precond(st)
            old = st['momentum_buffer']
            new = self._normal_like(p, draw, PURPOSE_STEP).mul_(group['noise_std'])
            new.add_(p.grad, alpha=-.5 * group['grad_v'] * group['bhn'] * M)
            if group['mom_decay'] > 0:
                new.add_(old, alpha=group['mom_decay'])
            c_gm = -.5 * group['bhn'] * M
            if kind == INITIAL:
                st['delta_energy'] = -self._point_energy(group, p, st)
            else:
                st['delta_energy'] += st['prev_new_momentum_delta']
                st['delta_energy'] += c_gm * _dot(p.grad, old)
            st['prev_new_momentum_delta'] = c_gm * _dot(p.grad, new)
            if calc_metrics:
                d = p.numel()
                which = new if kind == FINAL else old
                st['est_temperature'] = _dot(which, which) / d
                st['est_config_temp'] = _dot(p, p.grad) * (group['num_data'] / d)
            st['momentum_buffer'] = new
            if kind != FINAL:
                p.add_(new, alpha=group['bh'] * M)
                self._rmsprop(group, p, st)
        return update

    def _go(self, kind, closure, save_state, calc_metrics):
        def scalars(g):
            self._scalars(g, kind)
            self._check_group(g)
        return self._sweep(scalars, self._update(kind, save_state, calc_metrics), closure)

    @torch.no_grad()
    def initial_step(self, closure=None, save_state=True, calc_metrics=True):
        self._step_count = getattr(self, '_step_count', 0) + 1
        return self._go(INITIAL, closure, save_state, calc_metrics)

    @torch.no_grad()
    def step(self, closure=None, calc_metrics=True):
        return self._go(MIDDLE, closure, False, calc_metrics)

    @torch.no_grad()
    def final_step(self, closure=None, calc_metrics=True):
        self._step_count = getattr(self, '_step_count', 0) + 1
        return self._go(FINAL, closure, False, calc_metrics)


class RefHMC(RefVerletSGLD):
    """Leapfrog HMC (a = 1, T = 1, no noise draw); mcmc/hmc.py:25-79.  ``temperature`` != 1 is the build's
    extension for BASELINE configs[4] (the reference asserts T == 1): same leapfrog map, momentum refresh
    N(0, T) and acceptance exp(-dH/T) as the base class already implements them."""

    def __init__(self, params, lr, num_data, raise_on_no_grad=True, raise_on_nan=True, noise=None,
                 temperature=1.):
        self._tempered = temperature != 1.
        super().__init__(params, lr, num_data, 1., temperature, raise_on_no_grad=raise_on_no_grad,
                         raise_on_nan=raise_on_nan, noise=noise)

    def _check_group(self, g):
        assert g['momentum'] == 1. and (self._tempered or g['temperature'] == 1.)  # mcmc/hmc.py:39

    def _point_energy(self, group, p, state):
        return .5 * _dot(state['momentum_buffer'], state['momentum_buffer'])  # mcmc/hmc.py:32-33

    def _update(self, kind, save_state, calc_metrics):
        def update(group, p, st, draw):
            # mcmc/hmc.py:41-79
            if save_state:
                self._save(group, p, st)
            M, mom, d = self._precond(st), st['momentum_buffer'], p.numel()
            if kind == INITIAL:
                kin = _dot(mom, mom)
                st['delta_energy'] = -.5 * kin
                if calc_metrics:
                    st['est_temperature'] = kin / d
            if calc_metrics:
                if kind == MIDDLE:
                    st['est_temperature'] = _dot(mom, mom) / d
                st['est_config_temp'] = _dot(p, p.grad) * (group['num_data'] / d)
            mom.add_(p.grad, alpha=-.5 * group['grad_v'] * group['bhn'] * M)
            if kind == FINAL:
                if calc_metrics:
                    st['est_temperature'] = _dot(mom, mom) / d
            else:
                p.add_(mom, alpha=group['bh'] * M)
                self._rmsprop(group, p, st)
        return update
