"""HMC with leapfrog integration on the HIP engine.  Drop-in for
``bnn_priors.mcmc.HMC`` (reference: bnn_priors/mcmc/hmc.py:10-79): VerletSGLD with
momentum = 1, temperature = 1, no noise draw and kinetic-energy M-H accounting.

Extension beyond the reference (keyword-only; BASELINE.json configs[4], the cold-posterior sweep):
``temperature=T`` samples exp(-U/T).  The leapfrog map is unchanged -- it integrates H = N*U + m.m/2
whatever T is -- and T enters exactly where the base class already puts it: the momentum refresh draws
m ~ N(0, T) (sgld.py:57-69) and the Metropolis-Hastings test accepts with min(1, exp(-dH/T))
(verlet_sgld.py:56-58), which leaves exp(-H/T) invariant.  The reference asserts T == 1 (hmc.py:39); with
the default ``temperature=1.`` this class does too."""
from .. import _hip
from .sgld import dot
from .verlet_sgld import VerletSGLD

__all__ = ("HMC",)


class HMC(VerletSGLD):
    _KIND = _hip.HMC

    def __init__(self, params, lr, num_data, raise_on_no_grad=True, raise_on_nan=True, *, temperature=1., **kw):
        self._tempered = temperature != 1.
        super().__init__(params, lr, num_data, 1., temperature, raise_on_no_grad=raise_on_no_grad,
                         raise_on_nan=raise_on_nan, **kw)
        # hmc.py:41-79 never writes state['prev_new_momentum_delta']
        self._engine.hidden_keys = frozenset({'prev_new_momentum_delta'})

    def _update_group_fn(self, g):
        super()._update_group_fn(g)
        assert g['momentum'] == 1.                                   # hmc.py:39
        assert self._tempered or g['temperature'] == 1.              # hmc.py:39 (unless constructed tempered)

    def _point_energy(self, group, p, state):
        return .5 * dot(state['momentum_buffer'], state['momentum_buffer'])  # hmc.py:32-33
