"""HMC with leapfrog integration on the HIP engine.  Drop-in for
``bnn_priors.mcmc.HMC`` (reference: bnn_priors/mcmc/hmc.py:10-79): VerletSGLD with
momentum = 1, temperature = 1, no noise draw and kinetic-energy M-H accounting."""
from .. import _hip
from .sgld import dot
from .verlet_sgld import VerletSGLD

__all__ = ("HMC",)


class HMC(VerletSGLD):
    _KIND = _hip.HMC

    def __init__(self, params, lr, num_data, raise_on_no_grad=True, raise_on_nan=True, **kw):
        super().__init__(params, lr, num_data, 1., 1., raise_on_no_grad=raise_on_no_grad,
                         raise_on_nan=raise_on_nan, **kw)
        # hmc.py:41-79 never writes state['prev_new_momentum_delta']
        self._engine.hidden_keys = frozenset({'prev_new_momentum_delta'})

    def _update_group_fn(self, g):
        super()._update_group_fn(g)
        assert g['momentum'] == 1. and g['temperature'] == 1.  # hmc.py:39

    def _point_energy(self, group, p, state):
        return .5 * dot(state['momentum_buffer'], state['momentum_buffer'])  # hmc.py:32-33
