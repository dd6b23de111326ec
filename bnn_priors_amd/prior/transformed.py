"""Priors on a TRANSFORMED parameter: the sampled tensor ``p`` is unconstrained, ``forward()`` maps it
to the constrained value and ``log_prob()`` is the density of that value (no Jacobian term -- as the
reference defines them, bnn_priors/prior/transformed.py:13-87).  They serve as the hyper-priors of
hierarchical scales (``hierarchical.py``); with one element and plain-number arguments the HIP prior
hook differentiates them in closed form (``fused_kind`` / ``fused_spec``).
"""
import math

import torch
import torch.distributions as td
import torch.nn.functional as F

from .base import Prior

__all__ = ("Uniform", "Gamma", "HalfCauchy", "DoubleGamma", "inv_softplus")

FUSED_GAMMA_SOFTPLUS, FUSED_UNIFORM_CDF, FUSED_HALFCAUCHY_SOFTPLUS = 6, 7, 8


def inv_softplus(x):
    "softplus^-1(x) = x + log(1 - exp(-x))"
    x = torch.as_tensor(x)
    return x + torch.log(-torch.expm1(-x))


def _plain(t):
    return isinstance(t, torch.Tensor) and not isinstance(t, torch.nn.Parameter) and t.numel() == 1


class Uniform(Prior):
    "value = low + (high - low) * Phi(p), p ~ N(0, 1): uniform on [low, high]; log_prob is the constant density"
    _dist = td.Uniform
    fused_kind = FUSED_UNIFORM_CDF

    def __init__(self, shape, low, high):
        super().__init__(shape, low=low, high=high)

    def _draw(self, shape):
        return torch.randn(shape)

    def forward(self):
        return self.low + (self.high - self.low) * td.Normal(0., 1.).cdf(self.p)

    def log_prob(self):
        width = self.high - self.low
        if isinstance(width, float):
            return -math.log(width) * self.p.numel()
        lp = -torch.log(width)
        return lp.sum() * (self.p.numel() / lp.numel())

    def fused_spec(self):
        # (the hook knows this family as a hyper-prior: one element, evaluated in double; a whole tensor with a
        #  uniform prior -- get_prior("uniform") -- stays in autograd)
        if not (_plain(self.low) and _plain(self.high)) or self.p.numel() != 1:
            return None
        return self.fused_kind, float(self.low), float(self.high), 0.0


class Gamma(Prior):
    "value = softplus(p) ~ Gamma(concentration, rate)"
    _dist = td.Gamma
    fused_kind = FUSED_GAMMA_SOFTPLUS

    def __init__(self, shape, concentration, rate):
        super().__init__(shape, concentration=concentration, rate=rate)

    def _draw(self, shape):
        return inv_softplus(super()._draw(shape))

    def forward(self):
        return F.softplus(self.p)

    def log_prob(self):
        return self._dist_obj().log_prob(self()).sum()

    def fused_spec(self):
        if not (_plain(self.concentration) and _plain(self.rate)):
            return None
        return self.fused_kind, float(self.concentration), float(self.rate), 0.0


class HalfCauchy(Prior):
    "value = softplus(p) * multiplier, with softplus(p) * multiplier scored under HalfCauchy(scale)"
    _dist = td.HalfCauchy
    fused_kind = FUSED_HALFCAUCHY_SOFTPLUS

    def __init__(self, shape, scale=1., multiplier=1.):
        super().__init__(shape, scale=scale)
        self.multiplier = multiplier

    def _draw(self, shape):
        return inv_softplus(super()._draw(shape))

    def forward(self):
        return F.softplus(self.p) * self.multiplier

    def log_prob(self):
        return self._dist_obj().log_prob(self()).sum()

    def fused_spec(self):
        if not (_plain(self.scale) and isinstance(self.multiplier, (int, float))):
            return None
        return self.fused_kind, float(self.multiplier), float(self.scale), 0.0


class DoubleGamma(Prior):
    """p - loc ~ DoubleGamma(concentration, rate = 1 / scale): the element-wise marginal of the reference's data-driven
    prior ``datadrivencorrdoublegamma`` (prior/transformed.py:83-95).  Not a family of the HIP hook: autograd."""
    fused_kind = None

    def __init__(self, shape, loc, scale, concentration):
        super().__init__(shape, loc=loc, scale=scale, concentration=concentration)

    def _dist(self, loc, scale, concentration):
        from .distributions import DoubleGamma as D
        return D(concentration=concentration, rate=1 / scale)

    def _draw(self, shape):
        return super()._draw(shape) + self.loc

    def log_prob(self):
        return self._dist_obj().log_prob(self.p - self.loc).sum()
