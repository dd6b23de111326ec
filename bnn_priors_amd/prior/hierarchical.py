"""Hierarchical priors: an element-wise family whose SCALE is itself a sampled parameter with a
hyper-prior (reference: bnn_priors/prior/hierarchical.py:17-104).  The scale prior is a sub-module
(``<prior>.scale.p`` in the ``state_dict``), initialised so that its value equals the requested scale
(Gamma, half-Cauchy) or sits in the middle of [0, 2 scale] (uniform).

On the HIP path the weight tensor's prior reads its scale from the hyper-parameter's segment at launch
time and the chain-rule term for the hyper-parameter comes from one extra reduction
(``sgmcmc_prior_grad``, SGMCMC_PRIOR_HAS_LINKS); see ``Prior.fused_spec`` / ``Prior.scale_link``.
"""
import torch

from .loc_scale import GenNorm, Laplace, Normal, StudentT
from .transformed import Gamma, HalfCauchy, Uniform, inv_softplus

__all__ = ("NormalGamma", "NormalUniform", "LaplaceGamma", "LaplaceUniform", "StudentTGamma",
           "StudentTUniform", "GenNormUniform", "Horseshoe")


def _gamma_scale(scale, rate):
    hyper = Gamma(shape=[], concentration=scale, rate=rate)
    with torch.no_grad():
        hyper.p.data = inv_softplus(torch.tensor(scale))
    return hyper


def _uniform(upto):
    hyper = Uniform(shape=[], low=0., high=upto * 2.)
    with torch.no_grad():
        hyper.p.data = torch.tensor(0.)
    return hyper


class NormalGamma(Normal):
    def __init__(self, shape, loc, scale, rate=1., gradient_clip=1.):
        super().__init__(shape, loc, _gamma_scale(scale, rate))


class NormalUniform(Normal):
    def __init__(self, shape, loc, scale, gradient_clip=1.):
        super().__init__(shape, loc, _uniform(scale))


class LaplaceGamma(Laplace):
    def __init__(self, shape, loc, scale, rate=1., gradient_clip=1.):
        super().__init__(shape, loc, _gamma_scale(scale, rate))


class LaplaceUniform(Laplace):
    def __init__(self, shape, loc, scale, gradient_clip=1.):
        super().__init__(shape, loc, _uniform(scale))


class StudentTGamma(StudentT):
    def __init__(self, shape, loc, scale, rate=1., df=2, gradient_clip=1.):
        super().__init__(shape, loc, _gamma_scale(scale, rate), df=df)


class StudentTUniform(StudentT):
    def __init__(self, shape, loc, scale, df=2, gradient_clip=1.):
        super().__init__(shape, loc, _uniform(scale), df=df)


class GenNormUniform(GenNorm):
    "hyper-prior on the shape parameter beta (stays in autograd: the hook links scales only)"
    def __init__(self, shape, loc, scale, beta=1., gradient_clip=1.):
        super().__init__(shape, loc, scale, beta=_uniform(beta))


class Horseshoe(Normal):
    def __init__(self, shape, loc, scale, hyperscale=1., gradient_clip=1.):
        hyper = HalfCauchy(shape=[], scale=hyperscale, multiplier=scale)
        with torch.no_grad():
            hyper.p.data = inv_softplus(torch.tensor(1.))
        super().__init__(shape, loc, hyper)
