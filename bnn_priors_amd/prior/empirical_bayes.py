"""Empirical-Bayes priors: an element-wise family whose scale (and shape parameter) is a LEARNABLE parameter without a
hyper-prior -- ``softplus(p)`` of a ``PositiveImproper`` sub-module initialised at the requested value (reference:
bnn_priors/prior/empirical_bayes.py:14-58).  ``state_dict`` keys as the reference's: ``p``, ``loc``, ``scale.p`` /
``scale.loc`` / ``scale.scale`` (+ ``df.*`` / ``beta.*`` / ``lengthscale.*``).

On the HIP path ``NormalEmpirical`` and ``LaplaceEmpirical`` are the hierarchical-scale mechanism of ``hierarchical.py``
with a hyper-prior kind that has no density (``PositiveImproper.fused_spec``): the tensor's prior reads its scale from the
hyper segment at launch time and the hyper-parameter's gradient is the chain-rule term of ``sgmcmc_prior_grad``'s ``dls``
reduction.  ``StudentTEmpirical`` / ``GenNormEmpirical`` also learn their shape parameter, which the hook does not
differentiate: autograd (``Potential.leftover``).
"""
import torch

from .correlated import ConvCorrelatedNormal
from .loc_scale import GenNorm, Laplace, Normal, PositiveImproper, StudentT
from .transformed import inv_softplus

__all__ = ("NormalEmpirical", "LaplaceEmpirical", "StudentTEmpirical", "GenNormEmpirical", "ConvCorrNormalEmpirical")


def _learnable(value):
    hyper = PositiveImproper(shape=[], loc=value, scale=1.)
    with torch.no_grad():
        hyper.p.data = inv_softplus(torch.tensor(value))
    return hyper


class NormalEmpirical(Normal):
    def __init__(self, shape, loc, scale):
        super().__init__(shape, loc, _learnable(scale))


class LaplaceEmpirical(Laplace):
    def __init__(self, shape, loc, scale):
        super().__init__(shape, loc, _learnable(scale))


class StudentTEmpirical(StudentT):
    def __init__(self, shape, loc, scale, df=2.):
        scale_prior, df_prior = _learnable(scale), _learnable(df)
        super().__init__(shape, loc, scale=scale_prior, df=df_prior)


class GenNormEmpirical(GenNorm):
    def __init__(self, shape, loc, scale, beta=0.5):
        scale_prior, beta_prior = _learnable(scale), _learnable(beta)
        super().__init__(shape, loc, scale=scale_prior, beta=beta_prior)


class ConvCorrNormalEmpirical(ConvCorrelatedNormal):
    def __init__(self, shape, loc, scale, lengthscale=1.0):
        lengthscale_prior, scale_prior = _learnable(lengthscale), _learnable(scale)
        super().__init__(shape, loc, scale=scale_prior, lengthscale=lengthscale_prior)
