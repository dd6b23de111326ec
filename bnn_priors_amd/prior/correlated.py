"""Multivariate priors over a convolution filter's spatial positions (the last two dimensions of the weight), all left
to autograd (``Potential.leftover``): none is an element-wise family of the HIP hook -- SURVEY.md section 8(f)4.

* ``ConvCorrelatedNormal``: a zero-mean-shifted Gaussian over the ``kh x kw`` positions with a squared-exponential
  covariance ``scale^2 exp(-d / lengthscale)`` of their Euclidean distances, independent across channels (reference:
  bnn_priors/prior/loc_scale.py:13-63); ``ConvCorrNormalGamma``: Gamma hyper-priors on scale and lengthscale
  (prior/hierarchical.py:32-39).
* ``FixedCovNormal`` / ``FixedCovGenNorm``: an element-wise base density pushed through the PCA transform of a given
  covariance of the positions (prior/conv_loc_scale.py:16-140).
The reference calls ``torch.cholesky`` / ``torch.symeig``, which this image's torch no longer has; ``torch.linalg.cholesky``
and ``torch.linalg.eigh`` compute the same factors.
"""
from numbers import Number

import numpy as np
import torch
import torch.distributions as td

from .base import Prior
from .distributions import GeneralizedNormal

__all__ = ("ConvCorrelatedNormal", "ConvCorrNormalGamma", "ConvCovariance", "FixedCovNormal", "FixedCovGenNorm")


class SquaredExponentialNormal(td.MultivariateNormal):
    def __init__(self, loc, scale, distance_matrix, lengthscale):
        cov = torch.exp(-distance_matrix / lengthscale) * scale ** 2.0
        super().__init__(loc=loc, scale_tril=torch.linalg.cholesky(cov))


class ConvCorrelatedNormal(Prior):
    _dist = SquaredExponentialNormal
    fused_kind = None

    def __init__(self, shape, loc, scale, *, lengthscale=1.0):
        # One location per spatial position.  The reference hands MultivariateNormal a ONE-element loc for a scalar
        # (loc_scale.py:41-43) and relies on its being broadcast against the covariance -- which torch >= 2 no longer does
        # (sampling and log_prob fail there with a shape error): the scalar is expanded here, the density is the same.
        loc = torch.as_tensor(loc, dtype=torch.get_default_dtype())
        if loc.dim() == 0 or loc.shape[-1] == 1:
            loc = loc.reshape(-1)[:1].expand(shape[-2] * shape[-1]).clone()
        pts = np.mgrid[:shape[-2], :shape[-1]].reshape(2, -1).T
        d = np.sum((pts[:, None, :] - pts[None, :, :]) ** 2.0, 2) ** 0.5
        super().__init__(shape, loc=loc, scale=scale, distance_matrix=d, lengthscale=lengthscale)

    def log_prob(self):
        return self._dist_obj().log_prob(self.p.reshape(self.p.shape[:-2] + (-1,))).sum()

    def _draw(self, shape):
        return torch.reshape(self._dist_obj().sample(sample_shape=shape[:-2]), shape)


class ConvCorrNormalGamma(ConvCorrelatedNormal):
    def __init__(self, shape, loc, scale, lengthscale=1., rate=1.):
        from .hierarchical import _gamma_scale
        super().__init__(shape, loc, scale=_gamma_scale(scale, rate), lengthscale=_gamma_scale(lengthscale, rate))


class _PCATransform(td.Transform):
    domain = td.constraints.real
    codomain = td.constraints.real
    event_dim = 2
    bijective = True

    def __init__(self, loc, scale, inv_scale, log_det, cache_size=0):
        super().__init__(cache_size=cache_size)
        self.loc, self.scale, self.inv_scale, self._log_det = loc, scale, inv_scale, log_det

    def log_abs_det_jacobian(self, x, y):
        return self._log_det

    def _call(self, x):
        flat = x.view(x.shape[:-2] + (-1,))
        return (flat @ self.scale + self.loc).view(x.shape)

    def _inverse(self, y):
        flat = y.view(y.shape[:-2] + (-1,))
        return ((flat - self.loc) @ self.inv_scale).view(y.shape)


class ConvCovariance(Prior):
    "base of the fixed-covariance priors: the covariance's PCA factors are buffers (``scale``, ``inv_scale``, ``log_sqrt_vals``)"
    fused_kind = None

    def __init__(self, shape, loc, cov, **kwargs):
        if isinstance(cov, Number) or len(cov.shape) == 0:
            cov = torch.eye(shape[-2] * shape[-1]) * cov ** 2          # (a number is a standard deviation)
            loc = torch.zeros(shape[-2] * shape[-1]) + loc
        scale, inv_scale, log_sqrt_vals = self._break_down_cov(cov)
        dt = torch.get_default_dtype()
        super().__init__(shape, loc=loc, scale=scale.to(dt), inv_scale=inv_scale.to(dt), log_sqrt_vals=log_sqrt_vals.to(dt),
                         event_shape=shape[-2:], **kwargs)

    @staticmethod
    def _break_down_cov(cov):
        vals, vecs = torch.linalg.eigh(cov.to(torch.float64))
        sqrt_vals = vals.sqrt()
        return sqrt_vals.unsqueeze(-1) * vecs.t(), vecs / sqrt_vals, vals.log().sum().view((1, 1)) / 2

    def assign_cov(self, cov):
        scale, inv_scale, log_sqrt_vals = self._break_down_cov(cov)
        self.scale.copy_(scale)
        self.inv_scale.copy_(inv_scale)
        self.log_sqrt_vals.copy_(log_sqrt_vals)


class FixedCovNormal(ConvCovariance):
    def __init__(self, shape, loc, cov):
        super().__init__(shape, loc, cov)

    def _dist(self, loc, scale, inv_scale, log_sqrt_vals, event_shape):
        zeros = torch.zeros((), device=loc.device, dtype=loc.dtype).expand(event_shape)
        return td.TransformedDistribution(td.Normal(zeros, zeros + 1), _PCATransform(loc, scale, inv_scale, log_sqrt_vals))


class FixedCovGenNorm(ConvCovariance):
    def __init__(self, shape, loc, cov, beta, base_scale=None):
        if base_scale is None:
            if isinstance(beta, Number):
                beta = torch.tensor(beta, dtype=torch.float64)
            base_scale = (torch.lgamma(1 / beta) - torch.lgamma(3 / beta)).div(2).exp()      # unit variance
        super().__init__(shape, loc, cov, beta=beta, base_scale=base_scale.to(torch.get_default_dtype()))

    def _dist(self, loc, scale, inv_scale, log_sqrt_vals, beta, base_scale, event_shape):
        zeros = torch.zeros((), device=loc.device, dtype=loc.dtype).expand(event_shape)
        return td.TransformedDistribution(
            GeneralizedNormal(loc=zeros, scale=base_scale.expand(event_shape), beta=beta.expand(event_shape)),
            _PCATransform(loc, scale, inv_scale, log_sqrt_vals))
