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


class _Whitening(td.Transform):
    "x -> x A + m on the flattened last two dimensions (A = the covariance's PCA factor); its log-determinant is constant"
    domain = td.constraints.real
    codomain = td.constraints.real
    event_dim = 2
    bijective = True

    def __init__(self, shift, factor, inverse_factor, log_det):
        super().__init__(cache_size=0)
        self.shift, self.factor, self.inverse_factor, self.log_det = shift, factor, inverse_factor, log_det

    def _flat(self, t):
        return t.view(t.shape[:-2] + (-1,))

    def _call(self, x):
        return (self._flat(x) @ self.factor + self.shift).view(x.shape)

    def _inverse(self, y):
        return ((self._flat(y) - self.shift) @ self.inverse_factor).view(y.shape)

    def log_abs_det_jacobian(self, x, y):
        return self.log_det


def _pca_factors(cov):
    "(A, A^-1, log det A) with A = diag(sqrt(lambda)) V^T of cov = V diag(lambda) V^T, computed in float64"
    lam, vec = torch.linalg.eigh(cov.to(torch.float64))
    root = lam.sqrt()
    return root.unsqueeze(-1) * vec.t(), vec / root, lam.log().sum().view((1, 1)) / 2


class ConvCovariance(Prior):
    """base of the fixed-covariance priors: an element-wise base density (``_base``) pushed through the whitening
    transform of a given covariance of the filter positions; the factors are buffers (``scale``, ``inv_scale``,
    ``log_sqrt_vals``: the reference's names, prior/conv_loc_scale.py:46-70).  A NUMBER as ``cov`` is a standard deviation."""
    fused_kind = None

    def __init__(self, shape, loc, cov, **kwargs):
        n_pos = shape[-2] * shape[-1]
        if isinstance(cov, Number) or len(cov.shape) == 0:
            cov, loc = torch.eye(n_pos) * cov ** 2, torch.zeros(n_pos) + loc
        dt = torch.get_default_dtype()
        factor, inverse, log_det = (t.to(dt) for t in _pca_factors(cov))
        super().__init__(shape, loc=loc, scale=factor, inv_scale=inverse, log_sqrt_vals=log_det, event_shape=shape[-2:],
                         **kwargs)

    def _base(self, zeros, **extra):
        raise NotImplementedError

    def _dist(self, loc, scale, inv_scale, log_sqrt_vals, event_shape, **extra):
        zeros = torch.zeros((), device=loc.device, dtype=loc.dtype).expand(event_shape)
        return td.TransformedDistribution(self._base(zeros, **extra), _Whitening(loc, scale, inv_scale, log_sqrt_vals))

    def assign_cov(self, cov):
        for buf, new in zip((self.scale, self.inv_scale, self.log_sqrt_vals), _pca_factors(cov)):
            buf.copy_(new)


class FixedCovNormal(ConvCovariance):
    def __init__(self, shape, loc, cov):
        super().__init__(shape, loc, cov)

    def _base(self, zeros):
        return td.Normal(zeros, zeros + 1)


class FixedCovGenNorm(ConvCovariance):
    "(the reference notes that sampling is slightly off -- the CDF's accuracy -- and irrelevant for inference)"

    def __init__(self, shape, loc, cov, beta, base_scale=None):
        if isinstance(beta, Number):
            beta = torch.tensor(beta, dtype=torch.float64)        # (stored in double, as the reference stores it)
        if base_scale is None:        # the scale that gives the base density unit variance
            base_scale = ((torch.lgamma(1 / beta) - torch.lgamma(3 / beta)) / 2).exp()
        super().__init__(shape, loc, cov, beta=beta, base_scale=torch.as_tensor(base_scale).to(torch.get_default_dtype()))

    def _base(self, zeros, beta, base_scale):
        shape = zeros.shape
        return GeneralizedNormal(loc=zeros, scale=base_scale.expand(shape), beta=beta.expand(shape))
