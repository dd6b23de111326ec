"""Element-wise location/scale priors used by the BASELINE configs.

Reference: ``bnn_priors/prior/loc_scale.py:21-103`` (Normal :34, Laplace :66,
Cauchy :70, StudentT :74-77, GenNorm :80-83, Improper :93-96).  Closed forms of log p and its
gradient, which the fused HIP prior hook implements, are in SURVEY.md
Appendix A and checked in tests/test_priors.py.
"""
import torch.distributions as td

from .base import Prior
from .distributions import GeneralizedNormal

__all__ = ("LocScale", "Normal", "Laplace", "Cauchy", "StudentT", "GenNorm", "LogNormal", "Improper", "get_prior",
           "FUSED_NONE", "FUSED_NORMAL", "FUSED_LAPLACE", "FUSED_STUDENT_T", "FUSED_CAUCHY", "FUSED_GENNORM")

FUSED_NONE, FUSED_NORMAL, FUSED_LAPLACE, FUSED_STUDENT_T, FUSED_CAUCHY, FUSED_GENNORM = 0, 1, 2, 3, 4, 5


class LocScale(Prior):
    def __init__(self, shape, loc, scale):
        super().__init__(shape, loc=loc, scale=scale)


class Normal(LocScale):
    _dist = td.Normal
    fused_kind = FUSED_NORMAL


class Laplace(LocScale):
    _dist = td.Laplace
    fused_kind = FUSED_LAPLACE


class Cauchy(LocScale):
    _dist = td.Cauchy
    fused_kind = FUSED_CAUCHY


class StudentT(LocScale):
    _dist = td.StudentT
    fused_kind = FUSED_STUDENT_T

    def __init__(self, shape, loc, scale, df=3):
        Prior.__init__(self, shape, df=df, loc=loc, scale=scale)


class GenNorm(LocScale):
    "generalised normal; the shape parameter ``beta`` rides in the hook's ``df`` slot"
    _dist = GeneralizedNormal
    fused_kind = FUSED_GENNORM
    _shape_arg = "beta"

    def __init__(self, shape, loc, scale, beta=0.5):
        Prior.__init__(self, shape, loc=loc, scale=scale, beta=beta)


class LogNormal(LocScale):
    """value = exp(p) with p ~ N(loc, scale): log p = Normal.log_prob(p) - sum(p)  (reference: prior/loc_scale.py:86-91).
    Not an element-wise family of the HIP hook (the extra -1 in the gradient and the transformed value): autograd."""
    _dist = td.Normal
    fused_kind = None

    def forward(self):
        return self.p.exp()

    def log_prob(self):
        return super().log_prob() - self.p.sum()


class Improper(Normal):
    "samples like a Normal, contributes nothing to the log-density"
    fused_kind = FUSED_NONE      # "fused" as a no-op: neither gradient nor log-density

    def log_prob(self):
        return 0.0


def _table():
    from . import hierarchical as H, mixture as M, transformed as T
    return {"gaussian": Normal, "laplace": Laplace, "student-t": StudentT, "cauchy": Cauchy,
            "improper": Improper, "gennorm": GenNorm,
            # by name, through autograd (Potential.leftover) -- the reference's table has them (prior/mixture.py:17-50):
            "lognormal": LogNormal, "uniform": T.Uniform, "mixture": M.Mixture, "scale_mixture": M.ScaleMixture,
            "gaussian_gamma": H.NormalGamma, "gaussian_uniform": H.NormalUniform, "horseshoe": H.Horseshoe,
            "laplace_gamma": H.LaplaceGamma, "laplace_uniform": H.LaplaceUniform,
            "student-t_gamma": H.StudentTGamma, "student-t_uniform": H.StudentTUniform,
            "gennorm_uniform": H.GenNormUniform}


def get_prior(name):
    """Name -> class (reference table: prior/mixture.py:17-50).  Element-wise families and hierarchical scales are
    differentiated by the HIP prior hook; ``lognormal``, ``uniform`` (as a tensor's prior), ``mixture`` and
    ``scale_mixture`` are built as the reference builds them and stay in autograd (``Potential.leftover``, with a
    one-time notice).  The correlated / data-driven / empirical-Bayes entries are out of scope."""
    if isinstance(name, type) and issubclass(name, Prior):
        return name
    table = _table()
    try:
        return table[name]
    except KeyError:
        raise KeyError(f"prior '{name}' is not built by this package (correlated / data-driven / empirical-Bayes "
                       f"families are out of scope); available: {sorted(table)}") from None
