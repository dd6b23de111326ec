"""Element-wise location/scale priors used by the BASELINE configs.

Reference: ``bnn_priors/prior/loc_scale.py:21-103`` (Normal :34, Laplace :66,
Cauchy :70, StudentT :74-77, GenNorm :80-83, Improper :93-96).  Closed forms of log p and its
gradient, which the fused HIP prior hook implements, are in SURVEY.md
Appendix A and checked in tests/test_priors.py.
"""
import torch.distributions as td

from .base import Prior
from .distributions import GeneralizedNormal

__all__ = ("LocScale", "Normal", "Laplace", "Cauchy", "StudentT", "GenNorm", "LogNormal", "Improper", "PositiveImproper",
           "get_prior",
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


class PositiveImproper(Improper):
    """value = softplus(p), no density: the learnable scale (shape parameter, lengthscale) of the empirical-Bayes priors
    (reference: prior/loc_scale.py:100-103, used by prior/empirical_bayes.py).  As the one-element scale of a Normal /
    Laplace / Student-t prior it is a hyper-prior kind of the HIP hook (SGMCMC_PRIOR_IMPROPER_SOFTPLUS: the chain rule
    through softplus, nothing of its own)."""
    fused_kind = 9      # _hip.PRIOR_IMPROPER_SOFTPLUS

    def forward(self):
        import torch.nn.functional as F
        return F.softplus(self.p)

    def fused_spec(self):
        if self.p.numel() != 1:
            return None
        return self.fused_kind, 0.0, 1.0, 0.0


def _table():
    from . import correlated as C, empirical_bayes as E, hierarchical as H, mixture as M, transformed as T
    return {"gaussian": Normal, "laplace": Laplace, "student-t": StudentT, "cauchy": Cauchy,
            "improper": Improper, "gennorm": GenNorm,
            "gaussian_gamma": H.NormalGamma, "gaussian_uniform": H.NormalUniform, "horseshoe": H.Horseshoe,
            "laplace_gamma": H.LaplaceGamma, "laplace_uniform": H.LaplaceUniform,
            "student-t_gamma": H.StudentTGamma, "student-t_uniform": H.StudentTUniform,
            # learnable scale without a hyper-prior (prior/empirical_bayes.py:24-58): Normal and Laplace in the HIP hook
            # (scale linked to a PositiveImproper segment), Student-t / generalised normal (their df / beta is learnable
            # too) through autograd
            "gaussian_empirical": E.NormalEmpirical, "laplace_empirical": E.LaplaceEmpirical,
            "student-t_empirical": E.StudentTEmpirical, "gennorm_empirical": E.GenNormEmpirical,
            # by name, through autograd (Potential.leftover) -- the rest of the reference's table (prior/mixture.py:17-50):
            "lognormal": LogNormal, "uniform": T.Uniform, "mixture": M.Mixture, "scale_mixture": M.ScaleMixture,
            "scale_mixture_empirical": M.ScaleMixtureEmpirical, "gennorm_uniform": H.GenNormUniform,
            "convcorrnormal": C.ConvCorrelatedNormal, "convcorrnormal_fitted_ls": C.ConvCorrelatedNormal,
            "convcorrnormal_empirical": E.ConvCorrNormalEmpirical, "convcorrnormal_gamma": C.ConvCorrNormalGamma,
            "fixedcov_normal": C.FixedCovNormal, "fixedcov_gennorm": C.FixedCovGenNorm,
            "datadrivencorrnormal": Normal, "datadrivencorrdoublegamma": T.DoubleGamma}


def get_prior(name):
    """Name -> class: ALL 31 names of the reference's table (prior/mixture.py:17-50).  Element-wise families, hierarchical
    scales and the learnable scales of ``gaussian_empirical`` / ``laplace_empirical`` are differentiated by the HIP prior
    hook; everything else -- mixtures, the correlated / fixed-covariance convolution priors, the double Gamma, families
    whose shape parameter is learnable -- is built as the reference builds it and stays in autograd
    (``Potential.leftover``, with a one-time notice; such a model's step is not captured into a hipGraph)."""
    if isinstance(name, type) and issubclass(name, Prior):
        return name
    table = _table()
    try:
        return table[name]
    except KeyError:
        raise KeyError(f"prior '{name}' is not in the reference's table (prior/mixture.py:17-50); available: "
                       f"{sorted(table)}") from None
