"""Distributions the prior modules need beyond ``torch.distributions``.

``GeneralizedNormal(loc, scale, beta)``: density  beta / (2 scale Gamma(1/beta)) * exp(-(|x - loc| / scale)^beta)
(reference: bnn_priors/prior/distributions.py:15-98).  ``DoubleGamma(concentration, rate)``: a Gamma density on |x|,
halved, with a random sign when sampled (distributions.py:97-112).  Sampling goes through ``scipy.stats.gennorm`` seeded from
torch's global generator, so a run under ``torch.manual_seed`` draws what the reference draws.
"""
import torch
from torch.distributions import constraints
from torch.distributions.distribution import Distribution
from torch.distributions.utils import broadcast_all

__all__ = ("GeneralizedNormal", "DoubleGamma")


class GeneralizedNormal(Distribution):
    arg_constraints = {"loc": constraints.real, "scale": constraints.positive, "beta": constraints.positive}
    support = constraints.real
    has_rsample = False

    def __init__(self, loc, scale, beta, validate_args=None):
        self.loc, self.scale = broadcast_all(loc, scale)
        (self.beta,) = broadcast_all(beta)
        super().__init__(self.loc.size(), validate_args=validate_args)

    @property
    def mean(self):
        return self.loc

    @property
    def variance(self):
        return self.scale.pow(2) * (torch.lgamma(3 / self.beta) - torch.lgamma(1 / self.beta)).exp()

    def log_prob(self, value):
        if self._validate_args:
            self._validate_sample(value)
        z = (value - self.loc).abs() / self.scale
        return torch.log(self.beta) - torch.log(2 * self.scale) - torch.lgamma(1 / self.beta) - z.pow(self.beta)

    def entropy(self):
        return 1 / self.beta - torch.log(self.beta) + torch.log(2 * self.scale) + torch.lgamma(1 / self.beta)

    def sample(self, sample_shape=torch.Size()):
        from scipy import stats
        frozen = stats.gennorm(loc=self.loc.detach().cpu().numpy(), scale=self.scale.detach().cpu().numpy(),
                               beta=self.beta.detach().cpu().numpy())
        seed = torch.randint(2 ** 32, ()).item()     # one draw from torch's generator per call
        shape = list(torch.Size(sample_shape) + self.loc.size())
        return torch.tensor(frozen.rvs(shape, random_state=seed), dtype=self.loc.dtype, device=self.loc.device)


class DoubleGamma(torch.distributions.Gamma):
    "density Gamma(|x|; concentration, rate) / 2 on the real line"
    mean = 0.

    @property
    def variance(self):
        return self.concentration * (1 + self.concentration) / self.rate.pow(2)

    def rsample(self, sample_shape=torch.Size()):
        x = super().rsample(sample_shape)
        sign = torch.randint(0, 2, x.size(), device=x.device, dtype=x.dtype).mul_(2).sub_(1)
        return x * sign

    def log_prob(self, value):
        import math
        return super().log_prob(value.abs()) - math.log(2)
