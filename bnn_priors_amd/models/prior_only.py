"""Likelihood-free models whose potential is just -log prior: the targets of the
reference's sampler tests (``bnn_priors/models/prior_only.py:11-53``)."""
import torch

from .. import prior
from .base import AbstractModel

__all__ = ("PriorOnlyModel", "GaussianModel", "NealFunnel", "NealFunnelT")


class PriorOnlyModel(AbstractModel):
    def __init__(self, priors):
        super().__init__(torch.nn.Identity())
        for i, pr in enumerate(priors):
            setattr(self, str(i), pr)

    def likelihood_dist(self, f):
        return torch.distributions.Normal(loc=f, scale=1.)

    def log_likelihood(self, x, y, eff_num_data):
        return torch.zeros((), requires_grad=True)

    def log_likelihood_avg(self, x, y):
        return torch.zeros((), requires_grad=True)

    def split_potential_and_acc(self, x, y, eff_num_data):
        zero = torch.zeros(())
        log_prior = self.log_prior()
        return zero, log_prior, -log_prior, zero, self.likelihood_dist(y)

    def potential_avg_closure(self):
        self.zero_grad()
        loss = self.potential_avg(None, None, 1.)
        loss.backward()
        return loss


class GaussianModel(PriorOnlyModel):
    def __init__(self, N, D, mean=0., std=1.):
        super().__init__([prior.Normal(torch.Size([D]), mean, std) for _ in range(N)])


class NealFunnel(PriorOnlyModel):
    def __init__(self):
        super().__init__([prior.Normal(torch.Size([]), 0., torch.linspace(0.01, 1, 100))])


class NealFunnelT(PriorOnlyModel):
    def __init__(self):
        super().__init__([prior.StudentT(torch.Size([]), 0., torch.linspace(0.01, 1, 100), df=3)])
