"""Likelihood-free sampling targets: the potential is minus the log-density of a list of priors.
These are the test distributions of the reference's sampler tests (a product of Gaussians, Neal's
funnel with Student-t marginals; reference: bnn_priors/models/prior_only.py)."""
import torch

from .. import prior
from .base import AbstractModel

__all__ = ("PriorOnlyModel", "GaussianModel", "NealFunnel", "NealFunnelT")


class PriorOnlyModel(AbstractModel):
    """``potential_avg = -sum_k log p_k(theta_k) / N``; no data term."""

    def __init__(self, priors):
        super().__init__(torch.nn.Identity())
        for k, pr in enumerate(priors):          # sub-module names "0", "1", ...
            self.add_module(str(k), pr)

    # the data term is identically zero (a leaf, so that .backward() on it is legal)
    @staticmethod
    def _no_data():
        return torch.zeros((), requires_grad=True)

    def log_likelihood(self, x, y, eff_num_data):
        return self._no_data()

    def log_likelihood_avg(self, x, y):
        return self._no_data()

    def likelihood_dist(self, f):
        return torch.distributions.Normal(f, 1.)

    def split_potential_and_acc(self, x, y, eff_num_data):
        lp = self.log_prior()
        z = torch.zeros(())
        return z, lp, -lp, z, self.likelihood_dist(y)

    def potential_avg_closure(self):
        "zero grads, evaluate and differentiate the potential at N = 1 (what the tests step with)"
        self.zero_grad()
        u = self.potential_avg(None, None, 1.)
        u.backward()
        return u


def GaussianModel(N, D, mean=0., std=1.):
    "N independent D-dimensional isotropic Gaussians"
    return PriorOnlyModel([prior.Normal(torch.Size([D]), mean, std) for _ in range(N)])


def _funnel_scales():
    return torch.linspace(0.01, 1, 100)


def NealFunnel():
    return PriorOnlyModel([prior.Normal(torch.Size([]), 0., _funnel_scales())])


def NealFunnelT():
    return PriorOnlyModel([prior.StudentT(torch.Size([]), 0., _funnel_scales(), df=3)])
