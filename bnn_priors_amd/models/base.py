"""Potential-energy interface of a BNN: what the samplers differentiate.

Mirrors ``bnn_priors/models/base.py:14-136,168-191`` of the reference
(``AbstractModel.log_prior / log_likelihood / potential_avg /
split_potential_and_acc``, ``ClassificationModel``, ``RegressionModel``).
The Rao-Blackwellised model (:194-311) is out of scope.

    potential_avg(x, y, N) = -(1/B) sum_i log p(y_i | x_i, theta) - log p(theta) / N
"""
from collections import OrderedDict

import torch
from torch import nn

from .. import prior

__all__ = ("AbstractModel", "ClassificationModel", "RegressionModel")


class AbstractModel(nn.Module):
    def __init__(self, net):
        super().__init__()
        self.net = net

    # -- densities -------------------------------------------------------------
    def log_prior(self):
        total = sum(pr.log_prob() for _, pr in prior.named_priors(self))
        return torch.tensor(total) if isinstance(total, float) else total

    def likelihood_dist(self, f):
        raise NotImplementedError

    def forward(self, x):
        return self.likelihood_dist(self.net(x))

    def _log_likelihood_preds(self, x, y, eff_num_data):
        assert x.shape[0] == y.shape[0]
        preds = self(x)
        return preds.log_prob(y).sum() * (eff_num_data / x.shape[0]), preds

    def log_likelihood(self, x, y, eff_num_data):
        return self._log_likelihood_preds(x, y, eff_num_data)[0]

    def log_likelihood_avg(self, x, y):
        return self._log_likelihood_preds(x, y, 1)[0]

    # -- potentials ------------------------------------------------------------
    def potential(self, x, y, eff_num_data):
        return -(self.log_likelihood(x, y, eff_num_data) + self.log_prior())

    def potential_avg(self, x, y, eff_num_data):
        return -(self.log_likelihood_avg(x, y) + self.log_prior() / eff_num_data)

    def _split_potential_preds(self, x, y, eff_num_data):
        lla, preds = self._log_likelihood_preds(x, y, 1)
        log_prior = self.log_prior()
        return -lla, log_prior, -lla - log_prior / eff_num_data, preds

    def split_potential_and_acc(self, x, y, eff_num_data):
        raise NotImplementedError

    def params_dict(self):
        return OrderedDict((n, p.detach()) for n, p in self.named_parameters())

    def sample_all_priors(self):
        for _, pr in prior.named_priors(self):
            pr.sample()


class RegressionModel(AbstractModel):
    "independent Gaussian likelihood with std ``noise_std`` (reference :139-165)"

    def __init__(self, net, noise_std):
        super().__init__(net)
        self.noise_std = noise_std

    def likelihood_dist(self, f):
        return torch.distributions.Normal(f, prior.value_or_call(self.noise_std))

    def acc_mse(self, preds, y):
        d = preds.mean - y
        return (d * d).sum(-1)

    def split_potential_and_acc(self, x, y, eff_num_data):
        loss, log_prior, pot, preds = self._split_potential_preds(x, y, eff_num_data)
        return loss, log_prior, pot, self.acc_mse(preds, y), preds


class ClassificationModel(AbstractModel):
    "categorical likelihood on ``net(x) / softmax_temp`` (reference :168-191)"

    def __init__(self, net, softmax_temp=1.):
        super().__init__(net)
        self.softmax_temp = softmax_temp

    def likelihood_dist(self, f):
        return torch.distributions.Categorical(logits=f / prior.value_or_call(self.softmax_temp))

    def acc_mse(self, preds, y):
        return preds.logits.argmax(dim=1).eq(y).to(torch.float32)

    def split_potential_and_acc(self, x, y, eff_num_data):
        loss, log_prior, pot, preds = self._split_potential_preds(x, y, eff_num_data)
        return loss, log_prior, pot, self.acc_mse(preds, y), preds
