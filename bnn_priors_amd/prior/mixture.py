"""Mixture priors: one sampled tensor ``p`` whose density is a weighted sum of element-wise families.

Reference: ``bnn_priors/prior/mixture.py:53-120`` (``Mixture``: the families named by an abbreviation string,
all with the same loc / scale) and ``:130-151`` (``ScaleMixture``: ONE family at five scales
``scale x {1/9, 1/3, 1, 3, 9}``).  The density is that of the whole tensor -- the mixture is over tensors, not
over elements:

    log p(p) = logsumexp_k( w_k + sum_elements log p_k(p) ) - logsumexp_k( w_k )        (mixture.py:75-87)

with the logits ``w`` (``mixture_weights``) a sampled parameter of the model without a prior of its own.

Every component is a Prior module registered as ``component_<k>`` that SHARES ``p`` with the mixture, so a stored
sample has the reference's keys (``p``, ``loc``, ``scale``, ``mixture_weights``, ``component_<k>.p`` / ``.loc`` /
``.scale`` [/ ``.df`` / ``.beta``]).  A component contributes nothing of its own to a model's summed log-prior
(``log_prob() == 0``; its density is ``component_log_prob()``) and is never handed to the HIP prior hook
(``is_component``): the mixture's log-density couples all elements of the tensor through one logsumexp, which is not
an element-wise hook -- these priors stay in autograd (``Potential.leftover``).
"""
import torch
import torch.distributions as td

from .base import Prior
from .loc_scale import LocScale, get_prior

__all__ = ("Mixture", "ScaleMixture", "ScaleMixtureEmpirical")

# abbreviation -> table name (mixture.py:100-126)
ABBREVIATIONS = {"g": "gaussian", "ln": "lognormal", "l": "laplace", "c": "cauchy", "s": "student-t", "u": "uniform",
                 "i": "improper", "gg": "gaussian_gamma", "gu": "gaussian_uniform", "h": "horseshoe",
                 "lg": "laplace_gamma", "lu": "laplace_uniform", "sg": "student-t_gamma", "su": "student-t_uniform",
                 "gn": "gennorm", "gnu": "gennorm_uniform", "ge": "gaussian_empirical", "le": "laplace_empirical",
                 "se": "student-t_empirical", "gne": "gennorm_empirical"}


def _as_component(comp, p):
    "``comp`` becomes a view of the mixture's tensor: same Parameter, no density of its own, never fused"
    comp.p = p
    density = comp.log_prob                      # bound method of the family
    comp.component_log_prob = density
    comp.log_prob = lambda: 0.                   # (a model sums log_prob over ALL Prior modules, models/base.py:26-28)
    comp.is_component = True
    comp.fused_spec = lambda: None
    return comp


class Mixture(LocScale):
    fused_kind = None

    def __init__(self, shape, loc, scale, components="g_l_s_c_gn"):
        names = self.get_components(components)
        assert len(names) > 0, "Too few mixture components"
        super().__init__(shape, loc, scale)
        self._install([get_prior(n)(shape, loc, scale) for n in names])

    def _install(self, comps):
        self.mixture_weights = torch.nn.Parameter(torch.zeros(len(comps)))
        self.components = [_as_component(c, self.p) for c in comps]
        for k, c in enumerate(self.components):
            self.add_module(f"component_{k}", c)
        self.sample()              # now from the mixture itself (the constructor drew a placeholder)

    def fused_spec(self):
        return None

    def log_prob(self):
        w = self.mixture_weights
        per = torch.stack([c.component_log_prob() for c in self.components])
        return torch.logsumexp(w + per, dim=0) - torch.logsumexp(w, dim=0)

    def _draw(self, shape):
        comps = self.__dict__.get("components")
        if comps is None:
            return torch.randn(shape)          # (called by Prior.__init__ before the components exist)
        k = td.Categorical(logits=self.mixture_weights).sample().item()
        return comps[k]._draw(shape)

    @staticmethod
    def get_components(comp_string):
        abbrs = comp_string.split("_")
        assert all(a in ABBREVIATIONS for a in abbrs), "Unknown mixture components"
        return [ABBREVIATIONS[a] for a in abbrs]


class ScaleMixture(Mixture):
    def __init__(self, shape, loc, scale, base_dist="gaussian", scales=None):
        self.scales = [scale / 9, scale / 3, scale, scale * 3, scale * 9] if scales is None else scales
        # The reference builds the DEFAULT five-family mixture first and then overwrites ``mixture_weights``,
        # ``components`` and the modules ``component_0 .. component_{n-1}`` (mixture.py:136-146).  Done the same way, so
        # that the random draws consumed and the stored keys are the reference's: with fewer than five scales the
        # default mixture's later components stay registered (``component_3`` / ``component_4`` are a Cauchy and a
        # generalised normal then) -- views of ``p`` that are part of every stored sample and of no density.
        Mixture.__init__(self, shape, loc, scale)
        self._install([get_prior(base_dist)(shape, loc, s) for s in self.scales])


class ScaleMixtureEmpirical(Mixture):
    """``ScaleMixture`` whose five scales are learnable (softplus of a parameter without a prior, initialised at
    scale x {1/9, 1/3, 1, 3, 9}): reference prior/mixture.py:154-178.  The scales are sampled parameters
    ``component_<k>.scale.p``; autograd."""
    def __init__(self, shape, loc, scale, base_dist="gaussian", scales=None):
        from .loc_scale import PositiveImproper
        from .transformed import inv_softplus
        self.scales = [scale / 9, scale / 3, scale, scale * 3, scale * 9] if scales is None else scales
        Mixture.__init__(self, shape, loc, scale)          # (the default mixture first, as ScaleMixture and the reference do)
        hypers = [PositiveImproper(shape=[], loc=s, scale=1.) for s in self.scales]
        for h, s in zip(hypers, self.scales):
            with torch.no_grad():
                h.p.data = inv_softplus(torch.tensor(s))
        self._install([get_prior(base_dist)(shape, loc, h) for h in hypers])
