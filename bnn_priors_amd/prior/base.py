"""Prior modules: each owns the sampled tensor ``p`` and its log-density.

Mirrors the interface of the reference's ``bnn_priors/prior/base.py:17-87``
(``Prior.p``, ``log_prob()``, ``forward()``, ``sample()``, ``named_priors``) so
models and stored ``state_dict``s keep the reference's key names
(``<layer>.weight_prior.p`` / ``.loc`` / ``.scale`` [/ ``.df``]).
"""
from numbers import Number

import numpy as np
import torch
from torch import nn

__all__ = ("Prior", "value_or_call", "named_priors", "named_params_with_prior")


def value_or_call(v):
    return v() if callable(v) else v


class Prior(nn.Module):
    """A parameter tensor ``p`` together with the distribution it is a priori drawn from.

    Distribution arguments given as numbers / arrays become buffers (and so
    appear in ``state_dict``), ``nn.Parameter``s and sub-modules are registered
    as such (reference: prior/base.py:25-45).  Sub-classes set ``_dist``.
    """
    _dist = None
    # (kind id understood by the fused HIP prior hook, or None when the prior
    #  is not an element-wise loc/scale family)
    fused_kind = None

    def __init__(self, shape, **dist_args):
        super().__init__()
        self._arg_names = tuple(dist_args)
        for name, value in dist_args.items():
            assert name != "p", "repeated name of parameter"
            if isinstance(value, Number):
                value = torch.tensor(value, dtype=torch.get_default_dtype())
            elif isinstance(value, np.ndarray):
                value = torch.from_numpy(value).to(torch.get_default_dtype())
            if isinstance(value, nn.Parameter):
                self.register_parameter(name, value)
            elif isinstance(value, nn.Module):
                self.add_module(name, value)
            elif isinstance(value, torch.Tensor):
                self.register_buffer(name, value)
            else:
                setattr(self, name, value)
        self.p = nn.Parameter(self._draw(torch.Size(shape)))

    def _dist_obj(self):
        return self._dist(**{k: value_or_call(getattr(self, k)) for k in self._arg_names})

    def _draw(self, shape):
        dist = self._dist_obj()
        covered = len(dist.batch_shape) + len(dist.event_shape)
        if covered:
            shape = shape[:-covered]
        return dist.sample(sample_shape=shape)

    def log_prob(self):
        "sum over elements of log p(p)  (reference: prior/base.py:57-58)"
        return self._dist_obj().log_prob(self.p).sum()

    @torch.no_grad()
    def sample(self):
        self.p.data = self._draw(self.p.size()).to(self.p.data)
        self.p.grad = None

    def forward(self):
        return self.p

    _shape_arg = "df"      # name of the family's extra shape argument, if it has one

    def scale_link(self):
        "the Prior module that produces this prior's scale (hierarchical priors), or None"
        sc = getattr(self, "scale", None)
        return sc if isinstance(sc, Prior) else None

    def fused_spec(self):
        """(kind, loc, scale, shape-parameter) with python floats when the prior is an element-wise
        family the HIP hook knows, with a plain-number loc and either a plain-number scale or a scale
        produced by a one-element hyper-prior the hook also knows (then scale is NaN here and the
        caller resolves ``scale_link()`` to a segment); else None."""
        if self.fused_kind is None:
            return None
        loc, scale = getattr(self, "loc", None), getattr(self, "scale", None)
        if not isinstance(loc, torch.Tensor) or loc.numel() != 1 or isinstance(loc, nn.Parameter):
            return None
        link = self.scale_link()
        if link is not None:
            if link.p.numel() != 1 or link.fused_spec() is None or self.fused_kind not in (1, 2, 3):
                return None
            scale_value = float("nan")
        else:
            if not isinstance(scale, torch.Tensor) or scale.numel() != 1 or isinstance(scale, nn.Parameter):
                return None
            scale_value = float(scale)
        extra = getattr(self, self._shape_arg, None)
        if isinstance(extra, (Prior, nn.Parameter)) or (isinstance(extra, torch.Tensor) and extra.numel() != 1):
            return None
        return self.fused_kind, float(loc), scale_value, (float(extra) if extra is not None else 0.0)


def named_priors(module):
    "(name, Prior) for every Prior inside ``module``"
    return ((n, m) for n, m in module.named_modules() if isinstance(m, Prior))


def named_params_with_prior(module):
    return ((n + ("p" if n == "" else ".p"), m.p) for n, m in named_priors(module))
