"""Linear / Conv2d whose weight and bias are read from Prior modules.

Same module/attribute names as ``bnn_priors/models/layers.py:5-46`` so that
``state_dict`` keys are ``<idx>.weight_prior.p`` / ``<idx>.bias_prior.p``.
"""
import torch.nn.functional as F
from torch import nn

__all__ = ("Linear", "Conv2d")


class Linear(nn.Module):
    def __init__(self, weight_prior, bias_prior=None):
        super().__init__()
        self.out_features, self.in_features = weight_prior.p.shape
        self.weight_prior = weight_prior
        self.bias_prior = bias_prior

    @property
    def weight(self):
        return self.weight_prior()

    @property
    def bias(self):
        return None if self.bias_prior is None else self.bias_prior()

    def forward(self, x):
        return F.linear(x, self.weight, self.bias)


class Conv2d(nn.Module):
    def __init__(self, weight_prior, bias_prior=None, stride=1, padding=0, dilation=1, groups=1):
        super().__init__()
        self.out_channels, cin, kh, kw = weight_prior.p.shape
        self.in_channels, self.kernel_size = cin * groups, (kh, kw)
        self.stride, self.padding, self.dilation, self.groups = stride, padding, dilation, groups
        self.weight_prior = weight_prior
        self.bias_prior = bias_prior

    @property
    def weight(self):
        return self.weight_prior()

    @property
    def bias(self):
        return None if self.bias_prior is None else self.bias_prior()

    def forward(self, x):
        return F.conv2d(x, self.weight, self.bias, self.stride, self.padding,
                        self.dilation, self.groups)
