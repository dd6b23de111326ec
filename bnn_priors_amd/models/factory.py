"""Model construction by name for the BASELINE configs.

Restates the on-path part of ``bnn_priors/exp_utils.py:63-69,99-105,108-234``:
``get_model`` for classificationdensenet / classificationconvnet / googleresnet /
densenet, ``he_initialize``, and the ``net.module.`` wrapper that gives stored
samples the reference's key prefix.  (The reference wraps GPU models in
``nn.DataParallel``; here one chain owns one GPU, so the wrapper is always the
plain ``DummyModule`` -- the keys are identical.)
"""
import math

import torch
from torch import nn

from ..prior import get_prior
from .nets import ClassificationConvNet, ClassificationDenseNet, DenseNet, ResNet

__all__ = ("get_model", "he_initialize", "DummyModule")


class DummyModule(nn.Module):
    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, *a, **kw):
        return self.module(*a, **kw)


def he_initialize(model):
    "Kaiming-normal weights, U(-1/sqrt(n_out), 1/sqrt(n_out)) biases (exp_utils.py:63-69)"
    for name, param in model.named_parameters():
        if "weight_prior.p" in name:
            nn.init.kaiming_normal_(param.data, mode='fan_in', nonlinearity='relu')
        elif "bias_prior.p" in name:
            bound = 1 / math.sqrt(param.size(0))
            nn.init.uniform_(param.data, -bound, bound)


def get_model(x_train, y_train, model, width=50, depth=3, weight_prior="gaussian", weight_loc=0.,
              weight_scale=2 ** .5, bias_prior="gaussian", bias_loc=0., bias_scale=1.,
              batchnorm=True, weight_prior_params={}, bias_prior_params={}):
    scaling_fn = (lambda std, dim: std / dim) if weight_prior == "cauchy" \
        else (lambda std, dim: std / dim ** 0.5)
    common = dict(prior_w=get_prior(weight_prior), loc_w=weight_loc, std_w=weight_scale,
                  prior_b=get_prior(bias_prior), loc_b=bias_loc, std_b=bias_scale,
                  scaling_fn=scaling_fn, weight_prior_params=weight_prior_params,
                  bias_prior_params=bias_prior_params)
    if model == "classificationdensenet":
        net = ClassificationDenseNet(x_train.size(-1), int(y_train.max()) + 1, width, depth,
                                     softmax_temp=1., **common)
    elif model == "classificationconvnet":
        if x_train.dim() == 4:
            in_channels, img_height = x_train.shape[1], x_train.shape[-2]
        else:
            in_channels, img_height = 1, int(math.sqrt(x_train.shape[-1]))
        net = ClassificationConvNet(in_channels, img_height, int(y_train.max()) + 1, width, depth,
                                    softmax_temp=1., **common)
    elif model == "googleresnet":
        # NB: conv_prior_w is not forwarded, so convolutions stay Gaussian whatever
        # weight_prior says (reference quirk, exp_utils.py:186-190).
        net = ResNet(depth=20, bn=batchnorm, softmax_temp=1., **common)
    elif model == "densenet":
        net = DenseNet(x_train.size(-1), y_train.size(-1), width, depth, noise_std=1., **common)
    else:
        raise ValueError(f"model='{model}' is outside the accelerated path")
    net = net.to(x_train.device if isinstance(x_train, torch.Tensor) else "cpu")
    inner = net.net
    del net.net
    net.net = DummyModule(inner)
    return net
