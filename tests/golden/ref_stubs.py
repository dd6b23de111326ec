"""Stub modules that let the reference import in the build container.

Used ONLY by tests/golden/make_goldens.py and tests that are skipped when
/root/reference is absent (it never exists on the GPU box).  The reference's
hot path needs none of these packages; they are imported at module scope by
off-path files.
"""
import sys
import types

import torch

REFERENCE_ROOT = "/root/reference"


def install():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules.setdefault(name, m)
        return sys.modules[name]

    def inv_softplus(x):
        return x + torch.log(-torch.expm1(-x))

    g = mod("gpytorch")
    g.utils = mod("gpytorch.utils")
    g.utils.transforms = mod("gpytorch.utils.transforms", inv_softplus=inv_softplus)
    g.distributions = mod("gpytorch.distributions",
                          MultivariateNormal=torch.distributions.MultivariateNormal)
    mod("h5py")
    s = mod("sacred")
    s.observers = mod("sacred.observers", FileStorageObserver=object)
    tv = mod("torchvision")
    tv.transforms = mod("torchvision.transforms")
    tv.datasets = mod("torchvision.datasets")
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    sys.dont_write_bytecode = True
