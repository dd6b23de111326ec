"""The three BASELINE networks (gradient providers for the samplers).

* ``ClassificationDenseNet`` -- reference ``models/dense_nets.py:16-67``
* ``ClassificationConvNet``  -- reference ``models/conv_nets.py:18-70``
* ``ResNet`` (googleresnet, depth 6n+2) -- reference ``models/google_resnet.py:11-78``

Layer order, module names and prior scales follow the reference so parameter
order (= the sampler's segment order) and ``state_dict`` keys match:
weight scale = std_w / sqrt(fan) with fan = in_features for Linear and
**in_channels** (not in_channels*k*k) for Conv2d (``conv_nets.py:26-31``);
bias prior is N(0, std_b) whatever ``loc_b`` (``dense_nets.py:22-23``);
googleresnet convolutions have no bias and always use ``conv_prior_w``
(default Normal; ``google_resnet.py:34-43``), BatchNorm affine parameters are
sampled without a prior.
"""
import torch
from torch import nn

from .. import bn as _bn
from .. import conv as _conv
from .. import pool as _pool
from .. import prior
from .. import resblock as _resblock
from .base import ClassificationModel, RegressionModel

__all__ = ("Linear", "Conv2d", "LinearPrior", "Conv2dPrior", "DenseNet", "ClassificationDenseNet",
           "ClassificationConvNet", "ResNet", "Reshape")



class _PriorBacked(nn.Module):
    """A layer whose ``weight`` / ``bias`` are produced by Prior modules (``weight_prior`` /
    ``bias_prior``), which keeps the reference's parameter names ``<idx>.weight_prior.p`` and
    ``<idx>.bias_prior.p`` (reference: models/layers.py:5-46)."""

    def __init__(self, weight_prior, bias_prior):
        super().__init__()
        self.weight_prior, self.bias_prior = weight_prior, bias_prior

    weight = property(lambda self: self.weight_prior())
    bias = property(lambda self: None if self.bias_prior is None else self.bias_prior())


class Linear(_PriorBacked):
    def __init__(self, weight_prior, bias_prior=None):
        super().__init__(weight_prior, bias_prior)
        self.out_features, self.in_features = weight_prior.p.shape

    def forward(self, x):
        w, b = self.weight, self.bias
        if _pool.linear_supported(x, w, b):        # a head with a few outputs: one launch each way (pool.linear)
            return _pool.linear(x, w, b, owner=self)
        _conv.library_path("linear", x)
        return nn.functional.linear(x, w, b)


class Conv2d(_PriorBacked):
    def __init__(self, weight_prior, bias_prior=None, stride=1, padding=0, dilation=1, groups=1):
        super().__init__(weight_prior, bias_prior)
        self.out_channels, cin, kh, kw = weight_prior.p.shape
        self.in_channels, self.kernel_size, self.groups = cin * groups, (kh, kw), groups
        self.conv_args = (stride, padding, dilation, groups)

    def forward(self, x, want_stats=False):
        """``want_stats``: return (y, stats) where stats is what ``bn.bn_train(stats=...)`` takes (the
        per-channel batch statistics of y, from the kernel's accumulators) or None on the library path"""
        w, b = self.weight, self.bias
        if _conv.supported(x, w, b, *self.conv_args):      # the trunk's 3x3s: fp32-MFMA kernels
            return _conv.conv3x3(x, w, want_stats)
        if _conv.stem_supported(x, w, b, *self.conv_args):
            return _conv.conv_stem(x, w, want_stats)
        _conv.library_path("conv2d", x)
        y = nn.functional.conv2d(x, w, b, *self.conv_args)
        return (y, None) if want_stats else y


def _default_scaling(std, dim):
    return std / dim ** 0.5


def LinearPrior(in_dim, out_dim, prior_w=prior.Normal, loc_w=0., std_w=1.,
                prior_b=prior.Normal, loc_b=0., std_b=1., scaling_fn=None,
                weight_prior_params={}, bias_prior_params={}):
    scaling_fn = scaling_fn or _default_scaling
    w = prior_w((out_dim, in_dim), loc_w, scaling_fn(std_w, in_dim), **weight_prior_params)
    b = prior_b((out_dim,), 0., std_b, **bias_prior_params) if prior_b is not None else None
    return Linear(w, b)


def Conv2dPrior(in_channels, out_channels, kernel_size=3, stride=1, padding=0, dilation=1,
                groups=1, prior_w=prior.Normal, loc_w=0., std_w=1., prior_b=prior.Normal,
                loc_b=0., std_b=1., scaling_fn=None, weight_prior_params={}, bias_prior_params={}):
    scaling_fn = scaling_fn or _default_scaling
    # bias drawn before the weight: keeps the RNG consumption order of conv_nets.py:28-29
    b = prior_b((out_channels,), 0., std_b, **bias_prior_params) if prior_b is not None else None
    w = prior_w((out_channels, in_channels // groups, kernel_size, kernel_size), loc_w,
                scaling_fn(std_w, in_channels), **weight_prior_params)
    return Conv2d(w, b, stride=stride, padding=padding, dilation=dilation, groups=groups)


class Reshape(nn.Module):
    def __init__(self, *shape):
        super().__init__()
        self.shape = shape

    def forward(self, x):
        return x.view(self.shape)


def _dense_stack(in_features, out_features, width, depth, kw):
    dims = [in_features] + [width] * (depth - 1) + [out_features]
    layers = []
    for i, (a, b) in enumerate(zip(dims[:-1], dims[1:])):
        layers.append(LinearPrior(a, b, **kw))
        if i < len(dims) - 2:
            layers.append(nn.ReLU())
    return nn.Sequential(*layers)


def _layer_kwargs(prior_w, loc_w, std_w, prior_b, loc_b, std_b, scaling_fn,
                  weight_prior_params, bias_prior_params):
    return dict(prior_w=prior_w, loc_w=loc_w, std_w=std_w, prior_b=prior_b, loc_b=loc_b,
                std_b=std_b, scaling_fn=scaling_fn, weight_prior_params=weight_prior_params,
                bias_prior_params=bias_prior_params)


def DenseNet(in_features, out_features, width, depth=3, noise_std=1.,
             prior_w=prior.Normal, loc_w=0., std_w=2 ** .5, prior_b=prior.Normal, loc_b=0.,
             std_b=1., scaling_fn=None, weight_prior_params={}, bias_prior_params={}):
    kw = _layer_kwargs(prior_w, loc_w, std_w, prior_b, loc_b, std_b, scaling_fn,
                       weight_prior_params, bias_prior_params)
    return RegressionModel(_dense_stack(in_features, out_features, width, depth, kw), noise_std)


def ClassificationDenseNet(in_features, out_features, width, depth=3, softmax_temp=1.,
                           prior_w=prior.Normal, loc_w=0., std_w=2 ** .5, prior_b=prior.Normal,
                           loc_b=0., std_b=1., scaling_fn=None, weight_prior_params={},
                           bias_prior_params={}):
    kw = _layer_kwargs(prior_w, loc_w, std_w, prior_b, loc_b, std_b, scaling_fn,
                       weight_prior_params, bias_prior_params)
    return ClassificationModel(_dense_stack(in_features, out_features, width, depth, kw),
                               softmax_temp)


def ClassificationConvNet(in_channels, img_height, out_features, width, depth=3, softmax_temp=1.,
                          prior_w=prior.Normal, loc_w=0., std_w=2 ** .5, prior_b=prior.Normal,
                          loc_b=0., std_b=1., scaling_fn=None, weight_prior_params={},
                          bias_prior_params={}):
    assert depth >= 2, "We can't have less than two layers"
    kw = _layer_kwargs(prior_w, loc_w, std_w, prior_b, loc_b, std_b, scaling_fn,
                       weight_prior_params, bias_prior_params)
    layers = [Reshape(-1, in_channels, img_height, img_height)]
    cin = in_channels
    for _ in range(depth - 1):
        layers += [Conv2dPrior(cin, width, kernel_size=3, padding=1, **kw), nn.ReLU(),
                   nn.MaxPool2d(2)]
        cin = width
    layers.append(nn.Flatten())
    flat = width * (img_height // 2 ** (depth - 1)) ** 2
    layers.append(LinearPrior(flat, out_features, **kw))
    return ClassificationModel(_ConvPoolTrunk(*layers), softmax_temp)


class _ConvPoolTrunk(nn.Sequential):
    "a Sequential that runs every Conv2d(+bias) -> ReLU -> MaxPool2d(2) triple as conv + one fused operator"

    def forward(self, x):
        mods, i = list(self), 0
        while i < len(mods):
            m = mods[i]
            if (isinstance(m, Conv2d) and i + 2 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
                    and _is_pool2(mods[i + 2]) and m.conv_args[3] == 1):
                w = m.weight                                                    # bias joins the fused tail
                if _conv.first_pool_supported(x, w, m.bias, *m.conv_args):         # ... or the whole triple is one
                    x, i = _conv.conv_first_pool(x, w, m.bias), i + 3
                    continue
                if _conv.conv50_pool_supported(x, w, m.bias, *m.conv_args):
                    x, i = _conv.conv50_pool(x, w, m.bias), i + 3
                    continue
                if _conv.first_supported(x, w, None, *m.conv_args):
                    y = _conv.conv_first(x, w)
                elif _conv.conv50_supported(x, w, None, *m.conv_args):
                    y = _conv.conv50(x, w)
                else:
                    _conv.library_path("conv2d", x)         # counted, announced once per shape, an error when strict
                    y = nn.functional.conv2d(x, w, None, *m.conv_args)
                b = m.bias
                if _pool.supported(y, b):
                    x, i = _pool.bias_relu_pool(y, b), i + 3
                    continue
                if b is not None:
                    y = y + b.view(1, -1, 1, 1)
                x, i = mods[i + 2](nn.functional.relu(y)), i + 3
            else:
                x, i = m(x), i + 1
        return x


def _all_of(kernel, spatial):
    k = kernel if isinstance(kernel, (tuple, list)) else (kernel, kernel)
    return tuple(k) == tuple(spatial)


def _is_pool2(m):
    return (isinstance(m, nn.MaxPool2d) and m.kernel_size in (2, (2, 2)) and m.stride in (2, (2, 2))
            and m.padding in (0, (0, 0)) and m.dilation in (1, (1, 1)) and not m.ceil_mode
            and not m.return_indices)


class _BatchNorm2d(nn.BatchNorm2d):
    """``nn.BatchNorm2d`` whose ``num_batches_tracked += 1`` is left to the enclosing
    ``_BNTrunk`` (one fused increment for all layers instead of one tiny launch per layer; with a
    fixed ``momentum`` the counter is bookkeeping only -- but it is part of every stored sample,
    so it must count exactly as the reference's layers do)."""

    def forward(self, x):
        return self.fused(x)

    def fused(self, x, residual=None, relu=False, stats=None):
        "relu?(BN(x) [+ residual]): one fused HIP operator in training mode (bn.py), ATen otherwise"
        if self.track_running_stats and _bn.supported(x, self.weight, self.bias, self.training, self.momentum):
            return _bn.bn_train(x, self.weight, self.bias, self.running_mean, self.running_var,
                                self.momentum, self.eps, residual, relu, stats)
        if (not self.training and self.track_running_stats
                and _bn.eval_supported(x, self.weight, self.bias, self.running_mean, self.running_var)):
            # evaluation passes (inference.py:199-213): running statistics, residual add and ReLU in one launch
            return _bn.bn_eval(x, self.weight, self.bias, self.running_mean, self.running_var, self.eps, residual, relu)
        if self.training:
            if _bn.log_active():
                raise _bn.LogModeUnsupported("a BatchNorm layer on the library path cannot log its batch statistics")
            _conv.library_path("batch_norm", x)
        if self.momentum is None or not self.track_running_stats:
            y = super().forward(x)
        else:
            y = nn.functional.batch_norm(x, self.running_mean, self.running_var, self.weight, self.bias,
                                         self.training, self.momentum, self.eps)
        if residual is not None:
            y = y + residual
        return nn.functional.relu(y) if relu else y


def _conv_bn(conv, bn, x, residual=None, relu=False):
    "relu?(bn(conv(x)) [+ residual]); the convolution's epilogue hands the batch statistics to the BN"
    if isinstance(conv, Conv2d) and isinstance(bn, _BatchNorm2d):
        if _conv.conv_bn_eval_supported(x, conv.weight, conv.bias, conv.conv_args, bn):
            # a test-set forward: the BatchNorm (running statistics), the shortcut's add and the ReLU ride in the
            # convolution's epilogue -- one launch (inference.py:199-213, exp_utils.py:250-340)
            return _conv.conv3x3_bn_eval(x, conv.weight, bn, residual, relu)
        want = bn.training and bn.track_running_stats and _bn.ENABLED
        y, stats = conv(x, want_stats=True) if want else (conv(x), None)
        return bn.fused(y, residual, relu, stats)
    y = bn(conv(x))
    if residual is not None:
        y = y + residual
    return nn.functional.relu(y) if relu else y


class _BNTrunk(nn.Sequential):
    "a Sequential that advances the batch counters of its ``_BatchNorm2d`` layers once per forward"

    def _flat_counters(self):
        """the BatchNorm layers' ``num_batches_tracked`` as 0-dim views of ONE int64 array, so that the head's launch
        (or one ATen launch) advances all of them; rebuilt whenever a layer's buffer is not such a view any more
        (``.to()``, ``deepcopy``, a re-registered buffer)"""
        bns = [m for m in self.modules() if isinstance(m, _BatchNorm2d) and m.track_running_stats
               and m.momentum is not None]
        if not bns:
            return None
        flat = self.__dict__.get("_bn_flat")
        if (flat is None or flat.numel() != len(bns) or flat.device != bns[0].num_batches_tracked.device
                or any(m.num_batches_tracked.data_ptr() != flat.data_ptr() + 8 * i for i, m in enumerate(bns))):
            with torch.no_grad():
                flat = torch.stack([m.num_batches_tracked.detach().reshape(()).to(torch.int64) for m in bns]).contiguous()
                for i, m in enumerate(bns):
                    m.num_batches_tracked = flat[i]
            self.__dict__["_bn_flat"] = flat
        return flat

    def forward(self, x):
        out, mods, i = x, list(self), 0
        # (log mode: the caller advances the counters afterwards)
        counters = self._flat_counters() if (self.training and not _bn.log_active()) else None
        if counters is not None and counters.numel() > 256:      # (the head's launch advances up to 256 counters)
            with torch.no_grad():
                counters.add_(1)
            counters = None
        counted = counters is None
        while i < len(mods):
            m = mods[i]
            if (isinstance(m, Conv2d) and i + 2 < len(mods) and isinstance(mods[i + 1], _BatchNorm2d)
                    and isinstance(mods[i + 2], nn.ReLU)):
                out, i = _conv_bn(m, mods[i + 1], out, relu=True), i + 3     # conv, BN + ReLU
            elif isinstance(m, _BatchNorm2d) and i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU):
                out, i = m.fused(out, relu=True), i + 2          # BN + ReLU in one operator
            elif (isinstance(m, nn.AvgPool2d) and i + 2 < len(mods) and isinstance(mods[i + 1], nn.Flatten)
                  and isinstance(mods[i + 2], Linear) and out.dim() == 4
                  and _all_of(m.kernel_size, out.shape[2:]) and m.padding in (0, (0, 0))
                  and _pool.head_supported(out, mods[i + 2].weight, mods[i + 2].bias)):
                lin = mods[i + 2]                                 # global average pool + linear head
                out, i = _pool.pool_linear(out, lin.weight, lin.bias, None if counted else counters), i + 3
                counted = True
            else:
                out, i = m(out), i + 1
        if not counted:
            with torch.no_grad():
                counters.add_(1)
        return out


class BasicBlock(nn.Module):
    "conv3x3-BN-ReLU-conv3x3-BN plus identity / 1x1-conv-BN shortcut, ReLU after the sum"

    def __init__(self, in_filters, filters, stride, conv_kwargs, batchnorm):
        super().__init__()
        self.main = nn.Sequential(
            Conv2dPrior(in_filters, filters, kernel_size=3, padding=1, stride=stride, **conv_kwargs),
            batchnorm(filters), nn.ReLU(),
            Conv2dPrior(filters, filters, kernel_size=3, padding=1, stride=1, **conv_kwargs),
            batchnorm(filters))
        if stride == 1:
            self.shortcut = nn.Identity()
        else:
            self.shortcut = nn.Sequential(
                Conv2dPrior(in_filters, filters, kernel_size=1, padding=0, stride=stride,
                            **conv_kwargs),
                batchnorm(filters))

    def _down_fused(self, x):
        c0, c1 = self.main[0], self.shortcut[0]
        return (isinstance(c0, Conv2d) and isinstance(c1, Conv2d) and c0.bias is None and c1.bias is None
                and c0.conv_args == (2, 1, 1, 1) and c1.conv_args == (2, 0, 1, 1)
                and isinstance(self.shortcut[1], _BatchNorm2d)
                and _conv.down_supported(x, c0.weight, c1.weight))

    def forward(self, x):
        m = self.main
        if isinstance(m[1], _BatchNorm2d):        # conv, BN+ReLU, conv, BN + shortcut + ReLU
            sc = self.shortcut
            if (isinstance(sc, nn.Identity) and isinstance(m[0], Conv2d) and isinstance(m[3], Conv2d)
                    and isinstance(m[4], _BatchNorm2d) and _bn.ENABLED
                    and _resblock.supported(x, m[0], m[1], m[3], m[4])):
                return _resblock.residual_block(x, m[0], m[1], m[3], m[4])      # the whole block: 3 launches each way
            if isinstance(sc, nn.Sequential) and self._down_fused(x):
                # down-sampling block: the strided 3x3 and the 1x1 shortcut read the same x -- one kernel
                want = m[1].training and m[1].track_running_stats and _bn.ENABLED
                ym, ys, *st = _conv.conv_down(x, m[0].weight, sc[0].weight, want)
                h = m[1].fused(ym, relu=True, stats=st[0] if want else None)
                if want and isinstance(m[3], Conv2d) and isinstance(m[4], _BatchNorm2d):
                    # the shortcut's BatchNorm is applied inside the block's last BatchNorm launch (bn.bn_train_dual)
                    y2, st2 = m[3](h, want_stats=True)
                    if _bn.dual_supported(y2, st2, ys, st[1], m[4], sc[1]):
                        return _bn.bn_train_dual(y2, st2, m[4], ys, st[1], sc[1])
                    return m[4].fused(y2, sc[1].fused(ys, stats=st[1]), True, st2)
                skip = sc[1].fused(ys, stats=st[1] if want else None)
            else:
                h = _conv_bn(m[0], m[1], x, relu=True)
                skip = _conv_bn(sc[0], sc[1], x) if isinstance(sc, nn.Sequential) else sc(x)
            return _conv_bn(m[3], m[4], h, residual=skip, relu=True)
        return nn.functional.relu(m(x) + self.shortcut(x))


def ResNet(softmax_temp=1., depth=20, num_classes=10, prior_w=prior.Normal, loc_w=0.,
           std_w=2 ** .5, prior_b=prior.Normal, loc_b=0., std_b=1., scaling_fn=None, bn=True,
           weight_prior_params={}, bias_prior_params={}, conv_prior_w=prior.Normal):
    if (depth - 2) % 6 != 0:
        raise ValueError('depth must be 6n+2 (e.g. 20, 32, 44).')
    conv_kwargs = dict(prior_w=conv_prior_w, loc_w=loc_w, std_w=std_w, prior_b=None,
                       scaling_fn=scaling_fn, weight_prior_params=weight_prior_params,
                       bias_prior_params=bias_prior_params)
    batchnorm = _BatchNorm2d if bn else nn.Identity
    blocks_per_stack, filters = (depth - 2) // 6, 16
    layers = [Conv2dPrior(3, filters, kernel_size=3, padding=1, stride=1, **conv_kwargs),
              batchnorm(filters), nn.ReLU()]
    for stack in range(3):
        stride = 1 if stack == 0 else 2
        prev, filters = filters, filters * stride
        layers.append(BasicBlock(prev, filters, stride, conv_kwargs, batchnorm))
        layers += [BasicBlock(filters, filters, 1, conv_kwargs, batchnorm)
                   for _ in range(blocks_per_stack - 1)]
    layers += [nn.AvgPool2d(8), nn.Flatten(),
               LinearPrior(filters, num_classes, prior_w=prior_w, loc_w=loc_w, std_w=std_w,
                           prior_b=prior_b, loc_b=loc_b, std_b=std_b, scaling_fn=scaling_fn,
                           weight_prior_params=weight_prior_params,
                           bias_prior_params=bias_prior_params)]
    return ClassificationModel((_BNTrunk if bn else nn.Sequential)(*layers), softmax_temp=softmax_temp)
