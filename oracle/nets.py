"""Plain-torch restatement of the three BASELINE networks and of the potential the runners
differentiate (TEST INFRASTRUCTURE; the model side of ``bench.py``'s ``cpu_baseline``).

The CPU baseline must not lean on the product: ``oracle/runner.py`` times the reference's loop body
on THESE modules -- ``torch.nn.functional`` layers, ``torch.distributions`` priors and likelihood,
evaluated statement by statement as the reference does:

* ``Classifier.split_potential_and_acc`` = ``models/base.py:57-62,72-77,187-191``: loss =
  -(1/B) sum_i Categorical(logits=f_i).log_prob(y_i); log_prior = sum over the priors, in module
  order, of ``dist(loc, scale).log_prob(p).sum()`` (``prior/base.py:57-58``, ``models/base.py:25-30``);
  potential_avg = loss - log_prior / N; accuracy per example;
* ``densenet`` = ``models/dense_nets.py:48-67`` (Linear priors N(0, std_w / sqrt(in)), bias N(0, std_b):
  ``dense_nets.py:16-23``); ``convnet`` = ``models/conv_nets.py:46-70`` (conv priors with scale
  std_w / sqrt(in_channels): ``conv_nets.py:26-31``); ``googleresnet`` = ``models/google_resnet.py:11-110``
  (depth 20, BatchNorm after every convolution, convolutions without bias and ALWAYS Gaussian --
  ``weight_prior`` reaches the final Linear only, ``exp_utils.py:186-190``);
* ``he_initialize`` = ``exp_utils.py:63-69``.

Parameter order equals the reference's ``named_parameters()`` order, so ``load_parameters`` can copy a
reference-shaped model's values one to one (tests/test_cpu_baseline_pin.py pins the arithmetic that way).
Never imported by ``bnn_priors_amd``.
"""
import math

import torch
import torch.nn.functional as F
from torch import nn

_DISTS = {"gaussian": torch.distributions.Normal, "laplace": torch.distributions.Laplace,
          "student-t": lambda loc, scale: torch.distributions.StudentT(3., loc, scale)}


class _Prior(nn.Module):
    "a parameter with an element-wise prior: ``.p`` and ``log_prob()`` (prior/base.py:40-58)"

    def __init__(self, shape, family, loc, scale):
        super().__init__()
        self.family = family
        self.register_buffer("loc", torch.tensor(float(loc)))
        self.register_buffer("scale", torch.tensor(float(scale)))
        self.p = nn.Parameter(torch.zeros(shape))

    def log_prob(self):
        return _DISTS[self.family](self.loc, self.scale).log_prob(self.p).sum()


class _Linear(nn.Module):
    def __init__(self, in_dim, out_dim, family, std_w, std_b):
        super().__init__()
        self.weight_prior = _Prior((out_dim, in_dim), family, 0., std_w / in_dim ** 0.5)
        self.bias_prior = _Prior((out_dim,), "gaussian", 0., std_b)

    def forward(self, x):
        return F.linear(x, self.weight_prior.p, self.bias_prior.p)


class _Conv(nn.Module):
    def __init__(self, cin, cout, k, family, std_w, std_b=None, stride=1, padding=0):
        super().__init__()
        self.stride, self.padding = stride, padding
        self.weight_prior = _Prior((cout, cin, k, k), family, 0., std_w / cin ** 0.5)
        self.bias_prior = None if std_b is None else _Prior((cout,), "gaussian", 0., std_b)

    def forward(self, x):
        return F.conv2d(x, self.weight_prior.p, None if self.bias_prior is None else self.bias_prior.p,
                        self.stride, self.padding)


class _Reshape(nn.Module):
    def __init__(self, *shape):
        super().__init__()
        self.shape = shape

    def forward(self, x):
        return x.view(self.shape)


class _Block(nn.Module):
    "models/google_resnet.py:11-32"

    def __init__(self, cin, cout, stride, std_w):
        super().__init__()
        self.main = nn.Sequential(_Conv(cin, cout, 3, "gaussian", std_w, stride=stride, padding=1),
                                  nn.BatchNorm2d(cout), nn.ReLU(),
                                  _Conv(cout, cout, 3, "gaussian", std_w, padding=1), nn.BatchNorm2d(cout))
        self.shortcut = (nn.Identity() if stride == 1 else
                         nn.Sequential(_Conv(cin, cout, 1, "gaussian", std_w, stride=stride), nn.BatchNorm2d(cout)))

    def forward(self, x):
        return F.relu(self.main(x) + self.shortcut(x))


class Classifier(nn.Module):
    "models/base.py:20-77,168-191 for a categorical likelihood with softmax temperature 1"

    def __init__(self, net):
        super().__init__()
        self.net = net

    def priors(self):
        return [m for m in self.modules() if isinstance(m, _Prior)]

    def log_prior(self):
        return sum(p.log_prob() for p in self.priors())

    def forward(self, x):
        return torch.distributions.Categorical(logits=self.net(x) / 1.)

    def split_potential_and_acc(self, x, y, eff_num_data):
        assert x.shape[0] == y.shape[0]
        preds = self(x)
        lla = preds.log_prob(y).sum() * (1 / x.shape[0])
        loss = -lla
        log_prior = self.log_prior()
        potential_avg = loss - log_prior / eff_num_data
        acc = torch.argmax(preds.logits, dim=1).eq(y).to(torch.float32)
        return loss, log_prior, potential_avg, acc, preds


def densenet(in_features=784, out_features=10, width=50, depth=3, weight_prior="gaussian", std_w=2 ** .5, std_b=1.):
    layers = [_Linear(in_features, width, weight_prior, std_w, std_b), nn.ReLU()]
    for _ in range(depth - 2):
        layers += [_Linear(width, width, weight_prior, std_w, std_b), nn.ReLU()]
    layers.append(_Linear(width, out_features, weight_prior, std_w, std_b))
    return Classifier(nn.Sequential(*layers))


def convnet(in_channels=1, img_height=28, out_features=10, width=50, depth=3, weight_prior="laplace", std_w=2 ** .5,
            std_b=1.):
    layers = [_Reshape(-1, in_channels, img_height, img_height),
              _Conv(in_channels, width, 3, weight_prior, std_w, std_b, padding=1), nn.ReLU(), nn.MaxPool2d(2)]
    for _ in range(depth - 2):
        layers += [_Conv(width, width, 3, weight_prior, std_w, std_b, padding=1), nn.ReLU(), nn.MaxPool2d(2)]
    layers += [nn.Flatten(),
               _Linear(width * (img_height // 2 ** (depth - 1)) ** 2, out_features, weight_prior, std_w, std_b)]
    return Classifier(nn.Sequential(*layers))


def googleresnet(depth=20, num_classes=10, weight_prior="gaussian", std_w=2 ** .5, std_b=1.):
    blocks, filters = (depth - 2) // 6, 16
    layers = [_Conv(3, filters, 3, "gaussian", std_w, padding=1), nn.BatchNorm2d(filters), nn.ReLU()]
    for stack in range(3):
        stride = 1 if stack == 0 else 2
        prev, filters = filters, filters * stride
        layers.append(_Block(prev, filters, stride, std_w))
        layers += [_Block(filters, filters, 1, std_w) for _ in range(blocks - 1)]
    layers += [nn.AvgPool2d(8), nn.Flatten(), _Linear(filters, num_classes, weight_prior, std_w, std_b)]
    return Classifier(nn.Sequential(*layers))


BUILDERS = {"classificationdensenet": densenet, "classificationconvnet": convnet, "googleresnet": googleresnet}


def he_initialize(model):
    "exp_utils.py:63-69"
    for name, param in model.named_parameters():
        if "weight_prior.p" in name:
            torch.nn.init.kaiming_normal_(param.data, mode="fan_in", nonlinearity="relu")
        elif "bias_prior.p" in name:
            bound = 1 / math.sqrt(param.size(0))
            torch.nn.init.uniform_(param.data, -bound, bound)


def load_parameters(model, other):
    "copy the values of a reference-shaped model (same parameter / buffer order) into ``model``"
    mine, theirs = list(model.parameters()), list(other.parameters())
    assert [tuple(p.shape) for p in mine] == [tuple(p.shape) for p in theirs]
    with torch.no_grad():
        for p, q in zip(mine, theirs):
            p.copy_(q)
    return model
