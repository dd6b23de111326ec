"""maxpool2x2(relu(x + bias)) as one operator (csrc/pool_hip.inc) against ATen, forward and backward."""
import pytest
import torch
import torch.nn.functional as F

from bnn_priors_amd import pool


def test_supported_is_false_on_cpu_and_odd_sizes():
    assert not pool.supported(torch.zeros(2, 3, 4, 4), None)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(128, 50, 28, 28), (128, 50, 14, 14), (5, 3, 6, 10), (1, 1, 2, 2)])
@pytest.mark.parametrize("with_bias,ties", [(True, False), (False, False), (True, True)])
def test_matches_aten(shape, with_bias, ties):
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(shape, generator=g)
    if ties:                      # coarse values: many equal maxima inside a window, many exact zeros
        x = (x * 2).round() / 2
    x = x.cuda()
    b = (torch.randn(shape[1], generator=g).cuda() * (0.5 if not ties else 0.0)) if with_bias else None
    dy = torch.randn(shape[0], shape[1], shape[2] // 2, shape[3] // 2, generator=g).cuda()
    assert pool.supported(x, b) and not pool.supported(x[:, :, :-1], b) and not pool.supported(x.double(), b)

    xr = x.clone().requires_grad_()
    br = b.clone().requires_grad_() if with_bias else None
    ref = F.max_pool2d(F.relu(xr + br.view(1, -1, 1, 1) if with_bias else xr), 2)
    ref.backward(dy)

    xg = x.clone().requires_grad_()
    bg = b.clone().requires_grad_() if with_bias else None
    y = pool.bias_relu_pool(xg, bg)
    y.backward(dy)
    assert torch.equal(y, ref)                       # max(x) + b == max(x + b) exactly
    assert torch.equal(xg.grad, xr.grad)             # same routing, ties included
    if with_bias:
        n_terms = shape[0] * shape[2] * shape[3] / 4
        torch.testing.assert_close(bg.grad, br.grad, rtol=1e-5, atol=2e-6 * n_terms ** .5)
        # accumulation (existing .grad): immediate reduction; reproducible
        g1 = bg.grad.clone()
        pool.bias_relu_pool(xg, bg).backward(dy)
        assert torch.equal(bg.grad, g1 + g1)


@pytest.mark.gpu
def test_nan_propagates_like_aten():
    x = torch.zeros(1, 1, 2, 4, device="cuda")
    x[0, 0, 1, 1] = float("nan")
    y = pool.bias_relu_pool(x, torch.ones(1, device="cuda"))
    ref = F.max_pool2d(F.relu(x + 1), 2)
    assert torch.isnan(y[0, 0, 0, 0]) and torch.isnan(ref[0, 0, 0, 0]) and y[0, 0, 0, 1] == ref[0, 0, 0, 1] == 1


@pytest.mark.gpu
def test_convnet_takes_the_fused_tail(monkeypatch):
    from bnn_priors_amd import models
    calls = []
    real = pool.bias_relu_pool
    monkeypatch.setattr(pool, "bias_relu_pool", lambda x, b=None: (calls.append(tuple(x.shape[1:])), real(x, b))[1])
    torch.manual_seed(0)
    x = torch.rand(6, 784)
    y = torch.randint(0, 10, (6,))
    kw = dict(width=50, depth=3, weight_prior="laplace", weight_loc=0., weight_scale=2 ** .5, bias_prior="gaussian",
              bias_loc=0., bias_scale=1., batchnorm=True, weight_prior_params={}, bias_prior_params={})
    net = models.get_model(x, y, "classificationconvnet", **kw).cuda()
    from bnn_priors_amd import conv
    fused = net.net(x.cuda())
    assert calls == []                  # round 3: each conv -> + bias -> ReLU -> pool triple is ONE operator ...
    monkeypatch.setattr(conv, "CONV_POOL", False)
    out = net.net(x.cuda())
    assert calls == [(50, 28, 28), (50, 14, 14)]       # ... and without it, convolution + fused tail
    assert torch.equal(fused, out)
    net.zero_grad()
    monkeypatch.setattr(conv, "CONV_POOL", True)
    F.cross_entropy(net.net(x.cuda()), y.cuda()).backward()
    fused_grads = {k: p.grad.clone() for k, p in net.named_parameters()}
    monkeypatch.setattr(conv, "CONV_POOL", False)
    monkeypatch.setattr(pool, "ENABLED", False)
    calls.clear()
    ref = net.net(x.cuda())
    assert not calls
    torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-5)
    # gradients through both routes
    grads = []
    for on in (True, False):
        monkeypatch.setattr(pool, "ENABLED", on)
        net.zero_grad()
        F.cross_entropy(net.net(x.cuda()), y.cuda()).backward()
        grads.append({k: p.grad.clone() for k, p in net.named_parameters()})
    for k in grads[0]:
        torch.testing.assert_close(grads[0][k], grads[1][k], rtol=1e-4, atol=1e-5 * max(1.0, grads[1][k].abs().max().item()), msg=k)
        torch.testing.assert_close(fused_grads[k], grads[1][k], rtol=1e-4, atol=1e-5 * max(1.0, grads[1][k].abs().max().item()), msg=k)


@pytest.mark.gpu
@pytest.mark.parametrize("n,c,hw,k", [(128, 64, 8, 10), (5, 64, 8, 10), (1, 16, 4, 3), (7, 48, 8, 16)])
@pytest.mark.parametrize("with_bias", [True, False])
def test_pool_linear_head_matches_aten(n, c, hw, k, with_bias):
    g = torch.Generator().manual_seed(n + c + k)
    h = torch.randn(n, c, hw, hw, generator=g).cuda()
    w = (torch.randn(k, c, generator=g) * 0.3).cuda()
    b = torch.randn(k, generator=g).cuda() if with_bias else None
    dl = torch.randn(n, k, generator=g).cuda()
    assert pool.head_supported(h, w, b) and not pool.head_supported(h.cpu(), w.cpu(), None)
    hd, wd = h.double().requires_grad_(), w.double().requires_grad_()
    bd = b.double().requires_grad_() if with_bias else None
    ref = F.linear(hd.mean(dim=(2, 3)), wd, bd)
    ref.backward(dl.double())
    hg, wg = h.clone().requires_grad_(), w.clone().requires_grad_()
    bg = b.clone().requires_grad_() if with_bias else None
    out = pool.pool_linear(hg, wg, bg)
    out.backward(dl)
    torch.testing.assert_close(out.double(), ref.detach(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(hg.grad.double(), hd.grad, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(wg.grad.double(), wd.grad, rtol=1e-4, atol=1e-5 * n ** .5)
    if with_bias:
        torch.testing.assert_close(bg.grad.double(), bd.grad, rtol=1e-4, atol=1e-5 * n ** .5)
    g1 = wg.grad.clone()                       # accumulation: immediate reduction, same bits
    pool.pool_linear(hg, wg, bg).backward(dl)
    assert torch.equal(wg.grad, g1 + g1) and not pool._conv._pending


@pytest.mark.gpu
@pytest.mark.parametrize("b,k", [(128, 10), (96, 10), (1, 2), (1024, 16), (77, 3)])
@pytest.mark.parametrize("reduction", ["mean", "sum"])
def test_softmax_cross_entropy_matches_aten(b, k, reduction):
    g = torch.Generator().manual_seed(b + k)
    logits = (torch.randn(b, k, generator=g) * 4).cuda()
    y = torch.randint(0, k, (b,), generator=g).cuda()
    assert pool.xent_supported(logits, y) and not pool.xent_supported(logits.cpu(), y.cpu())
    assert not pool.xent_supported(torch.zeros(2000, 10, device="cuda"), torch.zeros(2000, dtype=torch.int64, device="cuda"))
    ld = logits.double().requires_grad_()
    ref = F.cross_entropy(ld, y, reduction=reduction)
    (ref * 0.37).backward()
    lg = logits.clone().requires_grad_()
    out = pool.cross_entropy(lg, y, reduction)
    (out * 0.37).backward()
    torch.testing.assert_close(out.double(), ref.detach(), rtol=2e-6, atol=1e-6 * (1 if reduction == "mean" else b))
    torch.testing.assert_close(lg.grad.double(), ld.grad, rtol=1e-5, atol=1e-7)
    out2 = pool.cross_entropy(logits, y, reduction)
    assert torch.equal(out2, out.detach())                       # reproducible
    big = torch.zeros(2000, 10, device="cuda")
    assert pool.cross_entropy(big, torch.zeros(2000, dtype=torch.int64, device="cuda")).item() == pytest.approx(2.302585, rel=1e-5)
    with pytest.raises(ValueError):
        pool.cross_entropy(logits, y, "none")


@pytest.mark.gpu
@pytest.mark.parametrize("b,k", [(128, 10), (77, 3), (1024, 16)])
@pytest.mark.parametrize("reduction,divide_by", [("mean", None), ("sum", 50000), ("sum", None)])
def test_cross_entropy_backward_in_one_launch_has_the_bits_of_loss_backward(b, k, reduction, divide_by):
    """pool.cross_entropy_backward (forward + autograd seed in one launch, what the captured step and the exact pass
    use) == cross_entropy(...)[/ divide_by] followed by .backward(), bit for bit, through a layer upstream."""
    g = torch.Generator().manual_seed(7 * b + k)
    x = torch.randn(b, 12, generator=g).cuda()
    w = (torch.randn(k, 12, generator=g) * .5).cuda()
    y = torch.randint(0, k, (b,), generator=g).cuda()
    w1 = w.clone().requires_grad_()
    loss1 = pool.cross_entropy(x @ w1.t(), y, reduction)
    if divide_by is not None:
        loss1 = loss1 / divide_by
    loss1.backward()
    w2 = w.clone().requires_grad_()
    loss2 = pool.cross_entropy_backward(x @ w2.t(), y, reduction, divide_by=divide_by)
    assert not loss2.requires_grad
    assert torch.equal(loss1.detach(), loss2)
    assert torch.equal(w1.grad, w2.grad)
    # unsupported inputs take the plain route
    w3 = w.double().clone().requires_grad_()
    loss3 = pool.cross_entropy_backward(x.double() @ w3.t(), y, reduction, divide_by=divide_by)
    torch.testing.assert_close(loss3.float(), loss2, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(w3.grad.float(), w2.grad, rtol=1e-4, atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("n,j,k,with_bias", [(128, 2450, 10, True), (5, 2450, 10, False), (1, 7, 1, True), (128, 50, 16, True)])
def test_narrow_linear_matches_float64(n, j, k, with_bias):
    "the convolutional classifier's head Linear(2450, 10) (models/conv_nets.py:57-70) and other narrow shapes"
    g = torch.Generator().manual_seed(n + j + k)
    x, w = torch.randn(n, j, generator=g), torch.randn(k, j, generator=g) / j ** .5
    b = torch.randn(k, generator=g) if with_bias else None
    dy = torch.randn(n, k, generator=g)
    assert pool.linear_supported(x.cuda(), w.cuda(), None if b is None else b.cuda())
    assert not pool.linear_supported(x, w, b) and not pool.linear_supported(x.cuda(), torch.zeros(17, j).cuda(), None)
    xr, wr = x.double().requires_grad_(), w.double().requires_grad_()
    br = b.double().requires_grad_() if with_bias else None
    ref = F.linear(xr, wr, br)
    ref.backward(dy.double())
    xg, wg = x.cuda().requires_grad_(), w.cuda().requires_grad_()
    bg = b.cuda().requires_grad_() if with_bias else None
    y = pool.linear(xg, wg, bg)
    y.backward(dy.cuda())
    torch.testing.assert_close(y.double().cpu(), ref.detach(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(xg.grad.double().cpu(), xr.grad, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(wg.grad.double().cpu(), wr.grad, rtol=1e-5, atol=2e-6 * n ** .5)
    if with_bias:
        torch.testing.assert_close(bg.grad.double().cpu(), br.grad, rtol=1e-5, atol=2e-6 * n ** .5)
    y2 = pool.linear(xg, wg, bg)
    assert torch.equal(y2, y)


@pytest.mark.gpu
@pytest.mark.parametrize("n,reduction,divide_by", [(128, "mean", None), (80, "sum", 50000.0), (3, "mean", None)])
def test_fused_head_loss_launch_equals_the_three_separate_kernels(n, reduction, divide_by, monkeypatch):
    """head_loss: logits, loss rows, d loss / d logits and the head's whole backward in ONE launch -- the same bits as
    pool_linear forward + softmax cross-entropy (fwd + seed) + pool_linear backward, counters advanced on the way"""
    from bnn_priors_amd import pool
    g = torch.Generator().manual_seed(n)
    h0 = torch.randn(n, 64, 8, 8, generator=g).cuda()
    W0, b0 = (torch.randn(10, 64, generator=g) * 0.2).cuda(), torch.randn(10, generator=g).cuda()
    y = torch.randint(0, 10, (n,), generator=g).cuda()
    outs = []
    for fused in (False, True):
        monkeypatch.setattr(pool, "FUSED_HEAD", fused)
        h, W, b = h0.clone().requires_grad_(), W0.clone().requires_grad_(), b0.clone().requires_grad_()
        counters = torch.arange(21, dtype=torch.int64).cuda()
        with pool.head_loss(y, reduction, divide_by):
            f = pool.pool_linear(h, W, b, counters)
        assert hasattr(f, "_sgmcmc_head_loss") == fused
        loss = pool.cross_entropy_backward(f, y, reduction, divide_by)
        assert torch.equal(counters.cpu(), torch.arange(21) + 1)
        outs.append((f.detach(), loss.detach(), h.grad, W.grad, b.grad))
    for a, b_, name in zip(outs[0], outs[1], ("logits", "loss", "dh", "dW", "db")):
        if name == "loss":          # the rows are summed by another (ATen) reduction: last-ulp differences
            torch.testing.assert_close(a, b_, rtol=1e-6, atol=0)
        else:
            assert torch.equal(a, b_), name
    # other labels / another reduction than announced: the tagged logits are not used
    monkeypatch.setattr(pool, "FUSED_HEAD", True)
    h, W = h0.clone().requires_grad_(), W0.clone().requires_grad_()
    with pool.head_loss(y, reduction, divide_by):
        f = pool.pool_linear(h, W, None)
    y2 = (y + 1) % 10
    pool.cross_entropy_backward(f, y2, "mean")
    h2, W2 = h0.clone().requires_grad_(), W0.clone().requires_grad_()
    torch.nn.functional.cross_entropy(torch.nn.functional.linear(h2.mean(dim=(2, 3)), W2), y2).backward()
    torch.testing.assert_close(h.grad, h2.grad, rtol=1e-4, atol=1e-7)
    torch.testing.assert_close(W.grad, W2.grad, rtol=1e-4, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("n,reduction,divide_by", [(128, "mean", None), (96, "sum", 60000.0)])
def test_fused_linear_forward_and_loss_equals_the_two_kernels(n, reduction, divide_by, monkeypatch):
    "the convolutional classifier's head: linear forward + softmax cross-entropy seed in one launch, same bits"
    from bnn_priors_amd import pool
    g = torch.Generator().manual_seed(n)
    x0 = torch.randn(n, 2450, generator=g).cuda()
    W0, b0 = (torch.randn(10, 2450, generator=g) * 0.02).cuda(), torch.randn(10, generator=g).cuda()
    y = torch.randint(0, 10, (n,), generator=g).cuda()
    outs = []
    for fused in (False, True):
        monkeypatch.setattr(pool, "FUSED_HEAD", fused)
        x, W, b = x0.clone().requires_grad_(), W0.clone().requires_grad_(), b0.clone().requires_grad_()
        with pool.head_loss(y, reduction, divide_by):
            f = pool.linear(x, W, b)
        assert hasattr(f, "_sgmcmc_head_loss") == fused
        loss = pool.cross_entropy_backward(f, y, reduction, divide_by)
        outs.append((f.detach(), loss.detach(), x.grad, W.grad, b.grad))
    for a, b_, name in zip(outs[0], outs[1], ("logits", "loss", "dx", "dW", "db")):
        if name == "loss":
            torch.testing.assert_close(a, b_, rtol=1e-6, atol=0)
        else:
            assert torch.equal(a, b_), name


@pytest.mark.gpu
def test_only_the_last_linear_layer_takes_the_likelihood():
    """A narrow hidden Linear (<= 16 outputs) passes ``linear_supported`` too: with ``head_loss(..., head=head_of(model))``
    only the net's LAST linear layer computes the loss rows and the seed; the hidden one runs the plain forward."""
    from bnn_priors_amd import pool, prior
    from bnn_priors_amd.models import nets
    torch.manual_seed(5)
    net = torch.nn.Sequential(nets.LinearPrior(40, 8), torch.nn.ReLU(), nets.LinearPrior(8, 10)).cuda()
    model = type("M", (), {"net": net})()
    assert pool.head_of(model) is net[2] and pool.head_of(model) is net[2]        # (cached)
    x = torch.randn(32, 40).cuda()
    y = torch.randint(0, 10, (32,)).cuda()
    seen = {}
    for i in (0, 2):
        net[i].register_forward_hook(lambda m, a, out, i=i: seen.__setitem__(i, hasattr(out, "_sgmcmc_head_loss")))
    with pool.head_loss(y, head=pool.head_of(model)):
        f = net(x)
    assert seen == {0: False, 2: True}
    loss = pool.cross_entropy_backward(f, y)
    ref = torch.nn.functional.cross_entropy(
        torch.nn.functional.linear(torch.relu(torch.nn.functional.linear(x, net[0].weight, net[0].bias)),
                                   net[2].weight, net[2].bias), y)
    torch.testing.assert_close(loss, ref.detach(), rtol=1e-5, atol=1e-6)
    with pool.head_loss(y):                       # no head named: any supported layer (direct callers)
        net(x)
    assert seen == {0: True, 2: True}
