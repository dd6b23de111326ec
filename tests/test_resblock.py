"""The fused residual block (csrc/conv_fused_hip.inc, bnn_priors_amd/resblock.py) against a float64 PyTorch
statement of the same block -- conv3x3 -> BatchNorm(train) -> ReLU -> conv3x3 -> BatchNorm(train) -> (+x) -> ReLU
(reference: bnn_priors/models/google_resnet.py:34-43, 77-90) -- forward value, running statistics and every
gradient, at the three trunk shapes and batch sizes 128 / 5 / 1; plus run-to-run bit reproducibility (the
cross-workgroup reductions are ordered) and agreement with this repo's layer-by-layer path."""
import pytest
import torch
import torch.nn.functional as F

from bnn_priors_amd import _hip, conv

pytestmark = pytest.mark.gpu
SHAPES = sorted(conv.SHAPES)
# the measured alternatives exist only in a library built and loaded with SGMCMC_ALTERNATIVES=1 (include/sgmcmc_hip_alternatives.h)
ALT = pytest.mark.skipif(not _hip.ALTERNATIVES, reason="measured alternative: needs the SGMCMC_ALTERNATIVES=1 build")


def _block(c, seed=0):
    from bnn_priors_amd.models import nets
    from bnn_priors_amd import prior
    torch.manual_seed(seed)
    kw = dict(prior_w=prior.Normal, loc_w=0., std_w=2 ** .5, prior_b=None, scaling_fn=None,
              weight_prior_params={}, bias_prior_params={})
    blk = nets.BasicBlock(c, c, 1, kw, nets._BatchNorm2d).cuda()
    with torch.no_grad():
        for m in blk.modules():
            if isinstance(m, nets._BatchNorm2d):
                m.weight.copy_(torch.rand(c) + 0.5)
                m.bias.copy_(torch.randn(c) * 0.3)
                m.running_mean.copy_(torch.randn(c) * 0.1)
                m.running_var.copy_(torch.rand(c) + 0.5)
            if isinstance(m, nets.Conv2d):
                m.weight_prior.p.mul_((2.0 / (9 * c)) ** .5 / m.weight_prior.p.std())
    return blk


def _reference(blk, x, dout):
    "float64 autograd through plain PyTorch ops"
    m = blk.main
    p = dict(w1=m[0].weight_prior.p, g1=m[1].weight, b1=m[1].bias, w2=m[3].weight_prior.p, g2=m[4].weight, b2=m[4].bias)
    q = {k: v.detach().double().requires_grad_() for k, v in p.items()}
    xd = x.detach().double().requires_grad_()
    rm = [m[1].running_mean.double().clone(), m[4].running_mean.double().clone()]
    rv = [m[1].running_var.double().clone(), m[4].running_var.double().clone()]
    y1 = F.conv2d(xd, q["w1"], None, 1, 1)
    h = F.relu(F.batch_norm(y1, rm[0], rv[0], q["g1"], q["b1"], True, m[1].momentum, m[1].eps))
    y2 = F.conv2d(h, q["w2"], None, 1, 1)
    out = F.relu(F.batch_norm(y2, rm[1], rv[1], q["g2"], q["b2"], True, m[4].momentum, m[4].eps) + xd)
    out.backward(dout.double())
    return out.detach(), xd.grad, {k: v.grad for k, v in q.items()}, rm, rv


def _run(blk, x, dout):
    for p in blk.parameters():
        p.grad = None
    xs = x.detach().clone().requires_grad_()
    out = blk(xs)
    out.backward(dout)
    m = blk.main
    grads = dict(w1=m[0].weight_prior.p.grad, g1=m[1].weight.grad, b1=m[1].bias.grad, w2=m[3].weight_prior.p.grad,
                 g2=m[4].weight.grad, b2=m[4].bias.grad)
    return out.detach(), xs.grad, {k: v.clone() for k, v in grads.items()}


def _route(monkeypatch, route):
    """the three backward routes, at every shape: BatchNorm-backward sums from the convolution gradient's epilogue
    (default) / BatchNorm backward inside the convolution-gradient launch / as two launches of its own"""
    from bnn_priors_amd import resblock
    monkeypatch.setattr(resblock, "EPILOGUE_SUMS", route == "epilogue_sums")
    monkeypatch.setattr(resblock, "FUSED_BN_BWD", set(SHAPES) if route == "fused" else set())


ROUTES = ["epilogue_sums", pytest.param("fused", marks=ALT), "two_launch"]


@pytest.mark.parametrize("fold", [pytest.param(True, marks=ALT), False])
@pytest.mark.parametrize("route", ROUTES)
@pytest.mark.parametrize("c,hw", SHAPES)
@pytest.mark.parametrize("n", [128, 5, 1])
def test_fused_block_matches_float64_reference(c, hw, n, route, fold, monkeypatch):
    "fold: the first BatchNorm + ReLU applied inside the second convolution's staging (statistics from integer fx slots)"
    from bnn_priors_amd import resblock
    _route(monkeypatch, route)
    monkeypatch.setattr(resblock, "FOLD_BN", fold)
    blk = _block(c)
    blk.train()
    g = torch.Generator().manual_seed(100 * c + n)
    x = torch.relu(torch.randn(n, c, hw, hw, generator=g)).cuda()           # a block's input is post-ReLU
    dout = torch.randn(n, c, hw, hw, generator=g).cuda()
    assert resblock.supported(x, blk.main[0], blk.main[1], blk.main[3], blk.main[4])
    ref_out, ref_dx, ref_g, rm, rv = _reference(blk, x, dout)                # (before the running stats move)
    out, dx, grads = _run(blk, x, dout)
    m = blk.main
    scale = lambda t: max(1.0, t.abs().max().item())
    torch.testing.assert_close(out.double(), ref_out, rtol=1e-4, atol=1e-4 * scale(ref_out))
    torch.testing.assert_close(m[1].running_mean.double(), rm[0], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(m[1].running_var.double(), rv[0], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(m[4].running_mean.double(), rm[1], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(m[4].running_var.double(), rv[1], rtol=1e-5, atol=1e-6)
    # gradients: fp32 accumulation over up to N*HW*HW terms; ReLU masks at |h| ~ rounding may flip single elements
    torch.testing.assert_close(dx.double(), ref_dx, rtol=2e-3, atol=2e-4 * scale(ref_dx))
    for k in ref_g:
        torch.testing.assert_close(grads[k].double(), ref_g[k], rtol=2e-3, atol=3e-4 * scale(ref_g[k])), k


@pytest.mark.parametrize("fold", [pytest.param(True, marks=ALT), False])
@pytest.mark.parametrize("route", ROUTES[:2])
@pytest.mark.parametrize("c,hw", SHAPES)
def test_fused_block_is_bitwise_reproducible_and_matches_the_layered_path(c, hw, route, fold, monkeypatch):
    from bnn_priors_amd import resblock
    _route(monkeypatch, route)
    monkeypatch.setattr(resblock, "FOLD_BN", fold)
    g = torch.Generator().manual_seed(c)
    x = torch.relu(torch.randn(64, c, hw, hw, generator=g)).cuda()
    dout = torch.randn(64, c, hw, hw, generator=g).cuda()
    runs = []
    for _ in range(3):
        blk = _block(c, seed=5)
        blk.train()
        runs.append(_run(blk, x, dout) + (blk.main[1].running_var.clone(),))
    for r in runs[1:]:
        assert torch.equal(r[0], runs[0][0]) and torch.equal(r[1], runs[0][1]) and torch.equal(r[3], runs[0][3])
        for k in r[2]:
            assert torch.equal(r[2][k], runs[0][2][k]), k
    old = resblock.ENABLED
    resblock.ENABLED = False
    try:
        blk = _block(c, seed=5)
        blk.train()
        layered = _run(blk, x, dout)
    finally:
        resblock.ENABLED = old
    torch.testing.assert_close(runs[0][0], layered[0], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(runs[0][1], layered[1], rtol=2e-3, atol=2e-4 * max(1.0, layered[1].abs().max().item()))
    for k in layered[2]:
        torch.testing.assert_close(runs[0][2][k], layered[2][k], rtol=2e-3,
                                   atol=3e-4 * max(1.0, layered[2][k].abs().max().item()))


def test_eval_mode_and_off_table_shapes_take_the_layered_path():
    from bnn_priors_amd import resblock
    blk = _block(16)
    x = torch.randn(4, 16, 32, 32).cuda()
    blk.eval()
    assert not resblock.supported(x, blk.main[0], blk.main[1], blk.main[3], blk.main[4])
    blk.train()
    assert not resblock.supported(torch.randn(4, 16, 16, 16).cuda(), blk.main[0], blk.main[1], blk.main[3], blk.main[4])
    assert resblock.supported(x, blk.main[0], blk.main[1], blk.main[3], blk.main[4])


def test_batchnorm_backward_sums_ride_in_the_upstream_convolution_gradient(monkeypatch):
    """googleresnet (depth 20), one forward + backward: ALL 21 BatchNorm layers take their backward sums from
    the launch that produced their incoming gradient -- the 7 first BatchNorms of the identity blocks inside their
    block; 14 across operators through bnlink tags (stem, the down-sampling blocks' main BatchNorms, the identity
    blocks' second ones incl. those in front of a down-sampling pair, and the two shortcut BatchNorms, whose sums
    ride in the dx launch of the BatchNorm that adds them, and the last block's, whose gradient comes from the pooling
    + linear head's backward launch); none launches its own.  Gradients agree with the route switched off."""
    from bnn_priors_amd import bnlink, models
    torch.manual_seed(0)
    x = torch.randn(16, 3, 32, 32).cuda()
    y = torch.randint(0, 10, (16,)).cuda()
    net = models.get_model(x.cpu()[:2], torch.tensor([0, 9]), "googleresnet", width=50, depth=3, weight_prior="gaussian",
                           weight_scale=2 ** .5, bias_prior="gaussian", bias_scale=1.).cuda()
    models.he_initialize(net)
    net.train()
    state = {k: v.clone() for k, v in net.state_dict().items()}

    def grads():
        net.load_state_dict(state)
        for p in net.parameters():
            p.grad = None
        loss = torch.nn.functional.cross_entropy(net.net(x), y)
        loss.backward()
        torch.cuda.synchronize()
        return loss.item(), [p.grad.clone() for p in net.parameters()]

    from bnn_priors_amd import bn as bnmod
    bnlink.STATS.update(upstream=0, own=0)
    loss1, g1 = grads()
    # (the two shortcut BatchNorms live inside their block's last BatchNorm operator since round 3 -- bn.bn_train_dual --
    # and take their sums from its dx launch directly, without a tag: 12 tagged hand-overs; 14 with the two operators)
    assert bnlink.STATS == {"upstream": 12, "own": 0}, bnlink.STATS
    monkeypatch.setattr(bnmod, "DUAL", False)
    bnlink.STATS.update(upstream=0, own=0)
    loss2, g2 = grads()
    assert bnlink.STATS == {"upstream": 14, "own": 0}, bnlink.STATS
    assert loss2 == loss1 and all(torch.equal(a, b) for a, b in zip(g1, g2))
    monkeypatch.setattr(bnlink, "ENABLED", False)
    bnlink.STATS.update(upstream=0, own=0)
    loss0, g0 = grads()
    assert bnlink.STATS["upstream"] == 0
    assert loss0 == loss1
    for a, b in zip(g1, g0):
        torch.testing.assert_close(a, b, rtol=2e-3, atol=2e-5 * max(1.0, b.abs().max().item()))
    # a gradient tensor that was touched after its producer tagged it is not trusted
    t = torch.zeros(4, device="cuda")
    monkeypatch.setattr(bnlink, "ENABLED", True)
    bnlink.tag_gradient(t, torch.zeros(1), 1)
    assert bnlink.sums_of(t) is not None
    t.add_(1.0)
    assert bnlink.sums_of(t) is None


@pytest.mark.parametrize("route", ["blocks", "layered", "persistent"])
def test_gradients_stored_masked_by_their_producers_change_no_bit(route, monkeypatch):
    """bnlink.PREMASK: a launch that leaves a BatchNorm's backward sums stores its data gradient as dx * [out > 0]; the
    BatchNorm's dx launch then runs without the ReLU mask (it does not read `out`) and the shortcut add takes the
    gradient as it is.  Every parameter gradient of googleresnet keeps its bits -- fused residual blocks, the
    layer-by-layer operators (conv3x3 / conv_down / bn_train / bn_train_dual) and the persistent convolutions."""
    from bnn_priors_amd import bnlink, models, resblock
    torch.manual_seed(1)
    x = torch.randn(24, 3, 32, 32).cuda()
    y = torch.randint(0, 10, (24,)).cuda()
    net = models.get_model(x.cpu()[:2], torch.tensor([0, 9]), "googleresnet", width=50, depth=3, weight_prior="gaussian",
                           weight_scale=2 ** .5, bias_prior="gaussian", bias_scale=1.).cuda()
    models.he_initialize(net)
    net.train()
    state = {k: v.clone() for k, v in net.state_dict().items()}
    if route == "layered":
        monkeypatch.setattr(resblock, "ENABLED", False)

    def grads():
        net.load_state_dict(state)
        for p in net.parameters():
            p.grad = None
        if route == "persistent":
            with conv.persistent():
                loss = torch.nn.functional.cross_entropy(net.net(x), y)
                loss.backward()
        else:
            loss = torch.nn.functional.cross_entropy(net.net(x), y)
            loss.backward()
        torch.cuda.synchronize()
        return loss.item(), [p.grad.clone() for p in net.parameters()]

    assert bnlink.PREMASK
    bnlink.STATS.update(upstream=0, own=0)
    loss1, g1 = grads()
    # layered: the 7 identity blocks' inputs take their gradient from two operators; autograd's sum carries no tag, the
    # BatchNorm in front launches its own sums and masks the (partly masked) sum again -- the fallback, same bits
    assert bnlink.STATS["own"] == (7 if route == "layered" else 0)
    monkeypatch.setattr(bnlink, "PREMASK", False)
    loss0, g0 = grads()
    assert loss0 == loss1
    assert all(torch.equal(a, b) for a, b in zip(g1, g0))


@ALT
def test_weight_gradients_on_a_side_stream_have_the_same_bits(monkeypatch):
    """conv.SIDE_STREAM (off by default: measured slower inside a replayed graph): the weight-gradient half of every
    trunk convolution's backward on a second stream, joined before the pass's slab reduction -- same workgroups, so
    every gradient is bit-identical to the merged launches'."""
    from bnn_priors_amd import conv, models
    torch.manual_seed(1)
    x = torch.randn(8, 3, 32, 32).cuda()
    y = (torch.arange(8) % 10).cuda()
    net = models.get_model(x.cpu()[:2], torch.tensor([0, 9]), "googleresnet", width=50, depth=3, weight_prior="gaussian",
                           weight_scale=2 ** .5, bias_prior="gaussian", bias_scale=1.).cuda()
    net.train()
    state = {k: v.clone() for k, v in net.state_dict().items()}
    outs = []
    for side in (False, True):
        monkeypatch.setattr(conv, "SIDE_STREAM", side)
        net.load_state_dict(state)
        for p in net.parameters():
            p.grad = None
        with conv.deferring():
            torch.nn.functional.cross_entropy(net.net(x), y).backward()
        torch.cuda.synchronize()
        assert not conv._pending and conv._side["forked"] is None and not conv._side["keep"]
        outs.append([p.grad.clone() for p in net.parameters()])
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@ALT
@pytest.mark.parametrize("c,hw", SHAPES)
def test_folded_first_batchnorm_logs_its_statistics_and_sums_exactly(c, hw, monkeypatch):
    """the folded route in logging mode (the exact pass's lanes): the first BatchNorm's batch mean / unbiased variance
    go to its slot, the running statistics stay; and the integer fx totals do not depend on the order of the atomics:
    the saved statistics of two runs -- one of them with the OTHER half of the GPU kept busy by a second stream -- are
    bit-identical"""
    from bnn_priors_amd import bn as bnmod
    from bnn_priors_amd import resblock
    from bnn_priors_amd.models import nets
    monkeypatch.setattr(resblock, "FOLD_BN", True)
    g = torch.Generator().manual_seed(7 * c)
    x = torch.relu(torch.randn(32, c, hw, hw, generator=g)).cuda()
    blk = _block(c, seed=9)
    blk.train()
    bns = [m for m in blk.modules() if isinstance(m, nets._BatchNorm2d)]
    slots = {id(m.running_mean): torch.zeros((c, 2), dtype=torch.float64, device="cuda") for m in bns}
    before = [(m.running_mean.clone(), m.running_var.clone()) for m in bns]
    with torch.no_grad(), bnmod.logging_running_stats(slots):
        out_log = blk(x)
    for m, (rm, rv) in zip(bns, before):
        assert torch.equal(m.running_mean, rm) and torch.equal(m.running_var, rv)
    with torch.no_grad():
        out = blk(x)                                   # the same batch, running statistics advanced this time
    assert torch.equal(out, out_log)
    mom = bns[0].momentum
    want_mean = (1 - mom) * before[0][0].double() + mom * slots[id(bns[0].running_mean)][:, 0]
    want_var = (1 - mom) * before[0][1].double() + mom * slots[id(bns[0].running_mean)][:, 1]
    assert torch.equal(bns[0].running_mean, want_mean.float()) and torch.equal(bns[0].running_var, want_var.float())
    y1 = torch.nn.functional.conv2d(x.double(), blk.main[0].weight.double(), padding=1)
    torch.testing.assert_close(slots[id(bns[0].running_mean)][:, 0], y1.mean(dim=(0, 2, 3)), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(slots[id(bns[0].running_mean)][:, 1], y1.var(dim=(0, 2, 3), unbiased=True), rtol=1e-5, atol=1e-6)
    # a busy neighbour stream changes when and where workgroups run, not the sums
    side = torch.cuda.Stream()
    junk = torch.randn(4096, 4096, device="cuda")
    with torch.no_grad():
        with torch.cuda.stream(side):
            for _ in range(20):
                junk = junk @ junk * 1e-3
        out2 = blk(x)
    torch.cuda.synchronize()
    assert torch.equal(out2, out)                      # (training mode: the output does not depend on the running statistics)
