"""Fused training-mode BatchNorm (+ residual + ReLU) kernels (csrc/bn_hip.inc) against ATen's
batch_norm / add / relu in float64, forward and backward, including the running statistics."""
import pytest
import torch
import torch.nn.functional as F

from bnn_priors_amd import bn


def _ref(x, w, b, rm, rv, res, relu, momentum=0.1, eps=1e-5, mask=None):
    """float64 reference; ``mask``: use THIS ReLU pattern (the fp32 output's) instead of the float64 one --
    a pre-activation of 1e-8 may land on either side of zero, and one flipped element moves its whole
    channel's gradient sums"""
    y = F.batch_norm(x, rm, rv, w, b, True, momentum, eps)
    if res is not None:
        y = y + res
    if not relu:
        return y
    return F.relu(y) if mask is None else y * mask


def test_supported_is_false_on_cpu_and_in_eval_mode():
    x, w = torch.zeros(2, 4, 8, 8), torch.ones(4)
    assert not bn.supported(x, w, w, True, 0.1)


@pytest.mark.gpu
@pytest.mark.parametrize("n,c,hw", [(128, 16, 32), (128, 32, 16), (128, 64, 8), (80, 16, 32), (96, 64, 8), (5, 16, 32), (1, 64, 8), (37, 3, 6)])
@pytest.mark.parametrize("relu,residual", [(True, False), (True, True), (False, False), (False, True)])
def test_fused_bn_matches_aten(n, c, hw, relu, residual):
    g = torch.Generator().manual_seed(n * 1000 + c)
    x = (torch.randn(n, c, hw, hw, generator=g) * 1.7 + 0.4).cuda()
    res = torch.randn(n, c, hw, hw, generator=g).cuda() if residual else None
    w = (torch.rand(c, generator=g) + 0.5).cuda()
    b = torch.randn(c, generator=g).cuda()
    dy = torch.randn(n, c, hw, hw, generator=g).cuda()
    rm0, rv0 = torch.randn(c, generator=g).cuda(), (torch.rand(c, generator=g) + 0.5).cuda()
    assert bn.supported(x, w, b, True, 0.1) and not bn.supported(x, w, b, False, 0.1)
    assert not bn.supported(x, w, b, True, None) and not bn.supported(x.double(), w, b, True, 0.1)

    leaves = [t.clone().requires_grad_() for t in (x, w, b)] + ([res.clone().requires_grad_()] if residual else [])
    rm, rv = rm0.clone(), rv0.clone()
    y = bn.bn_train(leaves[0], leaves[1], leaves[2], rm, rv, 0.1, 1e-5, leaves[3] if residual else None, relu)
    y.backward(dy)

    # float64 reference (with the fp32 output's ReLU pattern)
    leaves64 = [t.double().requires_grad_() for t in (x, w, b)] + ([res.double().requires_grad_()] if residual else [])
    rm64, rv64 = rm0.double().clone(), rv0.double().clone()
    y64 = _ref(leaves64[0], leaves64[1], leaves64[2], rm64, rv64, leaves64[3] if residual else None, relu,
               mask=(y.detach() > 0).double())
    y64.backward(dy.double())

    M = n * hw * hw
    torch.testing.assert_close(y.double(), y64.detach(), rtol=1e-5, atol=2e-5)
    torch.testing.assert_close(rm.double(), rm64, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(rv.double(), rv64, rtol=1e-5, atol=1e-6)
    if relu:    # the pattern itself: identical to float64's except where the pre-activation is ~1e-7
        pre = _ref(x.double(), w.double(), b.double(), None, None, res.double() if residual else None, False)
        flipped = (y > 0) != (pre > 0)
        assert flipped.sum() <= 8 and (pre[flipped].abs() < 1e-5).all()
    torch.testing.assert_close(leaves[0].grad.double(), leaves64[0].grad, rtol=2e-4, atol=2e-5 + 2e-4 / M ** .5)
    scale = M ** .5
    torch.testing.assert_close(leaves[1].grad.double(), leaves64[1].grad, rtol=1e-4, atol=3e-5 * scale)
    torch.testing.assert_close(leaves[2].grad.double(), leaves64[2].grad, rtol=1e-4, atol=3e-5 * scale)
    if residual:
        torch.testing.assert_close(leaves[3].grad.double(), leaves64[3].grad, rtol=1e-6, atol=1e-6)

    # reproducible bit for bit
    leaves2 = [t.clone().requires_grad_() for t in (x, w, b)] + ([res.clone().requires_grad_()] if residual else [])
    y2 = bn.bn_train(leaves2[0], leaves2[1], leaves2[2], rm0.clone(), rv0.clone(), 0.1, 1e-5,
                     leaves2[3] if residual else None, relu)
    y2.backward(dy)
    assert torch.equal(y2, y)
    for a, bb in zip(leaves, leaves2):
        assert torch.equal(a.grad, bb.grad)


@pytest.mark.gpu
def test_resnet_forward_backward_matches_the_library_path(monkeypatch):
    "whole googleresnet: fused conv / BN operators on vs off -- same loss, gradients, running statistics"
    from bnn_priors_amd import conv, models
    torch.manual_seed(0)
    x = torch.randn(16, 3, 32, 32)
    y = torch.randint(0, 10, (16,))

    def run(on):
        monkeypatch.setattr(conv, "ENABLED", on)
        monkeypatch.setattr(bn, "ENABLED", on)
        torch.manual_seed(1)
        net = models.get_model(x, y, "googleresnet", width=50, depth=3, weight_prior="gaussian", weight_loc=0.,
                               weight_scale=2 ** .5, bias_prior="gaussian", bias_loc=0., bias_scale=1.,
                               batchnorm=True, weight_prior_params={}, bias_prior_params={}).cuda()
        net.train()
        loss = F.cross_entropy(net.net(x.cuda()), y.cuda())
        loss.backward()
        grads = {k: p.grad.clone() for k, p in net.named_parameters()}
        return loss.item(), grads, {k: v.clone() for k, v in net.state_dict().items()}

    l1, g1, s1 = run(True)
    l0, g0, s0 = run(False)
    assert abs(l1 - l0) < 1e-4 * max(1.0, abs(l0))
    # Two fp32 evaluation orders of a 20-layer backward pass agree to ~1e-3 of a tensor's scale -- except
    # where a ReLU whose pre-activation is ~1e-7 lands on different sides of zero in the two paths: that one
    # element moves its channel's BatchNorm sums and with them a few rows of the neighbouring weight
    # gradients by several per cent (either path shows the same against a float64 run).  A wrong kernel is
    # off by O(1) everywhere.
    rel = {k: (g1[k] - g0[k]).abs().max().item() / (g0[k].abs().max().item() + 1e-12) for k in g0}
    assert max(rel.values()) < 0.25, max(rel.items(), key=lambda kv: kv[1])
    assert sorted(rel.values())[int(0.8 * len(rel))] < 5e-3, sorted(rel.items(), key=lambda kv: -kv[1])[:5]
    for k in s0:
        if "running" in k or "num_batches" in k:
            torch.testing.assert_close(s1[k], s0[k], rtol=1e-4, atol=1e-5, msg=k)


@pytest.mark.gpu
@pytest.mark.parametrize("offset", [0.0, 30.0, 3000.0])
def test_statistics_do_not_cancel_against_the_mean(offset):
    "a channel whose mean dwarfs its spread: variance and output still match a float64 evaluation"
    g = torch.Generator().manual_seed(7)
    x = (torch.randn(64, 16, 32, 32, generator=g) * 0.25 + offset).cuda()
    w, b = torch.ones(16, device="cuda"), torch.zeros(16, device="cuda")
    rm, rv = torch.zeros(16, device="cuda"), torch.ones(16, device="cuda")
    y = bn.bn_train(x, w, b, rm, rv, 1.0, 1e-5)                      # momentum 1: running stats = batch stats
    xd = x.double()
    var = xd.var(dim=(0, 2, 3), unbiased=True)
    torch.testing.assert_close(rv.double(), var, rtol=2e-4, atol=0)
    torch.testing.assert_close(rm.double(), xd.mean(dim=(0, 2, 3)), rtol=1e-6, atol=1e-6)
    ref = (xd - xd.mean(dim=(0, 2, 3), keepdim=True)) / (xd.var(dim=(0, 2, 3), unbiased=False, keepdim=True) + 1e-5).sqrt()
    # the input itself carries offset * 2^-24 of rounding: that, not the statistics, bounds the agreement
    torch.testing.assert_close(y.double(), ref, rtol=0, atol=1e-4 + 4 * offset * 2.0 ** -24 / 0.25)


@pytest.mark.gpu
@pytest.mark.parametrize("c,hw,n", [(16, 32, 128), (32, 16, 5), (64, 8, 1)])
@pytest.mark.parametrize("relu,with_residual", [(True, False), (True, True), (False, False)])
def test_eval_mode_batchnorm_kernel_matches_aten(c, hw, n, relu, with_residual):
    "model.eval(): running statistics + residual + ReLU in one launch == F.batch_norm(training=False) (+ add, relu)"
    from bnn_priors_amd import bn
    g = torch.Generator().manual_seed(c + n)
    x = (torch.randn(n, c, hw, hw, generator=g) * 2 + 0.5).cuda()
    res = torch.randn(n, c, hw, hw, generator=g).cuda() if with_residual else None
    w, b = (torch.rand(c, generator=g) + 0.5).cuda(), torch.randn(c, generator=g).cuda()
    rm, rv = torch.randn(c, generator=g).cuda(), (torch.rand(c, generator=g) + 0.1).cuda()
    with torch.no_grad():
        assert bn.eval_supported(x, w, b, rm, rv)
        got = bn.bn_eval(x, w, b, rm, rv, 1e-5, res, relu)
        ref = torch.nn.functional.batch_norm(x, rm, rv, w, b, False, 0.1, 1e-5)
        if res is not None:
            ref = ref + res
        if relu:
            ref = torch.relu(ref)
    torch.testing.assert_close(got, ref, rtol=1e-5, atol=1e-5)
    assert not bn.eval_supported(x, w, b, rm, rv)            # with grad mode on: the differentiable library operator


@pytest.mark.gpu
def test_eval_mode_forward_of_the_resnet_runs_on_this_packages_kernels():
    """the per-epoch evaluation pass (inference.py:199-213): same logits as the library path, and no ATen batch_norm /
    MIOpen convolution in it"""
    from bnn_priors_amd import bn, conv, models, pool, resblock
    torch.manual_seed(0)
    x = torch.randn(64, 3, 32, 32).cuda()
    net = models.get_model(x.cpu()[:2], torch.tensor([0, 9]), "googleresnet", width=50, depth=3, weight_prior="gaussian",
                           weight_scale=2 ** .5, bias_prior="gaussian", bias_scale=1.).cuda()
    models.he_initialize(net)
    net.train()
    with torch.no_grad():
        net.net(x)                       # running statistics away from their initial values
    net.eval()
    with torch.no_grad():
        got = net.net(x)
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CPU]) as prof:
            net.net(x)
        names = {e.key for e in prof.key_averages()}
        assert not any("batch_norm" in k or "convolution" in k for k in names), sorted(names)
        mods = (bn, conv, pool, resblock)
        old = [m.ENABLED for m in mods]
        try:
            for m in mods:
                m.ENABLED = False
            ref = net.net(x)
        finally:
            for m, v in zip(mods, old):
                m.ENABLED = v
    torch.testing.assert_close(got, ref, rtol=1e-3, atol=1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("cin,hw,n", [(16, 32, 128), (32, 16, 128), (16, 32, 5)])
def test_down_sampling_block_with_the_shortcut_batchnorm_applied_on_the_fly_has_the_bits_of_two_launches(cin, hw, n, monkeypatch):
    """relu(BN(conv2(h)) + BN_s(shortcut(x))) in ONE launch (bn.bn_train_dual): output, every gradient, the saved and the
    running statistics of both layers are bit-identical to the two-operator route; so is the logging mode."""
    from bnn_priors_amd import bn as bnmod
    from bnn_priors_amd import conv
    from bnn_priors_amd.models import nets
    from bnn_priors_amd import prior

    def make():
        torch.manual_seed(3)
        kw = dict(prior_w=prior.Normal, loc_w=0., std_w=2 ** .5, prior_b=None, scaling_fn=None, weight_prior_params={},
                  bias_prior_params={})
        blk = nets.BasicBlock(cin, 2 * cin, 2, kw, nets._BatchNorm2d).cuda().train()
        with torch.no_grad():
            for m in blk.modules():
                if isinstance(m, nets._BatchNorm2d):
                    m.weight.uniform_(0.5, 1.5)
                    m.bias.uniform_(-0.5, 0.5)
        return blk

    g = torch.Generator().manual_seed(cin + n)
    x = torch.randn(n, cin, hw, hw, generator=g).cuda()
    dout = torch.randn(n, 2 * cin, hw // 2, hw // 2, generator=g).cuda()
    results = []
    for dual in (True, False):
        monkeypatch.setattr(bnmod, "DUAL", dual)
        blk = make()
        calls = []
        real = bnmod.bn_train_dual
        monkeypatch.setattr(bnmod, "bn_train_dual", lambda *a: (calls.append(1), real(*a))[1])
        xg = x.clone().requires_grad_()
        out = blk(xg)
        out.backward(dout)
        assert bool(calls) == dual
        monkeypatch.setattr(bnmod, "bn_train_dual", real)
        results.append((out.detach(), xg.grad, {k: p.grad for k, p in blk.named_parameters()},
                        {k: v.clone() for k, v in blk.named_buffers()}))
    (o1, gx1, gp1, b1), (o0, gx0, gp0, b0) = results
    assert torch.equal(o1, o0) and torch.equal(gx1, gx0)
    for k in gp0:
        assert torch.equal(gp1[k], gp0[k]), k
    for k in b0:
        assert torch.equal(b1[k], b0[k]), k
    # logging mode: the batch statistics go to the slots, the running statistics stay
    blk = make()
    bns = [m for m in blk.modules() if isinstance(m, nets._BatchNorm2d)]
    logs = {}
    for dual in (True, False):
        monkeypatch.setattr(bnmod, "DUAL", dual)
        slots = {id(m.running_mean): torch.zeros((m.num_features, 2), dtype=torch.float64, device="cuda") for m in bns}
        before = [m.running_mean.clone() for m in bns]
        with torch.no_grad(), bnmod.logging_running_stats(slots):
            out = blk(x)
        assert all(torch.equal(m.running_mean, b) for m, b in zip(bns, before))
        logs[dual] = (out, [slots[id(m.running_mean)].clone() for m in bns])
    assert torch.equal(logs[True][0], logs[False][0])
    for a, b in zip(logs[True][1], logs[False][1]):
        assert torch.equal(a, b) and a.abs().sum() > 0


# ---- several minibatches per launch (bn.grouped; csrc/bn_hip.inc GROUPS) ---------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("G,nb,c,hw", [(4, 128, 16, 32), (3, 128, 32, 16), (2, 128, 64, 8), (4, 5, 16, 32), (3, 1, 64, 8)])
@pytest.mark.parametrize("relu,residual", [(True, False), (True, True), (False, False)])
def test_grouped_batchnorm_has_the_bits_of_the_groups_run_alone(G, nb, c, hw, relu, residual):
    """G minibatches in one launch: output, input gradient, residual gradient and the logged statistics of every group
    are BIT-IDENTICAL to a launch on that group alone; (dgamma, dbeta) are the groups' sums (fixed order)."""
    g = torch.Generator().manual_seed(G * 1000 + c)
    n = G * nb
    x = (torch.randn(n, c, hw, hw, generator=g) * 1.7 + 0.4).cuda()
    x[nb:] += 0.8                                            # the groups' statistics differ
    res = torch.randn(n, c, hw, hw, generator=g).cuda() if residual else None
    w, b = (torch.rand(c, generator=g) + 0.5).cuda(), torch.randn(c, generator=g).cuda()
    dy = torch.randn(n, c, hw, hw, generator=g).cuda()
    rm, rv = torch.zeros(c).cuda(), torch.ones(c).cuda()

    def run(xs, rs, dys, groups):
        leaves = [t.clone().requires_grad_() for t in (xs, w, b)] + ([rs.clone().requires_grad_()] if residual else [])
        slot = torch.zeros((groups, c, 2) if groups > 1 else (c, 2), dtype=torch.float64, device="cuda")
        with bn.logging_running_stats({id(rm): slot}), bn.grouped(groups):
            y = bn.bn_train(leaves[0], leaves[1], leaves[2], rm, rv, 0.1, 1e-5, leaves[3] if residual else None, relu)
            y.backward(dys)
        return y.detach(), [t.grad for t in leaves], slot.view(-1, c, 2)

    y, grads, log = run(x, res, dy, G)
    assert torch.equal(rm, torch.zeros_like(rm)) and torch.equal(rv, torch.ones_like(rv))      # logged, not advanced
    dw, db = torch.zeros_like(w, dtype=torch.float64), torch.zeros_like(b, dtype=torch.float64)
    for k in range(G):
        sl = slice(k * nb, (k + 1) * nb)
        y1, g1, log1 = run(x[sl], res[sl] if residual else None, dy[sl], 1)
        assert torch.equal(y[sl], y1), k
        assert torch.equal(grads[0][sl], g1[0]), k
        if residual:
            assert torch.equal(grads[3][sl], g1[3]), k
        assert torch.equal(log[k], log1[0]), k
        dw += g1[1].double()
        db += g1[2].double()
    torch.testing.assert_close(grads[1].double(), dw, rtol=1e-6, atol=1e-6 * max(1.0, dw.abs().max().item()))
    torch.testing.assert_close(grads[2].double(), db, rtol=1e-6, atol=1e-6 * max(1.0, db.abs().max().item()))
    with pytest.raises(bn.LogModeUnsupported):               # without the log the groups' updates would be unordered
        with bn.grouped(G):
            bn.bn_train(x, w, b, rm, rv, 0.1, 1e-5, None, relu)
    if n > G:
        with pytest.raises(ValueError):                      # G * nb - 1 rows are not G equal minibatches
            with bn.grouped(G), bn.logging_running_stats({id(rm): torch.zeros((G, c, 2), dtype=torch.float64, device="cuda")}):
                bn.bn_train(x[:n - 1], w, b, rm, rv, 0.1, 1e-5, None, relu)


@pytest.mark.gpu
@pytest.mark.parametrize("persistent", [False, True])
@pytest.mark.parametrize("G,nb", [(4, 128), (3, 32), (2, 5)])
def test_grouped_gradient_evaluation_of_the_resnet_equals_the_minibatches_one_by_one(G, nb, persistent):
    """the whole googleresnet under bn.grouped(G) -- fused residual blocks, down-sampling pairs with the dual BatchNorm,
    the stem, the fused head + loss, every BatchNorm backward's sums from an upstream epilogue looked up per group --
    against the G minibatches evaluated one by one: logits and logged statistics bit-identical, the summed gradient to
    rounding (the groups' slabs are added in another order); on the default and on the persistent convolutions (what the
    grouped exact pass runs, conv.persistent)."""
    import runner_cases as RC
    from bnn_priors_amd import conv, models, pool
    g = torch.Generator().manual_seed(77)
    x = torch.randn(G * nb, 3, 32, 32, generator=g).cuda()
    y = torch.randint(0, 10, (G * nb,), generator=g).cuda()
    model = RC.make_net(models, x[:2].cpu(), torch.tensor([0, 9]), device="cuda:0", cfg=dict(model="googleresnet"))
    model.train()
    bns = [m for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    params = [p for p in model.parameters()]
    N = 50000.0

    def run(xs, ys, groups):
        cmax = max(m.num_features for m in bns)
        cur = torch.zeros((groups, len(bns), cmax, 2), dtype=torch.float64, device="cuda")
        slots = {id(m.running_mean): (cur[:, i, :m.num_features] if groups > 1 else cur[0, i, :m.num_features])
                 for i, m in enumerate(bns)}
        for p in params:
            p.grad = None
        with bn.logging_running_stats(slots), bn.grouped(groups), conv.persistent(persistent), conv.deferring(model):
            with pool.head_loss(ys, "sum", N):
                f = model.net(xs)
            loss = pool.cross_entropy_backward(f, ys, reduction="sum", divide_by=N)
        torch.cuda.synchronize()
        return f.detach().clone(), loss.double().item(), [p.grad.clone() for p in params], cur.clone()

    f, loss, grads, log = run(x, y, G)
    acc = [torch.zeros_like(p, dtype=torch.float64) for p in params]
    loss1 = 0.0
    for k in range(G):
        sl = slice(k * nb, (k + 1) * nb)
        fk, lk, gk, logk = run(x[sl], y[sl], 1)
        assert torch.equal(f[sl], fk), k
        assert torch.equal(log[k], logk[0]), k
        loss1 += lk
        for a, t in zip(acc, gk):
            a += t.double()
    assert abs(loss - loss1) <= 1e-12 * abs(loss1)
    for (name, _), a, t in zip(model.named_parameters(), acc, grads):
        torch.testing.assert_close(t.double(), a, rtol=0, atol=2e-6 * max(1e-30, a.abs().max().item()), msg=name)
