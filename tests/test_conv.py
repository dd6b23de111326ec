"""fp32-MFMA convolution kernels of the ResNet trunk (csrc/conv_hip.inc) against a float64
reference of the same operator, through the C ABI and through autograd."""
import pytest
import torch
import torch.nn.functional as F

from bnn_priors_amd import _hip, conv

SHAPES = sorted(conv.SHAPES)
# the measured alternatives exist only in a library built and loaded with SGMCMC_ALTERNATIVES=1 (include/sgmcmc_hip_alternatives.h)
ALT = pytest.mark.skipif(not _hip.ALTERNATIVES, reason="measured alternative: needs the SGMCMC_ALTERNATIVES=1 build")


def _check_stats(stats, ref, parts_per_image):
    """stats[c][slice] = (sum, sum of squared deviations from the slice's own mean) over equal slices of
    (N, H, W); ref: float64 [N, C, H, W]"""
    n, c = ref.shape[0], ref.shape[1]
    assert stats.shape == (c, n * parts_per_image, 2) and stats.dtype == torch.float64
    parts = ref.reshape(n, c, parts_per_image, -1).permute(1, 0, 2, 3).reshape(c, n * parts_per_image, -1)
    scale = max(1.0, ref.abs().max().item())
    torch.testing.assert_close(stats[:, :, 0], parts.sum(-1), rtol=1e-5, atol=2e-5 * parts.shape[-1] ** .5 * scale)
    m2 = ((parts - parts.mean(-1, keepdim=True)) ** 2).sum(-1)
    torch.testing.assert_close(stats[:, :, 1], m2, rtol=1e-4, atol=1e-5 * parts.shape[-1] * scale ** 2)


def _data(c, hw, n, seed=0):
    g = torch.Generator().manual_seed(1000 * c + n + seed)
    x = torch.randn(n, c, hw, hw, generator=g)
    w = torch.randn(c, c, 3, 3, generator=g) * (2.0 / (9 * c)) ** .5
    dy = torch.randn(n, c, hw, hw, generator=g)
    return x, w, dy


def test_supported_is_false_off_the_table():
    x, w, _ = _data(16, 32, 2)
    assert not conv.supported(x, w, None, 1, 1, 1, 1)                 # CPU tensor
    assert not conv.supported(x.double(), w.double(), None, 1, 1, 1, 1)


@pytest.mark.gpu
@pytest.mark.parametrize("persistent", [False, True])
@pytest.mark.parametrize("c,hw", SHAPES)
@pytest.mark.parametrize("n", [128, 80, 5, 1])
def test_kernels_match_float64_reference(c, hw, n, persistent, monkeypatch):
    """forward, data gradient, weight gradient: error of the order of MIOpen's own (fp32 summation order) -- for the
    default kernels (csrc/conv_hip.inc) and for the persistent ones on prepared weight fragments (csrc/conv2_hip.inc)"""
    monkeypatch.setattr(conv, "PERSISTENT", persistent)
    x, w, dy = (t.cuda() for t in _data(c, hw, n))
    assert conv.supported(x, w, None, 1, 1, 1, 1) and conv.supported(x, w, None, (1, 1), (1, 1), (1, 1), 1)
    for bad in (dict(stride=2), dict(padding=0), dict(dilation=2), dict(groups=2)):
        kw = dict(stride=1, padding=1, dilation=1, groups=1, **{}) | bad
        assert not conv.supported(x, w, None, kw["stride"], kw["padding"], kw["dilation"], kw["groups"])
    assert not conv.supported(x, w, torch.zeros(c, device="cuda"), 1, 1, 1, 1)

    xd, wd, dyd = x.double(), w.double(), dy.double()
    ref_y = F.conv2d(xd, wd, padding=1)
    ref_dx = torch.nn.grad.conv2d_input(x.shape, wd, dyd, padding=1)
    ref_dw = torch.nn.grad.conv2d_weight(xd, w.shape, dyd, padding=1)

    xg, wg = x.clone().requires_grad_(), w.clone().requires_grad_()
    y = conv.conv3x3(xg, wg)
    y.backward(dy)
    # K = 9c products of O(1) magnitude per output; n*hw*hw of them per weight-gradient element
    tol_y = 64 * torch.finfo(torch.float32).eps * (9 * c) ** .5
    assert (y.double() - ref_y).abs().max() <= tol_y * max(1.0, ref_y.abs().max().item())
    assert (xg.grad.double() - ref_dx).abs().max() <= tol_y * max(1.0, ref_dx.abs().max().item())
    tol_w = 64 * torch.finfo(torch.float32).eps * (n * hw * hw) ** .5
    assert (wg.grad.double() - ref_dw).abs().max() <= tol_w * max(1.0, ref_dw.abs().max().item())

    # and the library operator it replaces, same tolerances
    xl, wl = x.clone().requires_grad_(), w.clone().requires_grad_()
    F.conv2d(xl, wl, padding=1).backward(dy)
    torch.testing.assert_close(y.detach(), F.conv2d(x, w, padding=1), rtol=1e-4, atol=tol_y * 8)
    torch.testing.assert_close(xg.grad, xl.grad, rtol=1e-4, atol=tol_y * 8)
    torch.testing.assert_close(wg.grad, wl.grad, rtol=1e-4, atol=tol_w * 8 * ref_dw.abs().max().item())


@pytest.mark.gpu
def test_gradients_are_reproducible_and_inputs_untouched():
    x, w, dy = (t.cuda() for t in _data(32, 16, 37))
    outs = []
    for _ in range(3):
        xg, wg = x.clone().requires_grad_(), w.clone().requires_grad_()
        y = conv.conv3x3(xg, wg)
        y.backward(dy)
        outs.append((y.detach(), xg.grad, wg.grad))
        assert torch.equal(xg.detach(), x) and torch.equal(wg.detach(), w)
    for o in outs[1:]:
        for a, b in zip(outs[0], o):
            assert torch.equal(a, b)
    with torch.no_grad():
        assert torch.equal(conv.conv3x3(x, w), outs[0][0])
    # the merged backward launch (data + weight gradient together) == the two separate entry points
    assert torch.equal(conv._run(dy, w, True)[0], outs[0][1])
    assert torch.equal(conv._weight_grad(x, dy, w), outs[0][2])
    # accumulation into an existing .grad reads the gradient at once: that route reduces immediately
    xg, wg = x.clone().requires_grad_(), w.clone().requires_grad_()
    conv.conv3x3(xg, wg).backward(dy)
    assert not conv._pending
    conv.conv3x3(xg, wg).backward(dy)
    assert not conv._pending
    assert torch.equal(wg.grad, outs[0][2] + outs[0][2]) and torch.equal(xg.grad, outs[0][1] + outs[0][1])
    # two convolutions in one backward pass: one deferred launch reduces both
    x2, w2, w3 = x.clone().requires_grad_(), w.clone().requires_grad_(), (w * 0.5).requires_grad_()
    conv.conv3x3(conv.conv3x3(x2, w2), w3).backward(dy)
    torch.cuda.synchronize()
    assert not conv._pending
    ref2, ref3 = w.clone().requires_grad_(), (w * 0.5).requires_grad_()
    F.conv2d(F.conv2d(x, ref2, padding=1), ref3, padding=1).backward(dy)
    torch.testing.assert_close(w2.grad, ref2.grad, rtol=1e-4, atol=1e-3 * ref2.grad.abs().max().item())
    torch.testing.assert_close(w3.grad, ref3.grad, rtol=1e-4, atol=1e-3 * ref3.grad.abs().max().item())
    xg, wg = x.clone().requires_grad_(), w.clone()          # only one gradient wanted: separate kernels
    conv.conv3x3(xg, wg).backward(dy)
    assert torch.equal(xg.grad, outs[0][1])


@pytest.mark.gpu
@pytest.mark.parametrize("c,hw", SHAPES)
@pytest.mark.parametrize("n", [128, 3])
def test_backward_with_the_shortcut_gradient_added_in_its_epilogue(c, hw, n):
    """sgmcmc_conv3x3_bwd_add: dx = data gradient + e_dout * [e_out > 0], same bits as the plain backward followed
    by the element-wise add; the weight gradient is untouched by the epilogue."""
    import ctypes
    lib = _hip.lib()
    x, w, dy = (t.cuda() for t in _data(c, hw, n, seed=5))
    g = torch.Generator().manual_seed(c + n)
    e_dout = torch.randn(n, c, hw, hw, generator=g).cuda()
    e_out = torch.relu(torch.randn(n, c, hw, hw, generator=g)).cuda()
    s = torch.cuda.current_stream().cuda_stream
    outs = []
    for add in (False, True):
        dx, dw = torch.empty_like(x), torch.empty_like(w)
        scratch = torch.empty(lib.sgmcmc_conv3x3_wrw_scratch_floats(n, c, hw), device="cuda")
        if add:
            err = lib.sgmcmc_conv3x3_bwd_add(x.data_ptr(), w.data_ptr(), dy.data_ptr(), dx.data_ptr(), e_dout.data_ptr(),
                                             e_out.data_ptr(), dw.data_ptr(), scratch.data_ptr(), n, c, hw, None, s)
        else:
            err = lib.sgmcmc_conv3x3_bwd(x.data_ptr(), w.data_ptr(), dy.data_ptr(), dx.data_ptr(), dw.data_ptr(),
                                         scratch.data_ptr(), n, c, hw, None, s)
        _hip.check(err, "bwd")
        outs.append((dx, dw))
    assert torch.equal(outs[1][0], outs[0][0] + e_dout * (e_out > 0))
    assert torch.equal(outs[1][1], outs[0][1])
    assert lib.sgmcmc_conv3x3_bwd_add(x.data_ptr(), w.data_ptr(), dy.data_ptr(), dx.data_ptr(), 0, e_out.data_ptr(),
                                      dw.data_ptr(), scratch.data_ptr(), n, c, hw, None, s) != 0


@pytest.mark.gpu
def test_abi_rejects_shapes_outside_the_table():
    lib = _hip.lib()
    x = torch.zeros(2, 48, 16, 16, device="cuda")
    w = torch.zeros(48, 48, 3, 3, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    assert lib.sgmcmc_conv3x3(x.data_ptr(), w.data_ptr(), x.data_ptr(), 2, 48, 16, 0, 0, st) != 0
    assert lib.sgmcmc_conv3x3(x.data_ptr(), w.data_ptr(), x.data_ptr(), 0, 16, 32, 0, 0, st) != 0
    assert lib.sgmcmc_conv3x3(0, w.data_ptr(), x.data_ptr(), 2, 16, 32, 0, 0, st) != 0
    assert lib.sgmcmc_conv3x3_wrw_scratch_floats(2, 48, 16) == -1
    assert lib.sgmcmc_conv3x3_wrw_scratch_floats(3, 16, 32) == 6 * 16 * 16 * 9
    assert lib.sgmcmc_conv3x3_wrw(x.data_ptr(), x.data_ptr(), w.data_ptr(), x.data_ptr(), 2, 48, 16, st) != 0


@pytest.mark.gpu
def test_resnet_layers_take_the_kernel_path(monkeypatch):
    """the googleresnet trunk routes its 16 stride-1 3x3 convolutions through the hand-written kernels: the seven
    identity-shortcut blocks as fused blocks (resblock.py, 2 convolutions each), the conv2 of the two down-sampling
    blocks through conv3x3, their strided pair through conv_down; switches = env / module flags"""
    from bnn_priors_amd import models, resblock
    blocks = []
    real_block = resblock.residual_block
    monkeypatch.setattr(resblock, "residual_block", lambda x, *a: (blocks.append(tuple(x.shape[1:3])), real_block(x, *a))[1])
    calls = []
    real = conv.conv3x3
    monkeypatch.setattr(conv, "conv3x3",
                        lambda x, w, want_stats=False: (calls.append(tuple(x.shape[1:3])), real(x, w, want_stats))[1])
    torch.manual_seed(0)
    x = torch.randn(4, 3, 32, 32)
    y = torch.randint(0, 10, (4,))
    net = models.get_model(x, y, "googleresnet", width=50, depth=3, weight_prior="gaussian", weight_loc=0.,
                           weight_scale=2 ** .5, bias_prior="gaussian", bias_loc=0., bias_scale=1.,
                           batchnorm=True, weight_prior_params={}, bias_prior_params={}).cuda()
    downs = []
    real_down = conv.conv_down
    monkeypatch.setattr(conv, "conv_down", lambda x, wm, ws, want_stats=False:
                        (downs.append(tuple(x.shape[1:3])), real_down(x, wm, ws, want_stats))[1])
    out = net.net(x.cuda())
    assert sorted(blocks) == [(16, 32)] * 3 + [(32, 16)] * 2 + [(64, 8)] * 2
    assert sorted(calls) == [(32, 16), (64, 8)]
    assert sorted(downs) == [(16, 32), (32, 16)]
    monkeypatch.setattr(resblock, "ENABLED", False)       # layer by layer: every stride-1 3x3 goes through conv3x3
    calls.clear(); blocks.clear(); downs.clear()
    layered = net.net(x.cuda())
    assert not blocks and sorted(calls) == [(16, 32)] * 6 + [(32, 16)] * 5 + [(64, 8)] * 5
    torch.testing.assert_close(layered, out, rtol=1e-3, atol=1e-3)
    monkeypatch.setattr(resblock, "ENABLED", True)
    monkeypatch.setattr(conv, "ENABLED", False)
    calls.clear()
    downs.clear()
    blocks.clear()
    net.eval()
    ref = net.net(x.cuda())
    assert not calls and not downs and not blocks
    monkeypatch.setattr(conv, "ENABLED", True)
    torch.testing.assert_close(net.net(x.cuda()), ref, rtol=5e-4, atol=1e-3)      # own kernels vs MIOpen, 20 layers deep
    assert out.shape == (4, 10)


@pytest.mark.gpu
@pytest.mark.parametrize("persistent", [False, True])
@pytest.mark.parametrize("c,hw", SHAPES)
def test_epilogue_statistics_feed_the_batchnorm(c, hw, persistent, monkeypatch):
    "conv3x3(want_stats=True): per-band sums of y; bn_train(stats=...) == bn_train() on the same y"
    from bnn_priors_amd import bn
    monkeypatch.setattr(conv, "PERSISTENT", persistent)
    x, w, _ = (t.cuda() for t in _data(c, hw, 19))
    y, stats = conv.conv3x3(x, w, want_stats=True)
    assert torch.equal(y, conv.conv3x3(x, w))
    _check_stats(stats, y.double(), hw // (4 if persistent else 8))     # one part per band of 8 (4) rows
    g = torch.Generator().manual_seed(5)
    gamma, beta = (torch.rand(c, generator=g) + 0.5).cuda(), torch.randn(c, generator=g).cuda()
    outs = []
    for st in (None, stats):
        rm, rv = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
        outs.append((bn.bn_train(y, gamma, beta, rm, rv, 0.1, 1e-5, None, True, st), rm, rv))
    for a, b in zip(*outs):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-5)
    with pytest.raises(ValueError, match="stats"):
        bn.bn_train(y, gamma, beta, None, None, 0.1, 1e-5, None, True, stats.float())


@pytest.mark.gpu
@pytest.mark.parametrize("cin,hwi", sorted(conv.DOWN_SHAPES))
@pytest.mark.parametrize("n", [128, 80, 7, 1])
def test_down_block_pair_matches_float64_reference(cin, hwi, n):
    "3x3/stride 2 + 1x1/stride 2 on the same input as one operator: outputs, statistics, all three gradients"
    g = torch.Generator().manual_seed(100 * cin + n)
    x = torch.randn(n, cin, hwi, hwi, generator=g).cuda()
    wm = (torch.randn(2 * cin, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** .5).cuda()
    ws = (torch.randn(2 * cin, cin, 1, 1, generator=g) * (2.0 / cin) ** .5).cuda()
    dym = torch.randn(n, 2 * cin, hwi // 2, hwi // 2, generator=g).cuda()
    dys = torch.randn(n, 2 * cin, hwi // 2, hwi // 2, generator=g).cuda()
    assert conv.down_supported(x, wm, ws) and not conv.down_supported(x.cpu(), wm.cpu(), ws.cpu())
    assert not conv.down_supported(x, wm, ws.reshape(2 * cin, cin))

    xd, wmd, wsd = (t.double().requires_grad_() for t in (x, wm, ws))
    rm, rs = F.conv2d(xd, wmd, stride=2, padding=1), F.conv2d(xd, wsd, stride=2)
    ((rm * dym.double()).sum() + (rs * dys.double()).sum()).backward()

    xg, wmg, wsg = (t.clone().requires_grad_() for t in (x, wm, ws))
    ym, ys, sm, ss = conv.conv_down(xg, wmg, wsg, True)
    ((ym * dym).sum() + (ys * dys).sum()).backward()
    eps = torch.finfo(torch.float32).eps
    tol = 64 * eps * (9 * cin) ** .5
    for got, ref in ((ym, rm), (ys, rs), (xg.grad, xd.grad)):
        assert (got.double() - ref.detach()).abs().max() <= tol * max(1.0, ref.abs().max().item())
    tol_w = 64 * eps * (n * hwi * hwi / 4) ** .5
    for got, ref in ((wmg.grad, wmd.grad), (wsg.grad, wsd.grad)):
        assert (got.double() - ref).abs().max() <= tol_w * max(1.0, ref.abs().max().item())
    for st, ref in ((sm, rm), (ss, rs)):
        _check_stats(st, ref.detach(), hwi // 16)
    # without statistics, and reproducibly
    x2, wm2, ws2 = (t.clone().requires_grad_() for t in (x, wm, ws))
    ym2, ys2 = conv.conv_down(x2, wm2, ws2)
    ((ym2 * dym).sum() + (ys2 * dys).sum()).backward()
    assert torch.equal(ym2, ym) and torch.equal(ys2, ys)
    for a, b in ((x2, xg), (wm2, wmg), (ws2, wsg)):
        assert torch.equal(a.grad, b.grad)
    # accumulating into existing gradients takes the immediate-reduction route: same bits
    ym3, ys3 = conv.conv_down(x2, wm2, ws2)
    ((ym3 * dym).sum() + (ys3 * dys).sum()).backward()
    assert torch.equal(wm2.grad, wmg.grad + wmg.grad) and torch.equal(ws2.grad, wsg.grad + wsg.grad)
    assert not conv._pending


@pytest.mark.gpu
@pytest.mark.parametrize("n", [128, 3, 1])
def test_stem_convolution_matches_float64_reference(n):
    g = torch.Generator().manual_seed(n)
    x = torch.randn(n, 3, 32, 32, generator=g).cuda()
    w = (torch.randn(16, 3, 3, 3, generator=g) * (2.0 / 27) ** .5).cuda()
    dy = torch.randn(n, 16, 32, 32, generator=g).cuda()
    assert conv.stem_supported(x, w, None, 1, 1, 1, 1) and not conv.stem_supported(x, w, None, 2, 1, 1, 1)
    assert not conv.stem_supported(x.clone().requires_grad_(), w, None, 1, 1, 1, 1)
    wd = w.double().requires_grad_()
    ref = F.conv2d(x.double(), wd, padding=1)
    (ref * dy.double()).sum().backward()
    wg = w.clone().requires_grad_()
    y, st = conv.conv_stem(x, wg, True)
    (y * dy).sum().backward()
    eps = torch.finfo(torch.float32).eps
    assert (y.double() - ref.detach()).abs().max() <= 64 * eps * 27 ** .5 * max(1.0, ref.abs().max().item())
    assert (wg.grad.double() - wd.grad).abs().max() <= 64 * eps * (n * 1024) ** .5 * max(1.0, wd.grad.abs().max().item())
    _check_stats(st, ref.detach(), 4)
    w2 = w.clone().requires_grad_()
    y2 = conv.conv_stem(x, w2)
    (y2 * dy).sum().backward()
    assert torch.equal(y2, y) and torch.equal(w2.grad, wg.grad)
    (conv.conv_stem(x, w2) * dy).sum().backward()          # accumulation: immediate reduction, same bits
    assert torch.equal(w2.grad, wg.grad + wg.grad) and not conv._pending


@pytest.mark.gpu
@pytest.mark.parametrize("n", [128, 3, 1])
def test_first_layer_convolution_matches_float64_reference(n):
    "1 -> 50 channels on 28x28: 9 taps as three padded k-steps, 50 channels as 3 + 1/8 tiles, ragged pixel tiles"
    g = torch.Generator().manual_seed(28 + n)
    x = torch.rand(n, 1, 28, 28, generator=g).cuda()
    w = (torch.randn(50, 1, 3, 3, generator=g) * (2.0 / 9) ** .5).cuda()
    dy = torch.randn(n, 50, 28, 28, generator=g).cuda()
    assert conv.first_supported(x, w, None, 1, 1, 1, 1) and not conv.first_supported(x, w, None, 1, 0, 1, 1)
    assert not conv.first_supported(x.clone().requires_grad_(), w, None, 1, 1, 1, 1)
    wd = w.double().requires_grad_()
    ref = F.conv2d(x.double(), wd, padding=1)
    ref.backward(dy.double())
    wg = w.clone().requires_grad_()
    y = conv.conv_first(x, wg)
    y.backward(dy)
    eps = torch.finfo(torch.float32).eps
    assert (y.double() - ref.detach()).abs().max() <= 32 * eps * max(1.0, ref.abs().max().item())
    assert (wg.grad.double() - wd.grad).abs().max() <= 64 * eps * (n * 784) ** .5 * max(1.0, wd.grad.abs().max().item())
    w2 = w.clone().requires_grad_()
    conv.conv_first(x, w2).backward(dy)
    assert torch.equal(w2.grad, wg.grad)
    conv.conv_first(x, w2).backward(dy)
    assert torch.equal(w2.grad, wg.grad + wg.grad) and not conv._pending


@pytest.mark.gpu
@pytest.mark.parametrize("n", [128, 5, 1])
@pytest.mark.parametrize("with_bias", [True, False])
def test_first_layer_with_its_tail_equals_the_two_operators(n, with_bias):
    """conv -> + bias -> ReLU -> MaxPool2d(2) in one launch each way (convfirst::fwd_pool_kernel / wrw_pool_kernel):
    the pooled map and the weight gradient carry the bits of conv_first -> bias_relu_pool (same contraction, same
    order), the bias gradient agrees to rounding (its partial sums are per band instead of per image slice); ties and
    exact zeros in the window go to the first position in scan order / do not pass the ReLU, as in ATen."""
    from bnn_priors_amd import conv, pool
    g = torch.Generator().manual_seed(11 + n)
    x = torch.randn(n, 1, 28, 28, generator=g)
    x[:, :, 4:8, 4:12] = 0.0                         # flat patches: tied windows
    x = x.cuda()
    w = (torch.randn(50, 1, 3, 3, generator=g) / 3).cuda()
    b = (torch.randn(50, generator=g) / 4).cuda() if with_bias else None
    if with_bias:
        b[:5] = 0.0                                  # max + bias == 0 on the flat patches: the ReLU's edge
    dy = torch.randn(n, 50, 14, 14, generator=g).cuda()
    w0 = w.clone().requires_grad_()
    b0 = b.clone().requires_grad_() if with_bias else None
    ref = pool.bias_relu_pool(conv.conv_first(x, w0), b0)
    ref.backward(dy)
    w1 = w.clone().requires_grad_()
    b1 = b.clone().requires_grad_() if with_bias else None
    assert conv.first_pool_supported(x, w1, b1, 1, 1, 1, 1)
    out = conv.conv_first_pool(x, w1, b1)
    out.backward(dy)
    assert torch.equal(out, ref)
    assert torch.equal(w1.grad, w0.grad) and not conv._pending
    if with_bias:
        torch.testing.assert_close(b1.grad, b0.grad, rtol=1e-5, atol=1e-4 * max(1.0, n ** .5))
    # ... and against ATen in float64
    wd = w.double().requires_grad_()
    bd = b.double().requires_grad_() if with_bias else None
    refd = F.max_pool2d(F.relu(F.conv2d(x.double(), wd, bd, padding=1)), 2)
    refd.backward(dy.double())
    eps = torch.finfo(torch.float32).eps
    assert (out.double() - refd.detach()).abs().max() <= 64 * eps * max(1.0, refd.abs().max().item())
    # (a window whose two largest entries differ by less than fp32 rounding may route its gradient elsewhere)
    close = (w1.grad.double() - wd.grad).abs().max() / max(1.0, wd.grad.abs().max().item())
    assert close <= 1e-3, close
    with torch.no_grad():                            # evaluation: forward alone
        assert torch.equal(conv.conv_first_pool(x, w, b), ref)


@pytest.mark.gpu
def test_first_layer_with_its_tail_defers_both_gradients_inside_the_samplers_pass():
    from bnn_priors_amd import conv
    g = torch.Generator().manual_seed(5)
    x = torch.randn(16, 1, 28, 28, generator=g).cuda()
    w = (torch.randn(50, 1, 3, 3, generator=g) / 3).cuda().requires_grad_()
    b = (torch.randn(50, generator=g) / 4).cuda().requires_grad_()
    dy = torch.randn(16, 50, 14, 14, generator=g).cuda()
    conv.conv_first_pool(x, w, b).backward(dy)
    gw, gb = w.grad.clone(), b.grad.clone()
    w.grad = b.grad = None
    with conv.deferring():
        conv.conv_first_pool(x, w, b).backward(dy)
    assert not conv._pending
    assert torch.equal(w.grad, gw) and torch.equal(b.grad, gb)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [128, 5, 1])
@pytest.mark.parametrize("with_bias", [True, False])
def test_second_layer_with_its_tail_equals_the_two_operators(n, with_bias):
    """conv50 -> + bias -> ReLU -> MaxPool2d(2) in one launch each way (conv50::conv_pool_kernel / bwd_pool_kernel):
    pooled map, data gradient and weight gradient carry the bits of conv50 -> bias_relu_pool (the 14 x 14 map and its
    gradient are rebuilt in LDS with the same values), the bias gradient agrees to rounding."""
    from bnn_priors_amd import conv, pool
    g = torch.Generator().manual_seed(50 + n)
    x = torch.randn(n, 50, 14, 14, generator=g)
    x[:, :, 2:6, 2:8] = 0.0                          # flat patches: tied windows, max + bias == 0 where bias is 0
    x = x.cuda()
    w = (torch.randn(50, 50, 3, 3, generator=g) / 21).cuda()
    b = (torch.randn(50, generator=g) / 4).cuda() if with_bias else None
    if with_bias:
        b[:5] = 0.0
    dy = torch.randn(n, 50, 7, 7, generator=g).cuda()
    x0, w0 = x.clone().requires_grad_(), w.clone().requires_grad_()
    b0 = b.clone().requires_grad_() if with_bias else None
    ref = pool.bias_relu_pool(conv.conv50(x0, w0), b0)
    ref.backward(dy)
    x1, w1 = x.clone().requires_grad_(), w.clone().requires_grad_()
    b1 = b.clone().requires_grad_() if with_bias else None
    assert conv.conv50_pool_supported(x1, w1, b1, 1, 1, 1, 1)
    out = conv.conv50_pool(x1, w1, b1)
    out.backward(dy)
    assert torch.equal(out, ref)
    assert torch.equal(x1.grad, x0.grad)
    assert torch.equal(w1.grad, w0.grad) and not conv._pending
    if with_bias:
        torch.testing.assert_close(b1.grad, b0.grad, rtol=1e-5, atol=1e-4 * max(1.0, n ** .5))
    wd, xd = w.double().requires_grad_(), x.double().requires_grad_()
    bd = b.double().requires_grad_() if with_bias else None
    refd = F.max_pool2d(F.relu(F.conv2d(xd, wd, bd, padding=1)), 2)
    eps = torch.finfo(torch.float32).eps
    assert (out.double() - refd.detach()).abs().max() <= 256 * eps * max(1.0, refd.abs().max().item())
    with torch.no_grad():
        assert torch.equal(conv.conv50_pool(x, w, b), ref)
    # only the input takes a gradient / only the parameters do
    x2 = x.clone().requires_grad_()
    conv.conv50_pool(x2, w, b).backward(dy)
    assert torch.equal(x2.grad, x0.grad)
    w3 = w.clone().requires_grad_()
    conv.conv50_pool(x, w3, b).backward(dy)
    assert torch.equal(w3.grad, w0.grad)


@pytest.mark.gpu
def test_second_layer_with_its_tail_defers_both_gradients_inside_the_samplers_pass():
    from bnn_priors_amd import conv
    g = torch.Generator().manual_seed(6)
    x = torch.randn(9, 50, 14, 14, generator=g).cuda().requires_grad_()
    w = (torch.randn(50, 50, 3, 3, generator=g) / 21).cuda().requires_grad_()
    b = (torch.randn(50, generator=g) / 4).cuda().requires_grad_()
    dy = torch.randn(9, 50, 7, 7, generator=g).cuda()
    conv.conv50_pool(x, w, b).backward(dy)
    gx, gw, gb = x.grad.clone(), w.grad.clone(), b.grad.clone()
    x.grad = w.grad = b.grad = None
    with conv.deferring():
        conv.conv50_pool(x, w, b).backward(dy)
    assert not conv._pending
    assert torch.equal(x.grad, gx) and torch.equal(w.grad, gw) and torch.equal(b.grad, gb)


# ------------------------------------------------------------------ deferred weight-gradient reduction
def _conv_mod():
    from bnn_priors_amd import conv
    return conv


@pytest.mark.gpu
def test_weight_gradient_is_reduced_immediately_outside_the_samplers_own_pass():
    "user code calling the operator directly never sees an unreduced gradient (no deferring() scope)"
    conv = _conv_mod()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(4, 16, 32, 32, generator=g).cuda().requires_grad_()
    w = torch.randn(16, 16, 3, 3, generator=g).cuda().requires_grad_()
    seen = {}
    w.register_hook(lambda gr: seen.setdefault("dw", gr.clone()))     # reads the gradient DURING backward
    conv.conv3x3(x, w).square().sum().backward()
    assert not conv._pending
    xr, wr = x.detach().clone().requires_grad_(), w.detach().clone().requires_grad_()
    torch.nn.functional.conv2d(xr, wr, None, 1, 1).square().sum().backward()
    torch.testing.assert_close(seen["dw"], wr.grad, rtol=2e-4, atol=2e-3)
    torch.testing.assert_close(w.grad, wr.grad, rtol=2e-4, atol=2e-3)


@pytest.mark.gpu
def test_deferring_scope_handles_shared_weights_and_existing_grads():
    """inside deferring(): a weight used by two operations (gradients summed mid-backward) and a weight that
    already has a .grad both take the immediate route; a weight used once is deferred -- all three correct"""
    conv = _conv_mod()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(3, 16, 32, 32, generator=g).cuda()
    ws = [torch.randn(16, 16, 3, 3, generator=g).cuda().mul_(0.1).requires_grad_() for _ in range(3)]
    ws[2].grad = torch.ones_like(ws[2])
    with conv.deferring():
        h = conv.conv3x3(conv.conv3x3(x, ws[0]), ws[0])          # shared
        h = conv.conv3x3(conv.conv3x3(h, ws[1]), ws[2])          # once / has grad
        h.square().mean().backward()
    assert not conv._pending and not conv._defer["active"]
    rs = [w.detach().clone().requires_grad_() for w in ws]
    rs[2].grad = torch.ones_like(rs[2])
    F = torch.nn.functional
    h = F.conv2d(F.conv2d(x, rs[0], None, 1, 1), rs[0], None, 1, 1)
    h = F.conv2d(F.conv2d(h, rs[1], None, 1, 1), rs[2], None, 1, 1)
    h.square().mean().backward()
    for w, r in zip(ws, rs):
        torch.testing.assert_close(w.grad, r.grad, rtol=1e-3, atol=1e-5)


@pytest.mark.gpu
def test_deferring_scope_recovers_after_an_exception():
    conv = _conv_mod()
    x = torch.randn(2, 16, 32, 32).cuda()
    w = torch.randn(16, 16, 3, 3).cuda().requires_grad_()
    with pytest.raises(RuntimeError, match="boom"):
        with conv.deferring():
            y = conv.conv3x3(x, w)
            raise RuntimeError("boom")
    assert not conv._defer["active"] and not conv._pending
    with conv.deferring():
        conv.conv3x3(x, w).sum().backward()
    ref = w.detach().clone().requires_grad_()
    torch.nn.functional.conv2d(x, ref, None, 1, 1).sum().backward()
    torch.testing.assert_close(w.grad, ref.grad, rtol=2e-4, atol=2e-3)


# ------------------------------------------------------------------ the convolutional classifier's 50 -> 50 layer
@pytest.mark.gpu
@pytest.mark.parametrize("n", [128, 5, 1])
def test_conv50_matches_float64(n):
    "conv2d(x, w, padding=1) for [N, 50, 14, 14] x [50, 50, 3, 3] (models/conv_nets.py:46-70): value and both gradients"
    g = torch.Generator().manual_seed(50 + n)
    x = torch.randn(n, 50, 14, 14, generator=g)
    w = torch.randn(50, 50, 3, 3, generator=g) * (2.0 / 450) ** .5
    dy = torch.randn(n, 50, 14, 14, generator=g)
    assert conv.conv50_supported(x.cuda(), w.cuda(), None, 1, 1, 1, 1)
    assert not conv.conv50_supported(x, w, None, 1, 1, 1, 1)                                    # CPU
    assert not conv.conv50_supported(x.cuda(), w.cuda(), torch.zeros(50).cuda(), 1, 1, 1, 1)    # bias joins the tail
    assert not conv.conv50_supported(x.cuda()[:, :, :12], w.cuda(), None, 1, 1, 1, 1)
    xr, wr = x.double().requires_grad_(), w.double().requires_grad_()
    ref = F.conv2d(xr, wr, None, 1, 1)
    ref.backward(dy.double())
    xg, wg = x.cuda().requires_grad_(), w.cuda().requires_grad_()
    y = conv.conv50(xg, wg)
    y.backward(dy.cuda())
    torch.testing.assert_close(y.double().cpu(), ref.detach(), rtol=1e-5, atol=2e-5)
    torch.testing.assert_close(xg.grad.double().cpu(), xr.grad, rtol=1e-5, atol=2e-5)
    torch.testing.assert_close(wg.grad.double().cpu(), wr.grad, rtol=1e-4, atol=2e-5 * (n * 196) ** .5)
    # reproducible bit for bit (fixed-order slab reduction); accumulation into an existing .grad
    g1 = wg.grad.clone()
    conv.conv50(xg, wg).backward(dy.cuda())
    assert torch.equal(wg.grad, g1 + g1)
    # data gradient only (frozen weight)
    xf = x.cuda().requires_grad_()
    conv.conv50(xf, w.cuda()).backward(dy.cuda())
    torch.testing.assert_close(xf.grad.double().cpu(), xr.grad, rtol=1e-5, atol=2e-5)


@pytest.mark.gpu
def test_convnet_has_no_library_convolution_left(monkeypatch):
    "classificationconvnet's two convolutions both run on this repo's kernels; F.conv2d is not called"
    from bnn_priors_amd import models
    torch.manual_seed(0)
    x = torch.rand(8, 784)
    y = torch.arange(8) % 10
    y[-1] = 9
    net = models.get_model(x, y, "classificationconvnet", width=50, depth=3, weight_prior="laplace", weight_loc=0.,
                           weight_scale=2 ** .5, bias_prior="gaussian", bias_loc=0., bias_scale=1.,
                           batchnorm=True, weight_prior_params={}, bias_prior_params={}).cuda()
    called = []
    real = torch.nn.functional.conv2d
    monkeypatch.setattr(torch.nn.functional, "conv2d", lambda *a, **k: (called.append(1), real(*a, **k))[1])
    out = net.net(x.cuda())
    out.sum().backward()
    assert not called and out.shape == (8, 10)
    monkeypatch.setattr(conv, "ENABLED", False)
    ref = net.net(x.cuda())
    assert called
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("name,xshape", [("googleresnet", (3, 32, 32)), ("classificationconvnet", (784,))])
def test_baseline_conv_models_take_no_library_path_in_a_gradient_evaluation(name, xshape):
    """configs[2] and [3]: a forward + backward of the model dispatches nothing to MIOpen / rocBLAS / ATen compute
    (conv.LIBRARY_CALLS stays empty; SGMCMC_STRICT=1 would raise instead of counting); an off-table shape is counted."""
    from bnn_priors_amd import models
    torch.manual_seed(0)
    x = torch.randn((8,) + xshape).cuda()
    y = (torch.arange(8) % 10).cuda()
    net = models.get_model(x.cpu()[:2], torch.tensor([0, 9]), name, width=50, depth=3, weight_prior="gaussian",
                           weight_scale=2 ** .5, bias_prior="gaussian", bias_scale=1.).cuda()     # (10 classes)
    net.train()
    conv.LIBRARY_CALLS.clear()
    F.cross_entropy(net.net(x), y).backward()
    assert not conv.LIBRARY_CALLS, dict(conv.LIBRARY_CALLS)
    torch.cuda.synchronize()
    # what an off-table layer would do: counted (and refused under SGMCMC_STRICT=1) -- without running the library here
    odd = torch.empty(4, 3, 24, 24, device="cuda")
    with pytest.warns(RuntimeWarning, match="library path"):      # loud, once per (operator, shape)
        conv.library_path("conv2d", odd)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        conv.library_path("conv2d", odd)                             # ... the second time only counted
    assert conv.LIBRARY_CALLS == {("conv2d", (3, 24, 24)): 2}
    conv.LIBRARY_CALLS.clear()
    old = conv.STRICT
    conv.STRICT = True
    try:
        with pytest.raises(RuntimeError, match="SGMCMC_STRICT"):
            conv.library_path("conv2d", odd)
        with torch.no_grad():
            conv.library_path("conv2d", odd)            # evaluation passes are not policed
    finally:
        conv.STRICT = old
        conv.LIBRARY_CALLS.clear()


@pytest.mark.gpu
def test_a_convnet_width_off_the_tables_is_loud_not_silent():
    "width != 50 (experiments/train_bnn.py:53-54 makes it an option): MIOpen runs the convolutions -- announced, counted"
    from bnn_priors_amd import models
    torch.manual_seed(0)
    x, y = torch.rand(8, 784).cuda(), (torch.arange(8) % 10).cuda()
    net = models.get_model(x.cpu()[:2], torch.tensor([0, 9]), "classificationconvnet", width=64, depth=3,
                           weight_prior="laplace", weight_scale=2 ** .5, bias_prior="gaussian", bias_scale=1.).cuda()
    conv.LIBRARY_CALLS.clear()
    with pytest.warns(RuntimeWarning, match="library path"):
        F.cross_entropy(net.net(x), y).backward()
    assert {k[0] for k in conv.LIBRARY_CALLS} == {"conv2d"} and len(conv.LIBRARY_CALLS) == 2, dict(conv.LIBRARY_CALLS)
    conv.LIBRARY_CALLS.clear()


@ALT
@pytest.mark.gpu
@pytest.mark.parametrize("c,hw", [(16, 32), (32, 16)])
@pytest.mark.parametrize("n", [128, 96, 5, 1])
@pytest.mark.parametrize("epi", ["none", "add_masked+sums", "add+sums+mask_dx"])
def test_uniform_backward_matches_the_merged_launch(c, hw, n, epi):
    """round 6's uniform backward convolution (csrc/conv_uni_hip.inc, a measured alternative): dx and the BatchNorm
    -backward sums carry the BITS of sgmcmc_conv3x3_bwd_ex, the weight gradient agrees with float64 like the default's"""
    import ctypes
    lib = _hip.lib()
    x, w, dy = (t.cuda() for t in _data(c, hw, n))
    g = torch.Generator().manual_seed(7 + n)
    e_dout, out = torch.randn(x.shape, generator=g).cuda(), torch.randn(x.shape, generator=g).relu().cuda()
    y_bn, mean, invstd = torch.randn(x.shape, generator=g).cuda(), torch.randn(c, generator=g).cuda() * .1, \
        torch.rand(c, generator=g).cuda() + .5
    s = torch.cuda.current_stream().cuda_stream
    slices = n * (hw // 8)

    def run(uniform):
        dx = torch.full_like(x, float("nan"))
        partial = torch.full((c, slices, 2), float("nan"), dtype=torch.float64, device="cuda")
        E = _hip.ConvBwdEpilogue()
        if epi != "none":
            E.e_dout = e_dout.data_ptr()
            E.e_out = out.data_ptr() if epi == "add_masked+sums" else None
            E.s_y, E.s_out, E.s_mean, E.s_invstd = y_bn.data_ptr(), out.data_ptr(), mean.data_ptr(), invstd.data_ptr()
            E.s_partial = partial.data_ptr()
            E.mask_dx = int(epi.endswith("mask_dx"))
        dw = torch.empty_like(w)
        if uniform:
            P = lib.sgmcmc_conv3x3_bwd_uniform_slabs(n, c, hw)
            part = torch.empty(P * w.numel(), device="cuda")
            _hip.check(lib.sgmcmc_conv3x3_bwd_uniform(x.data_ptr(), w.data_ptr(), dy.data_ptr(), dx.data_ptr(), ctypes.byref(E),
                                                      part.data_ptr(), n, c, hw, s), "sgmcmc_conv3x3_bwd_uniform")
            job = (_hip.ReduceJob * 1)()
            job[0].part, job[0].out, job[0].n_slabs, job[0].numel, job[0].taps = part.data_ptr(), dw.data_ptr(), P, w.numel(), 9
            _hip.check(lib.sgmcmc_wrw_reduce_many(ctypes.cast(job, ctypes.c_void_p), 1, s), "sgmcmc_wrw_reduce_many")
        else:
            part = torch.empty(lib.sgmcmc_conv3x3_wrw_scratch_floats(n, c, hw), device="cuda")
            _hip.check(lib.sgmcmc_conv3x3_bwd_ex(x.data_ptr(), w.data_ptr(), dy.data_ptr(), dx.data_ptr(), ctypes.byref(E),
                                                 dw.data_ptr(), part.data_ptr(), n, c, hw, None, s), "sgmcmc_conv3x3_bwd_ex")
        torch.cuda.synchronize()
        return dx, dw, partial

    dx0, dw0, p0 = run(False)
    dx1, dw1, p1 = run(True)
    assert torch.equal(dx0, dx1)
    if epi != "none":
        assert torch.equal(p0, p1)
    ref = torch.nn.grad.conv2d_weight(x.double(), w.shape, dy.double(), padding=1)
    scale = ref.abs().max().item()
    assert (dw1.double() - ref).abs().max().item() <= 2e-6 * scale * max(1.0, (n * hw * hw) ** .5 / 32)
    assert (dw1 - dw0).abs().max().item() <= 4e-6 * scale * max(1.0, (n * hw * hw) ** .5 / 32)
