"""Check and time the fused down-sampling convolution kernels against ATen/MIOpen (needs a GPU)."""
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from bnn_priors_amd import _hip

lib = _hip.lib()
dev = "cuda:0"
torch.backends.cudnn.benchmark = True
st = lambda: torch.cuda.current_stream().cuda_stream


def timeit(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def fwd(x, wm, ws, stats=False):
    n, cin, hwi = x.shape[0], x.shape[1], x.shape[2]
    ym = torch.empty(n, 2 * cin, hwi // 2, hwi // 2, device=x.device)
    ys = torch.empty_like(ym)
    sm = ss = None
    if stats:
        sl = lib.sgmcmc_conv_down_stat_slices(n, cin, hwi)
        sm = torch.empty(2 * cin, sl, 2, dtype=torch.float64, device=x.device)
        ss = torch.empty_like(sm)
    err = lib.sgmcmc_conv_down_fwd(x.data_ptr(), wm.data_ptr(), ws.data_ptr(), ym.data_ptr(), ys.data_ptr(),
                                   0 if sm is None else sm.data_ptr(), 0 if ss is None else ss.data_ptr(), n, cin, hwi, st())
    assert err == 0, err
    return ym, ys, sm, ss


for cin, hwi in ((16, 32), (32, 16)):
    g = torch.Generator(device=dev).manual_seed(cin)
    x = torch.randn(128, cin, hwi, hwi, device=dev, generator=g)
    wm = torch.randn(2 * cin, cin, 3, 3, device=dev, generator=g) * 0.1
    ws = torch.randn(2 * cin, cin, 1, 1, device=dev, generator=g) * 0.2
    rm = F.conv2d(x.double(), wm.double(), stride=2, padding=1)
    rs = F.conv2d(x.double(), ws.double(), stride=2)
    ym, ys, sm, ss = fwd(x, wm, ws, True)
    print(f"cin={cin} hwi={hwi} fwd: main err {(ym.double() - rm).abs().max().item():.3e} short err {(ys.double() - rs).abs().max().item():.3e}"
          f"  stats err {(sm[:, :, 0].sum(1) - rm.sum(dim=(0, 2, 3))).abs().max().item():.2e} {(ss[:, :, 1].sum(1) - (rs * rs).sum(dim=(0, 2, 3))).abs().max().item():.2e}")
    t = timeit(lambda: fwd(x, wm, ws, True))
    tl = timeit(lambda: (F.conv2d(x, wm, stride=2, padding=1), F.conv2d(x, ws, stride=2)))
    print(f"    fwd {t:.1f} us vs MIOpen (two calls) {tl:.1f} us")
    dym = torch.randn_like(ym)
    dys = torch.randn_like(ys)
    xd = x.double().requires_grad_()
    wmd, wsd = wm.double().requires_grad_(), ws.double().requires_grad_()
    (F.conv2d(xd, wmd, stride=2, padding=1) * dym.double()).sum().backward(retain_graph=False)
    gx_m = xd.grad.clone(); xd.grad = None
    (F.conv2d(xd, wsd, stride=2) * dys.double()).sum().backward()
    ref_dx = gx_m + xd.grad
    n = x.shape[0]
    scratch = torch.empty(lib.sgmcmc_conv_down_scratch_floats(n, cin, hwi), device=dev)
    dx, dwm, dws = torch.empty_like(x), torch.empty_like(wm), torch.empty_like(ws)

    def bwd():
        err = lib.sgmcmc_conv_down_bwd(x.data_ptr(), wm.data_ptr(), ws.data_ptr(), dym.data_ptr(), dys.data_ptr(),
                                       dx.data_ptr(), dwm.data_ptr(), dws.data_ptr(), scratch.data_ptr(), n, cin, hwi,
                                       None, st())
        assert err == 0, err
    bwd()
    print(f"    bwd: dx err {(dx.double() - ref_dx).abs().max().item():.3e} (|dx| max {ref_dx.abs().max().item():.2f})  "
          f"dw_main err {(dwm.double() - wmd.grad).abs().max().item():.3e} (max {wmd.grad.abs().max().item():.1f})  "
          f"dw_short err {(dws.double() - wsd.grad).abs().max().item():.3e} (max {wsd.grad.abs().max().item():.1f})")
    t = timeit(bwd)

    def lib_bwd():
        torch.nn.grad.conv2d_input(x.shape, wm, dym, stride=2, padding=1)
        torch.nn.grad.conv2d_weight(x, wm.shape, dym, stride=2, padding=1)
        torch.nn.grad.conv2d_input(x.shape, ws, dys, stride=2)
        torch.nn.grad.conv2d_weight(x, ws.shape, dys, stride=2)
    print(f"    bwd {t:.1f} us vs MIOpen (four calls) {timeit(lib_bwd):.1f} us")
