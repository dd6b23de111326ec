"""Dispatch-packet timing of the BatchNorm kernels at the trunk's three shapes: apply (statistics supplied by the
convolution epilogue), backward sums, backward dx.  python tools/bn_micro.py [--iters 40] [--groups G]
(--groups G: G minibatches of --n images per launch, the exact pass's grouped mode; logging statistics)"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import PacketTimer
from bnn_priors_amd import _hip

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=40)
ap.add_argument("--n", type=int, default=128)
ap.add_argument("--groups", type=int, default=1)
a = ap.parse_args()
G = a.groups
lib, dev, n = _hip.lib(), torch.device("cuda", 0), a.n * G
s = torch.cuda.current_stream(dev).cuda_stream
for c, hw in ((16, 32), (32, 16), (64, 8)):
    g = torch.Generator(device=dev).manual_seed(c)
    x = torch.randn((n, c, hw, hw), generator=g, device=dev)
    w = torch.randn((c, c, 3, 3), generator=g, device=dev) * (2.0 / (9 * c)) ** .5
    res = torch.randn((n, c, hw, hw), generator=g, device=dev)
    dout = torch.randn((n, c, hw, hw), generator=g, device=dev)
    y, out, dx_, dres = (torch.empty_like(x) for _ in range(4))
    slices = lib.sgmcmc_conv3x3_stat_slices(n, c, hw)
    stats = torch.empty((c, slices, 2), dtype=torch.float64, device=dev)
    _hip.check(lib.sgmcmc_conv3x3(x.data_ptr(), w.data_ptr(), y.data_ptr(), n, c, hw, 0, stats.data_ptr(), s), "conv")
    gamma, beta = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
    saved = torch.empty((2, G * c), device=dev)
    dgb = torch.empty((G, 2, c), device=dev)
    log = torch.empty((G, c, 2), dtype=torch.float64, device=dev)
    scratch = torch.empty(lib.sgmcmc_bn_scratch_doubles(n, c, hw * hw, G), dtype=torch.float64, device=dev)

    def fwd(r):
        if G > 1:
            return lib.sgmcmc_bn_train_fwd_log(y.data_ptr(), r, gamma.data_ptr(), beta.data_ptr(), log.data_ptr(), 2 * c,
                                               1e-5, 1, n, c, hw * hw, out.data_ptr(), saved[0].data_ptr(),
                                               saved[1].data_ptr(), 0, stats.data_ptr(), slices, G, s)
        return lib.sgmcmc_bn_train_fwd(y.data_ptr(), r, gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(), rv.data_ptr(), 0.1,
                                       1e-5, 1, n, c, hw * hw, out.data_ptr(), saved[0].data_ptr(), saved[1].data_ptr(), 0,
                                       stats.data_ptr(), slices, 1, s)
    stats8 = stats[:, :8 * G].contiguous()      # (timing only) what a prologue over 8 partial pairs per channel costs
    part = torch.randn((c, slices, 2), generator=g, device=dev, dtype=torch.float64)
    part8 = part[:, :8 * G].contiguous()

    def fwd8(r):
        if G > 1:
            return lib.sgmcmc_bn_train_fwd_log(y.data_ptr(), r, gamma.data_ptr(), beta.data_ptr(), log.data_ptr(), 2 * c,
                                               1e-5, 1, n, c, hw * hw, out.data_ptr(), saved[0].data_ptr(),
                                               saved[1].data_ptr(), 0, stats8.data_ptr(), 8 * G, G, s)
        return lib.sgmcmc_bn_train_fwd(y.data_ptr(), r, gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(), rv.data_ptr(), 0.1,
                                       1e-5, 1, n, c, hw * hw, out.data_ptr(), saved[0].data_ptr(), saved[1].data_ptr(), 0,
                                       stats8.data_ptr(), 8, 1, s)

    def dx(pt, k):
        return lib.sgmcmc_bn_bwd_dx(dout.data_ptr(), out.data_ptr(), y.data_ptr(), gamma.data_ptr(), saved[0].data_ptr(),
                                    saved[1].data_ptr(), 1, n, c, hw * hw, pt.data_ptr(), k, dx_.data_ptr(), 0,
                                    dgb.data_ptr(), None, G, s)
    cases = {
        "apply, 8 partials per channel": lambda: fwd8(0),
        f"bwd_dx, {slices // G} partials per channel": lambda: dx(part, slices),
        "bwd_dx, 8 partials per channel": lambda: dx(part8, 8 * G),
        "apply": lambda: fwd(0),
        "apply+res": lambda: fwd(res.data_ptr()),
        "bwd_sums (first launch of bwd)": lambda: lib.sgmcmc_bn_train_bwd(
            dout.data_ptr(), out.data_ptr(), y.data_ptr(), gamma.data_ptr(), saved[0].data_ptr(), saved[1].data_ptr(), 1, n, c,
            hw * hw, dx_.data_ptr(), 0, dgb.data_ptr(), scratch.data_ptr(), G, s),
    }
    for name, fn in cases.items():
        for _ in range(5):
            _hip.check(fn(), name)
        torch.cuda.synchronize(dev)
        t = PacketTimer()
        for _ in range(a.iters):
            t.arm()
            _hip.check(fn(), name)
        ms = t.collect_ms()
        print(f"C={c:2d} HW={hw:2d} {name:32s} avg {1e3 * sum(ms) / len(ms):7.2f} us  min {1e3 * min(ms):7.2f} us", flush=True)
    # the second launch of the backward: total of both launches between two events minus the first
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(dev)
    e0.record()
    for _ in range(a.iters):
        _hip.check(cases["bwd_sums (first launch of bwd)"](), "bwd")
    e1.record()
    torch.cuda.synchronize(dev)
    print(f"C={c:2d} HW={hw:2d} {'bwd (sums + dx) back to back':32s} avg {1e3 * e0.elapsed_time(e1) / a.iters:7.2f} us per pair", flush=True)
