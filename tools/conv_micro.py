"""Kernel-by-kernel timing of the trunk convolutions (dispatch-packet timestamps, as bench.py's roofline rows):
forward, data gradient alone, weight gradient alone (incl. its slab reduction), merged backward, BatchNorm-fused
backward, for the three trunk shapes.  python tools/conv_micro.py [--iters 40]"""
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
a = ap.parse_args()
lib, dev, n = _hip.lib(), torch.device("cuda", 0), a.n
s = torch.cuda.current_stream(dev).cuda_stream
for c, hw in ((16, 32), (32, 16), (64, 8)):
    g = torch.Generator(device=dev).manual_seed(c)
    x = torch.randn((n, c, hw, hw), generator=g, device=dev)
    dy = torch.randn((n, c, hw, hw), generator=g, device=dev)
    w = torch.randn((c, c, 3, 3), generator=g, device=dev) * (2.0 / (9 * c)) ** .5
    y, dx, out, dw = torch.empty_like(x), torch.empty_like(x), torch.relu(torch.randn_like(x)), torch.empty_like(w)
    slices = lib.sgmcmc_conv3x3_stat_slices(n, c, hw)
    stats = torch.empty((c, slices, 2), dtype=torch.float64, device=dev)
    scratch = torch.empty(lib.sgmcmc_conv3x3_wrw_scratch_floats(n, c, hw), device=dev)
    sums = torch.zeros(lib.sgmcmc_bn_scratch_doubles(n, c, hw * hw, 1), dtype=torch.float64, device=dev)
    saved = torch.stack([torch.zeros(c, device=dev), torch.ones(c, device=dev)])
    gamma, dgb = torch.ones(c, device=dev), torch.empty((2, c), device=dev)
    slabs, n_sums = ctypes.c_int(0), ctypes.c_int(0)
    _hip.check(lib.sgmcmc_bn_bwd_sums(dy.data_ptr(), out.data_ptr(), y.data_ptr(), saved[0].data_ptr(),
                                      saved[1].data_ptr(), sums.data_ptr(), ctypes.byref(n_sums), n, c, hw * hw, 1, s), "sums")
    A = _hip.ConvBnBwdArgs(dout=dy.data_ptr(), mask_out=out.data_ptr(), y=y.data_ptr(), mean=saved[0].data_ptr(),
                           invstd=saved[1].data_ptr(), gamma=gamma.data_ptr(), sums=sums.data_ptr(),
                           n_sums=n_sums.value, reserved=0, dgamma=dgb[0].data_ptr(), dbeta=dgb[1].data_ptr(),
                           e_dout=0, e_out=0)
    part = torch.empty((c, slices, 2), dtype=torch.float64, device=dev)
    E_s, E_a, E_as = _hip.ConvBwdEpilogue(), _hip.ConvBwdEpilogue(), _hip.ConvBwdEpilogue()
    for E in (E_s, E_as):
        E.s_y, E.s_out, E.s_mean, E.s_invstd, E.s_partial = y.data_ptr(), out.data_ptr(), saved[0].data_ptr(), saved[1].data_ptr(), part.data_ptr()
    for E in (E_a, E_as):
        E.e_dout, E.e_out = dy.data_ptr(), out.data_ptr()
    cases = {
        "fwd+stats": lambda: lib.sgmcmc_conv3x3(x.data_ptr(), w.data_ptr(), y.data_ptr(), n, c, hw, 0, stats.data_ptr(), s),
        "dgrad": lambda: lib.sgmcmc_conv3x3(dy.data_ptr(), w.data_ptr(), dx.data_ptr(), n, c, hw, 1, 0, s),
        "wrw(first launch)": lambda: lib.sgmcmc_conv3x3_wrw(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), scratch.data_ptr(), n, c, hw, s),
        "bwd": lambda: lib.sgmcmc_conv3x3_bwd(x.data_ptr(), w.data_ptr(), dy.data_ptr(), dx.data_ptr(), 0, scratch.data_ptr(),
                                              n, c, hw, ctypes.byref(slabs), s),
        "bwd+sums": lambda: lib.sgmcmc_conv3x3_bwd_ex(x.data_ptr(), w.data_ptr(), dy.data_ptr(), dx.data_ptr(), ctypes.byref(E_s), 0,
                                                      scratch.data_ptr(), n, c, hw, ctypes.byref(slabs), s),
        "bwd+add": lambda: lib.sgmcmc_conv3x3_bwd_ex(x.data_ptr(), w.data_ptr(), dy.data_ptr(), dx.data_ptr(), ctypes.byref(E_a), 0,
                                                     scratch.data_ptr(), n, c, hw, ctypes.byref(slabs), s),
        "bwd+add+sums": lambda: lib.sgmcmc_conv3x3_bwd_ex(x.data_ptr(), w.data_ptr(), dy.data_ptr(), dx.data_ptr(), ctypes.byref(E_as), 0,
                                                          scratch.data_ptr(), n, c, hw, ctypes.byref(slabs), s),
        **({"bn_bwd": lambda: lib.sgmcmc_conv3x3_bn_bwd(x.data_ptr(), w.data_ptr(), dx.data_ptr(), scratch.data_ptr(),
                                                       ctypes.byref(A), n, c, hw, ctypes.byref(slabs), s)}
           if _hip.ALTERNATIVES else {}),       # (SGMCMC_ALTERNATIVES=1 builds only)
        "bn_bwd_sums": lambda: lib.sgmcmc_bn_bwd_sums(dy.data_ptr(), out.data_ptr(), y.data_ptr(), saved[0].data_ptr(),
                                                      saved[1].data_ptr(), sums.data_ptr(), ctypes.byref(n_sums), n, c, hw * hw, 1, s),
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
        print(f"C={c:2d} HW={hw:2d} {name:18s} avg {1e3 * sum(ms) / len(ms):7.2f} us  min {1e3 * min(ms):7.2f} us", flush=True)
