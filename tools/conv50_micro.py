"""Dispatch-packet timing of the convolutional classifier's kernels (50->50 @ 14x14 convolution both ways, the first
convolution, the pooling and linear operators).  python tools/conv50_micro.py [--iters 40]"""
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
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn((n, 50, 14, 14), generator=g, device=dev)
dy = torch.randn((n, 50, 14, 14), generator=g, device=dev)
w = torch.randn((50, 50, 3, 3), generator=g, device=dev) * .05
y, dx = torch.empty_like(x), torch.empty_like(x)
scratch = torch.empty(lib.sgmcmc_conv50_scratch_floats(n), device=dev)
slabs = ctypes.c_int(0)
cases = {
    "conv50 fwd": lambda: lib.sgmcmc_conv50(x.data_ptr(), w.data_ptr(), y.data_ptr(), n, 0, s),
    "conv50 dgrad": lambda: lib.sgmcmc_conv50(dy.data_ptr(), w.data_ptr(), dx.data_ptr(), n, 1, s),
    "conv50 wrw only": lambda: lib.sgmcmc_conv50_bwd(x.data_ptr(), w.data_ptr(), dy.data_ptr(), 0, 0, scratch.data_ptr(), n,
                                                     ctypes.byref(slabs), s),
    "conv50 bwd": lambda: lib.sgmcmc_conv50_bwd(x.data_ptr(), w.data_ptr(), dy.data_ptr(), dx.data_ptr(), 0, scratch.data_ptr(),
                                                n, ctypes.byref(slabs), s),
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
    print(f"{name:18s} avg {1e3 * sum(ms) / len(ms):7.2f} us  min {1e3 * min(ms):7.2f} us", flush=True)
