"""The trunk's BatchNorm kernels of ONE stage alone, for counter-only rocprofv3 passes (tools/pmc_hbm.sh): apply (ReLU;
ReLU + residual) from a convolution epilogue's statistics partials and the backward dx launch from upstream sums
partials, a few times each through the C ABI.  One stage per process: the kernels' names and grids do not tell the
stages apart.    python tools/bn_pmc.py --shape 16x32 [--iters 6] [--n 128]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bnn_priors_amd import _hip

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="16x32")
ap.add_argument("--iters", type=int, default=6)
ap.add_argument("--n", type=int, default=128)
a = ap.parse_args()
c, hw = (int(v) for v in a.shape.split("x"))
lib, dev, n = _hip.lib(), torch.device("cuda", 0), a.n
s = torch.cuda.current_stream(dev).cuda_stream
g = torch.Generator(device=dev).manual_seed(c)
x = torch.randn((n, c, hw, hw), generator=g, device=dev)
w = torch.randn((c, c, 3, 3), generator=g, device=dev) * (2.0 / (9 * c)) ** .5
res = torch.randn((n, c, hw, hw), generator=g, device=dev)
dout = torch.randn((n, c, hw, hw), generator=g, device=dev)
y, out, dx = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
slices = lib.sgmcmc_conv3x3_stat_slices(n, c, hw)
stats = torch.empty((c, slices, 2), dtype=torch.float64, device=dev)
_hip.check(lib.sgmcmc_conv3x3(x.data_ptr(), w.data_ptr(), y.data_ptr(), n, c, hw, 0, stats.data_ptr(), s), "conv")
gamma, beta = torch.ones(c, device=dev), torch.zeros(c, device=dev)
rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
saved, dgb = torch.empty((2, c), device=dev), torch.empty((1, 2, c), device=dev)
part = torch.randn((c, slices, 2), generator=g, device=dev, dtype=torch.float64)
# a working set far beyond the L2s between the launches, so that every launch finds its operands where the step's does
# not help it: the counters then show what the kernel itself moves
flush = torch.empty(96 << 20, dtype=torch.float32, device=dev)
for _ in range(a.iters):
    for r in (0, res.data_ptr()):
        flush.add_(1.0)
        _hip.check(lib.sgmcmc_bn_train_fwd(y.data_ptr(), r, gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(), rv.data_ptr(),
                                           0.1, 1e-5, 1, n, c, hw * hw, out.data_ptr(), saved[0].data_ptr(),
                                           saved[1].data_ptr(), 0, stats.data_ptr(), slices, 1, s), "apply")
    flush.add_(1.0)
    # (the step's backward launches: the incoming gradient arrives masked -- bnlink.PREMASK -- relu = 0, `out` not read)
    _hip.check(lib.sgmcmc_bn_bwd_dx(dout.data_ptr(), 0, y.data_ptr(), gamma.data_ptr(), saved[0].data_ptr(),
                                    saved[1].data_ptr(), 0, n, c, hw * hw, part.data_ptr(), slices, dx.data_ptr(), 0,
                                    dgb.data_ptr(), None, 1, s), "bwd_dx")
torch.cuda.synchronize(dev)
print("done")
