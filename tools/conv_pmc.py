"""The trunk's convolution kernels alone, for counter-only rocprofv3 passes (tools/pmc_hbm.sh, tools/pmc_mfma.sh):
each shape's forward (conv3x3 + statistics), backward (conv3x3_bwd, plain and with the sums epilogue the step uses)
and the alternative BatchNorm-fused backward (conv3x3_bn_bwd) launched a few times through the C ABI on synthetic tensors of the step's shapes.
    python tools/conv_pmc.py [--iters 6] [--n 128]"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bnn_priors_amd import _hip

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=6)
ap.add_argument("--n", type=int, default=128)
a = ap.parse_args()
lib, dev, n = _hip.lib(), torch.device("cuda", 0), a.n
s = torch.cuda.current_stream(dev).cuda_stream
for c, hw in ((16, 32), (32, 16), (64, 8)):
    g = torch.Generator(device=dev).manual_seed(c)
    x = torch.randn((n, c, hw, hw), generator=g, device=dev)
    dy = torch.randn((n, c, hw, hw), generator=g, device=dev)
    w = torch.randn((c, c, 3, 3), generator=g, device=dev) * (2.0 / (9 * c)) ** .5
    y, dx, out = torch.empty_like(x), torch.empty_like(x), torch.relu(torch.randn_like(x))
    slices = lib.sgmcmc_conv3x3_stat_slices(n, c, hw)
    stats = torch.empty((c, slices, 2), dtype=torch.float64, device=dev)
    scratch = torch.empty(lib.sgmcmc_conv3x3_wrw_scratch_floats(n, c, hw), device=dev)
    sums = torch.zeros(lib.sgmcmc_bn_scratch_doubles(n, c, hw * hw, 1), dtype=torch.float64, device=dev)
    saved = torch.stack([torch.zeros(c, device=dev), torch.ones(c, device=dev)])
    gamma, dgb = torch.ones(c, device=dev), torch.empty((2, c), device=dev)
    slabs, n_sums = ctypes.c_int(0), ctypes.c_int(0)
    part = torch.empty((c, slices, 2), dtype=torch.float64, device=dev)
    ALT = _hip.ALTERNATIVES         # the measured alternatives exist in SGMCMC_ALTERNATIVES=1 builds only
    # the persistent alternative (csrc/conv2_hip.inc): fragments, forward with statistics, backward with the sums epilogue
    f_fwd, f_dg = torch.empty_like(w).view(-1), torch.empty_like(w).view(-1)
    if ALT:
        slices2 = lib.sgmcmc_conv3x3_frag_stat_slices(n, c, hw)
        stats2 = torch.empty((c, slices2, 2), dtype=torch.float64, device=dev)
        part2 = torch.empty((c, slices2, 2), dtype=torch.float64, device=dev)
        scratch2 = torch.empty(lib.sgmcmc_conv3x3_frag_scratch_floats(n, c, hw), device=dev)
        job = (_hip.FragJob * 1)()
        job[0].w, job[0].fwd, job[0].dgrad, job[0].channels = w.data_ptr(), f_fwd.data_ptr(), f_dg.data_ptr(), c
    for _ in range(a.iters):
        if ALT:
            _hip.check(lib.sgmcmc_conv3x3_prepare_weights(ctypes.cast(job, ctypes.c_void_p), 1, s), "frag")
            _hip.check(lib.sgmcmc_conv3x3_frag_fwd(x.data_ptr(), f_fwd.data_ptr(), y.data_ptr(), n, c, hw, stats2.data_ptr(), s), "frag_fwd")
            E2 = _hip.ConvBwdEpilogue(s_y=y.data_ptr(), s_out=out.data_ptr(), s_mean=saved[0].data_ptr(),
                                      s_invstd=saved[1].data_ptr(), s_partial=part2.data_ptr())
            _hip.check(lib.sgmcmc_conv3x3_frag_bwd(x.data_ptr(), f_dg.data_ptr(), dy.data_ptr(), dx.data_ptr(), ctypes.byref(E2), 0,
                                                   scratch2.data_ptr(), n, c, hw, ctypes.byref(slabs), s), "frag_bwd")
        _hip.check(lib.sgmcmc_conv3x3(x.data_ptr(), w.data_ptr(), y.data_ptr(), n, c, hw, 0, stats.data_ptr(), s), "fwd")
        _hip.check(lib.sgmcmc_conv3x3_bwd(x.data_ptr(), w.data_ptr(), dy.data_ptr(), dx.data_ptr(), 0, scratch.data_ptr(),
                                          n, c, hw, ctypes.byref(slabs), s), "bwd")
        E = _hip.ConvBwdEpilogue(s_y=y.data_ptr(), s_out=out.data_ptr(), s_mean=saved[0].data_ptr(), s_invstd=saved[1].data_ptr(),
                                 s_partial=part.data_ptr(), mask_dx=1)
        _hip.check(lib.sgmcmc_conv3x3_bwd_ex(x.data_ptr(), w.data_ptr(), dy.data_ptr(), dx.data_ptr(), ctypes.byref(E), 0,
                                             scratch.data_ptr(), n, c, hw, ctypes.byref(slabs), s), "bwd_ex")    # as the step runs it
        E3 = _hip.ConvBwdEpilogue(e_dout=dy.data_ptr(), e_out=0, s_y=y.data_ptr(), s_out=out.data_ptr(),
                                  s_mean=saved[0].data_ptr(), s_invstd=saved[1].data_ptr(), s_partial=part.data_ptr(), mask_dx=1)
        _hip.check(lib.sgmcmc_conv3x3_bwd_ex(x.data_ptr(), w.data_ptr(), dy.data_ptr(), dx.data_ptr(), ctypes.byref(E3), 0,
                                             scratch.data_ptr(), n, c, hw, ctypes.byref(slabs), s), "bwd_ex(add)")   # an identity block's first convolution
        _hip.check(lib.sgmcmc_bn_bwd_sums(dy.data_ptr(), out.data_ptr(), y.data_ptr(), saved[0].data_ptr(),
                                          saved[1].data_ptr(), sums.data_ptr(), ctypes.byref(n_sums), n, c, hw * hw, 1, s), "sums")
        if ALT:
            A = _hip.ConvBnBwdArgs(dout=dy.data_ptr(), mask_out=out.data_ptr(), y=y.data_ptr(), mean=saved[0].data_ptr(),
                                   invstd=saved[1].data_ptr(), gamma=gamma.data_ptr(), sums=sums.data_ptr(),
                                   n_sums=n_sums.value, reserved=0, dgamma=dgb[0].data_ptr(), dbeta=dgb[1].data_ptr(),
                                   e_dout=0, e_out=0)
            _hip.check(lib.sgmcmc_conv3x3_bn_bwd(x.data_ptr(), w.data_ptr(), dx.data_ptr(), scratch.data_ptr(), ctypes.byref(A),
                                                 n, c, hw, ctypes.byref(slabs), s), "bn_bwd")
    torch.cuda.synchronize(dev)
print("done")
