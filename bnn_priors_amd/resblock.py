"""A residual block of the ResNet trunk as ONE autograd operator over three launches each way
(``csrc/conv_fused_hip.inc``; C ABI ``sgmcmc_block_*``):

    y1 = conv1(x)   h = relu(bn1(y1))   y2 = conv2(h)   out = relu(bn2(y2) + x)

Reference: ``BasicBlock`` with the identity shortcut, bnn_priors/models/google_resnet.py:34-43, 77-90, inside the
gradient evaluation of inference.py:215-223.  Compared with running the block layer by layer (4 launches forward,
7 backward + an ATen add for the shortcut's gradient) the intermediate activation ``h`` is never materialised
-- BatchNorm + ReLU are applied while the next convolution stages its operand -- and both BatchNorm backward
passes live inside the convolutions' gradient launches (operand staging and data-gradient epilogue).  Batch
statistics and the backward sums are finished inside the producing launch by its last workgroup (deterministic,
no floating-point atomics).  Training mode only; everything else takes the layer-by-layer path (``models/nets.py``).
"""
import ctypes
import os

import torch

from . import _hip
from . import conv as _conv

ENABLED = os.environ.get("SGMCMC_BLOCK", "1") != "0"


def supported(x, conv1, bn1, conv2, bn2):
    "identity-shortcut block on one of the trunk's (channels, side) shapes, training-mode BatchNorm with running stats"
    if not (ENABLED and _conv.ENABLED and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[0] > 0):
        return False
    c, hw = x.shape[1], x.shape[2]
    if (c, hw) not in _conv.SHAPES or x.shape[3] != hw:
        return False
    for cv in (conv1, conv2):
        if cv.bias_prior is not None or cv.conv_args != (1, 1, 1, 1) or tuple(cv.weight_prior.p.shape) != (c, c, 3, 3):
            return False
    for bn in (bn1, bn2):
        if not (bn.training and bn.track_running_stats and bn.momentum is not None and bn.affine
                and bn.weight.dtype == torch.float32 and bn.num_features == c):
            return False
    return True


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return 0 if t is None else t.data_ptr()


class _Block(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w1, g1, b1, w2, g2, b2, rm1, rv1, rm2, rv2, mom1, eps1, mom2, eps2, tickets):
        lib = _hip.lib()
        _conv._note_use(w1, w2)
        x, w1, w2 = x.contiguous(), w1.contiguous(), w2.contiguous()
        n, c, hw = x.shape[0], x.shape[1], x.shape[2]
        dev = x.device
        slices = lib.sgmcmc_conv3x3_stat_slices(n, c, hw)
        stats = torch.empty((c, slices, 2), dtype=torch.float64, device=dev)   # consumed inside each launch
        coef = torch.empty((2, 4, c), dtype=torch.float32, device=dev)
        y1, y2, out = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
        s = _stream()
        err = lib.sgmcmc_block_conv_fwd(x.data_ptr(), w1.data_ptr(), 0, y1.data_ptr(), stats.data_ptr(),
                                        coef[0].data_ptr(), g1.data_ptr(), b1.data_ptr(), _p(rm1), _p(rv1),
                                        float(mom1), float(eps1), tickets[0:].data_ptr(), n, c, hw, s)
        if err:
            _hip.check(err, "sgmcmc_block_conv_fwd")
        err = lib.sgmcmc_block_conv_fwd(y1.data_ptr(), w2.data_ptr(), coef[0].data_ptr(), y2.data_ptr(),
                                        stats.data_ptr(), coef[1].data_ptr(), g2.data_ptr(), b2.data_ptr(),
                                        _p(rm2), _p(rv2), float(mom2), float(eps2), tickets[1:].data_ptr(), n, c, hw, s)
        if err:
            _hip.check(err, "sgmcmc_block_conv_fwd")
        err = lib.sgmcmc_block_apply(y2.data_ptr(), x.data_ptr(), coef[1].data_ptr(), out.data_ptr(), n, c, hw * hw, s)
        if err:
            _hip.check(err, "sgmcmc_block_apply")
        ctx.save_for_backward(x, y1, y2, out, w1, w2, g1, g2, coef, tickets)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dout):
        lib = _hip.lib()
        x, y1, y2, out, w1, w2, g1, g2, coef, tickets = ctx.saved_tensors
        dout = dout.contiguous()
        n, c, hw = x.shape[0], x.shape[1], x.shape[2]
        dev, s = x.device, _stream()
        slices = lib.sgmcmc_conv3x3_stat_slices(n, c, hw)
        small = torch.empty((2, 5, c), dtype=torch.float32, device=dev)       # per BN: bcoef[3], dgamma, dbeta
        sums = torch.empty(lib.sgmcmc_block_sums_doubles(n, c, hw * hw), dtype=torch.float64, device=dev)
        esums = torch.empty((c, slices, 2), dtype=torch.float64, device=dev)
        # B1: the last BatchNorm's backward sums
        err = lib.sgmcmc_block_bwd_sums(dout.data_ptr(), out.data_ptr(), y2.data_ptr(), coef[1].data_ptr(),
                                        g2.data_ptr(), sums.data_ptr(), small[1, 0:3].data_ptr(),
                                        small[1, 3].data_ptr(), small[1, 4].data_ptr(), tickets[2:].data_ptr(),
                                        n, c, hw * hw, s)
        if err:
            _hip.check(err, "sgmcmc_block_bwd_sums")
        n_scr = lib.sgmcmc_conv3x3_wrw_scratch_floats(n, c, hw)
        scr = torch.empty((2, n_scr), dtype=torch.float32, device=dev)
        dz1, dx = torch.empty_like(x), torch.empty_like(x)
        dw1, dw2 = torch.empty_like(w1), torch.empty_like(w2)
        slabs = ctypes.c_int(0)
        # B2: conv2's gradients (+ bn2 backward while staging, bn1 backward sums in the epilogue)
        A = _hip.BlockBwdArgs(dz=dout.data_ptr(), mask_out=out.data_ptr(), y=y2.data_ptr(), coef=coef[1].data_ptr(),
                              bcoef=small[1].data_ptr(), xcoef=coef[0].data_ptr(), ye=y1.data_ptr(),
                              ecoef=coef[0].data_ptr(), egamma=g1.data_ptr(), esums=esums.data_ptr(),
                              ebcoef=small[0].data_ptr(), edgamma=small[0, 3].data_ptr(),
                              edbeta=small[0, 4].data_ptr(), e_dout=0, e_out=0, ticket=tickets[3:].data_ptr())
        err = lib.sgmcmc_block_conv_bwd(2, y1.data_ptr(), w2.data_ptr(), dz1.data_ptr(), scr[1].data_ptr(),
                                        ctypes.byref(A), n, c, hw, ctypes.byref(slabs), s)
        if err:
            _hip.check(err, "sgmcmc_block_conv_bwd(2)")
        n_slabs = slabs.value
        # B3: conv1's gradients (+ bn1 backward while staging, shortcut gradient added in the epilogue)
        A = _hip.BlockBwdArgs(dz=dz1.data_ptr(), mask_out=0, y=y1.data_ptr(), coef=coef[0].data_ptr(),
                              bcoef=small[0].data_ptr(), xcoef=0, ye=0, ecoef=0, egamma=0, esums=0, ebcoef=0,
                              edgamma=0, edbeta=0, e_dout=dout.data_ptr(), e_out=out.data_ptr(), ticket=0)
        err = lib.sgmcmc_block_conv_bwd(3, x.data_ptr(), w1.data_ptr(), dx.data_ptr(), scr[0].data_ptr(),
                                        ctypes.byref(A), n, c, hw, ctypes.byref(slabs), s)
        if err:
            _hip.check(err, "sgmcmc_block_conv_bwd(3)")
        outs = []
        for w, dw, part in ((w1, dw1, scr[0]), (w2, dw2, scr[1])):
            if _conv._may_defer(w):      # summed with the pass's other slabs by ONE launch at its end
                torch.autograd.Variable._execution_engine.queue_callback(_conv._flush_pending)
                _conv._pending.append((part, dw, n_slabs))
                outs.append(dw.view(dw.shape))
            else:
                job = (_hip.ReduceJob * 1)()
                job[0].part, job[0].out, job[0].n_slabs, job[0].numel = part.data_ptr(), dw.data_ptr(), n_slabs, dw.numel()
                err = lib.sgmcmc_wrw_reduce_many(ctypes.cast(job, ctypes.c_void_p), 1, s)
                if err:
                    _hip.check(err, "sgmcmc_wrw_reduce_many")
                outs.append(dw)
        return (dx, outs[0], small[0, 3], small[0, 4], outs[1], small[1, 3], small[1, 4]) + (None,) * 9


def residual_block(x, conv1, bn1, conv2, bn2):
    "relu(bn2(conv2(relu(bn1(conv1(x))))) + x) for modules that pass ``supported``"
    tk = bn1.__dict__.get("_block_tickets")
    if tk is None or tk.device != x.device:
        tk = bn1.__dict__["_block_tickets"] = torch.zeros(8, dtype=torch.int32, device=x.device)
    return _Block.apply(x, conv1.weight, bn1.weight, bn1.bias, conv2.weight, bn2.weight, bn2.bias,
                        bn1.running_mean, bn1.running_var, bn2.running_mean, bn2.running_var,
                        bn1.momentum, bn1.eps, bn2.momentum, bn2.eps, tk)
