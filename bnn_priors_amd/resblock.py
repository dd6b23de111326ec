"""A residual block of the ResNet trunk as ONE autograd operator:

    y1 = conv1(x)   h = relu(bn1(y1))   y2 = conv2(h)   out = relu(bn2(y2) + x)

Reference: ``BasicBlock`` with the identity shortcut, bnn_priors/models/google_resnet.py:34-43, 77-90, inside the
gradient evaluation of inference.py:215-223.

Forward: the same four launches as the layer-by-layer path (``conv.conv3x3`` with the batch statistics from its
epilogue, ``bn.bn_train``).  Backward: FOUR launches (the layered path: seven + an ATen add):

    dx(bn2)  ->  conv2 gradients [+ sums(bn1) in the data gradient's epilogue]
             ->  dx(bn1)  ->  conv1 gradients [+ dout*[out>0], + sums(previous BatchNorm) in the epilogue]

A BatchNorm backward needs two per-channel sums over the batch before its element-wise pass.  They are left, per
workgroup band, by the epilogue of the convolution-gradient launch that PRODUCES the BatchNorm's incoming gradient
(``sgmcmc_conv3x3_bwd_ex``, csrc/conv_hip.inc ``band_sums``) -- inside the block for bn1, and across operators for
bn2 through ``bnlink`` tags (the next block's conv1 gradient, a down-sampling pair's data gradient); only a
BatchNorm whose gradient comes from somewhere else (the head) launches ``sgmcmc_bn_bwd_sums`` itself.
Alternatives kept behind switches, both measured slower: the BatchNorm backward formed inside the
convolution-gradient launch while it stages its operands (``FUSED_BN_BWD`` / ``sgmcmc_conv3x3_bn_bwd``: its prologue
needs few partials per channel, so it cannot take epilogue sums) and finishing the sums inside the producing launch
by a last-arriver ticket (7-12 us per hand-off on MI355X against ~1.5 us for a kernel boundary -- DESIGN.md).
Training mode only; anything else takes the layer-by-layer path (``models/nets.py``).
"""
import ctypes
import os

import torch

from . import _hip
from . import bn as _bn
from . import bnlink as _bnlink
from . import conv as _conv

ENABLED = True          # (module attribute, not an environment switch)
# (channels, side) for which the BatchNorm backward is formed inside the convolution-gradient launch.  Measured on
# MI355X (profiles/): 32 @ 16x16 and 64 @ 8x8 gain 1-5 us per pair over bwd_dx + conv3x3_bwd; at 16 @ 32x32 the three
# 8 MB operand tensors staged by BOTH the data- and the weight-gradient workgroups cost 4 us more than the extra
# launch saves, so that shape keeps the two-launch BatchNorm backward (and only fuses the shortcut's add).
FUSED_BN_BWD = {(32, 16), (64, 8)}          # (module attribute: only consulted when SGMCMC_BLOCK_EPILOGUE_SUMS=0)
# The route that replaced it (default): the BatchNorm backward's SUMS come out of the epilogue of the
# convolution-gradient launch that produces the BatchNorm's incoming gradient (``sgmcmc_conv3x3_bwd_ex``), so a
# BatchNorm backward is the dx launch alone and a block's backward is four launches whatever the shape:
#     [sums(bn2): epilogue of the NEXT block's conv1 gradient, else its own launch]
#     dx(bn2) -> conv2 gradients [+ sums(bn1)] -> dx(bn1) -> conv1 gradients [+ dout*[out>0]] [+ sums(previous bn2)]
# SGMCMC_BLOCK_EPILOGUE_SUMS=0 restores the routes above.
EPILOGUE_SUMS = os.environ.get("SGMCMC_BLOCK_EPILOGUE_SUMS", "1") != "0"
if not EPILOGUE_SUMS and FUSED_BN_BWD and not _hip.ALTERNATIVES:
    raise RuntimeError("SGMCMC_BLOCK_EPILOGUE_SUMS=0 with fused shapes selects sgmcmc_conv3x3_bn_bwd, a measured "
                       "alternative: build and load the library with SGMCMC_ALTERNATIVES=1")


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


def _conv_bn_fwd(lib, x, w, g, b, rm, rv, mom, eps, residual, s, G=1):
    "y = conv3x3(x, w); out = relu(bn(y) [+ residual]) -> y, out, saved (mean, invstd; [2, G * C])"
    n, c, hw = x.shape[0], x.shape[1], x.shape[2]
    y, out = torch.empty_like(x), torch.empty_like(x)
    saved = torch.empty((2, G * c), dtype=torch.float32, device=x.device)
    if _conv.PERSISTENT:
        slices = lib.sgmcmc_conv3x3_frag_stat_slices(n, c, hw)
        stats = torch.empty((c, slices, 2), dtype=torch.float64, device=x.device)
        err = lib.sgmcmc_conv3x3_frag_fwd(x.data_ptr(), _conv.frags(w)[0].data_ptr(), y.data_ptr(), n, c, hw,
                                          stats.data_ptr(), s)
    else:
        slices = lib.sgmcmc_conv3x3_stat_slices(n, c, hw)
        stats = torch.empty((c, slices, 2), dtype=torch.float64, device=x.device)
        err = lib.sgmcmc_conv3x3(x.data_ptr(), w.data_ptr(), y.data_ptr(), n, c, hw, 0, stats.data_ptr(), s)
    if err:
        _hip.check(err, "sgmcmc_conv3x3")
    err = _bn.train_fwd(lib, y, residual, g, b, rm, rv, mom, eps, 1, n, c, hw * hw, out, saved, None, stats, slices, s, G)
    if err:
        _hip.check(err, "sgmcmc_bn_train_fwd")
    return y, out, saved


# The block's FIRST BatchNorm + ReLU folded into the second convolution's operand staging (round 3): conv1 leaves its
# batch statistics in integer fx slots (csrc/conv_hip.inc: order-independent, hence bit-reproducible, atomics into per-XCD
# slots), conv2 forms the BatchNorm's coefficients from them in its prologue, applies BatchNorm + ReLU to its operands
# on their way into LDS and writes h as a side output -- three launches forward instead of four.  MEASURED, OFF BY
# DEFAULT: 83 -> 77 launches (7 apply launches go, one zeroing launch comes) but the same 1,183 steps/s -- the two
# convolutions get slower by what the apply launch cost (csrc/conv_hip.inc, BnIn).  SGMCMC_FOLD_BN=1 enables it (its
# statistics, exact integer totals, differ from the per-slice pairs' combination in rounding; tested at both settings).
FOLD_BN = os.environ.get("SGMCMC_FOLD_BN", "0") == "1"
if FOLD_BN and not _hip.ALTERNATIVES:
    raise RuntimeError("SGMCMC_FOLD_BN=1 selects a measured alternative: build and load the library with "
                       "SGMCMC_ALTERNATIVES=1")


def _conv_bn_conv_fwd(lib, x, w1, g1, b1, rm1, rv1, mom1, eps1, w2, s):
    "y1 = conv3x3(x, w1); h = relu(bn1(y1)); y2 = conv3x3(h, w2) + y2's statistics -> y1, h, saved1, y2, stats2, slices"
    n, c, hw = x.shape[0], x.shape[1], x.shape[2]
    y1, h, y2 = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    saved1 = torch.empty((2, c), dtype=torch.float32, device=x.device)
    fx = _conv.fx_take(c, x.device)
    err = lib.sgmcmc_conv3x3_fx(x.data_ptr(), w1.data_ptr(), y1.data_ptr(), n, c, hw, fx.data_ptr(), s)
    if err:
        _hip.check(err, "sgmcmc_conv3x3_fx")
    slices = lib.sgmcmc_conv3x3_stat_slices(n, c, hw)
    stats2 = torch.empty((c, slices, 2), dtype=torch.float64, device=x.device)
    slot = _bn.log_slot(rm1)
    B = _hip.BnIn(fx=fx.data_ptr(), gamma=g1.data_ptr(), beta=b1.data_ptr(), save_mean=saved1[0].data_ptr(),
                  save_invstd=saved1[1].data_ptr(), running_mean=0 if slot is not None else rm1.data_ptr(),
                  running_var=0 if slot is not None else rv1.data_ptr(), stat_log=_p(slot), momentum=float(mom1),
                  eps=float(eps1), h=h.data_ptr())
    err = lib.sgmcmc_conv3x3_bnin(y1.data_ptr(), w2.data_ptr(), y2.data_ptr(), n, c, hw, stats2.data_ptr(),
                                  ctypes.byref(B), s)
    if err:
        _hip.check(err, "sgmcmc_conv3x3_bnin")
    return y1, h, saved1, y2, stats2, slices


def _reduce_or_defer(lib, w, part, n_slabs, s):
    dw = torch.empty_like(w)
    if _conv._may_defer(w):      # summed with the pass's other slabs by ONE launch at its end
        torch.autograd.Variable._execution_engine.queue_callback(_conv._flush_pending)
        _conv._pending.append((part, dw, n_slabs, 9))
        return dw.view(dw.shape)
    job = (_hip.ReduceJob * 1)()
    job[0].part, job[0].out, job[0].n_slabs, job[0].numel, job[0].taps = part.data_ptr(), dw.data_ptr(), n_slabs, dw.numel(), 9
    err = lib.sgmcmc_wrw_reduce_many(ctypes.cast(job, ctypes.c_void_p), 1, s)
    if err:
        _hip.check(err, "sgmcmc_wrw_reduce_many")
    return dw


def _conv_bn_bwd_two_launch(lib, x, w, y, out, dout, saved, g, dgb, e_dout, e_out, s):
    """the same gradients with the BatchNorm backward as its own two launches (sums, dx) before conv3x3_bwd;
    e_dout / e_out: dx += e_dout * [e_out > 0] in the data gradient's epilogue (what the shortcut carries)"""
    n, c, hw = x.shape[0], x.shape[1], x.shape[2]
    scratch = torch.empty(lib.sgmcmc_bn_scratch_doubles(n, c, hw * hw, 1), dtype=torch.float64, device=x.device)
    dy = torch.empty_like(x)
    err = lib.sgmcmc_bn_train_bwd(dout.data_ptr(), out.data_ptr(), y.data_ptr(), g.data_ptr(), saved[0].data_ptr(),
                                  saved[1].data_ptr(), 1, n, c, hw * hw, dy.data_ptr(), 0, dgb.data_ptr(),
                                  scratch.data_ptr(), 1, s)
    if err:
        _hip.check(err, "sgmcmc_bn_train_bwd")
    part = torch.empty(lib.sgmcmc_conv3x3_wrw_scratch_floats(n, c, hw), dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x)
    slabs = ctypes.c_int(0)
    if e_dout is None:
        err = lib.sgmcmc_conv3x3_bwd(x.data_ptr(), w.data_ptr(), dy.data_ptr(), dx.data_ptr(), 0, part.data_ptr(), n, c, hw,
                                     ctypes.byref(slabs), s)
    else:
        err = lib.sgmcmc_conv3x3_bwd_add(x.data_ptr(), w.data_ptr(), dy.data_ptr(), dx.data_ptr(), e_dout.data_ptr(),
                                         e_out.data_ptr(), 0, part.data_ptr(), n, c, hw, ctypes.byref(slabs), s)
    if err:
        _hip.check(err, "sgmcmc_conv3x3_bwd")
    return dx, _reduce_or_defer(lib, w, part, slabs.value, s)


def _conv_bn_bwd(lib, x, w, y, out, dout, saved, g, dgb, e_dout, e_out, s):
    """gradients of (x, w) through out = relu(bn(conv3x3(x, w)) [+ r]) given dout, BatchNorm backward inside the
    convolution-gradient launch; dgb [2][C] <- dgamma, dbeta; e_dout / e_out: dx += e_dout * [e_out > 0]"""
    n, c, hw = x.shape[0], x.shape[1], x.shape[2]
    sums = torch.empty(lib.sgmcmc_bn_scratch_doubles(n, c, hw * hw, 1), dtype=torch.float64, device=x.device)
    n_sums = ctypes.c_int(0)
    err = lib.sgmcmc_bn_bwd_sums(dout.data_ptr(), out.data_ptr(), y.data_ptr(), saved[0].data_ptr(),
                                 saved[1].data_ptr(), sums.data_ptr(), ctypes.byref(n_sums), n, c, hw * hw, 1, s)
    if err:
        _hip.check(err, "sgmcmc_bn_bwd_sums")
    part = torch.empty(lib.sgmcmc_conv3x3_wrw_scratch_floats(n, c, hw), dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x)
    A = _hip.ConvBnBwdArgs(dout=dout.data_ptr(), mask_out=out.data_ptr(), y=y.data_ptr(), mean=saved[0].data_ptr(),
                           invstd=saved[1].data_ptr(), gamma=g.data_ptr(), sums=sums.data_ptr(), n_sums=n_sums.value,
                           reserved=0, dgamma=dgb[0].data_ptr(), dbeta=dgb[1].data_ptr(), e_dout=_p(e_dout),
                           e_out=_p(e_out))
    slabs = ctypes.c_int(0)
    err = lib.sgmcmc_conv3x3_bn_bwd(x.data_ptr(), w.data_ptr(), dx.data_ptr(), part.data_ptr(), ctypes.byref(A), n, c,
                                    hw, ctypes.byref(slabs), s)
    if err:
        _hip.check(err, "sgmcmc_conv3x3_bn_bwd")
    return dx, _reduce_or_defer(lib, w, part, slabs.value, s)


def _bn_dx(lib, dout, out, y, saved, g, dgb, partial, n_partials, s, G=1, masked=False):
    """dy of out = relu(bn(y) [+ r]) given dout and the channel sums' partials; dgb [G][2][C] <- the groups' dgamma, dbeta;
    ``masked``: dout arrives as dout * [out > 0] (bnlink.PREMASK) and ``out`` is not read"""
    n, c, hw = y.shape[0], y.shape[1], y.shape[2]
    dy = torch.empty_like(y)
    err = lib.sgmcmc_bn_bwd_dx(dout.data_ptr(), 0 if masked else out.data_ptr(), y.data_ptr(), g.data_ptr(),
                               saved[0].data_ptr(), saved[1].data_ptr(), 0 if masked else 1, n, c, hw * hw,
                               partial.data_ptr(), n_partials, dy.data_ptr(), 0, dgb.data_ptr(), None, G, s)
    if err:
        _hip.check(err, "sgmcmc_bn_bwd_dx")
    return dy


def _bn_sums(lib, dout, out, y, saved, s, G=1):
    "the sums launch on its own -> (partial, n_partials)"
    n, c, hw = y.shape[0], y.shape[1], y.shape[2]
    sums = torch.empty(lib.sgmcmc_bn_scratch_doubles(n, c, hw * hw, G), dtype=torch.float64, device=y.device)
    n_sums = ctypes.c_int(0)
    err = lib.sgmcmc_bn_bwd_sums(dout.data_ptr(), out.data_ptr(), y.data_ptr(), saved[0].data_ptr(),
                                 saved[1].data_ptr(), sums.data_ptr(), ctypes.byref(n_sums), n, c, hw * hw, G, s)
    if err:
        _hip.check(err, "sgmcmc_bn_bwd_sums")
    return sums, n_sums.value


def _conv_bwd_ex(lib, x, w, dy, s, add=None, sums_for=None, G=1):
    """both gradients of y = conv3x3(x, w) given dy; ``add`` = (e_dout, e_out): dx += e_dout * [e_out > 0] (e_out None:
    e_dout is masked already); ``sums_for`` = (y_bn, out_bn, saved_bn): also the partial sums of the BatchNorm backward
    whose incoming gradient dx is, and (bnlink.PREMASK) dx is stored as dx * [out_bn > 0]
    -> (dx, dw, partial or None, n_partials)"""
    n, c, hw = x.shape[0], x.shape[1], x.shape[2]
    if _conv.persistent_bwd(c, hw) and not (_conv.SIDE_STREAM and _conv._may_defer(w)):
        defer = _conv._may_defer(w)
        if defer:      # summed with the pass's other slabs by ONE launch at its end
            torch.autograd.Variable._execution_engine.queue_callback(_conv._flush_pending)
        return _conv.frag_backward(lib, x, w, dy, defer, add=add, sums_for=sums_for)
    part = torch.empty(lib.sgmcmc_conv3x3_wrw_scratch_floats(n, c, hw), dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x)
    E = _hip.ConvBwdEpilogue()
    E.wrw_mult = _conv.WRW_GROUP_MULT and G       # (several minibatches: the slab count of one)
    if add is not None:
        E.e_dout, E.e_out = add[0].data_ptr(), _p(add[1])
    partial, n_partials = None, lib.sgmcmc_conv3x3_stat_slices(n, c, hw)
    if sums_for is not None:
        y_bn, out_bn, saved_bn = sums_for
        partial = torch.empty((c, n_partials, 2), dtype=torch.float64, device=x.device)
        E.s_y, E.s_out, E.s_mean, E.s_invstd = y_bn.data_ptr(), out_bn.data_ptr(), saved_bn[0].data_ptr(), saved_bn[1].data_ptr()
        E.s_partial = partial.data_ptr()
        E.group_imgs = n // G if G > 1 else 0
        E.mask_dx = int(_bnlink.PREMASK)
    slabs = ctypes.c_int(0)
    if _conv.SIDE_STREAM and _conv._may_defer(w):
        # the weight-gradient slabs leave the critical path: side stream, joined before the pass's slab reduction
        _conv.split_backward(lib, x, w, dy, dx, E, part, slabs)
    else:
        err = lib.sgmcmc_conv3x3_bwd_ex(x.data_ptr(), w.data_ptr(), dy.data_ptr(), dx.data_ptr(), ctypes.byref(E), 0,
                                        part.data_ptr(), n, c, hw, ctypes.byref(slabs), s)
        if err:
            _hip.check(err, "sgmcmc_conv3x3_bwd_ex")
    return dx, _reduce_or_defer(lib, w, part, slabs.value, s), partial, n_partials


class _Block(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w1, g1, b1, w2, g2, b2, rm1, rv1, rm2, rv2, mom1, eps1, mom2, eps2, src_y, src_saved):
        lib = _hip.lib()
        _conv._note_use(w1, w2)
        x, w1, w2 = x.contiguous(), w1.contiguous(), w2.contiguous()
        s = _stream()
        G = ctx.groups = _bn._groups_of(x)
        if G > 1:
            if FOLD_BN or not EPILOGUE_SUMS or _conv.SIDE_STREAM:
                raise RuntimeError("bn.grouped(G > 1): only the epilogue-sums route (default or persistent convolutions)")
            _conv._note_use(g1, b1, g2, b2)
        if FOLD_BN and not _conv.PERSISTENT:
            y1, h, saved1, y2, stats2, slices = _conv_bn_conv_fwd(lib, x, w1, g1, b1, rm1, rv1, mom1, eps1, w2, s)
            n, c, hw = x.shape[0], x.shape[1], x.shape[2]
            out = torch.empty_like(x)
            saved2 = torch.empty((2, c), dtype=torch.float32, device=x.device)
            err = _bn.train_fwd(lib, y2, x, g2, b2, rm2, rv2, mom2, eps2, 1, n, c, hw * hw, out, saved2, None, stats2,
                                slices, s)
            if err:
                _hip.check(err, "sgmcmc_bn_train_fwd")
        else:
            y1, h, saved1 = _conv_bn_fwd(lib, x, w1, g1, b1, rm1, rv1, mom1, eps1, None, s, G)
            y2, out, saved2 = _conv_bn_fwd(lib, h, w2, g2, b2, rm2, rv2, mom2, eps2, x, s, G)
        ctx.save_for_backward(x, y1, h, y2, out, w1, w2, g1, g2, saved1, saved2, src_y, src_saved, b1, b2)
        ctx.mark_non_differentiable(y2, saved2)
        ctx.set_materialize_grads(False)
        return out, y2, saved2        # (y2, saved2: where `out` came from, for the next operator -- bnlink)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dout, *_):
        lib = _hip.lib()
        x, y1, h, y2, out, w1, w2, g1, g2, saved1, saved2, src_y, src_saved, b1, b2 = ctx.saved_tensors
        if dout is None:
            return (None,) * 17
        up = _bnlink.sums_of(dout) if EPILOGUE_SUMS else None
        dout = dout.contiguous()
        s = _stream()
        G = ctx.groups
        dgb = torch.empty((2, G, 2, x.shape[1]), dtype=torch.float32, device=x.device)     # [layer][group][gamma, beta][C]
        if EPILOGUE_SUMS:
            # bn2's sums: left by the launch that produced dout (the next block's conv1 gradient), else a launch here
            # (masked: the launch that produced dout stored it as dout * [out > 0] -- bnlink.PREMASK)
            sums2, n2, masked = up if up is not None else (*_bn_sums(lib, dout, out, y2, saved2, s, G), False)
            dy2 = _bn_dx(lib, dout, out, y2, saved2, g2, dgb[1], sums2, n2, s, G, masked)
            dh, dw2, sums1, n1 = _conv_bwd_ex(lib, h, w2, dy2, s, sums_for=(y1, h, saved1), G=G)
            dy1 = _bn_dx(lib, dh, h, y1, saved1, g1, dgb[0], sums1, n1, s, G, _bnlink.PREMASK)
            # ... and this block's input came out of a BatchNorm + ReLU too: its sums ride in conv1's gradient launch
            dx, dw1, sums0, n0 = _conv_bwd_ex(lib, x, w1, dy1, s, add=(dout, None if masked else out),
                                              sums_for=None if src_y is None else (src_y, x, src_saved), G=G)
            if sums0 is not None:
                _bnlink.tag_gradient(dx, sums0, n0, _bnlink.PREMASK)
        elif (x.shape[1], x.shape[2]) in FUSED_BN_BWD:
            dh, dw2 = _conv_bn_bwd(lib, h, w2, y2, out, dout, saved2, g2, dgb[1, 0], None, None, s)
            # the shortcut carries dz2 = dout * [out > 0] back to x: added in conv1's data-gradient epilogue
            dx, dw1 = _conv_bn_bwd(lib, x, w1, y1, h, dh, saved1, g1, dgb[0, 0], dout, out, s)
        else:
            dh, dw2 = _conv_bn_bwd_two_launch(lib, h, w2, y2, out, dout, saved2, g2, dgb[1], None, None, s)
            dx, dw1 = _conv_bn_bwd_two_launch(lib, x, w1, y1, h, dh, saved1, g1, dgb[0], dout, out, s)
        dg1, db1 = _bn.sum_groups(dgb[0], g1, b1)
        dg2, db2 = _bn.sum_groups(dgb[1], g2, b2)
        return (dx, dw1, dg1, db1, dw2, dg2, db2) + (None,) * 10


def residual_block(x, conv1, bn1, conv2, bn2):
    "relu(bn2(conv2(relu(bn1(conv1(x))))) + x) for modules that pass ``supported``"
    src_y, src_saved = _bnlink.source_of(x) if (EPILOGUE_SUMS and x.requires_grad) else (None, None)
    out, y2, saved2 = _Block.apply(x, conv1.weight, bn1.weight, bn1.bias, conv2.weight, bn2.weight, bn2.bias,
                                   bn1.running_mean, bn1.running_var, bn2.running_mean, bn2.running_var,
                                   bn1.momentum, bn1.eps, bn2.momentum, bn2.eps, src_y, src_saved)
    return _bnlink.tag_output(out, y2, saved2)
