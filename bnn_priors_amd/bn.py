"""Training-mode BatchNorm2d fused with the residual add and ReLU that follow it in the ResNet
trunk, on hand-written HIP kernels (``csrc/bn_hip.inc``; C ABI ``sgmcmc_bn_train_fwd/_bwd``).

``bn_train(x, weight, bias, running_mean, running_var, momentum, eps, residual=None, relu=False)``
equals ``relu?(F.batch_norm(x, ..., training=True) [+ residual])`` up to fp32 rounding, updates
the running statistics like ``nn.BatchNorm2d`` and is differentiable in x, weight, bias and
residual.  Two launches forward (one when the producing convolution hands over the batch statistics), two
backward, deterministic statistics (no atomics).
Reference: the BatchNorm layers of bnn_priors/models/google_resnet.py:34-43, 77-90.
"""
import contextlib
import ctypes
import os

import torch

from . import _hip
from . import bnlink as _bnlink
from . import conv as _conv

ENABLED = True          # (module attribute, not an environment switch)


def supported(x, weight, bias, training, momentum):
    return (ENABLED and training and momentum is not None and weight is not None and bias is not None
            and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[0] > 0
            and (x.shape[2] * x.shape[3]) % 4 == 0 and weight.dtype == torch.float32)


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def eval_supported(x, weight, bias, running_mean, running_var):
    "evaluation mode (running statistics) without a graph: the per-epoch test pass and the stored samples' predictions"
    return (ENABLED and not torch.is_grad_enabled() and weight is not None and bias is not None
            and running_mean is not None and running_var is not None and x.is_cuda and x.dtype == torch.float32
            and x.dim() == 4 and x.shape[0] > 0 and (x.shape[2] * x.shape[3]) % 4 == 0
            and weight.dtype == torch.float32 and not torch._C._functorch.is_batchedtensor(weight))


def bn_eval(x, weight, bias, running_mean, running_var, eps, residual=None, relu=False):
    "relu?(batch_norm(x; running statistics) [+ residual]) in one launch (sgmcmc_bn_eval_fwd)"
    x = x.contiguous()
    if residual is not None:
        residual = residual.contiguous()
    y = torch.empty_like(x)
    n, c, plane = x.shape[0], x.shape[1], x.shape[2] * x.shape[3]
    err = _hip.lib().sgmcmc_bn_eval_fwd(x.data_ptr(), _ptr(residual), weight.data_ptr(), bias.data_ptr(),
                                        running_mean.data_ptr(), running_var.data_ptr(), float(eps), int(bool(relu)), n, c,
                                        plane, y.data_ptr(), torch.cuda.current_stream().cuda_stream)
    if err:
        _hip.check(err, "sgmcmc_bn_eval_fwd")
    return y


# ---- running statistics logged instead of updated ---------------------------------------------------------------
# Inside ``with logging_running_stats(slots):`` (slots: id(running_mean tensor) -> float64 [C, 2] device tensor) a
# training-mode forward leaves its batch mean / unbiased variance in the layer's slot and does NOT touch the running
# statistics; ``replay_running_stats`` advances them later by a sequence of logged batches, in order (the same bits
# as forwards in that order).  Used by the exact full-data pass when its minibatches run on several streams at once
# (graphed.ConcurrentAccumulate).  A BatchNorm that cannot log (library path) raises LogModeUnsupported.
_log = {"slots": None}


class LogModeUnsupported(RuntimeError):
    pass


@contextlib.contextmanager
def logging_running_stats(slots):
    old = _log["slots"]
    _log["slots"] = slots
    try:
        yield
    finally:
        _log["slots"] = old


def log_active():
    return _log["slots"] is not None


def log_slot(running_mean):
    "the slot of the BatchNorm that owns ``running_mean`` while logging is active, else None"
    slots = _log["slots"]
    if slots is None or running_mean is None:
        return None
    slot = slots.get(id(running_mean))
    if slot is None:
        raise LogModeUnsupported("a BatchNorm layer outside the logged set ran in log mode")
    return slot


# ---- several minibatches per launch (round 4) -------------------------------------------------------------------
# Inside ``with grouped(G):`` a batch of G * B rows is G independent minibatches of B rows stored one after the other
# (csrc/bn_hip.inc, GROUPS): every training-mode BatchNorm of this package normalises each group with its own batch
# statistics -- the bits of a forward on that group alone -- keeps [G][C] saved statistics, logs G rows (log mode is
# required: G in-place updates of the running statistics would have to be ordered), and its backward returns the sum of
# the groups' (dgamma, dbeta).  The convolutions, the pooling head and the loss are per image and need nothing; the
# launches that leave a BatchNorm's backward sums in their epilogues (bnlink) look the group's statistics up by image.
# Used by the exact full-data gradient (graphed.py): G minibatches per launch chain instead of one.
_grp = {"G": 1}


@contextlib.contextmanager
def grouped(G):
    old = _grp["G"]
    _grp["G"] = int(G)
    try:
        yield
    finally:
        _grp["G"] = old


def groups():
    return _grp["G"]


def _groups_of(x):
    "the active group count, checked against the batch"
    G = _grp["G"]
    if G > 1 and x.shape[0] % G:
        raise ValueError(f"bn.grouped({G}): a batch of {x.shape[0]} rows is not {G} equal minibatches")
    return G


def _slot_ptr_stride(slot, G):
    "(data pointer, doubles between two groups' rows) of a log slot: [C, 2] for one group, [G, C, 2] (any group stride) else"
    if G == 1:
        return slot.data_ptr(), 0
    if slot.dim() != 3 or slot.shape[0] != G or slot.stride(1) != 2 or slot.stride(2) != 1:
        raise LogModeUnsupported(f"bn.grouped({G}) needs [G, C, 2] log slots")
    return slot.data_ptr(), slot.stride(0)


def train_fwd(lib, x, residual, weight, bias, running_mean, running_var, momentum, eps, relu, n, c, plane, y, saved,
              scratch, stats_in, stats_slices, stream, G=1):
    """sgmcmc_bn_train_fwd, or its logging variant while ``logging_running_stats`` is active (all arguments tensors /
    None); ``saved``: [2, G * C] (mean, invstd per group and channel)"""
    slot = log_slot(running_mean)
    if slot is None:
        if G > 1 and running_mean is not None:
            raise LogModeUnsupported("bn.grouped(G > 1) advances running statistics through the log only")
        return lib.sgmcmc_bn_train_fwd(x.data_ptr(), _ptr(residual), weight.data_ptr(), bias.data_ptr(),
                                       _ptr(running_mean), _ptr(running_var), float(momentum), float(eps), int(relu),
                                       n, c, plane, y.data_ptr(), saved[0].data_ptr(), saved[1].data_ptr(),
                                       _ptr(scratch), _ptr(stats_in), stats_slices, G, stream)
    ptr, stride = _slot_ptr_stride(slot, G)
    return lib.sgmcmc_bn_train_fwd_log(x.data_ptr(), _ptr(residual), weight.data_ptr(), bias.data_ptr(), ptr, stride,
                                       float(eps), int(relu), n, c, plane, y.data_ptr(), saved[0].data_ptr(),
                                       saved[1].data_ptr(), _ptr(scratch), _ptr(stats_in), stats_slices, G, stream)


def sum_groups(dgb, weight, bias):
    """(dgamma, dbeta) of parameters shared by the G groups from dgb [G, 2, C]: views of ONE [2, C] tensor that the
    pass's deferred slab reduction fills (conv._pending: G slabs of 2 C numbers, fixed order), or an immediate sum"""
    G, _, c = dgb.shape
    if G == 1:
        return dgb[0, 0], dgb[0, 1]
    out = torch.empty((2, c), dtype=torch.float32, device=dgb.device)
    if _conv._may_defer(weight, bias):
        torch.autograd.Variable._execution_engine.queue_callback(_conv._flush_pending)
        _conv._pending.append((dgb, out, G, 1))
        return out[0], out[1]
    torch.sum(dgb, dim=0, out=out)
    return out[0], out[1]


def replay_running_stats(log, entry_stride, n_entries, momentum, running_mean, running_var, stream):
    "running_mean / running_var advanced by n_entries logged batches (log: data pointer of the first entry's slot)"
    err = _hip.lib().sgmcmc_bn_running_replay(log, entry_stride, n_entries, float(momentum), running_mean.data_ptr(),
                                              running_var.data_ptr(), running_mean.numel(), stream)
    if err:
        _hip.check(err, "sgmcmc_bn_running_replay")


def replay_running_stats_many(layers, log_all, n_entries, stream):
    """every layer's running statistics advanced by the first ``n_entries`` rows of ``log_all`` ([capacity, n_layers,
    cmax, 2] float64: row j, column i = layer i's batch mean / unbiased variance of logged minibatch j) -- one launch"""
    n = len(layers)
    arr = (_hip.BnReplayLayer * n)()
    for i, (a, m) in enumerate(zip(arr, layers)):
        a.log, a.running_mean, a.running_var = log_all[0, i].data_ptr(), m.running_mean.data_ptr(), m.running_var.data_ptr()
        a.momentum, a.channels = float(m.momentum), m.num_features
    err = _hip.lib().sgmcmc_bn_running_replay_many(ctypes.cast(arr, ctypes.c_void_p), n, log_all.stride(0), n_entries, stream)
    if err:
        _hip.check(err, "sgmcmc_bn_running_replay_many")


class _BNTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, residual, running_mean, running_var, momentum, eps, relu, stats_in,
                res_y=None, res_saved=None):
        lib = _hip.lib()
        x = x.contiguous()
        if residual is not None:
            residual = residual.contiguous()
        n, c, plane = x.shape[0], x.shape[1], x.shape[2] * x.shape[3]
        G = ctx.groups = _groups_of(x)
        if G > 1:
            _conv._note_use(weight, bias)      # (their groups' gradients may join the pass's deferred reduction)
        y = torch.empty_like(x)
        stats = torch.empty((2, G * c), dtype=torch.float32, device=x.device)
        scratch = None
        if stats_in is None:
            scratch = torch.empty(lib.sgmcmc_bn_scratch_doubles(n, c, plane, G), dtype=torch.float64,
                                  device=x.device)
        elif stats_in.dtype != torch.float64 or stats_in.dim() != 3 or stats_in.shape[0] != c \
                or stats_in.shape[2] != 2 or not stats_in.is_contiguous():
            raise ValueError("stats must be a contiguous float64 [channels][slices][2] tensor")
        err = train_fwd(lib, x, residual, weight, bias, running_mean, running_var, momentum, eps, relu, n, c, plane, y,
                        stats, scratch, stats_in, 0 if stats_in is None else stats_in.shape[1],
                        torch.cuda.current_stream().cuda_stream, G)
        if err:
            _hip.check(err, "sgmcmc_bn_train_fwd")
        ctx.save_for_backward(x, weight, y if relu else None, stats, res_y, res_saved, bias)
        ctx.relu, ctx.has_residual = bool(relu), residual is not None
        ctx.mark_non_differentiable(stats)
        ctx.set_materialize_grads(False)
        return y, stats

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy, *_):
        lib = _hip.lib()
        x, weight, y, stats, res_y, res_saved, bias = ctx.saved_tensors
        if dy is None:
            return (None,) * 12
        # the launch that produced dy may have left this BatchNorm's channel sums with it (bnlink)
        up = _bnlink.sums_of(dy)
        dy = dy.contiguous()
        n, c, plane = x.shape[0], x.shape[1], x.shape[2] * x.shape[3]
        G = ctx.groups
        dx = torch.empty_like(x)
        want_res = ctx.has_residual and ctx.needs_input_grad[3]
        if want_res and not ctx.relu:
            dres = dy                    # no mask: the residual's gradient IS dy
        else:
            dres = torch.empty_like(x) if want_res else None
        dgb = torch.empty((G, 2, c), dtype=torch.float32, device=x.device)
        st = torch.cuda.current_stream().cuda_stream
        # the residual is a ReLU-less BatchNorm's output (the down-sampling shortcut): its sums ride in the dx launch
        rsums = res_y is not None and want_res and ctx.relu
        if up is not None or rsums:
            if up is None:
                sums = torch.empty(lib.sgmcmc_bn_scratch_doubles(n, c, plane, G), dtype=torch.float64, device=x.device)
                n_sums = ctypes.c_int(0)
                err = lib.sgmcmc_bn_bwd_sums(dy.data_ptr(), _ptr(y), x.data_ptr(), stats[0].data_ptr(),
                                             stats[1].data_ptr(), sums.data_ptr(), ctypes.byref(n_sums), n, c, plane, G, st)
                if err:
                    _hip.check(err, "sgmcmc_bn_bwd_sums")
                up = (sums, n_sums.value, False)
            # dy stored as dy * [y > 0] by the launch that produced it (bnlink.PREMASK): no mask to apply, y is not
            # read, and the residual's gradient IS dy
            masked = ctx.relu and up[2]
            if masked and want_res:
                dres = dy
            R = None
            if rsums:
                r_part = torch.empty(lib.sgmcmc_bn_scratch_doubles(n, c, plane, G), dtype=torch.float64, device=x.device)
                R = _hip.BnResidualSums(y=res_y.data_ptr(), mean=res_saved[0].data_ptr(),
                                        invstd=res_saved[1].data_ptr(), partial=r_part.data_ptr())
            err = lib.sgmcmc_bn_bwd_dx(dy.data_ptr(), 0 if masked else _ptr(y), x.data_ptr(), weight.data_ptr(),
                                       stats[0].data_ptr(), stats[1].data_ptr(), int(ctx.relu and not masked), n, c, plane,
                                       up[0].data_ptr(), up[1], dx.data_ptr(), 0 if dres is dy else _ptr(dres),
                                       dgb.data_ptr(), None if R is None else ctypes.byref(R), G, st)
            if err:
                _hip.check(err, "sgmcmc_bn_bwd_dx")
            if rsums:
                _bnlink.tag_gradient(dres, r_part, r_part.numel() // (2 * c))
            return (dx, *sum_groups(dgb, weight, bias), dres) + (None,) * 8
        scratch = torch.empty(lib.sgmcmc_bn_scratch_doubles(n, c, plane, G), dtype=torch.float64, device=x.device)
        err = lib.sgmcmc_bn_train_bwd(dy.data_ptr(), _ptr(y), x.data_ptr(), weight.data_ptr(),
                                      stats[0].data_ptr(), stats[1].data_ptr(), int(ctx.relu), n, c, plane,
                                      dx.data_ptr(), 0 if dres is dy else _ptr(dres), dgb.data_ptr(), scratch.data_ptr(),
                                      G, torch.cuda.current_stream().cuda_stream)
        if err:
            _hip.check(err, "sgmcmc_bn_train_bwd")
        return (dx, *sum_groups(dgb, weight, bias), dres) + (None,) * 8


# ---- relu(BN(x) + BN_s(r)): the last BatchNorm of a down-sampling block with the shortcut's BatchNorm applied on the fly
# (csrc/bn_hip.inc apply_dual_kernel).  ``DUAL = False`` restores the two operators (the tests' cross-check).
DUAL = True


def dual_supported(x, stats, r, r_stats, bn, bn_s):
    "both layers in training mode with running statistics, float32 affine, statistics from their convolutions' epilogues"
    if not (DUAL and stats is not None and r_stats is not None and r.shape == x.shape and r.dtype == x.dtype):
        return False
    for layer in (bn, bn_s):
        if not (layer.training and layer.track_running_stats and layer.momentum is not None and layer.affine
                and supported(x, layer.weight, layer.bias, True, layer.momentum)):
            return False
    return True


class _BNTrainDual(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stats_in, r, r_weight, r_bias, r_stats, main, short):
        "main / short: (running_mean, running_var, momentum, eps) of the two layers"
        lib = _hip.lib()
        x, r = x.contiguous(), r.contiguous()
        n, c, plane = x.shape[0], x.shape[1], x.shape[2] * x.shape[3]
        G = ctx.groups = _groups_of(x)
        if G > 1:
            _conv._note_use(weight, bias, r_weight, r_bias)
        y = torch.empty_like(x)
        saved = torch.empty((2, G * c), dtype=torch.float32, device=x.device)
        r_saved = torch.empty((2, G * c), dtype=torch.float32, device=x.device)
        slot, r_slot = log_slot(main[0]), log_slot(short[0])
        if G > 1 and (slot is None or r_slot is None):
            raise LogModeUnsupported("bn.grouped(G > 1) advances running statistics through the log only")
        (lp, ls), (rp, rs) = ((0, 0) if t is None else _slot_ptr_stride(t, G) for t in (slot, r_slot))
        R = _hip.BnDual(r=r.data_ptr(), gamma=r_weight.data_ptr(), beta=r_bias.data_ptr(), partial=r_stats.data_ptr(),
                        n_partials=r_stats.shape[1], reserved=0, eps=float(short[3]), momentum=float(short[2]),
                        save_mean=r_saved[0].data_ptr(), save_invstd=r_saved[1].data_ptr(),
                        running_mean=0 if r_slot is not None else short[0].data_ptr(),
                        running_var=0 if r_slot is not None else short[1].data_ptr(), stat_log=rp, log_stride=rs)
        err = lib.sgmcmc_bn_train_fwd_dual(x.data_ptr(), weight.data_ptr(), bias.data_ptr(),
                                           0 if slot is not None else main[0].data_ptr(),
                                           0 if slot is not None else main[1].data_ptr(), float(main[2]), float(main[3]),
                                           n, c, plane, y.data_ptr(), saved[0].data_ptr(), saved[1].data_ptr(),
                                           stats_in.data_ptr(), stats_in.shape[1], lp, ls, ctypes.byref(R), G,
                                           torch.cuda.current_stream().cuda_stream)
        if err:
            _hip.check(err, "sgmcmc_bn_train_fwd_dual")
        ctx.save_for_backward(x, weight, y, saved, r, r_weight, r_saved, bias, r_bias)
        ctx.mark_non_differentiable(saved)
        ctx.set_materialize_grads(False)
        return y, saved

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy, *_):
        "the two layers' backward as _BNTrain runs them: dx with the shortcut's sums riding along, then the shortcut's dx"
        lib = _hip.lib()
        x, weight, y, saved, r, r_weight, r_saved, bias, r_bias = ctx.saved_tensors
        if dy is None:
            return (None,) * 10
        up = _bnlink.sums_of(dy)
        dy = dy.contiguous()
        n, c, plane = x.shape[0], x.shape[1], x.shape[2] * x.shape[3]
        G = ctx.groups
        st = torch.cuda.current_stream().cuda_stream
        if up is None:
            sums = torch.empty(lib.sgmcmc_bn_scratch_doubles(n, c, plane, G), dtype=torch.float64, device=x.device)
            n_sums = ctypes.c_int(0)
            err = lib.sgmcmc_bn_bwd_sums(dy.data_ptr(), y.data_ptr(), x.data_ptr(), saved[0].data_ptr(),
                                         saved[1].data_ptr(), sums.data_ptr(), ctypes.byref(n_sums), n, c, plane, G, st)
            if err:
                _hip.check(err, "sgmcmc_bn_bwd_sums")
            up = (sums, n_sums.value, False)
        masked = up[2]         # dy stored as dy * [y > 0] by its producer (bnlink.PREMASK): dz IS dy
        dx, dr = torch.empty_like(x), torch.empty_like(x)
        dz = dy if masked else torch.empty_like(x)
        dgb = torch.empty((2, G, 2, c), dtype=torch.float32, device=x.device)
        r_part = torch.empty(lib.sgmcmc_bn_scratch_doubles(n, c, plane, G), dtype=torch.float64, device=x.device)
        R = _hip.BnResidualSums(y=r.data_ptr(), mean=r_saved[0].data_ptr(), invstd=r_saved[1].data_ptr(),
                                partial=r_part.data_ptr())
        err = lib.sgmcmc_bn_bwd_dx(dy.data_ptr(), 0 if masked else y.data_ptr(), x.data_ptr(), weight.data_ptr(),
                                   saved[0].data_ptr(), saved[1].data_ptr(), 0 if masked else 1, n, c, plane,
                                   up[0].data_ptr(), up[1], dx.data_ptr(), 0 if masked else dz.data_ptr(),
                                   dgb[0].data_ptr(), ctypes.byref(R), G, st)
        if err:
            _hip.check(err, "sgmcmc_bn_bwd_dx")
        err = lib.sgmcmc_bn_bwd_dx(dz.data_ptr(), 0, r.data_ptr(), r_weight.data_ptr(), r_saved[0].data_ptr(),
                                   r_saved[1].data_ptr(), 0, n, c, plane, r_part.data_ptr(), r_part.numel() // (2 * c),
                                   dr.data_ptr(), 0, dgb[1].data_ptr(), None, G, st)
        if err:
            _hip.check(err, "sgmcmc_bn_bwd_dx(shortcut)")
        dg, db = sum_groups(dgb[0], weight, bias)
        rdg, rdb = sum_groups(dgb[1], r_weight, r_bias)
        return dx, dg, db, None, dr, rdg, rdb, None, None, None


def bn_train_dual(x, stats, bn, r, r_stats, bn_s):
    "relu(bn(x) + bn_s(r)) for two training-mode BatchNorm2d modules that pass ``dual_supported``"
    out, saved = _BNTrainDual.apply(x, bn.weight, bn.bias, stats, r, bn_s.weight, bn_s.bias, r_stats,
                                    (bn.running_mean, bn.running_var, bn.momentum, bn.eps),
                                    (bn_s.running_mean, bn_s.running_var, bn_s.momentum, bn_s.eps))
    _bnlink.tag_output(out, x, saved)     # a convolution that consumes `out` can produce this BatchNorm's backward sums
    return out


def bn_train(x, weight, bias, running_mean, running_var, momentum, eps, residual=None, relu=False, stats=None):
    """``stats``: per-slice partial (sum, sum of squared deviations from the slice mean) of x over equal parts of (N, H, W) per channel, float64
    [channels][slices][2], if the producer of x already has them (``conv.conv3x3(..., want_stats=True)``);
    the statistics pass over x is skipped then."""
    res_y, res_saved = _bnlink.linear_source_of(residual) if relu else (None, None)
    out, saved = _BNTrain.apply(x, weight, bias, residual, running_mean, running_var, momentum, eps, relu, stats,
                                res_y, res_saved)
    if relu:      # a convolution that consumes `out` can produce this BatchNorm's backward sums (bnlink)
        _bnlink.tag_output(out, x, saved)
    elif residual is None:
        _bnlink.tag_linear_output(out, x, saved)
    return out
