"""``maxpool2x2(relu(x + bias))`` of the convolutional classifier as one HIP operator, forward and
backward (``csrc/pool_hip.inc``; C ABI ``sgmcmc_bias_relu_pool_fwd/_bwd``): what follows every
convolution of bnn_priors/models/conv_nets.py:44-56.  The convolution then runs without its bias; the
bias gradient comes out of the same backward pass (per-block partials, fixed-order reduction).
"""
import contextlib
import os

import torch

from . import _hip
from . import bnlink as _bnlink
from . import conv as _conv

ENABLED = True          # (module attribute, not an environment switch)


def supported(x, bias):
    return (ENABLED and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[0] > 0
            and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0
            and (bias is None or (bias.dtype == torch.float32 and bias.shape == (x.shape[1],))))


class _BiasReluPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bias):
        if bias is not None:
            _conv._note_use(bias)
        x = x.contiguous()
        n, c, h, w = x.shape
        y = torch.empty((n, c, h // 2, w // 2), dtype=torch.float32, device=x.device)
        err = _hip.lib().sgmcmc_bias_relu_pool_fwd(x.data_ptr(), 0 if bias is None else bias.data_ptr(),
                                                   y.data_ptr(), n, c, h, w, _conv._stream())
        if err:
            _hip.check(err, "sgmcmc_bias_relu_pool_fwd")
        ctx.save_for_backward(x, bias)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        import ctypes
        lib = _hip.lib()
        x, bias = ctx.saved_tensors
        dy = dy.contiguous()
        n, c, h, w = x.shape
        dx = torch.empty_like(x)
        want_b = bias is not None and ctx.needs_input_grad[1]
        part = db = None
        if want_b:
            slices = lib.sgmcmc_pool_slices(n, c, h, w)
            part = torch.empty((slices, c), dtype=torch.float32, device=x.device)
            db = torch.empty_like(bias)
        err = lib.sgmcmc_bias_relu_pool_bwd(x.data_ptr(), 0 if bias is None else bias.data_ptr(), dy.data_ptr(),
                                            dx.data_ptr(), 0 if part is None else part.data_ptr(), n, c, h, w,
                                            _conv._stream())
        if err:
            _hip.check(err, "sgmcmc_bias_relu_pool_bwd")
        if not want_b:
            return dx, None
        if _conv._may_defer(bias):
            # summed with the convolutions' weight-gradient slabs at the end of the backward pass
            torch.autograd.Variable._execution_engine.queue_callback(_conv._flush_pending)
            _conv._pending.append((part, db, part.shape[0], 1))
            return dx, db.view(db.shape)
        job = (_hip.ReduceJob * 1)()
        job[0].part, job[0].out, job[0].n_slabs, job[0].numel = part.data_ptr(), db.data_ptr(), part.shape[0], c
        err = lib.sgmcmc_wrw_reduce_many(ctypes.cast(job, ctypes.c_void_p), 1, _conv._stream())
        if err:
            _hip.check(err, "sgmcmc_wrw_reduce_many")
        return dx, db


def bias_relu_pool(x, bias=None):
    "max_pool2d(relu(x + bias[None, :, None, None]), 2) for NCHW float32 tensors with even height and width"
    return _BiasReluPool.apply(x, bias)


# ------------------------------------------------------------------ global average pool -> Linear
def head_supported(h, weight, bias):
    "AvgPool2d over the whole map -> Flatten -> Linear as one operator (csrc/pool_hip.inc, namespace head)"
    if not (ENABLED and h.is_cuda and h.dtype == torch.float32 and h.dim() == 4 and h.shape[0] > 0):
        return False
    c, plane = h.shape[1], h.shape[2] * h.shape[3]
    return (c <= 64 and plane % 16 == 0 and weight.dim() == 2 and weight.shape[1] == c and weight.shape[0] <= 16
            and weight.dtype == torch.float32 and (bias is None or bias.shape == (weight.shape[0],)))


def _reduce_rows(slabs, out, defer):
    "out <- sum over rows of slabs [n][numel]: with the pass's other slabs if nothing reads `out` before"
    import ctypes
    if defer:
        torch.autograd.Variable._execution_engine.queue_callback(_conv._flush_pending)
        _conv._pending.append((slabs, out, slabs.shape[0], 1))
        return out.view(out.shape)
    job = (_hip.ReduceJob * 1)()
    job[0].part, job[0].out, job[0].n_slabs, job[0].numel = slabs.data_ptr(), out.data_ptr(), slabs.shape[0], out.numel()
    err = _hip.lib().sgmcmc_wrw_reduce_many(ctypes.cast(job, ctypes.c_void_p), 1, _conv._stream())
    if err:
        _hip.check(err, "sgmcmc_wrw_reduce_many")
    return out


# ---- head + loss + both backward passes in one launch (csrc/pool_hip.inc, head::loss_kernel) ----------------------
# A caller that is about to take ``cross_entropy_backward(model.net(x), y, ...)`` announces the labels first:
#
#     with pool.head_loss(y, reduction, divide_by):
#         f = model.net(x)
#     loss = pool.cross_entropy_backward(f, y, reduction, divide_by)
#
# A trunk that ends in the fused head (``pool_linear``) then computes, in the head's ONE launch, the logits, the loss
# rows, d loss / d logits and the head's whole backward (dh, the rows of dW / db, the last BatchNorm's backward sums) --
# everything between the last activation and its gradient is per image.  ``cross_entropy_backward`` recognises the
# tagged logits and seeds autograd with the stashed gradient; ``_PoolLinear.backward`` hands out what the launch
# already produced.  Three launches at the launch floor become one; any other use of the logits (another loss, a
# temperature, other labels) takes the separate kernels as before.
FUSED_HEAD = True
_head = {"spec": None, "last": None, "counted": False}


@contextlib.contextmanager
def head_loss(y, reduction="mean", divide_by=None, head=None):
    """``head``: the module whose launch may take the likelihood (``head_of(model)``: the net's LAST linear layer).  A
    narrow hidden ``Linear`` (<= 16 outputs, a user's ``width``) passes ``linear_supported`` too; without this it would
    compute a softmax and a loss row nobody reads.  None: any supported layer (direct callers of ``pool.linear``)."""
    old = _head["spec"], _head.get("head")
    _head["spec"], _head["head"] = (y, reduction, divide_by), head
    try:
        yield
    finally:
        _head["spec"], _head["head"] = old


def head_of(model):
    "the last module of ``model.net`` with a ``weight_prior`` and 2-D weights (models/nets.py Linear), cached on the net"
    net = getattr(model, "net", model)
    hit = net.__dict__.get("_sgmcmc_head")
    if hit is None:
        last = None
        for m in net.modules():
            wp = getattr(m, "weight_prior", None)
            if wp is not None and getattr(getattr(wp, "p", None), "dim", lambda: 0)() == 2:
                last = m
        hit = net.__dict__["_sgmcmc_head"] = (last,)
    return hit[0]


def _grad_scale(rows, reduction, divide_by):
    "(scale of the loss, scale of d loss / d logits) as cross_entropy_backward forms them (float32 arithmetic)"
    import numpy as np
    scale = 1.0 / rows if reduction == "mean" else 1.0
    seed = np.float32(1.0) if divide_by is None else np.float32(1.0) / np.float32(divide_by)
    return scale, float(seed * np.float32(scale))


class _PoolLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, weight, bias, src_y=None, src_saved=None, counters=None):
        _conv._note_use(*((weight,) if bias is None else (weight, bias)))
        h, weight = h.contiguous(), weight.contiguous()
        n, c, plane, k = h.shape[0], h.shape[1], h.shape[2] * h.shape[3], weight.shape[0]
        pooled = torch.empty((n, c), dtype=torch.float32, device=h.device)
        logits = torch.empty((n, k), dtype=torch.float32, device=h.device)
        ctx.fused = None
        _head["last"], _head["counted"] = None, False
        spec = _head["spec"]
        if (FUSED_HEAD and spec is not None and ctx.needs_input_grad[0] and n <= 1024 and spec[0].is_cuda
                and spec[0].dtype == torch.int64 and spec[0].dim() == 1 and spec[0].shape[0] == n
                and spec[1] in ("mean", "sum")):
            y, reduction, divide_by = spec
            y = y.contiguous()
            dev = h.device
            want_w, want_b = ctx.needs_input_grad[1], bias is not None and ctx.needs_input_grad[2]
            d = torch.empty((n, k), dtype=torch.float32, device=dev)
            loss_rows = torch.empty((n,), dtype=torch.float32, device=dev)
            dh = torch.empty_like(h)
            sw = torch.empty((n, k * c), dtype=torch.float32, device=dev) if want_w else None
            sb = torch.empty((n, k), dtype=torch.float32, device=dev) if want_b else None
            partial = torch.empty((c, n, 2), dtype=torch.float64, device=dev) if src_y is not None else None
            _, gscale = _grad_scale(n, reduction, divide_by)
            p = lambda t: 0 if t is None else t.data_ptr()
            err = _hip.lib().sgmcmc_pool_linear_loss(
                h.data_ptr(), weight.data_ptr(), p(bias), y.data_ptr(), pooled.data_ptr(), logits.data_ptr(), d.data_ptr(),
                loss_rows.data_ptr(), dh.data_ptr(), p(sw), p(sb), p(src_y), h.data_ptr() if src_y is not None else 0,
                p(src_saved[0]) if src_y is not None else 0, p(src_saved[1]) if src_y is not None else 0, p(partial),
                _conv._group_imgs(n) if src_y is not None else 0,
                p(counters), 0 if counters is None else counters.numel(), n, c, plane, k, gscale, _conv._stream())
            if err:
                _hip.check(err, "sgmcmc_pool_linear_loss")
            ctx.fused = (d, dh, sw, sb, partial)
            _head["last"] = (spec, d, loss_rows)
            _head["counted"] = counters is not None
            ctx.save_for_backward(pooled, weight, bias, src_y, src_saved, h if src_y is not None else None)
            ctx.h_shape = tuple(h.shape)
            return logits
        err = _hip.lib().sgmcmc_pool_linear_fwd(h.data_ptr(), weight.data_ptr(), 0 if bias is None else bias.data_ptr(),
                                                pooled.data_ptr(), logits.data_ptr(), n, c, plane, k, _conv._stream())
        if err:
            _hip.check(err, "sgmcmc_pool_linear_fwd")
        ctx.save_for_backward(pooled, weight, bias, src_y, src_saved, h if src_y is not None else None)
        ctx.h_shape = tuple(h.shape)
        return logits

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dlogits):
        pooled, weight, bias, src_y, src_saved, h = ctx.saved_tensors
        if ctx.fused is not None and dlogits.data_ptr() == ctx.fused[0].data_ptr():
            # the gradient the forward launch already propagated (head_loss): nothing left to launch
            _, dh, sw, sb, partial = ctx.fused
            ctx.fused = None
            if partial is not None:
                _bnlink.tag_gradient(dh, partial, ctx.h_shape[0])
            dw = _reduce_rows(sw, torch.empty_like(weight), _conv._may_defer(weight)) if sw is not None else None
            db = _reduce_rows(sb, torch.empty_like(bias), _conv._may_defer(bias)) if sb is not None else None
            return dh, dw, db, None, None, None
        ctx.fused = None
        dlogits = dlogits.contiguous()
        n, c, hh, ww = ctx.h_shape
        k = weight.shape[0]
        dev = pooled.device
        dh = torch.empty(ctx.h_shape, dtype=torch.float32, device=dev)
        want_w, want_b = ctx.needs_input_grad[1], bias is not None and ctx.needs_input_grad[2]
        sw = torch.empty((n, k * c), dtype=torch.float32, device=dev) if want_w else None
        sb = torch.empty((n, k), dtype=torch.float32, device=dev) if want_b else None
        if src_y is not None:      # h came out of a BatchNorm + ReLU: its backward sums ride in this launch (bnlink)
            partial = torch.empty((c, n, 2), dtype=torch.float64, device=dev)
            err = _hip.lib().sgmcmc_pool_linear_bwd_sums(
                dlogits.data_ptr(), pooled.data_ptr(), weight.data_ptr(), dh.data_ptr(),
                0 if sw is None else sw.data_ptr(), 0 if sb is None else sb.data_ptr(), src_y.data_ptr(), h.data_ptr(),
                src_saved[0].data_ptr(), src_saved[1].data_ptr(), partial.data_ptr(), _conv._group_imgs(n), n, c, hh * ww,
                k, _conv._stream())
            _bnlink.tag_gradient(dh, partial, n)
        else:
            err = _hip.lib().sgmcmc_pool_linear_bwd(dlogits.data_ptr(), pooled.data_ptr(), weight.data_ptr(), dh.data_ptr(),
                                                    0 if sw is None else sw.data_ptr(), 0 if sb is None else sb.data_ptr(),
                                                    n, c, hh * ww, k, _conv._stream())
        if err:
            _hip.check(err, "sgmcmc_pool_linear_bwd")
        dw = db = None
        if want_w:
            dw = _reduce_rows(sw, torch.empty_like(weight), _conv._may_defer(weight))
        if want_b:
            db = _reduce_rows(sb, torch.empty_like(bias), _conv._may_defer(bias))
        return dh, dw, db, None, None, None


def pool_linear(h, weight, bias=None, counters=None):
    """linear(h.mean(dim=(2, 3)), weight, bias) for NCHW float32 h with <= 64 channels and <= 16 outputs.
    ``counters``: a contiguous int64 tensor that gets + 1 along the way (the trunk's BatchNorm batch counters) --
    inside the head's launch when it is the fused one (``head_loss``), by one ATen launch otherwise."""
    src_y, src_saved = (None, None)
    if h.requires_grad and h.shape[2] * h.shape[3] == 64:
        src_y, src_saved = _bnlink.source_of(h)
    logits = _PoolLinear.apply(h, weight, bias, src_y, src_saved, counters)
    if _head["last"] is not None:
        logits._sgmcmc_head_loss = _head["last"] + (logits._version,)
        _head["last"] = None
    if counters is not None and not _head["counted"]:
        with torch.no_grad():
            counters.add_(1)
    _head["counted"] = False
    return logits


# ------------------------------------------------------------------ narrow linear layer
def linear_supported(x, weight, bias):
    "x [N, J] float32 on the GPU, weight [K <= 16, J]: the convolutional classifier's head (csrc/pool_hip.inc, lin)"
    return (ENABLED and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[0] > 0
            and weight.dim() == 2 and weight.shape[1] == x.shape[1] and weight.shape[0] <= 16
            and weight.dtype == torch.float32
            and (bias is None or (bias.dtype == torch.float32 and bias.shape == (weight.shape[0],))))


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, owner=None):
        _conv._note_use(weight)
        x, weight = x.contiguous(), weight.contiguous()
        n, k = x.shape[0], weight.shape[0]
        y = torch.empty((n, k), dtype=torch.float32, device=x.device)
        spec = _head["spec"]
        _head["last"] = None
        is_head = _head.get("head") is None or _head["head"] is owner
        if (FUSED_HEAD and spec is not None and is_head and n <= 1024 and spec[0].is_cuda and spec[0].dtype == torch.int64
                and spec[0].dim() == 1 and spec[0].shape[0] == n and spec[1] in ("mean", "sum")
                and any(ctx.needs_input_grad)):
            # ``head_loss``: the likelihood's forward + seed in the last layer's launch (the backward stays lin::bwd)
            lab, reduction, divide_by = spec
            d = torch.empty_like(y)
            loss_rows = torch.empty((n,), dtype=torch.float32, device=x.device)
            _, gscale = _grad_scale(n, reduction, divide_by)
            err = _hip.lib().sgmcmc_linear_fwd_loss(x.data_ptr(), weight.data_ptr(), 0 if bias is None else bias.data_ptr(),
                                                    lab.contiguous().data_ptr(), y.data_ptr(), d.data_ptr(),
                                                    loss_rows.data_ptr(), n, x.shape[1], k, gscale, _conv._stream())
            if err:
                _hip.check(err, "sgmcmc_linear_fwd_loss")
            _head["last"] = (spec, d, loss_rows)
            ctx.save_for_backward(x, weight)
            ctx.has_bias = bias is not None
            return y
        err = _hip.lib().sgmcmc_linear_fwd(x.data_ptr(), weight.data_ptr(), 0 if bias is None else bias.data_ptr(),
                                           y.data_ptr(), x.shape[0], x.shape[1], weight.shape[0], _conv._stream())
        if err:
            _hip.check(err, "sgmcmc_linear_fwd")
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        lib = _hip.lib()
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        slabs = None
        if ctx.needs_input_grad[1]:
            slabs = torch.empty((lib.sgmcmc_linear_row_groups(x.shape[0]),) + tuple(weight.shape), dtype=torch.float32,
                                device=x.device)
        db = (torch.empty(weight.shape[0], dtype=torch.float32, device=x.device)
              if ctx.has_bias and ctx.needs_input_grad[2] else None)
        p = lambda t: 0 if t is None else t.data_ptr()
        err = lib.sgmcmc_linear_bwd(x.data_ptr(), weight.data_ptr(), dy.data_ptr(), p(dx), p(slabs), p(db),
                                    x.shape[0], x.shape[1], weight.shape[0], _conv._stream())
        if err:
            _hip.check(err, "sgmcmc_linear_bwd")
        dw = None
        if slabs is not None:     # the row groups' slabs: summed with the pass's other slabs when nothing reads dw earlier
            dw = _reduce_rows(slabs.view(slabs.shape[0], -1), torch.empty_like(weight), _conv._may_defer(weight))
        return dx, dw, db, None


def linear(x, weight, bias=None, owner=None):
    "F.linear(x, weight, bias) for 2-D float32 x and at most 16 output features; ``owner``: the calling module (``head_loss``)"
    y = _Linear.apply(x, weight, bias, owner)
    if _head["last"] is not None:          # (``head_loss`` was active: the launch also produced the likelihood's seed)
        y._sgmcmc_head_loss = _head["last"] + (y._version,)
        _head["last"] = None
    return y


# ------------------------------------------------------------------ softmax cross-entropy
def xent_supported(logits, y):
    return (ENABLED and logits.is_cuda and logits.dtype == torch.float32 and logits.dim() == 2
            and 0 < logits.shape[0] <= 1024 and 0 < logits.shape[1] <= 16 and y.dtype == torch.int64
            and y.shape == (logits.shape[0],))


class _SoftmaxXent(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, y, scale):
        logits, y = logits.contiguous(), y.contiguous()
        b, k = logits.shape
        probs = torch.empty_like(logits)
        loss = torch.empty((), dtype=torch.float32, device=logits.device)
        err = _hip.lib().sgmcmc_softmax_xent_fwd(logits.data_ptr(), y.data_ptr(), probs.data_ptr(), loss.data_ptr(),
                                                 b, k, float(scale), _conv._stream())
        if err:
            _hip.check(err, "sgmcmc_softmax_xent_fwd")
        ctx.save_for_backward(probs, y)
        ctx.scale = float(scale)
        return loss

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        probs, y = ctx.saved_tensors
        g = g.contiguous().float()
        d = torch.empty_like(probs)
        err = _hip.lib().sgmcmc_softmax_xent_bwd(probs.data_ptr(), y.data_ptr(), g.data_ptr(), d.data_ptr(),
                                                 probs.shape[0], probs.shape[1], ctx.scale, _conv._stream())
        if err:
            _hip.check(err, "sgmcmc_softmax_xent_bwd")
        return d, None, None


def cross_entropy_backward(logits, y, reduction="mean", divide_by=None, want_loss=True):
    """``loss = F.cross_entropy(logits, y, reduction=...) [/ divide_by]; loss.backward()`` with the likelihood's
    forward and backward in ONE launch: the kernel leaves the loss and d loss / d logits, and autograd is seeded
    with the latter (``logits.backward(dlogits)``).  Same bits as ``cross_entropy(...)`` followed by
    ``backward()``.  Returns the detached loss (None with ``want_loss=False``: a captured step that logs nothing does
    not spend a launch on summing the loss rows of the fused head)."""
    if reduction not in ("mean", "sum"):
        raise ValueError("reduction must be 'mean' or 'sum'")
    st = getattr(logits, "_sgmcmc_head_loss", None)
    if st is not None:
        (sy, sred, sdiv), d, loss_rows, version = st
        if sy is y and sred == reduction and sdiv == divide_by and logits._version == version and logits.requires_grad:
            # the head's launch already holds d loss / d logits and everything behind it (``head_loss``)
            logits.backward(d)
            if not want_loss:
                return None
            scale, _ = _grad_scale(logits.shape[0], reduction, divide_by)
            from . import bn as _bn
            G = _bn.groups()
            if G > 1 and reduction == "sum" and logits.shape[0] % G == 0:
                # several minibatches in one launch (bn.grouped): every minibatch's rows are summed as a pass over that
                # minibatch alone sums them, the minibatches' sums added in double -- what the caller's accumulator does
                loss = loss_rows.view(G, -1).sum(dim=1).mul_(scale)
                loss = loss if divide_by is None else loss / divide_by
                return loss.double().sum()
            loss = loss_rows.sum() * scale
            return loss if divide_by is None else loss / divide_by
    if not (xent_supported(logits, y) and logits.requires_grad):
        loss = cross_entropy(logits, y, reduction)
        if divide_by is not None:
            loss = loss / divide_by
        loss.backward()
        return loss.detach()
    src, yc = logits.detach().contiguous(), y.contiguous()
    b, k = src.shape
    scale = 1.0 / b if reduction == "mean" else 1.0
    import numpy as np
    # the gradient autograd would hand the kernel: ones for the loss itself, (1 / divide_by) in float32 after a division
    seed = np.float32(1.0) if divide_by is None else np.float32(1.0) / np.float32(divide_by)
    gscale = float(seed * np.float32(scale))
    d = torch.empty_like(src)
    loss = torch.empty((), dtype=torch.float32, device=src.device)
    err = _hip.lib().sgmcmc_softmax_xent_fwd_grad(src.data_ptr(), yc.data_ptr(), loss.data_ptr(), d.data_ptr(), b, k,
                                                  float(scale), gscale, _conv._stream())
    if err:
        _hip.check(err, "sgmcmc_softmax_xent_fwd_grad")
    logits.backward(d)
    return loss if divide_by is None else loss / divide_by


def cross_entropy(logits, y, reduction="mean"):
    """``F.cross_entropy(logits, y, reduction=...)`` ("mean" or "sum") -- one launch each way for float32
    logits of up to 1024 rows x 16 classes on the GPU, ATen otherwise"""
    if reduction not in ("mean", "sum"):
        raise ValueError("reduction must be 'mean' or 'sum'")
    if xent_supported(logits, y):
        return _SoftmaxXent.apply(logits, y, 1.0 / logits.shape[0] if reduction == "mean" else 1.0)
    return torch.nn.functional.cross_entropy(logits, y, reduction=reduction)
