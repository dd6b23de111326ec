"""The ResNet trunk's 3x3 convolutions on hand-written fp32-MFMA kernels
(``csrc/conv_hip.inc``; C ABI ``sgmcmc_conv3x3`` / ``sgmcmc_conv3x3_wrw``), as an autograd
function: forward, data gradient (the same kernel on transposed + flipped weights) and weight
gradient (partial slabs + a fixed-order reduction: reproducible bit for bit).

``conv3x3(x, w)`` equals ``F.conv2d(x, w, None, 1, 1)`` up to fp32 summation order for the shapes of
``SHAPES``; ``supported(...)`` tells the layer whether a call qualifies -- anything else (strided
and 1x1 convolutions, the 3-channel stem, other dtypes, CPU tensors) stays on ATen/MIOpen.
Reference: the convolutions of bnn_priors/models/google_resnet.py:34-43 inside the gradient
evaluation of inference.py:215-223.
"""
import collections
import contextlib
import ctypes
import os
import warnings
import weakref

import torch

from . import _hip
from . import bnlink as _bnlink

# Layers of a GPU model that did NOT take one of this package's kernels (a shape off the tables, a switch set to 0)
# are counted here per (operator, per-sample shape); SGMCMC_STRICT=1 turns such a library dispatch into an error --
# for production runs of the BASELINE configurations, whose steps contain no library compute kernel.
STRICT = os.environ.get("SGMCMC_STRICT", "0") == "1"
LIBRARY_CALLS = collections.Counter()


def library_path(op, x):
    if x.is_cuda and torch.is_grad_enabled():      # (gradient evaluations only: evaluation passes are not the hot path)
        key = (op, tuple(x.shape[1:]))
        first = key not in LIBRARY_CALLS
        LIBRARY_CALLS[key] += 1
        if STRICT:
            raise RuntimeError(f"SGMCMC_STRICT=1: {op} on a per-sample shape {tuple(x.shape[1:])} has no kernel in "
                               "bnn_priors_amd and would run on the library path (MIOpen / rocBLAS / ATen)")
        if first:       # loud, once per (operator, shape): a width / depth off the kernel tables is several times slower
            warnings.warn(f"bnn_priors_amd: {op} on a per-sample shape {tuple(x.shape[1:])} has no hand-written kernel and "
                          "runs on the library path (MIOpen / rocBLAS / ATen) inside a gradient evaluation; kernel "
                          "tables: conv.SHAPES, conv.DOWN_SHAPES, the 50-channel classifier -- set SGMCMC_STRICT=1 to "
                          "make this an error", RuntimeWarning, stacklevel=3)


SHAPES = {(16, 32), (32, 16), (64, 8)}      # (channels, image side)
ENABLED = True          # (module attribute, not an environment switch)
DEFER_REDUCE = True
# the persistent kernels on prepared weight fragments (csrc/conv2_hip.inc): double-buffered workgroups over a stream of
# 4-row items.  WHERE THEY RUN: launches that carry several minibatches (``persistent()`` below: the grouped exact
# full-data pass, graphed.py) -- 10-20 % faster per launch at 512 images (tools/conv_lab), the googleresnet pass 190 ->
# 172 ms.  NOT the 128-image leapfrog step: there they are slower inside the captured graph (1,154 -> 1,131 steps/s:
# cold operands of the epilogues, the fragment launch, twice the slab bytes at 64 channels, twice the statistics slices
# for the BatchNorm kernels -- DESIGN.md section 3).  SGMCMC_CONV_PERSISTENT=1 selects them everywhere (same results up
# to fp32 summation order: tested).
PERSISTENT = os.environ.get("SGMCMC_CONV_PERSISTENT", "0") == "1"


# ... and, per trunk stage, for the BACKWARD launch alone (forward on the default kernels, so the BatchNorm that follows
# reads the default statistics slices): "32x16,64x8" style list of (channels x side) -- an A/B switch of round 4
PERSISTENT_BWD = set()          # (module attribute, not an environment switch)


def persistent_bwd(c, hw):
    return PERSISTENT or (c, hw) in PERSISTENT_BWD


@contextlib.contextmanager
def persistent(on=True):
    "scope in which the trunk's 3x3 convolutions (forward and both gradients) run on the persistent kernels"
    global PERSISTENT
    old, PERSISTENT = PERSISTENT, bool(on) or PERSISTENT
    try:
        yield
    finally:
        PERSISTENT = old


# ---- prepared weight fragments (csrc/conv2_hip.inc) ------------------------------------------------------------
# The persistent kernels read a convolution's weights in MFMA fragment order (forward, and transposed + flipped for
# the data gradient).  ``frags(w)`` returns the two buffers of weight tensor w, current: inside a ``deferring(owner)``
# scope -- ONE forward + backward evaluation, during which the weights do not change -- every weight the owner used
# in its previous scope is prepared by ONE launch at the scope's entry; anything else is prepared by a launch of its
# own right where it is needed (always, outside a scope: nothing is ever assumed about a weight's history).
_frag_cache = {}      # id(weight tensor) -> [weakref, forward fragments, data-gradient fragments]
_frag_valid = set()   # ids whose fragments are current in the active scope
_owner_weights = {}   # id(owner) -> weak references of the weights its last scope used


def _frag_entry(w):
    e = _frag_cache.get(id(w))
    if e is None or e[0]() is not w:
        key = id(w)
        ref = weakref.ref(w, lambda _r, key=key: _frag_cache.pop(key, None))
        e = _frag_cache[key] = [ref, torch.empty(w.numel(), dtype=torch.float32, device=w.device),
                                torch.empty(w.numel(), dtype=torch.float32, device=w.device)]
    return e


def _prepare(weights):
    "fragments of all ``weights`` in one launch (sgmcmc_conv3x3_prepare_weights splits beyond SGMCMC_FRAG_JOBS)"
    jobs = (_hip.FragJob * len(weights))()
    for j, w in zip(jobs, weights):
        e = _frag_entry(w)
        j.w, j.fwd, j.dgrad, j.channels = w.data_ptr(), e[1].data_ptr(), e[2].data_ptr(), w.shape[0]
    err = _hip.lib().sgmcmc_conv3x3_prepare_weights(ctypes.cast(jobs, ctypes.c_void_p), len(weights), _stream())
    if err:
        _hip.check(err, "sgmcmc_conv3x3_prepare_weights")


def frags(w):
    "(forward fragments, data-gradient fragments) of the contiguous [C, C, 3, 3] float32 weight w, current"
    e = _frag_entry(w)
    if id(w) not in _frag_valid:
        _prepare([w])
        if _defer["active"]:
            _frag_valid.add(id(w))
    if _defer["active"] and id(w) not in _defer["weights"]:
        _defer["weights"][id(w)] = e[0]
    return e[1], e[2]

# ---- deferred weight-gradient reduction: opt-in, per backward pass ---------------------------------------
# Inside ``with deferring():`` (the samplers' own gradient evaluations: potential.py, graphed.py) a
# convolution's backward hands autograd a weight gradient whose memory is filled by ONE reduction launch at
# the end of the pass.  That is only sound if AccumulateGrad adopts exactly that tensor and nothing reads it
# earlier, so a call defers only when ALL of these hold: the context is active, grad mode is off inside
# backward (no create_graph), the weight is a leaf without a ``.grad`` yet, and the weight took part in ONE
# forward call of this pass (a weight shared by two operations has its gradients summed mid-backward).
# Everywhere else -- user code calling these operators directly -- the reduction is launched immediately.
_defer = {"active": False, "uses": {}, "weights": {}}

# ---- zeroed integer slots for grid-wide sums (csrc/conv_hip.inc namespace fx) ---------------------------------------
# A launch that leaves per-channel sums in fx slots ADDS to them, so they have to be zero beforehand.  Inside a
# ``deferring`` scope the slots of a pass are slices of ONE arena zeroed by one launch: a fresh arena per captured graph
# (its memset is a node of that graph; graphs of one model replay concurrently on their own streams), a persistent one
# per (owner, stream) for eager passes.  Outside a scope every request is its own ``torch.zeros``.
FX_ARENA_INT64 = 1 << 18          # 2 MiB (googleresnet's nine folded BatchNorms take 1.7 MiB)
_fx = {"arena": None, "offset": 0, "owner": None, "eager": {}, "kept": []}


def _fx_reset(owner):
    _fx["arena"], _fx["offset"], _fx["owner"] = None, 0, owner


def fx_take(channels, device):
    "a zeroed int64 [8][5][channels][16] slot set (sgmcmc_fx_slot_int64) for one launch pair of this pass"
    n = _hip.lib().sgmcmc_fx_slot_int64(channels)
    if not _defer["active"] or n > FX_ARENA_INT64:
        return torch.zeros(n, dtype=torch.int64, device=device)
    if _fx["arena"] is None:
        if torch.cuda.is_current_stream_capturing():
            arena = torch.zeros(FX_ARENA_INT64, dtype=torch.int64, device=device)
            # the graph replays on this memory: it lives as long as the model that owns the captured pass
            owner = _fx["owner"]
            (owner.__dict__.setdefault("_fx_arenas", []) if hasattr(owner, "__dict__") else _fx["kept"]).append(arena)
        else:
            key = (id(_fx["owner"]), torch.cuda.current_stream(device).cuda_stream, str(device))
            arena = _fx["eager"].get(key)
            if arena is None:
                if len(_fx["eager"]) >= 16:
                    _fx["eager"].clear()
                arena = _fx["eager"][key] = torch.empty(FX_ARENA_INT64, dtype=torch.int64, device=device)
            arena.zero_()
        _fx["arena"], _fx["offset"] = arena, 0
    if _fx["offset"] + n > FX_ARENA_INT64:
        return torch.zeros(n, dtype=torch.int64, device=device)
    out = _fx["arena"][_fx["offset"]:_fx["offset"] + n]
    _fx["offset"] += n
    return out


@contextlib.contextmanager
def deferring(owner=None):
    """scope of one forward + backward evaluation whose convolution weight gradients may be reduced at the end.
    ``owner`` (any object, e.g. the model): the trunk weights this owner used in its previous scope get their MFMA
    fragments prepared by one launch right here (see ``frags``)."""
    if not DEFER_REDUCE or _defer["active"]:
        yield
        return
    _defer["active"] = True
    _defer["uses"] = {}
    _defer["weights"] = {}
    _fx_reset(owner)
    _frag_valid.clear()
    _pending.clear()                # leftovers of a pass that raised before its final callback
    if (PERSISTENT or PERSISTENT_BWD) and owner is not None:
        known = [w for w in (r() for r in _owner_weights.get(id(owner), ())) if w is not None and w.is_cuda]
        if known:
            _prepare(known)
            _frag_valid.update(id(w) for w in known)
    try:
        yield
    except BaseException:
        _pending.clear()
        _join_side()
        raise
    finally:
        if owner is not None:
            _owner_weights[id(owner)] = list(_defer["weights"].values())
        _defer["active"] = False
        _defer["uses"] = {}
        _defer["weights"] = {}
        _fx_reset(None)
        _frag_valid.clear()
    _flush_pending()                # no-op when the backward's final callback already ran


def _note_use(*weights):
    if _defer["active"]:
        for w in weights:
            _defer["uses"][id(w)] = _defer["uses"].get(id(w), 0) + 1


def _may_defer(*weights):
    return (_defer["active"] and not torch.is_grad_enabled()
            and all(w.is_leaf and w.grad is None and _defer["uses"].get(id(w), 0) == 1 for w in weights))


def supported(x, w, bias, stride, padding, dilation, groups):
    if not ENABLED or bias is not None or not x.is_cuda or x.dtype != torch.float32 or x.dim() != 4:
        return False
    c, hw = x.shape[1], x.shape[2]
    return ((c, hw) in SHAPES and x.shape[3] == hw and tuple(w.shape) == (c, c, 3, 3)
            and w.dtype == torch.float32 and groups == 1 and x.shape[0] > 0
            and _pair(stride) == (1, 1) and _pair(padding) == (1, 1) and _pair(dilation) == (1, 1))


def _pair(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _run(x, w, transpose_w, want_stats=False):
    lib = _hip.lib()
    y = torch.empty_like(x)
    stats = None
    if PERSISTENT:
        f_fwd, f_dgrad = frags(w)
        if want_stats:
            slices = lib.sgmcmc_conv3x3_frag_stat_slices(x.shape[0], x.shape[1], x.shape[2])
            stats = torch.empty((x.shape[1], slices, 2), dtype=torch.float64, device=x.device)
        err = lib.sgmcmc_conv3x3_frag_fwd(x.data_ptr(), (f_dgrad if transpose_w else f_fwd).data_ptr(), y.data_ptr(),
                                          x.shape[0], x.shape[1], x.shape[2], 0 if stats is None else stats.data_ptr(),
                                          _stream())
        if err:
            _hip.check(err, "sgmcmc_conv3x3_frag_fwd")
        return y, stats
    if want_stats:   # [channels][slices][2] partial (sum, centred sum of squares) of y, for the BatchNorm that follows
        slices = lib.sgmcmc_conv3x3_stat_slices(x.shape[0], x.shape[1], x.shape[2])
        stats = torch.empty((x.shape[1], slices, 2), dtype=torch.float64, device=x.device)
    err = lib.sgmcmc_conv3x3(x.data_ptr(), w.data_ptr(), y.data_ptr(), x.shape[0], x.shape[1], x.shape[2],
                             int(transpose_w), 0 if stats is None else stats.data_ptr(), _stream())
    if err:
        _hip.check(err, "sgmcmc_conv3x3")
    return y, stats


def _weight_grad(x, dy, w=None):
    lib = _hip.lib()
    n, c, hw = x.shape[0], x.shape[1], x.shape[2]
    if PERSISTENT and w is not None:
        # the persistent launch always carries both halves (they share the GPU): the data gradient is discarded
        return frag_backward(lib, x, w, dy, False)[1]
    scratch = torch.empty(lib.sgmcmc_conv3x3_wrw_scratch_floats(n, c, hw), dtype=torch.float32, device=x.device)
    dw = torch.empty((c, c, 3, 3), dtype=torch.float32, device=x.device)
    err = lib.sgmcmc_conv3x3_wrw(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), scratch.data_ptr(), n, c, hw,
                                 _stream())
    if err:
        _hip.check(err, "sgmcmc_conv3x3_wrw")
    return dw


def split_backward(lib, x, w, dy, dx, E, scratch, slabs):
    "weight-gradient slabs on the side stream (forked here: dy is ready), data gradient (+ epilogues E) on the main one"
    n, c, hw = x.shape[0], x.shape[1], x.shape[2]
    side = side_stream_for(x)
    keep_until_join(x, dy, scratch)
    err = lib.sgmcmc_conv3x3_bwd_part(x.data_ptr(), 0, dy.data_ptr(), 0, None, scratch.data_ptr(), n, c, hw, 2,
                                      ctypes.byref(slabs), side.cuda_stream)
    if err:
        _hip.check(err, "sgmcmc_conv3x3_bwd_part(weights)")
    err = lib.sgmcmc_conv3x3_bwd_part(0, w.data_ptr(), dy.data_ptr(), dx.data_ptr(), None if E is None else ctypes.byref(E),
                                      0, n, c, hw, 1, None, _stream())
    if err:
        _hip.check(err, "sgmcmc_conv3x3_bwd_part(data)")


def frag_backward(lib, x, w, dy, defer, add=None, sums_for=None):
    """Both gradients of y = conv3x3(x, w) by the persistent launch (sgmcmc_conv3x3_frag_bwd).  ``add`` = (e_dout,
    e_out): dx += e_dout * [e_out > 0]; ``sums_for`` = (y_bn, out_bn, saved_bn): also the partial sums of the BatchNorm
    backward whose incoming gradient dx is.  -> (dx, dw, partial or None, n_partials); with ``defer`` dw is an alias
    whose memory the pass's final reduction fills (one job per 16-output-channel tile)."""
    n, c, hw = x.shape[0], x.shape[1], x.shape[2]
    scratch = torch.empty(lib.sgmcmc_conv3x3_frag_scratch_floats(n, c, hw), dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x)
    dw = torch.empty((c, c, 3, 3), dtype=torch.float32, device=x.device)
    E = _hip.ConvBwdEpilogue()
    if add is not None:      # (e_out None: e_dout is masked already -- bnlink.PREMASK)
        E.e_dout, E.e_out = add[0].data_ptr(), 0 if add[1] is None else add[1].data_ptr()
    partial, n_part = None, lib.sgmcmc_conv3x3_frag_stat_slices(n, c, hw)
    if sums_for is not None:
        y_bn, out_bn, saved_bn = sums_for
        partial = torch.empty((c, n_part, 2), dtype=torch.float64, device=x.device)
        E.s_y, E.s_out, E.s_mean, E.s_invstd = (y_bn.data_ptr(), out_bn.data_ptr(), saved_bn[0].data_ptr(),
                                                saved_bn[1].data_ptr())
        E.s_partial = partial.data_ptr()
        E.group_imgs = _group_imgs(n)
        E.mask_dx = int(_bnlink.PREMASK)         # dx leaves as dx * [out_bn > 0]: what its consumers form from it
    slabs = ctypes.c_int(0)
    err = lib.sgmcmc_conv3x3_frag_bwd(x.data_ptr(), frags(w)[1].data_ptr(), dy.data_ptr(), dx.data_ptr(), ctypes.byref(E),
                                      0 if defer else dw.data_ptr(), scratch.data_ptr(), n, c, hw,
                                      ctypes.byref(slabs) if defer else None, _stream())
    if err:
        _hip.check(err, "sgmcmc_conv3x3_frag_bwd")
    if defer:
        P, slab = slabs.value, 144 * c          # [tile][P][9][16][C]: one reduction job per output-channel tile
        for t in range(c // 16):
            _pending.append((scratch[t * P * slab:(t + 1) * P * slab], dw[16 * t:16 * (t + 1)], P, 9))
        return dx, dw.view(dw.shape), partial, n_part
    return dx, dw, partial, n_part


# launches that carry G minibatches (bn.grouped): every weight-gradient workgroup of the trunk's 3x3 convolutions walks G
# times as many items, so a pass leaves the slabs of ONE minibatch (``WRW_GROUP_MULT = False``: G times as many slabs)
WRW_GROUP_MULT = True


def _group_imgs(n):
    "images per group for a launch that looks a BatchNorm's saved statistics up by image (bn.grouped); 0: one batch"
    from . import bn as _bn
    G = _bn.groups()
    if G > 1 and SIDE_STREAM:
        raise RuntimeError("bn.grouped(G > 1) does not run with the side-stream alternative")
    return n // G if G > 1 else 0


def _both_grads(x, w, dy, defer, sums_for=None):
    "``sums_for`` = (y_bn, out_bn, saved_bn) of the BatchNorm + ReLU that produced x: its backward sums ride along"
    lib = _hip.lib()
    n, c, hw = x.shape[0], x.shape[1], x.shape[2]
    if persistent_bwd(c, hw) and not (defer and SIDE_STREAM):
        dx, dw, partial, n_part = frag_backward(lib, x, w, dy, defer, sums_for=sums_for)
        if partial is not None:
            _bnlink.tag_gradient(dx, partial, n_part, _bnlink.PREMASK)
        return dx, dw
    scratch = torch.empty(lib.sgmcmc_conv3x3_wrw_scratch_floats(n, c, hw), dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x)
    dw = torch.empty((c, c, 3, 3), dtype=torch.float32, device=x.device)
    slabs = ctypes.c_int(0)
    if defer and SIDE_STREAM:
        partial, E = None, None
        if sums_for is not None:
            y_bn, out_bn, saved_bn = sums_for
            n_part = lib.sgmcmc_conv3x3_stat_slices(n, c, hw)
            partial = torch.empty((c, n_part, 2), dtype=torch.float64, device=x.device)
            E = _hip.ConvBwdEpilogue(s_y=y_bn.data_ptr(), s_out=out_bn.data_ptr(), s_mean=saved_bn[0].data_ptr(),
                                     s_invstd=saved_bn[1].data_ptr(), s_partial=partial.data_ptr(),
                                     mask_dx=int(_bnlink.PREMASK))
        split_backward(lib, x, w, dy, dx, E, scratch, slabs)
        if partial is not None:
            _bnlink.tag_gradient(dx, partial, n_part, _bnlink.PREMASK)
        _pending.append((scratch, dw, slabs.value, 9))
        return dx, dw.view(dw.shape)
    if sums_for is None:
        err = lib.sgmcmc_conv3x3_bwd(x.data_ptr(), w.data_ptr(), dy.data_ptr(), dx.data_ptr(), dw.data_ptr(),
                                     scratch.data_ptr(), n, c, hw, ctypes.byref(slabs) if defer else None, _stream())
    else:
        y_bn, out_bn, saved_bn = sums_for
        n_part = lib.sgmcmc_conv3x3_stat_slices(n, c, hw)
        partial = torch.empty((c, n_part, 2), dtype=torch.float64, device=x.device)
        E = _hip.ConvBwdEpilogue(s_y=y_bn.data_ptr(), s_out=out_bn.data_ptr(), s_mean=saved_bn[0].data_ptr(),
                                 s_invstd=saved_bn[1].data_ptr(), s_partial=partial.data_ptr(),
                                 group_imgs=_group_imgs(n), mask_dx=int(_bnlink.PREMASK))
        from . import bn as _bn
        E.wrw_mult = WRW_GROUP_MULT and _bn.groups()
        err = lib.sgmcmc_conv3x3_bwd_ex(x.data_ptr(), w.data_ptr(), dy.data_ptr(), dx.data_ptr(), ctypes.byref(E),
                                        dw.data_ptr(), scratch.data_ptr(), n, c, hw,
                                        ctypes.byref(slabs) if defer else None, _stream())
        _bnlink.tag_gradient(dx, partial, n_part, _bnlink.PREMASK)
    if err:
        _hip.check(err, "sgmcmc_conv3x3_bwd")
    if defer:
        # autograd gets an ALIAS: AccumulateGrad only adopts a gradient nobody else references, and it
        # must adopt (not copy) this one -- its memory is filled at the end of the pass
        _pending.append((scratch, dw, slabs.value, 9))     # 3x3 slabs are tap-major (sgmcmc_reduce_job.taps)
        return dx, dw.view(dw.shape)
    return dx, dw


# Weight-gradient slabs whose reduction waits for the end of the running backward pass, where ONE launch
# sums all of them (autograd's final callback); see ``deferring`` above for when a call may defer.
_pending = []


# ---- weight gradients off the critical path (MEASURED SLOWER: off by default) ------------------------------------
# Inside a deferring pass the two halves of a trunk convolution's backward -- data gradient (on the critical path:
# the next layer's backward waits for it) and weight-gradient slabs (needed only by the reduction at the END of the
# pass) -- can be launched separately: the slabs on a SIDE stream that forks off the main one where dy is ready and
# joins it again before the reduction, captured into the step's hipGraph as parallel branches.  The dependent chain
# would carry 10 us data-gradient launches instead of 17 - 22 us merged ones.  On MI355X / ROCm 7.2 every
# fork + join edge of a replayed graph costs ~19 us of cross-queue signalling: 16 of them per step, googleresnet
# 1,137 -> 845 steps/s (round 2; round 6 on the current tree: 1,272 -> 307, and 415 with ONE fork per pass -- a graph with
# any parallel branch replays 2-3x slower than a chain: profiles/r06_lab_side_queue.txt).  The merged launch (both halves'
# workgroups in one grid) stays the default;
# SGMCMC_CONV_SIDE_STREAM=1 enables this route (same workgroups, same bits: tested).
SIDE_STREAM = os.environ.get("SGMCMC_CONV_SIDE_STREAM", "0") == "1"
if SIDE_STREAM and not _hip.ALTERNATIVES:
    raise RuntimeError("SGMCMC_CONV_SIDE_STREAM=1 selects a measured alternative: build and load the library with "
                       "SGMCMC_ALTERNATIVES=1")
_side = {"streams": {}, "forked": None, "keep": []}


def side_stream_for(t):
    "fork: the side stream of t's device, made to wait for everything enqueued on the current stream so far"
    main = torch.cuda.current_stream(t.device)
    side = _side["streams"].get(t.device)
    if side is None:
        side = _side["streams"][t.device] = torch.cuda.Stream(device=t.device)
    side.wait_stream(main)
    _side["forked"] = (main, side)
    return side


def keep_until_join(*tensors):
    "operands of side-stream launches must outlive them: held until the join (their memory is not reused before)"
    _side["keep"].extend(tensors)


def _join_side():
    if _side["forked"] is not None:
        main, side = _side["forked"]
        main.wait_stream(side)
        _side["forked"] = None
    _side["keep"].clear()


def _flush_pending():
    _join_side()
    if not _pending:
        return
    jobs = (_hip.ReduceJob * len(_pending))()
    for j, (scratch, dw, slabs, taps) in zip(jobs, _pending):
        j.part, j.out, j.n_slabs, j.numel, j.taps = scratch.data_ptr(), dw.data_ptr(), slabs, dw.numel(), taps
    err = _hip.lib().sgmcmc_wrw_reduce_many(ctypes.cast(jobs, ctypes.c_void_p), len(_pending), _stream())
    _pending.clear()
    if err:
        _hip.check(err, "sgmcmc_wrw_reduce_many")


class _Conv3x3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, want_stats, src_y=None, src_saved=None):
        _note_use(w)
        x, w = x.contiguous(), w.contiguous()
        ctx.save_for_backward(x, w, src_y, src_saved)
        ctx.set_materialize_grads(False)     # no zero tensors for the (non-differentiable) statistics output
        y, stats = _run(x, w, False, want_stats)
        if not want_stats:
            return y
        ctx.mark_non_differentiable(stats)
        return y, stats

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy, *_):
        x, w, src_y, src_saved = ctx.saved_tensors
        if dy is None:                 # the output took no gradient (grads are not materialised here)
            return None, None, None, None, None
        dy = dy.contiguous()
        if ctx.needs_input_grad[0] and ctx.needs_input_grad[1]:
            defer = _may_defer(w)
            if defer:   # every deferring call queues it; the first one to run does the work
                torch.autograd.Variable._execution_engine.queue_callback(_flush_pending)
            # one launch for the two of them (+ the sums of the BatchNorm that produced x, if x is tagged)
            return (*_both_grads(x, w, dy, defer, None if src_y is None else (src_y, x, src_saved)), None, None, None)
        dx = _run(dy, w, True)[0] if ctx.needs_input_grad[0] else None
        dw = _weight_grad(x, dy, w) if ctx.needs_input_grad[1] else None
        return dx, dw, None, None, None


# Evaluation mode: conv3x3 -> BatchNorm (running statistics) -> (+ residual) -> ReLU in ONE launch (csrc/conv_hip.inc,
# EvalBn): the BatchNorm is a per-channel affine map of the accumulator tile, so its launch, the convolution's output
# tensor and the read of it disappear from every test-set forward; the bits of ``conv3x3`` followed by ``bn.bn_eval``.
# SGMCMC_CONV_BN_EVAL=0 restores the two launches (A/B, the tests' cross-check).
CONV_BN_EVAL = True


def conv_bn_eval_supported(x, w, bias, conv_args, bn):
    "a trunk 3x3 convolution followed by an eval-mode BatchNorm2d with running statistics, outside autograd"
    from . import bn as _bn
    return (CONV_BN_EVAL and not PERSISTENT and not bn.training and bn.track_running_stats
            and supported(x, w, bias, *conv_args)
            and _bn.eval_supported(x, bn.weight, bn.bias, bn.running_mean, bn.running_var))


def conv3x3_bn_eval(x, w, bn, residual=None, relu=False):
    "relu?(bn_eval(conv3x3(x, w)) [+ residual]) -- sgmcmc_conv3x3_bn_eval"
    x, w = x.contiguous(), w.contiguous()
    if residual is not None:
        residual = residual.contiguous()
    y = torch.empty_like(x)
    err = _hip.lib().sgmcmc_conv3x3_bn_eval(x.data_ptr(), w.data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr(),
                                            bn.running_mean.data_ptr(), bn.running_var.data_ptr(), float(bn.eps),
                                            0 if residual is None else residual.data_ptr(), int(bool(relu)), y.data_ptr(),
                                            x.shape[0], x.shape[1], x.shape[2], _stream())
    if err:
        _hip.check(err, "sgmcmc_conv3x3_bn_eval")
    return y


def conv3x3(x, w, want_stats=False):
    """3x3 / stride 1 / zero-pad 1 convolution, no bias, for the (channels, side) pairs in SHAPES.
    ``want_stats``: also return the per-band (sum, sum of squared deviations from the band mean) of every output channel, float64
    [channels][slices][2] -- what ``bn.bn_train(..., stats=...)`` needs instead of a pass over y."""
    src_y, src_saved = _bnlink.source_of(x) if (x.requires_grad and w.requires_grad) else (None, None)
    return _Conv3x3.apply(x, w, want_stats, src_y, src_saved)


# ------------------------------------------------------------------ the down-sampling block's pair
DOWN_SHAPES = {(16, 32), (32, 16)}          # (input channels, input side): 3x3 / stride 2 + 1x1 / stride 2


def down_supported(x, w_main, w_short):
    "the two convolutions that open a down-sampling block, as one operator (csrc/conv_down_hip.inc)"
    if not ENABLED or not x.is_cuda or x.dtype != torch.float32 or x.dim() != 4 or x.shape[0] == 0:
        return False
    c, hw = x.shape[1], x.shape[2]
    return ((c, hw) in DOWN_SHAPES and x.shape[3] == hw and tuple(w_main.shape) == (2 * c, c, 3, 3)
            and tuple(w_short.shape) == (2 * c, c, 1, 1) and w_main.dtype == torch.float32
            and w_short.dtype == torch.float32)


class _ConvDown(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w_main, w_short, want_stats, src_y=None, src_saved=None):
        lib = _hip.lib()
        _note_use(w_main, w_short)
        x, w_main, w_short = x.contiguous(), w_main.contiguous(), w_short.contiguous()
        n, c, hw = x.shape[0], x.shape[1], x.shape[2]
        ym = torch.empty((n, 2 * c, hw // 2, hw // 2), dtype=torch.float32, device=x.device)
        ys = torch.empty_like(ym)
        sm = ss = None
        if want_stats:
            slices = lib.sgmcmc_conv_down_stat_slices(n, c, hw)
            sm = torch.empty((2 * c, slices, 2), dtype=torch.float64, device=x.device)
            ss = torch.empty_like(sm)
        err = lib.sgmcmc_conv_down_fwd(x.data_ptr(), w_main.data_ptr(), w_short.data_ptr(), ym.data_ptr(),
                                       ys.data_ptr(), 0 if sm is None else sm.data_ptr(),
                                       0 if ss is None else ss.data_ptr(), n, c, hw, _stream())
        if err:
            _hip.check(err, "sgmcmc_conv_down_fwd")
        ctx.save_for_backward(x, w_main, w_short, src_y, src_saved)
        ctx.set_materialize_grads(False)
        if not want_stats:
            return ym, ys
        ctx.mark_non_differentiable(sm, ss)
        return ym, ys, sm, ss

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dym, dys, *_):
        lib = _hip.lib()
        x, w_main, w_short, src_y, src_saved = ctx.saved_tensors
        if dym is None and dys is None:
            return None, None, None, None, None, None
        n_, c_, hw_ = x.shape[0], x.shape[1], x.shape[2]
        zeros = lambda: torch.zeros((n_, 2 * c_, hw_ // 2, hw_ // 2), dtype=torch.float32, device=x.device)
        dym = zeros() if dym is None else dym.contiguous()
        dys = zeros() if dys is None else dys.contiguous()
        n, c, hw = x.shape[0], x.shape[1], x.shape[2]
        scratch = torch.empty(lib.sgmcmc_conv_down_scratch_floats(n, c, hw), dtype=torch.float32, device=x.device)
        dx = torch.empty_like(x)
        dwm, dws = torch.empty_like(w_main), torch.empty_like(w_short)
        defer = _may_defer(w_main, w_short)
        slabs = ctypes.c_int(0)
        if defer:
            torch.autograd.Variable._execution_engine.queue_callback(_flush_pending)
        if src_y is None:
            err = lib.sgmcmc_conv_down_bwd(x.data_ptr(), w_main.data_ptr(), w_short.data_ptr(), dym.data_ptr(),
                                           dys.data_ptr(), dx.data_ptr(), dwm.data_ptr(), dws.data_ptr(),
                                           scratch.data_ptr(), n, c, hw, ctypes.byref(slabs) if defer else None,
                                           _stream())
        else:       # x came out of a BatchNorm + ReLU: that BatchNorm's backward sums ride in this launch (bnlink)
            n_part = lib.sgmcmc_conv_down_bwd_sum_slices(n, c, hw)
            partial = torch.empty((c, n_part, 2), dtype=torch.float64, device=x.device)
            E = _hip.ConvBwdEpilogue(s_y=src_y.data_ptr(), s_out=x.data_ptr(), s_mean=src_saved[0].data_ptr(),
                                     s_invstd=src_saved[1].data_ptr(), s_partial=partial.data_ptr(),
                                     group_imgs=_group_imgs(n), mask_dx=int(_bnlink.PREMASK))
            err = lib.sgmcmc_conv_down_bwd_ex(x.data_ptr(), w_main.data_ptr(), w_short.data_ptr(), dym.data_ptr(),
                                              dys.data_ptr(), dx.data_ptr(), ctypes.byref(E), dwm.data_ptr(),
                                              dws.data_ptr(), scratch.data_ptr(), n, c, hw,
                                              ctypes.byref(slabs) if defer else None, _stream())
            _bnlink.tag_gradient(dx, partial, n_part, _bnlink.PREMASK)
        if err:
            _hip.check(err, "sgmcmc_conv_down_bwd")
        if defer:   # scratch = [slabs][dwm.numel()] then [slabs][dws.numel()]
            _pending.append((scratch, dwm, slabs.value, 9))
            _pending.append((scratch[slabs.value * dwm.numel():], dws, slabs.value, 1))
            return dx, dwm.view(dwm.shape), dws.view(dws.shape), None, None, None
        return dx, dwm, dws, None, None, None


def conv_down(x, w_main, w_short, want_stats=False):
    """(conv2d(x, w_main, stride=2, padding=1), conv2d(x, w_short, stride=2)) in one operator, for the
    (channels, side) pairs in DOWN_SHAPES; with ``want_stats`` also the two outputs' batch statistics
    (see ``conv3x3``)."""
    src_y, src_saved = _bnlink.source_of(x) if x.requires_grad else (None, None)
    return _ConvDown.apply(x, w_main, w_short, want_stats, src_y, src_saved)


# ------------------------------------------------------------------ the stem
def stem_supported(x, w, bias, stride, padding, dilation, groups):
    "3 -> 16 channels, 3x3 / stride 1 / pad 1 on 32x32 images that take no gradient (csrc/conv_down_hip.inc)"
    return (ENABLED and bias is None and x.is_cuda and x.dtype == torch.float32 and not x.requires_grad
            and tuple(x.shape[1:]) == (3, 32, 32) and x.shape[0] > 0 and tuple(w.shape) == (16, 3, 3, 3)
            and w.dtype == torch.float32 and groups == 1 and _pair(stride) == (1, 1)
            and _pair(padding) == (1, 1) and _pair(dilation) == (1, 1))


class _ConvStem(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, want_stats):
        lib = _hip.lib()
        _note_use(w)
        x, w = x.contiguous(), w.contiguous()
        n = x.shape[0]
        y = torch.empty((n, 16, 32, 32), dtype=torch.float32, device=x.device)
        stats = torch.empty((16, 4 * n, 2), dtype=torch.float64, device=x.device) if want_stats else None
        err = lib.sgmcmc_conv_stem_fwd(x.data_ptr(), w.data_ptr(), y.data_ptr(),
                                       0 if stats is None else stats.data_ptr(), n, _stream())
        if err:
            _hip.check(err, "sgmcmc_conv_stem_fwd")
        ctx.save_for_backward(x, w)
        ctx.set_materialize_grads(False)     # no zero tensors for the (non-differentiable) statistics output
        if not want_stats:
            return y
        ctx.mark_non_differentiable(stats)
        return y, stats

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy, *_):
        lib = _hip.lib()
        x, w = ctx.saved_tensors
        if dy is None or not ctx.needs_input_grad[1]:
            return None, None, None
        dy = dy.contiguous()
        n = x.shape[0]
        scratch = torch.empty(lib.sgmcmc_conv_stem_scratch_floats(n), dtype=torch.float32, device=x.device)
        dw = torch.empty_like(w)
        defer = _may_defer(w)
        slabs = ctypes.c_int(0)
        if defer:
            torch.autograd.Variable._execution_engine.queue_callback(_flush_pending)
        err = lib.sgmcmc_conv_stem_wrw(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), scratch.data_ptr(), n,
                                       ctypes.byref(slabs) if defer else None, _stream())
        if err:
            _hip.check(err, "sgmcmc_conv_stem_wrw")
        if defer:
            _pending.append((scratch, dw, slabs.value, 1))
            return None, dw.view(dw.shape), None
        return None, dw, None


def conv_stem(x, w, want_stats=False):
    "conv2d(x, w, padding=1) for x [N, 3, 32, 32] (no gradient), w [16, 3, 3, 3]; see ``conv3x3``"
    return _ConvStem.apply(x, w, want_stats)


# ------------------------------------------------------------------ the convolutional classifier's first layer
def first_supported(x, w, bias, stride, padding, dilation, groups):
    "1 -> 50 channels, 3x3 / stride 1 / pad 1 on 28x28 images that take no gradient (namespace convfirst)"
    return (ENABLED and bias is None and x.is_cuda and x.dtype == torch.float32 and not x.requires_grad
            and x.dim() == 4 and tuple(x.shape[1:]) == (1, 28, 28) and x.shape[0] > 0
            and tuple(w.shape) == (50, 1, 3, 3) and w.dtype == torch.float32 and groups == 1
            and _pair(stride) == (1, 1) and _pair(padding) == (1, 1) and _pair(dilation) == (1, 1))


class _ConvFirst(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        _note_use(w)
        x, w = x.contiguous(), w.contiguous()
        n = x.shape[0]
        y = torch.empty((n, 50, 28, 28), dtype=torch.float32, device=x.device)
        err = _hip.lib().sgmcmc_conv_first_fwd(x.data_ptr(), w.data_ptr(), y.data_ptr(), n, _stream())
        if err:
            _hip.check(err, "sgmcmc_conv_first_fwd")
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        lib = _hip.lib()
        x, w = ctx.saved_tensors
        if not ctx.needs_input_grad[1]:
            return None, None
        dy = dy.contiguous()
        n = x.shape[0]
        scratch = torch.empty(lib.sgmcmc_conv_first_scratch_floats(n), dtype=torch.float32, device=x.device)
        dw = torch.empty_like(w)
        defer = _may_defer(w)
        slabs = ctypes.c_int(0)
        if defer:
            torch.autograd.Variable._execution_engine.queue_callback(_flush_pending)
        err = lib.sgmcmc_conv_first_wrw(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), scratch.data_ptr(), n,
                                        ctypes.byref(slabs) if defer else None, _stream())
        if err:
            _hip.check(err, "sgmcmc_conv_first_wrw")
        if defer:
            _pending.append((scratch, dw, slabs.value, 1))
            return None, dw.view(dw.shape)
        return None, dw


def conv_first(x, w):
    "conv2d(x, w, padding=1) for x [N, 1, 28, 28] (no gradient), w [50, 1, 3, 3]"
    return _ConvFirst.apply(x, w)


# ---- the first layer with its tail (csrc/conv_down_hip.inc, convfirst::fwd_pool_kernel / wrw_pool_kernel) -----------
# ``CONV_POOL = False`` restores conv -> bias_relu_pool as two operators (the tests' cross-check).
CONV_POOL = True


def first_pool_supported(x, w, bias, stride, padding, dilation, groups):
    "``first_supported`` + a float32 bias (or none) for the fused conv -> + bias -> ReLU -> MaxPool2d(2)"
    return (CONV_POOL and first_supported(x, w, None, stride, padding, dilation, groups)
            and (bias is None or (bias.dtype == torch.float32 and tuple(bias.shape) == (50,) and bias.is_cuda)))


class _ConvFirstPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias):
        _note_use(w) if bias is None else _note_use(w, bias)
        x, w = x.contiguous(), w.contiguous()
        n = x.shape[0]
        pooled = torch.empty((n, 50, 14, 14), dtype=torch.float32, device=x.device)
        code = torch.empty((n, 50, 14, 14), dtype=torch.uint8, device=x.device)
        err = _hip.lib().sgmcmc_conv_first_pool_fwd(x.data_ptr(), w.data_ptr(), 0 if bias is None else bias.data_ptr(),
                                                    pooled.data_ptr(), code.data_ptr(), n, _stream())
        if err:
            _hip.check(err, "sgmcmc_conv_first_pool_fwd")
        ctx.save_for_backward(x, w, bias, code)
        return pooled

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dpooled):
        lib = _hip.lib()
        x, w, bias, code = ctx.saved_tensors
        want_w, want_b = ctx.needs_input_grad[1], bias is not None and ctx.needs_input_grad[2]
        if not (want_w or want_b):
            return None, None, None
        dpooled = dpooled.contiguous()
        n = x.shape[0]
        scratch = torch.empty(lib.sgmcmc_conv_first_pool_scratch_floats(n), dtype=torch.float32, device=x.device)
        dw = torch.empty_like(w)
        db = torch.empty_like(bias) if want_b else None
        defer = _may_defer(w) and (not want_b or _may_defer(bias))
        slabs = ctypes.c_int(0)
        if defer:
            torch.autograd.Variable._execution_engine.queue_callback(_flush_pending)
        err = lib.sgmcmc_conv_first_pool_bwd(x.data_ptr(), dpooled.data_ptr(), code.data_ptr(), dw.data_ptr(),
                                             0 if db is None else db.data_ptr(), scratch.data_ptr(), n, int(want_b),
                                             ctypes.byref(slabs) if defer else None, _stream())
        if err:
            _hip.check(err, "sgmcmc_conv_first_pool_bwd")
        if defer:
            P = slabs.value
            _pending.append((scratch[:P * 450], dw, P, 1))
            if want_b:
                _pending.append((scratch[P * 450:], db, P, 1))
                return None, dw.view(dw.shape), db.view(db.shape)
            return None, dw.view(dw.shape), None
        return None, dw, db


def conv_first_pool(x, w, bias=None):
    "max_pool2d(relu(conv2d(x, w, bias, padding=1)), 2) for x [N, 1, 28, 28] (no gradient), w [50, 1, 3, 3]"
    return _ConvFirstPool.apply(x, w, bias)


# ------------------------------------------------------------------ the convolutional classifier's second layer
def conv50_supported(x, w, bias, stride, padding, dilation, groups):
    "50 -> 50 channels, 3x3 / stride 1 / pad 1 on 14x14 maps (csrc/conv50_hip.inc); the bias joins the fused tail"
    return (ENABLED and bias is None and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
            and tuple(x.shape[1:]) == (50, 14, 14) and x.shape[0] > 0 and tuple(w.shape) == (50, 50, 3, 3)
            and w.dtype == torch.float32 and groups == 1 and _pair(stride) == (1, 1) and _pair(padding) == (1, 1)
            and _pair(dilation) == (1, 1))


class _Conv50(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        _note_use(w)
        x, w = x.contiguous(), w.contiguous()
        y = torch.empty_like(x)
        # the forward launch also leaves the weights as the data gradient reads them (transposed, taps flipped)
        wT = torch.empty_like(w) if ctx.needs_input_grad[0] else None
        err = _hip.lib().sgmcmc_conv50_fwd(x.data_ptr(), w.data_ptr(), y.data_ptr(), 0 if wT is None else wT.data_ptr(),
                                           x.shape[0], _stream())
        if err:
            _hip.check(err, "sgmcmc_conv50_fwd")
        ctx.save_for_backward(x, w)
        ctx.wT = wT
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        lib = _hip.lib()
        x, w = ctx.saved_tensors
        wT = ctx.wT
        dy = dy.contiguous()
        n = x.shape[0]
        if not ctx.needs_input_grad[1]:
            dx = torch.empty_like(x)
            if wT is not None:
                err = lib.sgmcmc_conv50_fwd(dy.data_ptr(), wT.data_ptr(), dx.data_ptr(), 0, n, _stream())
            else:
                err = lib.sgmcmc_conv50(dy.data_ptr(), w.data_ptr(), dx.data_ptr(), n, 1, _stream())
            if err:
                _hip.check(err, "sgmcmc_conv50")
            return dx, None
        scratch = torch.empty(lib.sgmcmc_conv50_scratch_floats(n), dtype=torch.float32, device=x.device)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw = torch.empty_like(w)
        defer = _may_defer(w)
        slabs = ctypes.c_int(0)
        if defer:
            torch.autograd.Variable._execution_engine.queue_callback(_flush_pending)
        if dx is not None and wT is not None:
            err = lib.sgmcmc_conv50_bwd_t(x.data_ptr(), wT.data_ptr(), dy.data_ptr(), dx.data_ptr(), dw.data_ptr(),
                                          scratch.data_ptr(), n, ctypes.byref(slabs) if defer else None, _stream())
        else:
            err = lib.sgmcmc_conv50_bwd(x.data_ptr(), w.data_ptr(), dy.data_ptr(), 0 if dx is None else dx.data_ptr(),
                                        dw.data_ptr(), scratch.data_ptr(), n, ctypes.byref(slabs) if defer else None,
                                        _stream())
        if err:
            _hip.check(err, "sgmcmc_conv50_bwd")
        if defer:
            _pending.append((scratch, dw, slabs.value, 9))
            return dx, dw.view(dw.shape)
        return dx, dw


def conv50(x, w):
    "conv2d(x, w, padding=1) for x [N, 50, 14, 14], w [50, 50, 3, 3]"
    return _Conv50.apply(x, w)


# ---- ... and the second layer with its tail (csrc/conv50_hip.inc, conv_pool_kernel / bwd_pool_kernel) ---------------
def conv50_pool_supported(x, w, bias, stride, padding, dilation, groups):
    "``conv50_supported`` + a float32 bias (or none) for the fused conv -> + bias -> ReLU -> MaxPool2d(2)"
    return (CONV_POOL and conv50_supported(x, w, None, stride, padding, dilation, groups)
            and (bias is None or (bias.dtype == torch.float32 and tuple(bias.shape) == (50,) and bias.is_cuda)))


class _Conv50Pool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias):
        _note_use(w) if bias is None else _note_use(w, bias)
        x, w = x.contiguous(), w.contiguous()
        n = x.shape[0]
        pooled = torch.empty((n, 50, 7, 7), dtype=torch.float32, device=x.device)
        code = torch.empty((n, 50, 7, 7), dtype=torch.uint8, device=x.device)
        # the forward launch also leaves the weights as the data gradient reads them (transposed, taps flipped)
        wT = torch.empty_like(w) if any(ctx.needs_input_grad) else None
        err = _hip.lib().sgmcmc_conv50_pool_fwd(x.data_ptr(), w.data_ptr(), 0 if bias is None else bias.data_ptr(),
                                                pooled.data_ptr(), code.data_ptr(), 0 if wT is None else wT.data_ptr(),
                                                n, _stream())
        if err:
            _hip.check(err, "sgmcmc_conv50_pool_fwd")
        ctx.save_for_backward(x, w, bias, code)
        ctx.wT = wT
        return pooled

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dpooled):
        lib = _hip.lib()
        x, w, bias, code = ctx.saved_tensors
        want_b = bias is not None and ctx.needs_input_grad[2]
        dpooled = dpooled.contiguous()
        n = x.shape[0]
        scratch = torch.empty(lib.sgmcmc_conv50_pool_scratch_floats(n), dtype=torch.float32, device=x.device)
        dx, dw = torch.empty_like(x), torch.empty_like(w)
        db = torch.empty_like(bias) if want_b else None
        want_w = ctx.needs_input_grad[1]
        defer = want_w and _may_defer(w) and (not want_b or _may_defer(bias))
        slabs = ctypes.c_int(0)
        if defer:
            torch.autograd.Variable._execution_engine.queue_callback(_flush_pending)
        err = lib.sgmcmc_conv50_pool_bwd(x.data_ptr(), ctx.wT.data_ptr(), dpooled.data_ptr(), code.data_ptr(),
                                         dx.data_ptr(), dw.data_ptr(), 0 if db is None else db.data_ptr(),
                                         scratch.data_ptr(), n, int(want_b), ctypes.byref(slabs) if defer else None,
                                         _stream())
        if err:
            _hip.check(err, "sgmcmc_conv50_pool_bwd")
        dx = dx if ctx.needs_input_grad[0] else None
        if not want_w:
            return dx, None, db
        if defer:
            P = slabs.value
            _pending.append((scratch[:P * 22500], dw, P, 9))
            if want_b:
                _pending.append((scratch[P * 22500:], db, P, 1))
            return dx, dw.view(dw.shape), None if db is None else db.view(db.shape)
        return dx, dw, db


def conv50_pool(x, w, bias=None):
    "max_pool2d(relu(conv2d(x, w, bias, padding=1)), 2) for x [N, 50, 14, 14], w [50, 50, 3, 3]"
    return _Conv50Pool.apply(x, w, bias)
