"""hipGraph-captured leapfrog step.

An ordinary leapfrog step of a small BNN is ~30 kernels of a few microseconds each; launched
eagerly the host (Python + ATen dispatch, ~5-10 us per op) is the bottleneck, not the GPU.
``GraphedLeapfrog`` captures ONE graph of

    zero-grad -> forward -> cross-entropy -> backward -> fused prior gradient
              -> fused sampler transition -> per-segment finalize

on static input buffers and replays it per minibatch.  What changes between replays -- the
learning-rate-derived scalars, the Philox draw counter -- lives in a device-resident
``sgmcmc_step_args`` that the captured kernels read when they run
(``sgmcmc_step_indirect``); the host refreshes it with one stream-ordered async copy from a
small ring of pinned slots before each replay, so a replay can never observe the next
step's scalars.

Two graphs are captured: the plain step and the metric step (every ``metrics_skip``-th), which
additionally evaluates the accuracy and the fused log-prior and packs everything the runner
logs -- loss, acc, log-prior, the transition's energy total, the per-segment temperature
estimates -- into one buffer that is read back with a single copy.  M-H points and off-shape
batches keep using the eager path; ``p.grad`` is re-bound to the replayed graph's static
gradient tensors whenever control moves between paths.
"""
import contextlib
import ctypes
import os

import torch

from . import bn as _bn
from . import conv as _conv
from . import pool as _pool

from . import _capture, _hip


def _stageable(dst, src):
    return (src.is_cuda and src.device == dst.device and src.dtype == dst.dtype and src.is_contiguous()
            and src.data_ptr() % 16 == 0 and dst.data_ptr() % 16 == 0 and (dst.numel() * dst.element_size()) % 4 == 0)


def stage_batch(x_dst, x, y_dst, y, args_dst=None, args_pinned=None, engine=None):
    """x_dst <- x, y_dst <- y (and the argument block from its pinned slot) in one launch when the batch is a
    contiguous device tensor of the static inputs' dtype; anything else (a host batch, a strided view, another
    dtype) goes through ``Tensor.copy_``, which converts.  With ``engine``: a transition whose bookkeeping is
    still pending (``engine.pending``) has it executed by the same launch."""
    jobs = []
    for dst, src in ((x_dst, x), (y_dst, y)):
        if src is None:
            continue
        if _stageable(dst, src):
            jobs.append((src.data_ptr(), dst.data_ptr(), dst.numel() * dst.element_size()))
        else:
            dst.copy_(src)
    if args_dst is not None:
        if args_pinned.numel() % 4 == 0:
            jobs.append((args_pinned.data_ptr(), args_dst.data_ptr(), args_pinned.numel()))
        else:
            args_dst.copy_(args_pinned, non_blocking=True)
    if not jobs:
        if engine is not None:
            engine.flush()
        return
    pending = None
    if engine is not None:
        pending, engine.pending = engine.pending, None
    n = len(jobs)
    src, dst, nb = (ctypes.c_void_p * n)(*[j[0] for j in jobs]), (ctypes.c_void_p * n)(*[j[1] for j in jobs]), \
        (ctypes.c_int64 * n)(*[j[2] for j in jobs])
    _hip.check(_hip.lib().sgmcmc_stage_batch(src, dst, nb, n,
                                             ctypes.byref(engine.layout) if pending is not None else None,
                                             ctypes.byref(pending) if pending is not None else None,
                                             torch.cuda.current_stream(x_dst.device).cuda_stream),
               "sgmcmc_stage_batch")


class PendingRow:
    """A metric step's read-back in flight: one async copy of ``report`` (+ optional head
    scalars) into a pinned slot, guarded by an event.  ``get()`` waits for it and returns
    (dict(loss, acc, log_prior, energy, nonfinite), per-segment state array)."""
    __slots__ = ("owner", "buf", "event", "parse")

    def __init__(self, owner, buf, event, parse):
        self.owner, self.buf, self.event, self.parse = owner, buf, event, parse

    def ready(self):
        return self.event.query()

    def get(self):
        self.event.synchronize()
        out = self.parse(self.buf.numpy())
        self.owner._free_slots.append((self.buf, self.event))
        return out


class _ReportSlots:
    "pool of pinned read-back buffers shared by the graphed step objects"

    def _init_slots(self, numel):
        self._slot_numel = numel
        self._free_slots = []

    def _take_slot(self):
        if self._free_slots:
            return self._free_slots.pop()
        return torch.empty(self._slot_numel, dtype=torch.float64).pin_memory(), torch.cuda.Event()


class GraphedLeapfrog(_ReportSlots):
    def __init__(self, potential, optimizer, x_example, y_example, ring=16, warmup=2):
        if not potential.fast or potential.leftover:
            raise ValueError("graph capture needs the fused-prior / cross-entropy potential")
        if len(optimizer.param_groups) != 1:
            raise ValueError("graph capture supports one parameter group")
        self.pot, self.opt, self.eng = potential, optimizer, optimizer.engine
        self.model = potential.model
        dev = self.eng.device
        self.x = torch.empty_like(x_example, device=dev)
        self.y = torch.empty_like(y_example, device=dev)
        self.shape = (tuple(self.x.shape), tuple(self.y.shape))
        nbytes = ctypes.sizeof(_hip.StepArgs)
        self.args_dev = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
        self._slots = [torch.zeros(nbytes, dtype=torch.uint8).pin_memory() for _ in range(ring)]
        # Slot reuse is guarded by an event recorded every ring / 2 steps, not after every staging launch: a
        # hipEventRecord between that launch and the graph replay put 5.8 us of idle GPU between them (tools/
        # gaps_around_staging.py: 5.8 -> 0.0 us; the convolutional classifier's step 123.2 -> 118.3 us).  Step k writes slot
        # k mod ring, last read by step k - ring: free once an event recorded at a step >= k - ring has completed.
        self._ring_events = {}
        self._every = max(1, ring // 2)
        self._k = 0
        # closed-form priors without linked scales: the update kernel adds their gradient in flight (and leaves the
        # log-density partials on metric steps) -- no prior launch in the captured step
        eng = self.eng
        self.inline_prior = (eng.small_finalize and not eng.prior_links and eng.prior_max_kind <= _hip.PRIOR_CAUCHY
                             and eng.layout.dtype == _hip.F32)
        self.defer_finalize = bool(eng.small_finalize)
        self._capture(x_example, y_example, warmup)

    # ------------------------------------------------------------------ args ring
    def _push_args(self, A, x=None, y=None):
        """The step's argument block into its device copy through a pinned ring slot -- and, with it, the
        minibatch into the graph's static inputs: ONE launch (``sgmcmc_stage_batch``) for all three."""
        k, ring, every = self._k, len(self._slots), self._every
        i = k % ring
        self._k += 1
        if k >= ring:
            j = ((k - ring) // every) * every + every - 1       # the first recording step at or after k - ring (< k)
            ev = self._ring_events.get(j)
            if ev is not None:
                ev.synchronize()        # the launch that last read this slot has executed
                for old in [q for q in self._ring_events if q < j]:
                    del self._ring_events[old]
        ctypes.memmove(self._slots[i].data_ptr(), ctypes.addressof(A), ctypes.sizeof(A))
        if y is None and x is not None and hasattr(x, "stage"):
            # a LazyBatch: gathered (and augmented) straight into the static inputs by the launch that also copies the
            # argument block and runs the previous transition's deferred bookkeeping
            eng = self.eng
            pending, eng.pending = eng.pending, None
            slot = self._slots[i]
            jobs = [(slot.data_ptr(), self.args_dev.data_ptr(), slot.numel())] if slot.numel() % 4 == 0 else []
            if not jobs:
                self.args_dev.copy_(slot, non_blocking=True)
            x.stage(self.x, self.y, jobs, eng.layout, pending, torch.cuda.current_stream(self.x.device).cuda_stream)
        else:
            stage_batch(self.x, x, self.y, y, self.args_dev, self._slots[i], self.eng)
        if k % every == every - 1:
            ev = self._ring_events[k] = torch.cuda.Event()
            ev.record()

    def _args(self, calc_metrics=False, variant=None):
        "``variant``: which captured graph runs the step (default: the one named by ``calc_metrics``)"
        variant = calc_metrics if variant is None else variant
        kind, flags, sc = self.opt._plain_step_spec(calc_metrics)
        if self.inline_prior:
            flags |= _hip.INLINE_PRIOR | (_hip.WITH_LOG_PRIOR if calc_metrics else 0)
        if self.defer_finalize and not variant:      # (the metric variant finalizes inside its graph)
            flags |= _hip.DEFER_FINALIZE    # the bookkeeping rides in the next step's staging launch
        return self.eng.make_args(0, kind, flags, self.eng.next_draw(),
                                  grad_clamp=self.opt.grad_clamp, **sc)

    # ------------------------------------------------------------------ capture
    N_HEAD = 6   # packed read-back: loss, acc, scalars[0..3], then the per-segment state array

    def _body(self, capturing, metrics):
        self.opt.zero_grad()
        with _conv.deferring(self.model):
            with _pool.head_loss(self.y, head=_pool.head_of(self.pot.model)):
                f = self.pot._logits(self.x)
            loss = _pool.cross_entropy_backward(f, self.y, want_loss=metrics)
        with torch.no_grad():
            self.opt._prepare_hyper_grads()
        self.eng.refresh(self.opt._preconditioners(), defer_upload=capturing)
        if not self.inline_prior:
            self.eng.prior_grad(self.pot.N, metrics)
        self.eng.step_indirect(self._A_host[metrics], self.args_dev.data_ptr())
        if not metrics:
            return None if loss is None else loss.detach(), None
        with torch.no_grad():
            acc = f.argmax(dim=1).eq(self.y).double().mean()
            packed = torch.cat([loss.detach().double().view(1), acc.view(1), self.eng.scalars[:4],
                                self.eng.state_dev])
        return loss.detach(), packed

    def _snapshot(self):
        eng = self.eng
        return dict(model={k: v.clone() for k, v in self.model.state_dict().items()},
                    m=eng.m.clone(), v=eng.v.clone(), state=eng.state_dev.clone(),
                    scalars=eng.scalars.clone(), draw=eng.draw, k=self._k)

    def _restore(self, snap):
        eng = self.eng
        with torch.no_grad():
            for k, v in self.model.state_dict().items():
                v.copy_(snap["model"][k])
            eng.m.copy_(snap["m"])
            eng.v.copy_(snap["v"])
            eng.state_dev.copy_(snap["state"])
            eng.scalars.copy_(snap["scalars"])
        eng.draw = snap["draw"]
        eng._touch()

    def _capture(self, x, y, warmup):
        dev = self.eng.device
        self.x.copy_(x)
        self.y.copy_(y)
        snap = self._snapshot()
        # (the host copies select kernels: INLINE_PRIOR / WITH_LOG_PRIOR differ between the two variants)
        self._A_host = {m: self._args(calc_metrics=m) for m in (False, True)}
        self._push_args(self._A_host[False])
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._body(False, False)
                self._body(False, True)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self._restore(snap)
        self.graphs, self.static_grads, self.outputs = {}, {}, {}
        for metrics in (False, True):
            self.opt.zero_grad()
            g = torch.cuda.CUDAGraph()
            with _capture.capture(g):
                self.outputs[metrics] = self._body(True, metrics)
            self.graphs[metrics] = g
            self.static_grads[metrics] = [p.grad for p in self.eng.params]
        self._init_slots(self.outputs[True][1].numel())
        self._bound = None
        torch.cuda.synchronize(dev)

    # ------------------------------------------------------------------ replay
    def matches(self, x, y):
        return (tuple(x.shape), tuple(y.shape)) == self.shape

    def replay(self, x, y, metrics=False, wait=True, calc_metrics=None):
        """One leapfrog step.  ``metrics=False``: nothing is read back, returns None.
        ``metrics=True``: the transition also updates the temperature estimates and the fused
        log-prior; returns dict(loss, acc, log_prior, energy, nonfinite) of Python floats after ONE
        device->host copy, and primes the engine's per-segment state cache with the same copy."""
        eng = self.eng
        if self._bound is not metrics:
            for p, g in zip(eng.params, self.static_grads[metrics]):
                p.grad = g
            self._bound = metrics
        elif eng.params[0].grad is not self.static_grads[metrics][0]:
            for p, g in zip(eng.params, self.static_grads[metrics]):
                p.grad = g
        eng.refresh(self.opt._preconditioners())
        # ``metrics`` picks the graph variant (with accuracy / log-prior / packed read-back);
        # ``calc_metrics`` (default: the same) is the sampler's own flag, read from the args at run time
        A = self._args(calc_metrics=metrics if calc_metrics is None else calc_metrics, variant=metrics)
        self._push_args(A, x, y)
        self.graphs[metrics].replay()
        if A.flags & _hip.DEFER_FINALIZE:
            eng.pending = A
        eng._touch()
        eng.energy_ready = True
        if not metrics:
            return None
        eng.metrics_ready = True
        buf, ev = self._take_slot()
        buf.copy_(self.outputs[True][1], non_blocking=True)
        ev.record()
        n_head, n_seg = self.N_HEAD, eng.n_seg

        def parse(v):
            r = dict(loss=float(v[0]), acc=float(v[1]), nonfinite=bool(v[3] != 0.0),
                     log_prior=float(v[4]), energy=float(v[5]))
            return r, v[n_head:].reshape(n_seg, -1).copy()
        row = PendingRow(self, buf, ev, parse)
        if not wait:
            return row
        r, state = row.get()
        eng._state_host = state
        if r["nonfinite"]:
            eng.scalars[1].zero_()
        return r


class GraphedAccumulate:
    """The body of the exact full-data gradient (reference: inference_reject.py:24-30),

        g += grad[ -sum_i log p(y_i | x_i) / N ]     loss += -sum_i log p(y_i | x_i) / N

    captured once on static inputs and replayed per full-size batch: each batch's gradient is added to
    the SAME accumulator tensors (``self.grads``) by one multi-tensor add, the loss into a float64
    device scalar.  An epoch-long pass is then L graph launches and no host read-back;
    off-shape batches (the ragged last one) run the same ops eagerly into the same accumulators.
    BatchNorm nets update their running statistics on every replay exactly as the eager pass does;
    the capture's warm-up runs are undone from a snapshot of the model's buffers.

    ``group`` = G > 1 (round 4): the static inputs hold G minibatches one after the other and ONE replay evaluates all
    of them -- every launch of the chain carries G times the work above the same launch floor.  Training-mode BatchNorm
    statistics stay per minibatch (``bn.grouped``: the reference computes them per minibatch, models/google_resnet.py:
    14-27 under inference_reject.py:18-33); the convolutions, the head and the loss are per image; the groups'
    weight-gradient slabs and (dgamma, dbeta) rows are summed by the pass's one deferred reduction.  Needs ``log_slots``
    (the groups' running-statistics updates are ordered: they are logged and replayed, ``ConcurrentAccumulate``)."""

    def __init__(self, potential, optimizer, x_example, y_example, warmup=2, log_slots=None, group=1, share=None):
        """``log_slots`` (id(running_mean) -> float64 [C, 2] tensor; [G, C, 2] with ``group`` = G > 1): the BatchNorm
        layers leave their batch statistics there instead of advancing their running statistics
        (bn.logging_running_stats) -- for a lane of ``ConcurrentAccumulate``, which advances them afterwards in batch
        order.  ``share``: another GraphedAccumulate whose accumulators (grads, loss) this one adds to."""
        self.pot, self.opt, self.eng = potential, optimizer, optimizer.engine
        self.model = potential.model
        self.log_slots, self.group = log_slots, int(group)
        if self.group > 1 and log_slots is None and any(
                isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and m.track_running_stats for m in self.model.modules()):
            raise _bn.LogModeUnsupported("a grouped accumulate body needs logged BatchNorm statistics")
        dev = self.eng.device
        self.rows = x_example.shape[0]
        rep = lambda t: t.repeat((self.group,) + (1,) * (t.dim() - 1)) if self.group > 1 else t
        self.x = torch.empty_like(rep(x_example), device=dev)
        self.y = torch.empty_like(rep(y_example), device=dev)
        self.shape = (tuple(x_example.shape), tuple(y_example.shape))      # of ONE minibatch
        if share is None:
            self.loss = torch.zeros((), dtype=torch.float64, device=dev)
            self.grads = [torch.zeros_like(p, memory_format=torch.preserve_format) for p in self.eng.params]
        else:
            self.loss, self.grads = share.loss, share.grads
        self.x.copy_(rep(x_example))
        self.y.copy_(rep(y_example))
        # What the warm-up runs and the capture can change is put back afterwards.  In log mode (a lane of
        # ConcurrentAccumulate -- possibly built in the MIDDLE of a pass, with other lanes' replays in flight) that is
        # only the shared accumulators: the bodies neither advance the running statistics nor the batch counters nor
        # touch a parameter, so nothing of the model is snapshotted or written back (a restore would race with the
        # other lanes' reads and could put stale running statistics over ones a log replay has advanced since).
        buffers = ({k: v.clone() for k, v in self.model.state_dict().items()} if log_slots is None else {})
        keep = ([g.clone() for g in self.grads], self.loss.clone()) if share is not None else None
        if share is None:
            self.begin()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._body()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with _capture.capture(self.graph):
            self._body()
        with torch.no_grad():
            for k, v in (self.model.state_dict().items() if buffers else ()):
                v.copy_(buffers[k])
            if keep is not None:        # (the warm-up runs added to accumulators that are somebody else's)
                for g, k in zip(self.grads, keep[0]):
                    g.copy_(k)
                self.loss.copy_(keep[1])
        torch.cuda.synchronize(dev)

    def _logging(self):
        return _bn.logging_running_stats(self.log_slots) if self.log_slots is not None else contextlib.nullcontext()

    def _accumulate(self, x, y, group=1):
        # every batch's gradient lands in FRESH tensors (autograd adopts them: no per-tensor `grad += ...`
        # launches, and the convolution slabs take their one deferred reduction), then ONE multi-tensor add
        # folds them into the accumulators
        params = self.eng.params
        for p in params:
            p.grad = None
        # (several minibatches per launch: the persistent convolutions, csrc/conv2_hip.inc -- see conv.persistent)
        with self._logging(), _bn.grouped(group), _conv.persistent(group > 1 and EXACT_PERSISTENT), \
                _conv.deferring(self.model):
            with _pool.head_loss(y, "sum", self.pot.N, head=_pool.head_of(self.pot.model)):
                f = self.pot._logits(x)
            this = _pool.cross_entropy_backward(f, y, reduction="sum", divide_by=self.pot.N)
        got = [(a, p.grad) for a, p in zip(self.grads, params) if p.grad is not None]   # hyper-parameters: none
        torch._foreach_add_([a for a, _ in got], [g for _, g in got])
        self.loss += this.detach().double()

    def _body(self):
        self._accumulate(self.x, self.y, self.group)

    def matches(self, x, y):
        return (tuple(x.shape), tuple(y.shape)) == self.shape

    def begin(self):
        "zero the accumulators"
        torch._foreach_zero_(self.grads)
        self.loss.zero_()

    def finish(self):
        "the accumulated full-data gradient becomes the parameters' gradient"
        for p, g in zip(self.eng.params, self.grads):
            p.grad = g

    def slot(self, j):
        "the static inputs of minibatch j of the group: (x [B, ...], y [B]) views to be filled by the producer"
        return self.x[j * self.rows:(j + 1) * self.rows], self.y[j * self.rows:(j + 1) * self.rows]

    def owns(self, x, y, j=0):
        "x, y ARE slot j (a producer that filled them in place: nothing to stage)"
        xs, ys = self.slot(j)
        return (x.data_ptr() == xs.data_ptr() and y.data_ptr() == ys.data_ptr() and x.shape == xs.shape
                and x.is_contiguous() and y.is_contiguous())

    def add(self, x, y):
        "one minibatch through a group-of-one body"
        if not self.owns(x, y):
            stage_batch(self.x, x, self.y, y)
        self.graph.replay()

    def add_eager(self, x, y):
        "same accumulation without the graph (any batch shape)"
        self._accumulate(x, y)


# ------------------------------------------------------------------ the exact pass on several streams
EXACT_LANES = int(os.environ.get("SGMCMC_EXACT_LANES", "3"))      # streams (3 with grouped launches: 189 vs 200 ms per googleresnet pass at 2; at one minibatch per launch 2 was the optimum)
# minibatches per launch chain: a number (1: one, as round 3), or "auto": 4 .. 8, preferring a count that divides the
# pass's full-size minibatches (no left-over one-minibatch replays) and deals the groups evenly to the lanes
EXACT_GROUP = os.environ.get("SGMCMC_EXACT_GROUP", "auto")
EXACT_GROUP = EXACT_GROUP if EXACT_GROUP == "auto" else int(EXACT_GROUP)


def pick_group(n_full, rows, lanes, want=None):
    "minibatches per replay for a pass of ``n_full`` full-size minibatches of ``rows`` rows on ``lanes`` streams"
    want = EXACT_GROUP if want is None else want
    if n_full is None:        # a source that cannot announce its full-size minibatches is never run in groups (run()):
        return 1              # no grouped bodies are captured for it -- their HBM and capture time would be spent unused
    cap = max(1, 1024 // max(rows, 1))            # (the fused head + loss launch takes up to 1,024 rows)
    if want != "auto":
        return max(1, min(int(want), cap))
    if not n_full or n_full < 8:
        return max(1, min(4, cap, n_full or 4))
    best, best_key = 1, None
    for g in range(min(8, cap), 0, -1):
        if g < 4 and best_key is not None:
            break
        groups, left = divmod(n_full, g)
        # fewest left-over minibatches, then evenly dealt groups, then the smaller group (less memory; 4 .. 8 measure alike)
        key = (left, (-groups) % max(lanes, 1), g)
        if best_key is None or key < best_key:
            best, best_key = g, key
    return best
# a group's G minibatches gathered (+ cropped / flipped) into the lane's static input by ONE launch instead of G
GROUP_GATHER = True
EXACT_PERSISTENT = True     # grouped bodies on the persistent convolutions
LOG_CAPACITY = 512        # minibatches whose BatchNorm statistics fit in the log before it is replayed and reused


class ConcurrentAccumulate:
    """The exact full-data gradient with its minibatches evaluated on ``lanes`` HIP streams at once, ``group`` of them
    per launch chain: every lane owns a grouped ``GraphedAccumulate`` (G minibatches per replay) and a single one (what
    is left over when the pass's full-size minibatches are not a multiple of G), with their own static inputs and ONE set
    of accumulators per lane; group q goes to lane q mod lanes.  A gradient evaluation is a chain of ~90 dependent
    launches at the launch floor that leaves the GPU partly idle at every boundary: G minibatches per launch put G times
    the work above the same floor, and two independent chains interleave.  The minibatches of this pass are independent
    given the parameters (inference_reject.py:18-33) -- except that training-mode BatchNorm advances its running
    statistics batch by batch: the bodies therefore LOG every layer's batch mean / unbiased variance per minibatch
    (``bn.logging_running_stats``, ``bn.grouped``) and the running statistics are advanced afterwards from the log in
    minibatch order, with the arithmetic of the in-kernel update (same bits as the sequential pass);
    ``num_batches_tracked`` is advanced by the number of minibatches.
    The gradient is the sum of the lanes' accumulators (a fixed order: reproducible; the sequential pass adds the
    minibatches in another order, so the two agree to rounding, not bit for bit).  Off-shape minibatches (the ragged
    last one) are evaluated eagerly, in order, after the lanes have been joined.
    A batch source that can fill buffers in place (``_BatchSource.filling``) writes every minibatch straight into the
    static inputs of the body that evaluates it: no staging copy."""

    def __init__(self, potential, optimizer, x_example, y_example, lanes=2, capacity=None, group=None, n_full=None):
        self.pot, self.opt, self.eng = potential, optimizer, optimizer.engine
        self.model = potential.model
        dev = self.dev = self.eng.device
        self.bn_layers = [m for m in self.model.modules()
                          if isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and m.running_mean is not None]
        if any(m.momentum is None for m in self.bn_layers):
            raise _bn.LogModeUnsupported("cumulative-average BatchNorm")
        self.cmax = max([m.num_features for m in self.bn_layers], default=1)
        self.n_bn = len(self.bn_layers)
        self.group = G = pick_group(n_full, x_example.shape[0], lanes, group)
        # lanes that share a hardware queue run back to back (round 5: the third lane was slower than two for that reason)
        from . import multichain
        # (distinct streams, none of them a chain's main stream: multichain.lanes)
        self.streams = multichain.lanes(lanes, dev, exclude=[torch.cuda.current_stream(dev)])
        self.lanes, self.singles, self.log_cur, self.log_cur1, self.slots1 = [], [], [], [], []
        self.off_shape = {}          # (lane, shapes) -> a captured body for a minibatch of another size (the ragged last one)
        main = torch.cuda.current_stream(dev)
        nb, cm = max(self.n_bn, 1), self.cmax
        for s in self.streams:
            s.wait_stream(main)
            with torch.cuda.stream(s):
                cur1 = torch.zeros((nb, cm, 2), dtype=torch.float64, device=dev)
                slots1 = {id(m.running_mean): cur1[i, :m.num_features] for i, m in enumerate(self.bn_layers)}
                single = GraphedAccumulate(potential, optimizer, x_example, y_example, log_slots=slots1)
                self.singles.append(single)
                self.log_cur1.append(cur1)
                self.slots1.append(slots1)
                if G > 1:
                    cur = torch.zeros((G, nb, cm, 2), dtype=torch.float64, device=dev)
                    slots = {id(m.running_mean): cur[:, i, :m.num_features] for i, m in enumerate(self.bn_layers)}
                    self.lanes.append(GraphedAccumulate(potential, optimizer, x_example, y_example, log_slots=slots,
                                                        group=G, share=single))
                    self.log_cur.append(cur)
            main.wait_stream(s)
        # (a group's minibatches are logged together: the log holds at least one group)
        self.log_all = torch.zeros((max(capacity or LOG_CAPACITY, G), nb, cm, 2), dtype=torch.float64, device=dev)
        self.loss = self.singles[0].loss
        self.count = self.logged_from = 0

    def matches(self, x, y):
        return self.singles[0].matches(x, y)

    def begin(self):
        main = torch.cuda.current_stream(self.dev)
        for s, lane in zip(self.streams, self.singles):
            s.wait_stream(main)
            with torch.cuda.stream(s):
                lane.begin()
        self.count = self.logged_from = 0

    def _room(self, n):
        "make sure the log takes n more minibatches"
        if self.count - self.logged_from + n > self.log_all.shape[0]:      # the log is full: advance and start over
            self._join_and_replay()

    def run(self, batches):
        "every minibatch of ``batches`` (an iterable of (x, y)), fetched under the stream of the lane that evaluates it"
        G, n_lanes = self.group, len(self.streams)
        n_full = batches.n_full_batches() if hasattr(batches, "n_full_batches") else None
        n_groups = n_full // G if (G > 1 and n_full is not None) else 0
        if n_groups and hasattr(batches, "example"):
            # a later pass over minibatches of ANOTHER size than the bodies were captured for (the cached accumulator of
            # potential.exact): no groups -- every minibatch goes through the one-minibatch / off-shape / eager route below
            ex = batches.example()
            if ex is None or not self.lanes[0].matches(*ex):
                n_groups = 0
        target = {"dst": None, "grp": None}
        try:       # (a source that can gather a whole group with one launch: inference._BatchSource)
            filling = batches.filling(lambda rows: target["dst"](rows) if target["dst"] else None,
                                      group=(G, lambda rows: target["grp"](rows) if target["grp"] else None)) \
                if hasattr(batches, "filling") else contextlib.nullcontext()
        except TypeError:
            filling = batches.filling(lambda rows: target["dst"](rows) if target["dst"] else None)
        with filling:
            it = iter(batches)
            turn = 0
            # ---- whole groups: G minibatches per replay, written straight into the lane's static inputs
            for q in range(n_groups):
                k = turn % n_lanes
                turn += 1
                lane = self.lanes[k]
                self._room(G)
                with torch.cuda.stream(self.streams[k]):
                    target["grp"] = (lambda rows: (lane.x, lane.y) if rows == G * lane.rows else None) if GROUP_GATHER else None
                    for j in range(G):
                        target["dst"] = lambda rows, j=j: lane.slot(j) if rows == lane.rows else None
                        x, y = next(it)
                        target["grp"] = None
                        if not lane.matches(x, y):      # (the contract of n_full_batches: the FIRST n_full ones are full)
                            raise RuntimeError("a batch source that announces n_full_batches() full-size minibatches must "
                                               f"yield them first: got {tuple(x.shape)} inside a group of {lane.shape[0]} "
                                               "(sources of another minibatch size are detected through example())")
                        if not lane.owns(x, y, j):
                            xs, ys = lane.slot(j)
                            stage_batch(xs, x, ys, y)
                    target["dst"] = None
                    lane.graph.replay()
                    if self.n_bn:
                        at = self.count - self.logged_from
                        self.log_all[at:at + G].copy_(self.log_cur[k], non_blocking=True)
                    self.count += G
            # ---- what is left: one minibatch per replay, then the off-shape ones eagerly
            while True:
                k = turn % n_lanes
                self._room(1)
                single = self.singles[k]
                with torch.cuda.stream(self.streams[k]):
                    target["dst"] = lambda rows: single.slot(0) if rows == single.rows else None
                    try:
                        x, y = next(it)
                    except StopIteration:
                        break
                    finally:
                        target["dst"] = None
                    body = single if self.matches(x, y) else self._off_shape_body(k, x, y)
                    if body is not None:
                        turn += 1
                        body.add(x, y)
                        if self.n_bn:
                            self.log_all[self.count - self.logged_from].copy_(self.log_cur1[k], non_blocking=True)
                        self.count += 1
                        continue
                # an off-shape minibatch: in order, on the main stream, with the ordinary running-statistics update
                self._join_and_replay()
                x, y = x.to(self.dev), y.to(self.dev)
                torch.cuda.current_stream(self.dev).wait_stream(self.streams[k])       # (x was produced there)
                self.singles[0].log_slots, keep = None, self.singles[0].log_slots
                try:
                    self.singles[0].add_eager(x, y)
                finally:
                    self.singles[0].log_slots = keep
                for s in self.streams:
                    s.wait_stream(torch.cuda.current_stream(self.dev))
        self._join_and_replay()

    def _off_shape_body(self, k, x, y):
        """a captured body for a minibatch of another size on lane k -- the ragged last minibatch of every pass, which ran
        eagerly (~90 launches from Python, 3-4 ms, behind a join of all lanes) until round 4; it is logged and replayed
        like any other minibatch.  At most two shapes per lane are captured; None = evaluate eagerly as before."""
        if not (x.is_cuda and y.is_cuda and x.shape[0] > 0):
            return None
        key = (k, tuple(x.shape), tuple(y.shape), x.dtype)
        body = self.off_shape.get(key)
        if body is None:
            if sum(1 for q in self.off_shape if q[0] == k) >= 2:
                return None
            try:
                body = GraphedAccumulate(self.pot, self.opt, x, y, log_slots=self.slots1[k], share=self.singles[k])
            except RuntimeError:
                body = False
            self.off_shape[key] = body
        return body or None

    def _join_and_replay(self):
        "lanes joined into the current stream; the running statistics advanced by the minibatches logged since the last join"
        main = torch.cuda.current_stream(self.dev)
        for s in self.streams:
            main.wait_stream(s)
        n = self.count - self.logged_from
        if n and self.n_bn:
            _bn.replay_running_stats_many(self.bn_layers, self.log_all, n, main.cuda_stream)
            with torch.no_grad():
                torch._foreach_add_([m.num_batches_tracked for m in self.bn_layers], n)
        self.logged_from = self.count
        for s in self.streams:
            s.wait_stream(main)            # the log may be overwritten / the statistics read from here on

    def finish(self):
        "the lanes' accumulators summed into lane 0's, which become the parameters' gradient"
        main = torch.cuda.current_stream(self.dev)
        for s in self.streams:
            main.wait_stream(s)
        for lane in self.singles[1:]:
            torch._foreach_add_(self.singles[0].grads, lane.grads)
            self.singles[0].loss += lane.loss
        self.singles[0].finish()
