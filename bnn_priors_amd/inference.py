"""Cyclical SG-MCMC drivers on the HIP samplers: ``SGLDRunner``, ``VerletSGLDRunner``,
``HMCRunner``.  Drop-in for ``bnn_priors/inference.py`` (reference :9-374): same
constructor, ``run(progressbar)``, ``get_samples()``, metric keys and step numbering
(quirks of SURVEY.md Appendix B are reproduced and marked ``# quirk``).

What differs, by design (DESIGN.md "Runner"):
* the gradient clamp (reference :219-220, one ``clamp_`` launch per tensor) is applied
  in flight by the fused step kernel (``optimizer.grad_clamp``);
* nothing is read back from the device on ordinary steps.  The reference's
  per-step ``isnan(potential).item()`` (:221) becomes a device-side non-finite flag
  that is tested whenever metrics are stored and at every epoch end, so
  "Potential is NaN" is still raised, at most ``metrics_skip`` steps later;
* minibatches of a ``TensorDataset`` are cut on the device from one permutation per
  epoch, drawn with exactly the RNG calls ``RandomSampler`` would make, instead of
  collating 128 single-example tensors per batch.
"""
import contextlib
import math
import os

import torch

from . import mcmc
from .evaluation import evaluate_model
from .schedule import get_cosine_schedule

__all__ = ("SGLDRunner", "VerletSGLDRunner", "HMCRunner")


class LambdaSchedule:
    """``torch.optim.lr_scheduler.LambdaLR`` for one multiplicative schedule, without its
    per-call bookkeeping (~10 us of Python per step): ``group['lr'] = base_lr * fn(k)`` with
    k = 0 at construction and +1 per ``step()`` -- the same expression and therefore the same
    doubles as LambdaLR (torch/optim/lr_scheduler.py, LambdaLR.get_lr)."""

    def __init__(self, optimizer, lr_lambda):
        self.optimizer, self.fn = optimizer, lr_lambda
        for g in optimizer.param_groups:
            g.setdefault('initial_lr', g['lr'])
        self.base_lrs = [g['initial_lr'] for g in optimizer.param_groups]
        self.last_epoch = -1
        self.step()

    def step(self):
        self.last_epoch += 1
        f = self.fn(self.last_epoch)
        for g, base in zip(self.optimizer.param_groups, self.base_lrs):
            g['lr'] = base * f

    def get_last_lr(self):
        return [g['lr'] for g in self.optimizer.param_groups]


def _is_hmc(optimizer):
    "momentum is fully refreshed before every initial step for HMC only (inference.py:312-315)"
    return isinstance(optimizer, mcmc.HMC) or getattr(optimizer, "is_hmc", False)


@contextlib.contextmanager
def _serial_cpu():
    """The host-side tensor work of a batch source (a permutation of N indices, a seed draw) with ONE intra-op thread.
    torch sizes its CPU thread pool by the machine (256 on the MI355X hosts) whatever share of it the process may use: a
    parallel region that wakes such a pool in the middle of a pass stalled the launching thread for 45-100 ms now and
    then (the convolutional classifier's exact pass: 31 ms -> 77-126 ms, every other pass; OMP_NUM_THREADS <= 8: gone)."""
    old = torch.get_num_threads()
    if old > 1:
        torch.set_num_threads(1)
    try:
        yield
    finally:
        if old > 1:
            torch.set_num_threads(old)


class LazyBatch:
    """A minibatch that is not gathered yet: rows ``idx`` (device int64) of a device-resident data set, with the
    augmentation draw of its traversal.  ``materialize()`` gathers it into new tensors -- what the batch source would have
    yielded; ``stage(...)`` gathers it straight into a captured step's static inputs in the SAME launch that copies the
    step's argument block and runs the previous transition's deferred bookkeeping (csrc/augment_hip.inc,
    gather_stage_kernel): one launch between two graph replays instead of gather + label index_select + staging copy."""
    __slots__ = ("src", "idx", "draw")

    def __init__(self, src, idx, draw):
        self.src, self.idx, self.draw = src, idx, draw

    def __len__(self):
        return self.idx.numel()

    @property
    def shapes(self):
        "(x shape, y shape) of the materialised minibatch"
        return (len(self),) + tuple(self.src.x.shape[1:]), (len(self),) + tuple(self.src.y.shape[1:])

    def materialize(self):
        s = self.src
        if s.augment is not None:
            return s.augment.gather(s.x, self.idx, self.draw), s.y.index_select(0, self.idx)
        return s.x.index_select(0, self.idx), s.y.index_select(0, self.idx)

    def stageable(self, x_dst, y_dst):
        s = self.src
        # (gather_stage_kernel reads rows as data[idx[i]] / labels[idx[i]] with unit strides, everything on ONE device)
        dev = s.x.device
        return (s.x.dtype == torch.float32 and s.x.is_contiguous() and s.y.dtype == torch.int64 and s.y.dim() == 1
                and s.y.is_contiguous() and self.idx.is_contiguous() and self.idx.dtype == torch.int64
                and s.y.device == dev and self.idx.device == dev and x_dst.device == dev and y_dst.device == dev
                and x_dst.dtype == torch.float32 and y_dst.dtype == torch.int64 and x_dst.is_contiguous()
                and y_dst.is_contiguous() and (tuple(x_dst.shape), tuple(y_dst.shape)) == self.shapes)

    def stage(self, x_dst, y_dst, jobs, layout, pending, stream):
        """x_dst / y_dst <- this minibatch; ``jobs``: up to three (src_ptr, dst_ptr, bytes) plain copies riding along;
        ``pending`` (with ``layout``): a transition whose deferred bookkeeping runs in the same launch, or None"""
        import ctypes
        from . import _hip
        s, aug = self.src, self.src.augment
        shape = tuple(s.x.shape[1:])
        if aug is not None and len(shape) == 3:
            c, h, w = shape
            pad, flip, seed, strm = aug.pad, int(aug.flip), aug.seed & (2 ** 64 - 1), aug.stream
            fill = 0
            if aug.fill is not None:
                if aug.fill.numel() != c:          # (as RandomCropFlip.gather checks: one fill value per channel)
                    raise ValueError(f"RandomCropFlip.fill has {aug.fill.numel()} values for {c} channels")
                if aug.fill.device != s.x.device or aug.fill.dtype != torch.float32 or not aug.fill.is_contiguous():
                    aug.fill = aug.fill.to(device=s.x.device, dtype=torch.float32).contiguous()
                fill = aug.fill.data_ptr()
        else:
            row = 1
            for d in shape:
                row *= d
            c, h, w, pad, flip, seed, strm, fill = 1, 1, row, 0, 0, 0, 0, 0
        G = _hip.Gather(data=s.x.data_ptr(), labels=s.y.data_ptr(), idx=self.idx.data_ptr(), out=x_dst.data_ptr(),
                        labels_out=y_dst.data_ptr(), fill=fill, batch=len(self), channels=c, height=h, width=w, pad=pad,
                        flip=flip, seed=seed, draw=int(self.draw or 0), stream=strm, reserved=0)
        n = len(jobs)
        arr = lambda vals, t: (t * max(n, 1))(*vals) if n else None
        err = _hip.lib().sgmcmc_gather_stage(ctypes.byref(G), arr([j[0] for j in jobs], ctypes.c_void_p),
                                             arr([j[1] for j in jobs], ctypes.c_void_p),
                                             arr([j[2] for j in jobs], ctypes.c_int64), n,
                                             ctypes.byref(layout) if pending is not None else None,
                                             ctypes.byref(pending) if pending is not None else None, stream)
        if err:
            _hip.check(err, "sgmcmc_gather_stage")


# minibatches handed to a captured step as LazyBatch objects (one launch between two replays); 0: gathered tensors
LAZY_BATCHES = True          # (module attribute, not an environment switch)


class _BatchSource:
    """Yields the minibatches ``dataloader`` would yield, in the same order and with
    the same RNG consumption, but sliced on the device when the dataset allows."""

    def __init__(self, dataloader, device):
        self.dl, self.device = dataloader, device
        ds = dataloader.dataset
        samplers = (torch.utils.data.RandomSampler, torch.utils.data.SequentialSampler)
        self.fast = (hasattr(ds, "tensors") and len(ds.tensors) == 2
                     and type(dataloader.sampler) in samplers
                     and dataloader.batch_size is not None and dataloader.num_workers == 0
                     and dataloader.collate_fn is torch.utils.data.default_collate)
        # a data set that augments on read (augment.AugmentedTensorDataset): whole minibatches are gathered
        # and augmented by one kernel launch, with a fresh draw counter per traversal
        self.augment = getattr(ds, "augment", None) if self.fast else None
        if self.fast:
            self.x, self.y = (t.to(device) for t in ds.tensors)
            if self.augment is not None:
                self.x = self.x.contiguous()
            if self.y.dim() == 1 and not self.y.is_contiguous():
                # a label column of a wider table (a strided view that .to(device) keeps as it is): the staging kernel
                # reads labels with unit stride, so the source owns a packed copy
                self.y = self.y.contiguous()

    def __len__(self):
        return len(self.dl)

    # ---- minibatches produced straight into a consumer's buffers -------------------------------------------------
    # ``with src.filling(provider):`` -- while the block iterates this source, every minibatch of ``rows`` rows is
    # written into the (x_dst, y_dst) pair that ``provider(rows)`` returns (contiguous tensors of the batch's shape and
    # dtype, e.g. slices of a captured graph's static inputs) and those are what the iteration yields; None from the
    # provider (or no provider) = freshly allocated tensors as always.  Same minibatches, same RNG consumption.
    # ``group=(G, group_provider)``: whenever G consecutive full-size minibatches are about to be produced and
    # ``group_provider(G * rows)`` returns an (x_dst, y_dst) pair of G * rows rows, ALL of them are gathered by one launch
    # (the rows are consecutive in the traversal order; the augmentation is keyed by data-set row and traversal, not by
    # position in a launch: the same bytes as G gathers) and yielded one by one as views of that pair.
    _provider = None
    _group = None

    @contextlib.contextmanager
    def filling(self, provider, group=None):
        old, self._provider, self._group = (self._provider, self._group), provider, group
        try:
            yield self
        finally:
            self._provider, self._group = old

    def _group_dst(self, i, n, bs):
        "(G, x_dst, y_dst) when a whole group of full-size minibatches starting at row i can be gathered at once, else None"
        if self._group is None:
            return None
        G, provider = self._group
        if G <= 1 or i + G * bs > (n // bs) * bs:
            return None
        d = provider(G * bs)
        return None if d is None else (G, d[0], d[1])

    def n_full_batches(self):
        """how many of this source's minibatches have the full batch size -- and come FIRST, in every traversal (the
        ragged one, if any, is the last: what a batch sampler over a map-style set yields); None: unknown"""
        if not self.fast:
            return None
        return len(self.dl.dataset) // self.dl.batch_size

    def example(self):
        "a full-size (x, y) minibatch of this source's shapes and dtypes without iterating it (None: unknown)"
        if not self.fast or len(self.dl.dataset) < self.dl.batch_size:
            return None
        bs = self.dl.batch_size
        return self.x[:bs], self.y[:bs]

    def _dst(self, rows):
        d = self._provider(rows) if self._provider is not None else None
        return (None, None) if d is None else d

    def _permutation(self):
        """(permutation, generator) exactly as torch/utils/data/sampler.py RandomSampler.__iter__
        draws them; ``_exhausted`` below replays what it does when the iteration runs out."""
        s, n = self.dl.sampler, len(self.dl.dataset)
        if isinstance(s, torch.utils.data.SequentialSampler):
            return None, None
        if s.replacement or s._num_samples is not None:
            return torch.tensor(list(iter(s)), dtype=torch.int64), None
        if s.generator is None:
            seed = int(torch.empty((), dtype=torch.int64).random_().item())
            gen = torch.Generator()
            gen.manual_seed(seed)
        else:
            gen = s.generator
        return torch.randperm(n, generator=gen), gen

    def _exhausted(self, gen):
        """RandomSampler ends with ``randperm(n)[: num_samples % n]`` -- an empty slice of a SECOND
        permutation, drawn when the iteration is exhausted.  With a generator shared across passes
        (the reject runner's per-cycle generator) that draw decides the batch composition of the
        next pass, which BatchNorm nets feel; reproduce it."""
        if gen is not None:
            with _serial_cpu():
                torch.randperm(len(self.dl.dataset), generator=gen)

    def __iter__(self):
        return self._iterate(False)

    def index_batches(self):
        """same minibatches, same RNG consumption, but named by row index (fused_dense.IndexBatch)
        so that a kernel can gather them itself; falls back to tensors when not ``fast``"""
        return self._iterate(True)

    def _stage_into(self, idx, draw, xd, yd):
        "rows ``idx`` (+ labels) gathered into (xd, yd) by one launch (LazyBatch.stage); False: not possible, gather as usual"
        if not self.x.is_cuda:
            return False
        lb = LazyBatch(self, idx, draw)
        if not lb.stageable(xd, yd):
            return False
        lb.stage(xd, yd, [], None, None, torch.cuda.current_stream(self.x.device).cuda_stream)
        return True

    def lazy_batches(self):
        """same minibatches, same RNG consumption, but as ``LazyBatch`` objects (x still in the data set): the consumer
        gathers them where it wants them (``LazyBatch.stage``); falls back to tensors when not ``fast``"""
        return self._iterate(False, lazy=True)

    def _iterate(self, by_index, lazy=False):
        if not self.fast:
            for x, y in self.dl:
                yield x.to(self.device), y.to(self.device)
            return
        n, bs = len(self.dl.dataset), self.dl.batch_size
        # DataLoader.__iter__ draws a worker base seed before the sampler runs
        # (torch/utils/data/dataloader.py, _BaseDataLoaderIter.__init__): keep the global
        # RNG stream aligned with a run that iterates the DataLoader itself
        with _serial_cpu():
            torch.empty((), dtype=torch.int64).random_(generator=self.dl.generator)
            perm, gen = self._permutation()
        stop = n - n % bs if self.dl.drop_last else n
        if self.augment is not None:
            draw = self.dl.dataset.next_draw()
            order = perm if perm is not None else torch.arange(n)
            order = order.to(self.device)
            i = 0
            while i < stop:
                grp = None if lazy else self._group_dst(i, n, bs)
                if grp is not None and self._stage_into(order[i:i + grp[0] * bs], draw, grp[1], grp[2]):
                    for j in range(grp[0]):         # (G minibatches gathered by ONE launch)
                        yield grp[1][j * bs:(j + 1) * bs], grp[2][j * bs:(j + 1) * bs]
                    i += grp[0] * bs
                    continue
                idx = order[i:i + bs]
                i += bs
                if lazy:
                    yield LazyBatch(self, idx, draw), None
                    continue
                xd, yd = self._dst(idx.numel())
                if xd is not None and self._stage_into(idx, draw, xd, yd):
                    yield xd, yd                    # (images + labels gathered into the consumer's buffers by ONE launch)
                    continue
                yield (self.augment.gather(self.x, idx, draw, out=xd),
                       self.y.index_select(0, idx) if yd is None else torch.index_select(self.y, 0, idx, out=yd))
            self._exhausted(gen)
            return
        if by_index:
            from .fused_dense import IndexBatch
            host = (perm if perm is not None else torch.arange(n)).numpy()
            for i in range(0, stop, bs):
                yield IndexBatch(host[i:i + bs], self.x, self.y), None
            self._exhausted(gen)
            return
        if perm is not None:
            perm = perm.to(self.device)
        if lazy:
            order = perm if perm is not None else torch.arange(n, device=self.device)
            for i in range(0, stop, bs):
                yield LazyBatch(self, order[i:i + bs], None), None
            self._exhausted(gen)
            return
        i = -bs
        skip_to = 0
        while i + bs < stop:
            i += bs
            if i < skip_to:
                continue
            grp = self._group_dst(i, n, bs) if perm is not None else None
            if grp is not None and self._stage_into(perm[i:i + grp[0] * bs], None, grp[1], grp[2]):
                for j in range(grp[0]):
                    yield grp[1][j * bs:(j + 1) * bs], grp[2][j * bs:(j + 1) * bs]
                skip_to = i + grp[0] * bs
                continue
            xd, yd = self._dst(min(bs, n - i))
            if perm is None:
                if xd is None:
                    yield self.x[i:i + bs], self.y[i:i + bs]
                else:
                    yield xd.copy_(self.x[i:i + bs]), yd.copy_(self.y[i:i + bs])
            else:
                idx = perm[i:i + bs]
                if xd is None:
                    yield self.x.index_select(0, idx), self.y.index_select(0, idx)
                elif self._stage_into(idx, None, xd, yd):
                    yield xd, yd
                else:
                    yield torch.index_select(self.x, 0, idx, out=xd), torch.index_select(self.y, 0, idx, out=yd)
        self._exhausted(gen)


class SGLDRunner:
    """Stochastic Gradient Langevin Dynamics for posterior sampling; arguments as the
    reference (bnn_priors/inference.py:10-58).  Extra keyword-only arguments:
    ``seed`` / ``chain_id`` for the in-kernel Philox noise."""
    _OPTIMIZER = mcmc.SGLD

    def __init__(self, model, dataloader, dataloader_test, epochs_per_cycle, warmup_epochs,
                 sample_epochs, learning_rate=1e-2, skip=1, metrics_skip=1, temperature=1.,
                 data_mult=1., momentum=0., sampling_decay=True, grad_max=1e6, cycles=1,
                 precond_update=None, metrics_saver=None, model_saver=None, reject_samples=False,
                 *, seed=None, chain_id=0, use_graph=True):
        self.model, self.dataloader, self.dataloader_test = model, dataloader, dataloader_test
        assert warmup_epochs >= 0
        assert sample_epochs >= 0
        assert epochs_per_cycle >= warmup_epochs + sample_epochs
        self.epochs_per_cycle = epochs_per_cycle
        self.descent_epochs = epochs_per_cycle - warmup_epochs - sample_epochs
        self.warmup_epochs, self.sample_epochs = warmup_epochs, sample_epochs
        self.skip, self.metrics_skip = skip, metrics_skip
        self.num_samples = sample_epochs // skip
        assert sample_epochs % skip == 0
        self.learning_rate, self.temperature = learning_rate, temperature
        self.eff_num_data = len(dataloader.dataset) * data_mult
        self.momentum, self.sampling_decay, self.grad_max = momentum, sampling_decay, grad_max
        self.cycles, self.precond_update = cycles, precond_update
        self.metrics_saver, self.model_saver = metrics_saver, model_saver
        if model_saver is None:
            self._samples = {
                name: torch.zeros(torch.Size([self.num_samples * cycles]) + t.shape, dtype=t.dtype)
                for name, t in model.state_dict().items()}
            self._samples["steps"] = torch.zeros(torch.Size([self.num_samples * cycles]),
                                                 dtype=torch.int64)
        self.param_names, self._params = zip(*model.named_parameters())
        self.reject_samples = reject_samples
        self.seed, self.chain_id = seed, chain_id
        self._device = self._params[0].device
        # ``use_graph``: run ordinary steps through a captured hipGraph (graphed.py) or, for the
        # dense classifier on a device-resident data set, the fused 3-launch step (fused_dense.py)
        self.use_graph = use_graph
        self._graphed = None
        self._fused = None
        self._rows, self._draining = [], False    # metric rows whose read-back is in flight

    # ------------------------------------------------------------------ factories
    def _sampler_kwargs(self):
        return dict(seed=self.seed, chain_id=self.chain_id, grad_clamp=self.grad_max or 0.0)

    def _make_optimizer(self, params):
        assert self.reject_samples is False, "SGLD cannot reject samples"
        return mcmc.SGLD(params=params, lr=self.learning_rate, num_data=self.eff_num_data,
                         momentum=self.momentum, temperature=self.temperature,
                         **self._sampler_kwargs())

    def _make_scheduler(self, optimizer):
        # inference.py:96-108
        if self.sampling_decay is True or self.sampling_decay == "cosine":
            schedule = get_cosine_schedule(len(self.dataloader) * self.epochs_per_cycle)
            return LambdaSchedule(optimizer, schedule)
        elif self.sampling_decay is False or self.sampling_decay == "stairs":
            return torch.optim.lr_scheduler.StepLR(optimizer, 150 * len(self.dataloader), gamma=0.1)
        elif self.sampling_decay == "flat":
            return torch.optim.lr_scheduler.StepLR(optimizer, 2 ** 30, gamma=1.0)
        raise ValueError(f"self.sampling_decay={self.sampling_decay}")

    def _is_sampling_epoch(self, epoch):
        epoch = epoch % self.epochs_per_cycle
        k = epoch - (self.descent_epochs + self.warmup_epochs)
        return (0 <= k) and (k % self.skip == 0)

    def _batches(self):
        try:
            return self._batch_source
        except AttributeError:
            self._batch_source = _BatchSource(self.dataloader, self._device)
            return self._batch_source

    def _fused_dense(self):
        """the 3-kernel fused step (fused_dense.py) when the model is the dense classifier and the
        data set is device resident; None otherwise"""
        if not self.use_graph or self._fused is False:
            return None
        if self._fused is None:
            from .fused_dense import FusedDenseLeapfrog
            src = self._batches()
            pot = self._potential()
            if (getattr(src, "fast", False) and pot.fast
                    and FusedDenseLeapfrog.supported(pot, self.optimizer)):
                self._fused = FusedDenseLeapfrog(pot, self.optimizer, src.x, src.y)
            else:
                self._fused = False
                return None
        return self._fused

    def _hot_batches(self):
        "minibatches of one epoch for the leapfrog loop: by index when the fused step can gather them"
        src = self._batches()
        if self._fused_dense() is not None and hasattr(src, "index_batches"):
            return src.index_batches()
        if (LAZY_BATCHES and self.use_graph and self._graphed is not False and getattr(src, "fast", False)
                and hasattr(src, "lazy_batches")):
            return src.lazy_batches()       # gathered inside the captured step's one staging launch (LazyBatch.stage)
        return iter(src)

    def _graph_for(self, x, y):
        """the captured graph if this step can use it (fused potential, matching batch shape)"""
        if not self.use_graph or self._graphed is False:
            return None
        if self._graphed is None:
            pot = self._potential()
            # the graphed metric rows read the transition's energy total from scalars[3], which only the
            # single-workgroup ("small") finalize writes: arenas beyond 4096 chunks keep the eager path
            if (not pot.fast or pot.leftover or len(self.optimizer.param_groups) != 1
                    or not self.optimizer.engine.small_finalize):
                self._graphed = False
                return None
            from .graphed import GraphedLeapfrog
            self._graphed = GraphedLeapfrog(pot, self.optimizer, x, y)
        return self._graphed if self._graphed.matches(x, y) else None

    def _graph_for_lazy(self, batch):
        "the captured graph for a LazyBatch it can gather into its static inputs itself, else None"
        if not self.use_graph or self._graphed is False:
            return None
        if self._graphed is None:
            x, y = batch.materialize()          # (the capture needs tensors once)
            if self._graph_for(x, y) is None:
                return None
        g = self._graphed
        if g in (None, False) or g.shape != batch.shapes or not batch.stageable(g.x, g.y):
            return None
        return g

    @staticmethod
    def _tensors_of(x, y):
        return x.materialize() if y is None else (x, y)

    def _fast_plain_step(self, x, y, store, log_row=None, want_acc=False):
        """Gradient + ordinary sampler ``step`` through the fastest available path.  Returns
        (handled, x, y): ``handled`` False means the caller runs the eager path on the returned
        tensors.  On metric steps ``log_row(r)`` is called with r = dict(loss, acc, log_prior,
        potential, energy) -- LATER: the read-back is asynchronous (one copy into a pinned slot,
        guarded by an event) so that logging never stalls the launch pipeline; rows are drained
        in order before anything else logs, evaluates or changes the sampler state."""
        by_index = y is None          # an IndexBatch from _BatchSource.index_batches(), or a LazyBatch
        lazy = by_index and isinstance(x, LazyBatch)
        if lazy:
            by_index = False
            graphed = self._graph_for_lazy(x)
            if graphed is None:
                x, y = x.materialize()
                graphed = self._graph_for(x, y)
        else:
            if by_index and self._fused_dense() is None:
                x, y = x.materialize()
                by_index = False
            graphed = self._fused_dense() if by_index else self._graph_for(x, y)
        if graphed is None:
            return False, x, y
        if not store:
            if want_acc:
                # the minibatch accuracy of THIS forward pass (no second forward: BatchNorm buffers
                # must see each batch once): metrics-variant replay, sampler metrics off
                r = (graphed.replay(x.idx, metrics=True, idx_ptr=x.ptr, calc_metrics=False) if by_index
                     else graphed.replay(x, y, metrics=True, calc_metrics=False))
                self.optimizer.engine._state_host = None
                self._drain_rows()            # older metric rows log (and set _last_acc) first
                self._last_acc = r["acc"]
            elif by_index:
                graphed.replay(x.idx, idx_ptr=x.ptr)
            else:
                graphed.replay(x, y)
            return True, x, y
        row = (graphed.replay(x.idx, metrics=True, idx_ptr=x.ptr, wait=False) if by_index
               else graphed.replay(x, y, metrics=True, wait=False))
        self._drain_rows(block=False)
        self._rows.append((row, log_row))
        return True, x, y

    def _drain_rows(self, block=True):
        "log the metric rows whose read-back has landed (all of them if ``block``), oldest first"
        rows = self._rows
        if not rows or self._draining:
            return
        self._draining = True
        try:
            eng = self.optimizer.engine
            while rows and (block or rows[0][0].ready()):
                row, log_row = rows.pop(0)
                r, state = row.get()
                if r["nonfinite"]:
                    eng.scalars[1].zero_()
                    raise ValueError("Gradient is not finite" if self.optimizer.raise_on_nan
                                     else "Potential is NaN")
                r["potential"] = r["loss"] - r["log_prior"] / self.eff_num_data
                eng._state_host = state          # what store_metrics reads the per-tensor values from
                try:
                    log_row(r)
                finally:
                    eng._state_host = None
        finally:
            self._draining = False

    # ------------------------------------------------------------------ the run
    def run(self, progressbar=False):
        "inference.py:110-187"
        for _ in self.run_iter():
            pass

    def run_iter(self):
        """``run()`` as a generator that yields after every minibatch step (and once per epoch end): what
        ``multichain.run_on_streams`` interleaves to drive several chains on several HIP streams of one GPU from
        one process.  Exhausting it IS ``run()``."""
        self.optimizer = self._make_optimizer(self._params)
        self.optimizer.sample_momentum()
        self.scheduler = self._make_scheduler(self.optimizer)
        self.metrics_saver.add_scalar("test/log_prob", math.nan, step=-1)
        self.metrics_saver.add_scalar("test/acc", math.nan, step=-1)

        step = -1
        for cycle in range(self.cycles):
            for epoch in range(self.epochs_per_cycle):
                for g in self.optimizer.param_groups:
                    g['temperature'] = 0. if epoch < self.descent_epochs else self.temperature
                for i, (x, y) in enumerate(self._hot_batches()):
                    step += 1
                    store_metrics = (i == 0 or step % self.metrics_skip == 0)
                    initial_step = (step == 0 or (i == 0 and self._is_sampling_epoch(epoch - 1)))
                    self.step(step, x, y, store_metrics=store_metrics, initial_step=initial_step)
                    yield step
                # rows still in flight were produced with the CURRENT preconditioners: log them first
                self._drain_rows()
                if self.precond_update is not None and epoch % self.precond_update == 0:
                    self.optimizer.update_preconditioner()
                self._check_finite()
                state_dict = self.model.state_dict()
                if self._is_sampling_epoch(epoch):
                    self._save_sample(state_dict, cycle, epoch, step)
                self._evaluate_model(state_dict, step)
                self.metrics_saver.flush(every_s=10)
                yield step
        self._drain_rows()
        # metrics for the last sample (inference.py:182-187)
        x, y = next(iter(self._batches()))
        self.step(step + 1, x, y, store_metrics=True, initial_step=self._is_sampling_epoch(-1))
        self._drain_rows()

    def _save_sample(self, state_dict, cycle, epoch, step):
        k = epoch - (self.descent_epochs + self.warmup_epochs)
        if self.model_saver is None:
            row = self.num_samples * cycle + k // self.skip
            for name, t in state_dict.items():
                self._samples[name][row] = t
        else:
            self.model_saver.add_state_dict(state_dict, step)
            self.model_saver.flush()

    def _evaluate_model(self, state_dict, step):
        self._drain_rows()
        if len(self.dataloader_test) == 0:
            return {}
        self.model.eval()
        one = {k: v.unsqueeze(0) for k, v in state_dict.items()}
        res = evaluate_model(self.model, self.dataloader_test, one, likelihood_eval=True,
                             accuracy_eval=True, calibration_eval=False)
        self.model.train()
        res = {"test/loss": -res["lp_last"], "test/acc": res["acc_last"]}
        for k, v in res.items():
            self.metrics_saver.add_scalar(k, v, step)
        return res

    def _check_finite(self):
        """The reference tests ``isnan(potential)`` after every gradient (inference.py:221) and, for HMC
        (``raise_on_nan=True``, mcmc/hmc.py:25-27), every gradient tensor inside ``step`` (sgld.py:101-104).
        Both are one device-side flag here (the sum of squared gradients of some tensor was not finite),
        tested whenever metrics are stored and at every epoch end: the same ``ValueError``, at most
        ``metrics_skip`` steps later and without a host synchronisation per step."""
        self._drain_rows()
        if self.optimizer.engine.nonfinite_seen():
            raise ValueError("Gradient is not finite" if self.optimizer.raise_on_nan else "Potential is NaN")

    def _potential(self):
        try:
            return self._potential_obj
        except AttributeError:
            from .potential import Potential
            self._potential_obj = Potential(self.model, self.optimizer, self.eff_num_data)
            self._potential_obj.graph_exact = bool(self.use_graph)
            return self._potential_obj

    def _model_potential_and_grad(self, x, y, want_metrics=True):
        """zero grads, g <- grad potential_avg(x, y) (inference.py:215-223).  The likelihood goes
        through autograd, element-wise priors through one fused launch (potential.py).  The
        +-grad_max clamp of :219-220 happens inside the step kernel; the NaN test of :221 is the
        deferred ``_check_finite``.  Returns device tensors (no sync); log_prior, potential and
        acc are None unless ``want_metrics``."""
        return self._potential().minibatch(x, y, want_metrics)

    def step(self, i, x, y, store_metrics, lr_decay=True, initial_step=False):
        "inference.py:225-249"
        lr = self.optimizer.param_groups[0]["lr"]

        def log_row(r, i=i, lr=lr, initial_step=initial_step):
            self.store_metrics(i=i - 1, loss=r["loss"], log_prior=r["log_prior"],   # quirk 6
                               potential=r["potential"], acc=r["acc"], lr=lr,
                               corresponds_to_sample=initial_step)
        handled, x, y = self._fast_plain_step(x, y, store_metrics, log_row)
        if handled:
            if lr_decay:
                self.scheduler.step()
            return None, None, None
        loss, log_prior, potential, acc = self._model_potential_and_grad(x, y, store_metrics)
        self.optimizer.step(calc_metrics=store_metrics)
        lr = self.optimizer.param_groups[0]["lr"]
        if lr_decay:
            self.scheduler.step()
        if store_metrics:
            self._check_finite()
            self.store_metrics(i=i - 1, loss=loss.item(), log_prior=log_prior.item(),   # quirk 6
                               potential=potential.item(), acc=acc.item(), lr=lr,
                               corresponds_to_sample=initial_step)
        return loss, acc, None

    def get_samples(self):
        if self.model_saver is None:
            return {k: v for k, v in self._samples.items() if k != "steps"}
        return self.model_saver.load_samples(keep_steps=False)

    def store_metrics(self, i, loss, log_prior, potential, acc, lr, corresponds_to_sample,
                      delta_energy=None, total_energy=None, rejected=None):
        "inference.py:262-294; one D2H copy serves every per-tensor scalar below"
        self._drain_rows()      # earlier rows first (no-op while a row is being logged)
        add = self.metrics_saver.add_scalar
        t_all = c_all = 0.
        numel = 0
        for n, p in zip(self.param_names, self.optimizer.param_groups[0]["params"]):
            st = self.optimizer.state[p]
            add("preconditioner/" + n, st["preconditioner"], i)
            add("est_temperature/" + n, st["est_temperature"], i)
            add("est_config_temp/" + n, st["est_config_temp"], i)
            t_all += st["est_temperature"] * p.numel()
            c_all += st["est_config_temp"] * p.numel()
            numel += p.numel()
        add("est_temperature/all", t_all / numel, i)
        add("est_config_temp/all", c_all / numel, i)
        add("temperature", self.optimizer.param_groups[0]["temperature"], i)
        add("loss", loss, i)
        add("acc", acc, i)
        add("log_prior", log_prior, i)
        add("potential", potential, i)
        add("lr", lr, i)
        add("acceptance/is_sample", int(corresponds_to_sample), i)
        if delta_energy is not None:
            add("delta_energy", delta_energy, i)
            add("total_energy", total_energy, i)
        if rejected is not None:
            add("acceptance/rejected", int(rejected), i)


class VerletSGLDRunner(SGLDRunner):
    "inference.py:297-365: stochastic-gradient energy differences ('illustrative' M-H)"

    def _make_optimizer(self, params):
        return mcmc.VerletSGLD(params=params, lr=self.learning_rate, num_data=self.eff_num_data,
                               momentum=self.momentum, temperature=self.temperature,
                               **self._sampler_kwargs())

    def step(self, i, x, y, store_metrics, lr_decay=True, initial_step=False):
        if i != 0 and not initial_step:
            lr0 = self.optimizer.param_groups[0]["lr"]

            def log_row(r, i=i, lr=lr0, u0=self._initial_potential, e0=self._total_energy):
                # quirk 13: the reference passes `loss` where a potential is expected
                de = self.optimizer.delta_energy_from_total(r["energy"], u0, r["loss"])
                self.store_metrics(i=i - 1, loss=r["loss"], log_prior=r["log_prior"],
                                   potential=r["potential"], acc=r["acc"], lr=lr, delta_energy=de,
                                   total_energy=e0 + de, rejected=None, corresponds_to_sample=False)
            handled, x, y = self._fast_plain_step(x, y, store_metrics, log_row)
            if handled:
                if lr_decay:
                    self.scheduler.step()
                return None, None, None
        x, y = self._tensors_of(x, y)
        # an M-H point or the very first step always stores metrics (see below)
        want = store_metrics or i == 0 or initial_step
        loss, log_prior, potential, acc = self._model_potential_and_grad(x, y, want)
        lr = self.optimizer.param_groups[0]["lr"]
        opt, is_hmc = self.optimizer, _is_hmc(self.optimizer)
        rejected = delta_energy = None
        if i == 0:
            if is_hmc:
                opt.sample_momentum()
            opt.initial_step(calc_metrics=True, save_state=self.reject_samples)
            if self.reject_samples:
                rejected = False
        elif initial_step:
            opt.final_step(calc_metrics=True)
            delta_energy = opt.delta_energy(self._initial_potential, potential)
            if self.reject_samples:
                rejected, _ = opt.maybe_reject(delta_energy)
            if is_hmc:
                opt.sample_momentum()
            opt.initial_step(calc_metrics=False, save_state=self.reject_samples)
        else:
            opt.step(calc_metrics=store_metrics)

        if i == 0:
            store_metrics = True
            total_energy = delta_energy = opt.delta_energy(0., 0.)                # quirk 13
            self._initial_potential = potential.item()
            self._total_energy = 0.
        elif initial_step:
            store_metrics = True
            self._initial_potential = potential.item()                            # quirk 1
            self._total_energy += delta_energy
            total_energy = self._total_energy
        elif store_metrics:
            delta_energy = opt.delta_energy(self._initial_potential, loss)        # quirk 13
            total_energy = self._total_energy + delta_energy

        if store_metrics:
            self._check_finite()
            self.store_metrics(i=i - 1, loss=loss.item(), log_prior=log_prior.item(),
                               potential=potential.item(), acc=acc.item(), lr=lr,
                               delta_energy=delta_energy, total_energy=total_energy,
                               rejected=rejected, corresponds_to_sample=initial_step)
        if lr_decay:
            self.scheduler.step()
        return loss, acc, delta_energy


class HMCRunner(VerletSGLDRunner):
    def _make_optimizer(self, params):
        # inference.py:367-374
        assert self.temperature == 1.0, "HMC only implemented for temperature=1."
        assert self.momentum == 1.0, "HMC only works with momentum=1."
        assert self.descent_epochs == 0, "HMC not implemented for descent epochs with temp=0."
        kw = self._sampler_kwargs()
        opt = mcmc.HMC(params=params, lr=self.learning_rate, num_data=self.eff_num_data, **kw)
        opt.defer_nan_check = True      # raise_on_nan stays True (hmc.py:25-27); tested by _check_finite
        return opt
