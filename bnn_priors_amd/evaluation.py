"""Posterior-predictive evaluation (SURVEY.md section 8 row f1 / e).

``evaluate_model`` restates ``bnn_priors/exp_utils.py:250-340`` for the
likelihood / accuracy outputs: for every stored sample load it, run the test
set, keep log p(y|x) ``lps[E,N]`` and the normalised logits ``acc[E,N,C]`` in
float64 ON THE DEVICE, then

    lp_ensemble  = mean_n( logsumexp_e lps - log E )
    ensemble logits = logsumexp_e acc - log E ,  acc = argmax == y

``ensemble_across_chains`` is the one exchange step of the multi-chain path:
each rank reduces its own samples to (max_e, sum_e exp(x - max)) and two small
all-reduces (MAX, SUM) over RCCL combine the chains -- N_test*(C+1) doubles,
instead of gathering [E, N, C] from every GPU.
"""
import contextlib
import math
import os
import warnings
import weakref

import torch

from . import _capture

__all__ = ("evaluate_model", "predictive_tables", "ensemble_across_chains", "ensemble_metrics",
           "gather_samples")


def _labels_of(dataloader):
    ds = dataloader.dataset
    if hasattr(ds, "tensors"):
        return ds.tensors[1]
    if hasattr(ds, "targets"):
        return torch.as_tensor(ds.targets)
    raise ValueError("I cannot find the labels in the dataloader.")


def _n_samples(samples):
    return min(len(v) for v in samples.values())


class _NotBatchable(Exception):
    "the grouped (vmap) evaluation cannot run this model / these samples: evaluate sample by sample instead"


# Stack the samples and run ONE grouped (vmap) forward per test batch on the library's batched layers.  OFF by default
# since round 4: once the test set is sliced instead of iterated (``_resident_tensors``) the sample-by-sample pass on
# this package's kernels -- a captured forward per 1,024 rows, ``load_state_dict`` per sample -- is 3-4x faster for all
# three BASELINE nets (10 samples x 10,000 rows, tools/eval_probe.py: googleresnet 141 vs 427 ms, convnet 24 vs 71,
# densenet 15 vs 58), needs no MIOpen kernel search (1.4-5 s on first use) and gives the per-epoch evaluation's bits.
BATCHED = False          # (module attribute, not an environment switch: tests set it)
SAMPLE_GROUP = 32       # samples per grouped forward: bounds the activation memory at E_group * batch images


@contextlib.contextmanager
def _plain_torch_layers():
    """torch.func.vmap batches ATen operators, not this package's custom-kernel autograd functions: inside the
    grouped forward the layers take their ATen route (the grouped evaluation is a throughput problem -- E samples x
    a test batch per launch -- that the library's batched convolutions / GEMMs handle well).
    The switch is process-wide but never observable by another chain: the block runs to completion inside ONE
    ``next()`` of a runner's generator (multichain.run_on_streams interleaves chains only at ``yield`` points, and
    nothing in here yields), on one Python thread."""
    from . import bn, conv, pool, resblock
    mods = (bn, conv, pool, resblock)
    old = [m.ENABLED for m in mods]
    for m in mods:
        m.ENABLED = False
    try:
        yield
    finally:
        for m, v in zip(mods, old):
            m.ENABLED = v


@torch.no_grad()
def _predictive_tables_batched(model, dataloader_test, samples, labels, E, C):
    """All samples at once (reference semantics: exp_utils.py:250-340, one load_state_dict + one pass over the test
    set PER SAMPLE): the stored samples stay stacked [E, ...], ``torch.func.functional_call`` under ``vmap`` runs the
    network once per test batch for a whole group of samples, and the normalisation (log-softmax), the gather of
    log p(y|x) and the fp64 tables are filled on the device without a Python loop over samples."""
    from torch.func import functional_call, vmap
    device = labels.device
    N = labels.shape[0]
    lps = torch.zeros((E, N), dtype=torch.float64, device=device)
    acc = torch.zeros((E, N, C), dtype=torch.float64, device=device)
    names = set(dict(model.named_parameters())) | set(dict(model.named_buffers()))
    state = {k: v[:E].to(device) for k, v in samples.items() if k in names}
    if set(state) != names:
        raise _NotBatchable("the samples do not cover the model's state")
    temp = model.softmax_temp
    if callable(temp) or isinstance(temp, torch.Tensor):
        raise _NotBatchable("the softmax temperature is not a plain number")

    def logits_of(st, x):
        return functional_call(model.net, {k[len("net."):]: v for k, v in st.items()}, (x,))
    grouped = vmap(logits_of, in_dims=(0, None), randomness="error")
    with _plain_torch_layers():
        i = 0
        for bx, by in dataloader_test:
            bx, by = bx.to(device), by.to(device)
            j = i + len(bx)
            for e0 in range(0, E, SAMPLE_GROUP):
                e1 = min(E, e0 + SAMPLE_GROUP)
                try:
                    f = grouped({k: v[e0:e1] for k, v in state.items()}, bx)           # [e, B, C]
                except RuntimeError as exc:
                    # only vmap's own refusals fall back (an operator without a batching rule, in-place writes to
                    # an unbatched tensor, data-dependent control flow); out-of-memory and real bugs propagate
                    msg = str(exc)
                    if any(t in msg for t in ("vmap", "batching rule", "Batching rule", "functorch")):
                        raise _NotBatchable(msg.splitlines()[0]) from exc
                    raise
                logp = torch.log_softmax(f / temp, dim=-1)        # Categorical(logits=...).logits, in the net's dtype
                acc[e0:e1, i:j] = logp
                lps[e0:e1, i:j] = logp.gather(-1, by.view(1, -1, 1).expand(e1 - e0, -1, 1)).squeeze(-1)
            i = j
    return lps, acc


# ---- sample-by-sample evaluation on a captured forward ---------------------------------------------------------
# The per-epoch evaluation (inference.py:199-213) and the stored samples' predictions (exp_utils.py:250-340) run the
# network once per test batch: ~80 kernels launched eagerly from Python, 1.7 ms per batch of which the GPU works 0.3.
# ``_GraphedLogits`` captures ``model.net(x)`` (eval mode: this package's convolution kernels, the running-statistics
# BatchNorm kernel, the fused head) on a static input batch once per (model, batch shape) and replays it per batch;
# ``load_state_dict`` writes into the parameters' own storage, so the graph sees every sample.
EVAL_GRAPH = True


# rows per evaluation forward (0: the loader's own batches)
EVAL_ROWS = 1024


def _resident_tensors(dataloader):
    """(x, y) when the loader is a plain in-order pass over a TensorDataset-like set (``.tensors``, SequentialSampler,
    default collation, no workers): its batches are consecutive slices of those tensors, so the evaluation can slice them
    itself instead of iterating the DataLoader -- which indexes the set item by item and stacks 128 views per batch,
    ~0.3 ms of host time each: 24 of the 28 ms a per-epoch evaluation of 10,000 rows took"""
    ds = getattr(dataloader, "dataset", None)
    t = getattr(ds, "tensors", None)
    if (t is None or len(t) != 2 or getattr(ds, "augment", None) is not None
            or not isinstance(getattr(dataloader, "sampler", None), torch.utils.data.SequentialSampler)
            or dataloader.batch_size is None or dataloader.num_workers != 0
            or dataloader.collate_fn is not torch.utils.data.default_collate):
        return None
    n = len(ds)
    if dataloader.drop_last:
        n -= n % dataloader.batch_size
    return t[0][:n], t[1][:n]


def _row_groups(dataloader, device, rows):
    "the loader's batches, consecutive ones concatenated up to ``rows`` rows (order kept; the last group is what is left)"
    if rows <= 0:
        yield from dataloader
        return
    res = _resident_tensors(dataloader)
    if res is not None:
        # the same groups: whole batches up to ``rows`` rows each, as slices of the set (moved once if it lives on the host)
        # DataLoader.__iter__ draws a worker base seed from the (global) generator before anything else
        # (torch/utils/data/dataloader.py, _BaseDataLoaderIter.__init__): a run that iterates the test loader -- the
        # reference's evaluate_model -- consumes it, and the plain runners seed their next shuffle from the same stream
        torch.empty((), dtype=torch.int64).random_(generator=dataloader.generator)
        bs = dataloader.batch_size
        per = max(bs, rows // bs * bs)
        x, y = _resident_on(dataloader, res, device)
        for i in range(0, x.shape[0], per):
            yield x[i:i + per], y[i:i + per]
        return
    xs, ys, have = [], [], 0
    for bx, by in dataloader:
        if xs and have + len(bx) > rows:
            yield (torch.cat(xs), torch.cat(ys)) if len(xs) > 1 else (xs[0], ys[0])
            xs, ys, have = [], [], 0
        xs.append(bx.to(device))
        ys.append(by.to(device))
        have += len(bx)
    if xs:
        yield (torch.cat(xs), torch.cat(ys)) if len(xs) > 1 else (xs[0], ys[0])


# loader -> (host x, host y, their versions, device, x on the device, y on the device): a host-resident test set is moved
# once per loader.  Weakly keyed by the loader OBJECT (an id() can be reused by the next loader of a loop that builds
# fresh ones) and holding the host tensors it was copied from, so neither their address nor the loader's can come
# back meaning other data; an in-place edit of the host tensors (normalisation after a first evaluation) bumps their
# version counters and invalidates the copy.  Tensors already on the device are never cached (nothing is copied).
_resident_cache = weakref.WeakKeyDictionary()


def _root(t):
    "the tensor a view was sliced from (views share its version counter), else the tensor itself"
    return t._base if t._base is not None else t


def _resident_on(dataloader, res, device):
    x, y = res
    device = torch.device(device)
    if x.device == device and y.device == device:
        return x, y
    c = _resident_cache.get(dataloader)
    if (c is not None and c[0] is _root(x) and c[1] is _root(y)
            and c[2] == (x._version, y._version, tuple(x.shape), tuple(y.shape)) and c[3] == device):
        return c[4], c[5]
    xd, yd = x.to(device), y.to(device)
    _resident_cache[dataloader] = (_root(x), _root(y), (x._version, y._version, tuple(x.shape), tuple(y.shape)),
                                   device, xd, yd)
    return xd, yd


class _GraphedLogits:
    def __init__(self, model, x):
        self.shape, self.dtype = tuple(x.shape), x.dtype
        self.x = torch.empty_like(x)
        self.x.copy_(x)
        dev = x.device
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(2):
                model.net(self.x)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with _capture.capture(self.graph):
            self.out = model.net(self.x)

    def __call__(self, x):
        self.x.copy_(x)
        self.graph.replay()
        return self.out


# model -> {key: _GraphedLogits}.  Kept OUTSIDE the module (a hipGraph is neither picklable nor deep-copyable: a cache
# in model.__dict__ broke copy.deepcopy(model) / torch.save(model) after the first evaluation) and weakly keyed, so the
# graphs and their private memory pools go when the model goes.
_eval_graphs = weakref.WeakKeyDictionary()


def _logits_fn(model, x):
    "model.net(x), through a captured graph when the model is on the GPU in eval mode (cached per batch shape)"
    if not (EVAL_GRAPH and x.is_cuda and not model.training and not torch.is_grad_enabled()):
        return model.net(x)
    cache = _eval_graphs.setdefault(model, {})
    # (the capture holds the storage addresses of the parameters AND of the buffers -- the running statistics the
    # BatchNorm kernels read: a model moved, re-built or given re-registered buffers since gets a new capture)
    key = (tuple(x.shape), x.dtype, tuple(p.data_ptr() for p in model.parameters()),
           tuple(b.data_ptr() for b in model.buffers()))
    g = cache.get(key)
    if g is None:
        if len(cache) >= 4:
            cache.clear()
        try:
            g = cache[key] = _GraphedLogits(model, x)
        except RuntimeError as exc:       # an operator that cannot be captured: stay eager for this model
            warnings.warn(f"evaluation forward not captured ({str(exc).splitlines()[0]}); running it eagerly")
            cache[key] = g = False
    return g(x) if g else model.net(x)


@torch.no_grad()
def predictive_tables(model, dataloader_test, samples):
    """lps [E, N] and acc_data [E, N, C] (float64, on the model's device)."""
    device = next(iter(model.parameters())).device
    labels = _labels_of(dataloader_test).to(device)
    N = labels.shape[0]
    C = int(labels.max().item()) + 1 if labels.dim() == 1 else labels.shape[1]
    E = _n_samples(samples)
    from .models.base import ClassificationModel
    if (BATCHED and E > 1 and isinstance(model, ClassificationModel) and labels.dim() == 1
            and not model.training and device.type == "cuda"):
        try:
            lps, acc = _predictive_tables_batched(model, dataloader_test, samples, labels, E, C)
        except _NotBatchable as exc:
            warnings.warn(f"grouped posterior-predictive evaluation unavailable ({exc}); evaluating sample by sample")
        else:
            # the reference leaves the model holding the LAST sample (exp_utils.py:262-264 loads each in turn)
            model.load_state_dict({k: v[E - 1] for k, v in samples.items()})
            return lps, acc, labels, "cat"
    lps = torch.zeros((E, N), dtype=torch.float64, device=device)
    acc = torch.zeros((E, N, C), dtype=torch.float64, device=device)
    kind = None
    graphed = (isinstance(model, ClassificationModel) and device.type == "cuda" and not model.training
               and labels.dim() == 1)
    plain_temp = graphed and isinstance(model.softmax_temp, (int, float))
    for e in range(E):
        model.load_state_dict({k: v[e] for k, v in samples.items()})
        i = 0
        # evaluation mode makes every row's prediction a function of that row alone, so the loader's batches are
        # evaluated several at a time (EVAL_ROWS rows per forward): 10 launch-bound forwards of 1,024 rows instead of
        # 79 of 128 for a CIFAR-10 test set -- same numbers row by row (the kernels' arithmetic is per image)
        for bx, by in (_row_groups(dataloader_test, device, EVAL_ROWS) if graphed and plain_temp else dataloader_test):
            bx, by = bx.to(device), by.to(device)
            if graphed and plain_temp:
                # the numbers ``model(bx)`` = Categorical(logits=net(x) / T) would hold -- normalised logits and
                # log p(y|x) -- without building the distribution object (its argument validation synchronises
                # with the host once per batch); the forward itself is replayed from a captured graph
                f = _logits_fn(model, bx)
                f = f if model.softmax_temp == 1 else f / model.softmax_temp
                a = f - f.logsumexp(dim=-1, keepdim=True)          # torch/distributions/categorical.py: logits
                lp = a.gather(-1, by.view(-1, 1)).squeeze(-1)
                kind = "cat"
                j = i + len(bx)
                lps[e, i:j] = lp
                acc[e, i:j] = a
                i = j
                continue
            preds = model(bx)
            if isinstance(preds, torch.distributions.Categorical):
                kind, a, lp = "cat", preds.logits, preds.log_prob(by)
            elif isinstance(preds, torch.distributions.Normal):
                kind, a, lp = "normal", preds.mean, preds.log_prob(by).sum(-1)
            else:
                raise ValueError(f"unknown likelihood {type(preds)}")
            j = i + len(bx)
            lps[e, i:j] = lp
            acc[e, i:j] = a
            i = j
    return lps, acc, labels, kind


def _log_mean_exp(t, dim):
    return t.logsumexp(dim) - math.log(t.size(dim))


def ensemble_metrics(model, lps, acc, labels, kind):
    lp_each = lps.mean(1)
    out = {"lp_ensemble": _log_mean_exp(lps, 0).mean().item(), "lp_last": lp_each[-1].item()}
    if kind == "cat":
        ens = torch.distributions.Categorical(logits=_log_mean_exp(acc, 0))
        last = torch.distributions.Categorical(logits=acc[-1])
    else:
        ens = torch.distributions.Normal(acc.mean(0), torch.ones_like(acc[0]))
        last = torch.distributions.Normal(acc[-1], ens.scale)
    out["acc_ensemble"] = model.acc_mse(ens, labels).double().mean(0).item()
    out["acc_last"] = model.acc_mse(last, labels).double().mean(0).item()
    return out


def evaluate_model(model, dataloader_test, samples, likelihood_eval=True, accuracy_eval=True,
                   calibration_eval=False):
    if calibration_eval:
        raise NotImplementedError("calibration metrics are outside the accelerated path")
    lps, acc, labels, kind = predictive_tables(model, dataloader_test, samples)
    res = ensemble_metrics(model, lps, acc, labels, kind)
    keep = (["lp_ensemble", "lp_last"] if likelihood_eval else []) + \
           (["acc_ensemble", "acc_last"] if accuracy_eval else [])
    return {k: res[k] for k in keep}


@torch.no_grad()
def ensemble_across_chains(lps, acc, group=None):
    """Combine the per-chain tables of all ranks into the ensemble over ALL chains'
    samples without gathering them: returns (log-mean-exp of lps [N], of acc [N, C]).

    Per rank: m = max_e x, s = sum_e exp(x - m).  all_reduce(MAX) gives the global
    max M; s * exp(m - M) summed with all_reduce(SUM), the sample counts likewise.
    Works on any backend (RCCL on the GPUs, gloo in the CPU tests)."""
    import torch.distributed as dist
    x = torch.cat([lps.unsqueeze(-1), acc], dim=-1)          # [E, N, C+1]
    m_local = x.max(dim=0).values
    s_local = (x - m_local).exp().sum(dim=0)
    if dist.is_available() and dist.is_initialized():
        m_glob = m_local.clone()
        dist.all_reduce(m_glob, op=dist.ReduceOp.MAX, group=group)
        packed = torch.cat([(s_local * (m_local - m_glob).exp()).reshape(-1),
                            torch.tensor([float(x.shape[0])], dtype=x.dtype, device=x.device)])
        dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
        s_glob, count = packed[:-1].reshape(s_local.shape), packed[-1]
    else:
        m_glob, s_glob, count = m_local, s_local, torch.tensor(float(x.shape[0]), dtype=x.dtype)
    lme = m_glob + s_glob.log() - count.log()
    return lme[..., 0], lme[..., 1:]


def gather_samples(samples, dst=0, group=None):
    """Collect every chain's stored samples (``runner.get_samples()``: name -> [E_c, ...]) on rank
    ``dst`` as name -> [sum_c E_c, ...], chain by chain -- what a single ``samples.pt`` of an
    8-chain run contains (reference: one file per chain, experiments/run_experiment.sh:15-34).
    One all_gather per tensor over RCCL (gloo in the CPU tests); returns None on other ranks."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return samples
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    out = {}
    for name in sorted(samples):
        t = samples[name].contiguous()
        n = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
        counts = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(counts, n, group=group)
        counts = [int(c.item()) for c in counts]
        pad = max(counts)
        buf = torch.zeros((pad,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        buf[:t.shape[0]] = t
        parts = [torch.zeros_like(buf) for _ in range(world)]
        dist.all_gather(parts, buf, group=group)
        if rank == dst:
            out[name] = torch.cat([p[:c] for p, c in zip(parts, counts)])
    return out if rank == dst else None
