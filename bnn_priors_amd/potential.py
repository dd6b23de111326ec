"""Gradient of the average potential, split the way the engine wants it.

    potential_avg(x, y) = -(1/B) sum_i log p(y_i | x_i, theta) - log p(theta) / N
                                                        (reference: models/base.py:72-77)

The reference sends both terms through autograd and ``torch.distributions`` (about ten
ATen launches and two validation syncs per prior tensor per step).  Here

* the likelihood term goes through autograd as one ``cross_entropy`` on the network's
  logits (same value as ``Categorical(logits=f).log_prob(y)``, models/base.py:181-182),
* element-wise priors with scalar loc/scale are differentiated by ONE fused HIP launch
  (``optimizer.add_prior_gradient``) that also leaves the summed log-density on the device,
* anything else (other prior families, tensor-valued scales, non-classification models)
  falls back to the reference's formulation unchanged.

Nothing here synchronises with the host; callers ``.item()`` what they log.
"""
import itertools
import warnings

import torch

from . import conv as _conv
from . import pool as _pool

from .models.base import ClassificationModel


_noticed = set()       # prior-class sets whose autograd route has been announced (once per process)


class Potential:
    def __init__(self, model, optimizer, num_data):
        self.model, self.opt, self.N = model, optimizer, float(num_data)
        self.fast = (isinstance(model, ClassificationModel)
                     and isinstance(model.softmax_temp, (int, float))
                     and hasattr(optimizer, "fuse_priors"))
        self.leftover = optimizer.fuse_priors(model) if self.fast else None
        if self.leftover:
            kinds = tuple(sorted({type(pr).__name__ for pr in self.leftover}))
            if kinds not in _noticed:
                _noticed.add(kinds)
                warnings.warn(f"priors {', '.join(kinds)} are not element-wise families of the HIP prior hook: their "
                              "log-density is differentiated by autograd in every gradient evaluation and the step is not "
                              "captured into a hipGraph (INTEGRATION.md section 1)", stacklevel=3)

    # ------------------------------------------------------------------ pieces
    def _logits(self, x):
        f = self.model.net(x)
        return f if self.model.softmax_temp == 1 else f / self.model.softmax_temp

    def _leftover_log_prior(self):
        if not self.leftover:
            return None
        return sum(pr.log_prob() for pr in self.leftover)

    def _log_prior(self, leftover_lp):
        lp = self.opt.fused_log_prior()          # float64, on the device
        return lp if leftover_lp is None else lp + leftover_lp.detach().double()

    # ------------------------------------------------------------------ minibatch gradient
    def minibatch(self, x, y, want_metrics):
        """zero grads; g <- grad potential_avg(x, y).  Returns (loss, log_prior, potential, acc)
        as device tensors; log_prior / potential / acc are None unless ``want_metrics``."""
        self.opt.zero_grad()
        if not self.fast:
            loss, log_prior, potential, accs, _ = self.model.split_potential_and_acc(x, y, self.N)
            potential.backward()
            return loss, log_prior, potential, accs.mean()
        with _conv.deferring(self.model):      # this pass's convolution weight-gradient slabs: one reduction at its end
            extra = self._leftover_log_prior()
            if extra is not None:
                # a tensor with a leftover prior receives TWO gradients in this backward (the likelihood's and the prior's
                # log-density's); autograd adds them as soon as both exist, so the likelihood's must be complete then:
                # count the prior as a use of its tensors -- their weight-gradient slabs are reduced at once (conv._may_defer)
                _conv._note_use(*{id(p): p for pr in self.leftover for p in pr.parameters()}.values())
            if extra is None:
                with _pool.head_loss(y, head=_pool.head_of(self.model)):                        # a fused head also takes the loss and both backward passes
                    f = self._logits(x)
                loss = _pool.cross_entropy_backward(f, y)       # likelihood forward + backward seed: one launch
            else:
                f = self._logits(x)
                loss = _pool.cross_entropy(f, y)
                (loss - extra / self.N).backward()
        self.opt.add_prior_gradient(calc_log_prior=want_metrics)
        if not want_metrics:
            return loss.detach(), None, None, None
        with torch.no_grad():
            log_prior = self._log_prior(extra)
            potential = loss.detach().double() - log_prior / self.N
            acc = f.argmax(dim=1).eq(y).float().mean()
        return loss.detach(), log_prior, potential, acc

    # ------------------------------------------------------------------ full-data gradient
    graph_exact = False     # set by the runner (``use_graph``): capture the exact pass's batch body

    def _make_exact_accumulator(self, x, y, n_full=None):
        """graph-captured accumulation: L launches, no read-back (graphed.GraphedAccumulate) -- for the launch-bound
        convolutional nets on EXACT_LANES streams at once (graphed.ConcurrentAccumulate)"""
        from . import bn as _bn, graphed
        from .models import nets
        if ((graphed.EXACT_LANES > 1 or graphed.EXACT_GROUP != 1)
                and any(isinstance(m, nets.Conv2d) for m in self.model.modules())):
            try:
                return graphed.ConcurrentAccumulate(self, self.opt, x, y, lanes=max(1, graphed.EXACT_LANES),
                                                    n_full=n_full)
            except _bn.LogModeUnsupported:
                pass
        return graphed.GraphedAccumulate(self, self.opt, x, y)

    def _may_graph_exact(self, batches):
        return (self.graph_exact and not self.leftover and self.opt.engine.device.type == "cuda"
                and len(batches) >= 4)      # fewer batches are not worth a capture

    def exact(self, batches):
        """g <- grad[-log_prior/N] + sum_batches grad[-sum_i log p_i / N]
        (reference: inference_reject.py:18-33).  Returns (loss, log_prior, potential)."""
        self.opt.zero_grad()
        if not self.fast:
            log_prior = self.model.log_prior()
            lnp = log_prior / -self.N
            lnp.backward()
            loss = 0.
            for x, y in batches:
                this = self.model.log_likelihood(x, y, -x.size(0) / self.N)
                this.backward()
                loss = loss + this.detach()
            return loss, log_prior.detach(), loss + lnp.detach()
        acc = getattr(self, "_exact_acc", None)
        may_graph = acc is not None or self._may_graph_exact(batches)
        loss = torch.zeros((), dtype=torch.float64, device=self.opt.engine.device)
        if acc is not None:
            acc.begin()
        if may_graph and acc is None:
            # the capture needs a minibatch's shapes: from the source itself when it can tell (it then stays the object
            # that ``ConcurrentAccumulate.run`` asks for in-place filling and the number of full-size minibatches),
            # else from the first minibatch (never peeked at beyond what is consumed: iterating the loader draws from its RNG)
            first = batches.example() if hasattr(batches, "example") else None
            if first is not None:
                n_full = batches.n_full_batches() if hasattr(batches, "n_full_batches") else None
                acc = self._exact_acc = self._make_exact_accumulator(*first, n_full=n_full)
                acc.begin()
            else:
                batches = iter(batches)
                first = next(batches, None)
                if first is not None:
                    acc = self._exact_acc = self._make_exact_accumulator(*first)
                    acc.begin()
                    batches = itertools.chain([first], batches)
        if acc is not None and hasattr(acc, "run"):
            acc.run(batches)         # minibatches on several streams at once, several per launch (graphed.ConcurrentAccumulate)
            batches = ()
        for x, y in batches:
            if acc is not None:
                (acc.add if acc.matches(x, y) else acc.add_eager)(x, y)
            else:
                # (gradients accumulate into existing .grad tensors here, so nothing is deferred)
                with _pool.head_loss(y, "sum", self.N, head=_pool.head_of(self.model)):
                    f = self._logits(x)
                this = _pool.cross_entropy_backward(f, y, reduction="sum", divide_by=self.N)
                loss = loss + this.double()
        if acc is not None:
            acc.finish()
            loss = acc.loss.clone()
        extra = self._leftover_log_prior()
        if extra is not None:
            (extra / -self.N).backward()
        self.opt.add_prior_gradient(calc_log_prior=True)
        log_prior = self._log_prior(extra)
        return loss, log_prior, loss - log_prior / self.N
