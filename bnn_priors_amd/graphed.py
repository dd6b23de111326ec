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

Steps that read metrics back (every ``metrics_skip``-th), M-H points and off-shape batches
keep using the eager path; ``p.grad`` is re-bound to the graph's static gradient tensors
whenever control returns to the graph.
"""
import ctypes

import torch
import torch.nn.functional as F

from . import _hip


class GraphedLeapfrog:
    def __init__(self, potential, optimizer, x_example, y_example, ring=8, warmup=2):
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
        self._slot_events = [None] * ring
        self._k = 0
        self._capture(x_example, y_example, warmup)

    # ------------------------------------------------------------------ args ring
    def _push_args(self, A):
        i = self._k % len(self._slots)
        self._k += 1
        ev = self._slot_events[i]
        if ev is not None:
            ev.synchronize()            # the copy that last used this slot has executed
        ctypes.memmove(self._slots[i].data_ptr(), ctypes.addressof(A), ctypes.sizeof(A))
        self.args_dev.copy_(self._slots[i], non_blocking=True)
        ev = self._slot_events[i] = ev or torch.cuda.Event()
        ev.record()

    def _args(self, calc_metrics=False):
        kind, flags, sc = self.opt._plain_step_spec(calc_metrics)
        return self.eng.make_args(0, kind, flags, self.eng.next_draw(),
                                  grad_clamp=self.opt.grad_clamp, **sc)

    # ------------------------------------------------------------------ capture
    def _body(self, capturing):
        self.opt.zero_grad()
        loss = F.cross_entropy(self.pot._logits(self.x), self.y)
        loss.backward()
        self.eng.refresh(self.opt._preconditioners(), defer_upload=capturing)
        self.eng.prior_grad(self.pot.N, False)
        self.eng.step_indirect(self._A_host, self.args_dev.data_ptr())
        return loss.detach()

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
        self.x.copy_(x)
        self.y.copy_(y)
        snap = self._snapshot()
        self._A_host = self._args()
        self._push_args(self._A_host)
        side = torch.cuda.Stream(device=self.eng.device)
        side.wait_stream(torch.cuda.current_stream(self.eng.device))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._body(False)
        torch.cuda.current_stream(self.eng.device).wait_stream(side)
        torch.cuda.synchronize(self.eng.device)
        self._restore(snap)
        self.opt.zero_grad()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = self._body(True)
        self.static_grads = [p.grad for p in self.eng.params]
        self.eng.refresh(self.opt._preconditioners())      # upload the captured pointers
        torch.cuda.synchronize(self.eng.device)

    # ------------------------------------------------------------------ replay
    def matches(self, x, y):
        return (tuple(x.shape), tuple(y.shape)) == self.shape

    def replay(self, x, y):
        """one ordinary leapfrog step (no metric read-back); returns the loss tensor (valid
        until the next replay)"""
        self.x.copy_(x)
        self.y.copy_(y)
        params = self.eng.params
        if params[0].grad is not self.static_grads[0]:
            for p, g in zip(params, self.static_grads):
                p.grad = g
        self.eng.refresh(self.opt._preconditioners())
        self._push_args(self._args())
        self.graph.replay()
        self.eng._touch()
        self.eng.energy_ready = True
        return self.loss
