"""Whole leapfrog step of the dense classifier as a 3-kernel hipGraph.

For ``ClassificationDenseNet`` (Linear-ReLU-Linear-ReLU-Linear, categorical likelihood,
element-wise priors; reference models/dense_nets.py:48-67) a leapfrog step is

    [one async copy: per-step scalars + the minibatch's row indices]
    mlp_fwdbwd_kernel       fused forward + backward on the matrix cores, rows gathered by
                            index from the device-resident data set      (csrc/mlp_hip.inc)
    prior_kernel_indirect   fixed-order sum of the per-slice partial gradients + prior
                            gradient (+ log-prior partials on metric steps)
    step_kernel_indirect    fused sampler transition (noise, momentum, position, RMSprop, dots)
    finalize_small_kernel   energy / temperature bookkeeping, energy total, log-prior total

captured once per batch size (by the library itself, on an internal stream:
``sgmcmc_dense_stepper_create``) and replayed with ONE native call per step that also ships the
per-step scalars and row indices through a ring of pinned slots; metric steps read ONE buffer back.  Numerically
the gradient differs from the autograd path only by fp32 summation order (tested against a
PyTorch reference in tests/test_fused_dense.py); everything downstream is the same code.
"""
import ctypes

import numpy as np
import torch
from torch import nn

from . import _hip
from .graphed import PendingRow, _ReportSlots
from .models.base import ClassificationModel


class IndexBatch:
    """A minibatch named by its row indices into a device-resident data set."""
    __slots__ = ("idx", "X", "Y", "ptr")

    def __init__(self, idx, X, Y):
        self.idx = np.ascontiguousarray(idx, dtype=np.int64)   # host
        self.X, self.Y = X, Y
        self.ptr = self.idx.__array_interface__["data"][0]

    def __len__(self):
        return len(self.idx)

    def materialize(self):
        i = torch.from_numpy(np.ascontiguousarray(self.idx)).to(self.X.device)
        return self.X.index_select(0, i), self.Y.index_select(0, i)


def _dense_layers(model):
    "the three Linear layers if ``model`` is the supported topology, else None"
    if not isinstance(model, ClassificationModel) or not isinstance(model.softmax_temp, (int, float)):
        return None
    net = getattr(model.net, "module", model.net)
    if not isinstance(net, nn.Sequential) or len(net) != 5:
        return None
    lin = [net[0], net[2], net[4]]
    if not all(isinstance(net[i], nn.ReLU) for i in (1, 3)):
        return None
    from .models.nets import Linear
    for l in lin:
        if not isinstance(l, Linear) or l.bias_prior is None:
            return None
    return lin


class FusedDenseLeapfrog(_ReportSlots):
    @staticmethod
    def supported(potential, optimizer):
        lin = _dense_layers(potential.model)
        eng = optimizer.engine
        if lin is None or not potential.fast or potential.leftover or not eng.small_finalize:
            return False
        if eng.dtype != torch.float32 or len(optimizer.param_groups) != 1:
            return False
        if eng.prior_links or eng.prior_max_kind > _hip.PRIOR_CAUCHY:
            return False       # the step kernel's in-flight prior covers the constant-scale families only
        want = [t for l in lin for t in (l.weight_prior.p, l.bias_prior.p)]
        if len(want) != len(eng.params) or any(a is not b for a, b in zip(want, eng.params)):
            return False
        (h1, i), (h2, h1b), (o, h2b) = (l.weight_prior.p.shape for l in lin)
        return (i % 4 == 0 and h1 == h1b and h2 == h2b and h1 <= 64 and h2 <= 64 and o <= 16
                and _hip.lib().sgmcmc_mlp_lds_bytes(i) <= 160 * 1024)

    def __init__(self, potential, optimizer, X, Y, ring=8, direct=None):
        assert self.supported(potential, optimizer)
        import os
        # three direct launches with by-value arguments vs replaying a captured graph
        self.direct = (os.environ.get("SGMCMC_DENSE_DIRECT", "1") == "1") if direct is None else direct
        # forward/backward as two launches over 4x more workgroups (csrc/mlp_hip.inc, 'two-launch split')
        self.split = True
        self.pot, self.opt, self.eng = potential, optimizer, optimizer.engine
        self.lib = _hip.lib()
        eng, dev = self.eng, optimizer.engine.device
        self.X_source = X      # identity of the data set this stepper gathers from
        self.X = X.to(dev, torch.float32).contiguous().view(X.shape[0], -1)
        self.Y = Y.to(dev, torch.int64).contiguous()
        self.lin = _dense_layers(potential.model)
        # static gradient buffer, packed like the noise index (4-aligned per tensor)
        self.offs = [int(o) for o in eng.seg_host["noise_base"]]
        self.stride = self.offs[-1] + -(-eng.params[-1].numel() // 4) * 4
        self.g_flat = torch.zeros(self.stride, device=dev)
        self.static_grads = [self.g_flat[o:o + p.numel()].view(p.shape)
                             for o, p in zip(self.offs, eng.params)]
        self.nbytes_args = ctypes.sizeof(_hip.StepArgs)
        self._ring, self._k, self._pp = ring, 0, 0
        self._by_batch = {}
        self._init_slots(eng.report.numel())

    # ------------------------------------------------------------------ per batch size
    def _setup(self, batch):
        eng, dev = self.eng, self.eng.device
        S = -(-batch // _hip.MLP_ROWS)
        slot_bytes = self.nbytes_args + 8 * batch
        st = dict(batch=batch, S=S)
        st["dev"] = torch.zeros(slot_bytes, dtype=torch.uint8, device=dev)
        st["pinned"] = torch.zeros(self._ring * slot_bytes, dtype=torch.uint8).pin_memory()
        st["gpart"] = torch.zeros(S * self.stride, device=dev)
        st["loss_part"] = torch.zeros(S, device=dev)
        st["corr_part"] = torch.zeros(S, device=dev)
        st["split"] = torch.zeros(self.lib.sgmcmc_mlp_split_scratch_floats(batch), device=dev)
        W1, b1, W2, b2, W3, b3 = eng.params
        o = self.offs
        st["mlp"] = _hip.MlpArgs(
            X=self.X.data_ptr(), Y=self.Y.data_ptr(), idx=st["dev"].data_ptr() + self.nbytes_args,
            W1=W1.data_ptr(), b1=b1.data_ptr(), W2=W2.data_ptr(), b2=b2.data_ptr(),
            W3=W3.data_ptr(), b3=b3.data_ptr(), gpart=st["gpart"].data_ptr(),
            loss_part=st["loss_part"].data_ptr(), correct_part=st["corr_part"].data_ptr(),
            gpart_stride=self.stride, off_W1=o[0], off_b1=o[1], off_W2=o[2], off_b2=o[3],
            off_W3=o[4], off_b3=o[5], batch=batch, in_features=W1.shape[1], hidden1=W1.shape[0],
            hidden2=W2.shape[0], out_features=W3.shape[0],
            inv_softmax_temp=1.0 / float(self.pot.model.softmax_temp), trace=None,
            args_src=None, args_dst=None, args_bytes=0, grad_scale=0.0,
            split_scratch=st["split"].data_ptr() if self.split else None)
        st["param_ptrs"] = [p.data_ptr() for p in eng.params]
        self._bind_grads()
        eng.refresh(self.opt._preconditioners())
        torch.cuda.synchronize(dev)
        st["A"] = self._args(st.get("A"), False, advance=False)
        handle = ctypes.c_void_p()
        _hip.check(self.lib.sgmcmc_dense_stepper_create(
            ctypes.byref(eng.layout), ctypes.byref(st["mlp"]), ctypes.byref(st["A"]), self.pot.N,
            st["dev"].data_ptr(), st["pinned"].data_ptr(), self._ring, slot_bytes,
            ctypes.byref(handle)), "sgmcmc_dense_stepper_create")
        st["handle"] = handle
        st["App"] = [self._args(None, False, advance=False), self._args(None, False, advance=False)]
        torch.cuda.synchronize(dev)
        return st

    def __del__(self):
        for st in getattr(self, "_by_batch", {}).values():
            try:
                self.lib.sgmcmc_dense_stepper_destroy(st["handle"])
            except Exception:
                pass

    def _args(self, A, calc_metrics, advance=True):
        "fill (or create) the per-step scalar struct for an ordinary step at the CURRENT lr"
        kind, flags, sc = self.opt._plain_step_spec(calc_metrics)
        eng = self.eng
        draw = eng.next_draw() if advance else eng.draw
        if A is None:
            return eng.make_args(0, kind, flags | _hip.WITH_LOG_PRIOR, draw,
                                 grad_clamp=self.opt.grad_clamp, **sc)
        A.flags = self._static_flags | flags
        A.draw = draw
        A.b2h2, A.bh, A.bhn = sc["b2h2"], sc["bh"], sc["bhn"]
        A.mom_decay, A.grad_v, A.noise_std = sc["mom_decay"], sc["grad_v"], sc["noise_std"]
        A.num_data, A.rmsprop_alpha = sc["num_data"], sc["rmsprop_alpha"]
        return A

    def _bind_grads(self):
        params = self.eng.params
        if params[0].grad is not self.static_grads[0]:
            for p, g in zip(params, self.static_grads):
                p.grad = g
            self.eng._seg_dirty = True

    # ------------------------------------------------------------------ exact full-data gradient
    MEGA = 2048   # rows per launch of the fused kernel in the full-data pass (128 workgroups)

    def exact(self):
        """g <- grad[-log_prior/N] + grad[-sum_i log p(y_i|x_i)/N] over the WHOLE device-resident
        data set (reference: inference_reject.py:18-33, one autograd pass per minibatch): the fused
        forward/backward kernel on mega-batches of 2048 rows, fp64 accumulation across them, then
        the prior gradient and log-density.  Returns (loss, log_prior, potential) as 0-d tensors."""
        eng, dev, lib = self.eng, self.eng.device, self.lib
        n_data = self.X.shape[0]
        ex = getattr(self, "_exact_state", None)
        if ex is None:
            S = self.MEGA // _hip.MLP_ROWS
            ex = self._exact_state = dict(
                gpart=torch.zeros(S * self.stride, device=dev), loss_part=torch.zeros(S, device=dev),
                corr_part=torch.zeros(S, device=dev),
                acc=torch.zeros(self.stride, dtype=torch.float64, device=dev),
                gsum=torch.zeros(self.stride, device=dev),
                stats=torch.zeros(2, dtype=torch.float64, device=dev))
        W1, b1, W2, b2, W3, b3 = eng.params
        o = self.offs
        stream = eng.stream()
        self._bind_grads()
        eng.refresh(self.opt._preconditioners())
        in_f = W1.shape[1]
        for start in range(0, n_data, self.MEGA):
            mb = min(self.MEGA, n_data - start)
            A = _hip.MlpArgs(
                X=self.X.data_ptr() + 4 * start * in_f, Y=self.Y.data_ptr() + 8 * start, idx=None,
                W1=W1.data_ptr(), b1=b1.data_ptr(), W2=W2.data_ptr(), b2=b2.data_ptr(),
                W3=W3.data_ptr(), b3=b3.data_ptr(), gpart=ex["gpart"].data_ptr(),
                loss_part=ex["loss_part"].data_ptr(), correct_part=ex["corr_part"].data_ptr(),
                gpart_stride=self.stride, off_W1=o[0], off_b1=o[1], off_W2=o[2], off_b2=o[3],
                off_W3=o[4], off_b3=o[5], batch=mb, in_features=in_f, hidden1=W1.shape[0],
                hidden2=W2.shape[0], out_features=W3.shape[0],
                inv_softmax_temp=1.0 / float(self.pot.model.softmax_temp), trace=None,
                args_src=None, args_dst=None, args_bytes=0, grad_scale=1.0 / self.pot.N,
                split_scratch=None)
            _hip.check(lib.sgmcmc_mlp_fwdbwd(ctypes.byref(A), stream), "sgmcmc_mlp_fwdbwd")
            last = start + mb >= n_data
            _hip.check(lib.sgmcmc_accumulate_parts(
                ex["gpart"].data_ptr(), -(-mb // _hip.MLP_ROWS), self.stride, ex["acc"].data_ptr(),
                ex["gsum"].data_ptr() if last else None, self.stride, ex["loss_part"].data_ptr(),
                ex["corr_part"].data_ptr(), ex["stats"].data_ptr(), int(start == 0), stream),
                "sgmcmc_accumulate_parts")
        # prior gradient + log-density on top of the accumulated likelihood gradient (one "slice")
        _hip.check(lib.sgmcmc_grad_reduce_prior(
            ctypes.byref(eng.layout), ex["gsum"].data_ptr(), 1, self.stride, ex["loss_part"].data_ptr(),
            ex["corr_part"].data_ptr(), 1, self.pot.N, _hip.CALC_METRICS, None, stream),
            "sgmcmc_grad_reduce_prior")
        eng._touch()
        loss = ex["stats"][0] / self.pot.N
        log_prior = eng.scalars[2].clone()
        return loss, log_prior, loss - log_prior / self.pot.N

    # ------------------------------------------------------------------ replay
    def replay(self, idx, metrics=False, idx_ptr=None, wait=True, calc_metrics=None):
        """One leapfrog step on the rows ``idx`` (host int64 array).  Returns None, or on a
        metric step dict(loss, acc, log_prior, energy, nonfinite) after one read-back."""
        batch = len(idx)
        st = self._by_batch.get(batch)
        eng = self.eng
        if st is not None:
            for p, q in zip(eng.params, st["param_ptrs"]):
                if p.data_ptr() != q:    # parameter storage moved (p.data = ...): re-capture
                    self.lib.sgmcmc_dense_stepper_destroy(st["handle"])
                    st = None
                    break
        if st is None:
            st = self._by_batch[batch] = self._setup(batch)
            self._static_flags = st["A"].flags & ~_hip.CALC_METRICS
        self._bind_grads()
        if eng._seg_dirty or eng._precond_dirty:
            eng.refresh(self.opt._preconditioners())
        if idx_ptr is None:
            idx_ptr = idx.__array_interface__["data"][0]
        if self.direct and batch <= 256:
            # ping-pong argument structs: the previous step's may still be pending
            self._pp ^= 1
            A = self._args(st["App"][self._pp], metrics if calc_metrics is None else calc_metrics)
            if not metrics and eng.small_finalize:
                A.flags |= _hip.DEFER_FINALIZE      # its bookkeeping rides in the NEXT launch
            pending, eng.pending = eng.pending, None
            err = self.lib.sgmcmc_dense_step_direct(eng.layout, st["mlp"], A, self.pot.N, idx_ptr,
                                                    pending, eng.stream())
            if A.flags & _hip.DEFER_FINALIZE:
                eng.pending = A
        else:
            eng.flush()
            A = self._args(st["A"], metrics if calc_metrics is None else calc_metrics)
            err = self.lib.sgmcmc_dense_stepper_step(st["handle"], A, idx_ptr, eng.stream())
        if err:
            _hip.check(err, "sgmcmc_dense_step")
        eng._state_host = None
        eng.energy_ready = True
        if not metrics:
            return None
        eng.metrics_ready = True
        buf, ev = self._take_slot()
        buf.copy_(eng.report, non_blocking=True)
        ev.record()
        n_seg = eng.n_seg

        def parse(v):
            r = dict(loss=float(v[4]), acc=float(v[5]), nonfinite=bool(v[1] != 0.0),
                     log_prior=float(v[2]), energy=float(v[3]))
            return r, v[8:].reshape(n_seg, -1).copy()
        row = PendingRow(self, buf, ev, parse)
        if not wait:
            return row
        r, state = row.get()
        eng._state_host = state
        if r["nonfinite"]:
            eng.scalars[1].zero_()
        return r


class MultiChainDense:
    """K independent chains of the dense classifier stepped in lock-step by THREE launches per leapfrog step
    (``sgmcmc_dense_step_multi``: grid dimension y = chain) instead of 3 K.

    One chain's launches occupy 33-46 workgroups of the 256 CUs and are latency-bound (a step is three dependent
    launches of 6-12 us whatever the net's size); chains are independent, so K of them -- each with its own
    ``FusedDenseLeapfrog`` (weights, sampler arena, data order, Philox stream) -- share the launches and the GPU does
    K steps in about the time of one.  All chains must have one architecture and one schedule (learning rate,
    temperature, metrics cadence): the transition's scalars travel once, by value.  Chain c's trajectory is
    bit-identical to the same chain stepped alone (tests/test_fused_dense.py).  This is an addition to the
    reference's one-process-per-chain model (experiments/run_experiment.sh:15-34), for the small nets only."""

    def __init__(self, steppers):
        self.steppers = list(steppers)
        K = len(self.steppers)
        if not 1 <= K <= _hip.MAX_CHAINS:
            raise ValueError(f"1..{_hip.MAX_CHAINS} chains per launch")
        s0 = self.steppers[0]
        for s in self.steppers:
            if not (s.direct and s.split and s.eng.small_finalize and s.eng.chunk == _hip.CHUNK_SMALL):
                raise ValueError("multi-chain stepping needs the direct, split fused dense path")
            if [tuple(p.shape) for p in s.eng.params] != [tuple(p.shape) for p in s0.eng.params]:
                raise ValueError("chains must share one architecture")
            if s.X.shape[0] > 65536:
                raise ValueError("16-bit row indices: data sets of up to 65,536 rows")
            if s.eng.prior_flags() != 0:
                raise ValueError("the multi-chain kernels carry the lean prior code only (kinds <= CAUCHY, no links)")
        self.lib, self.device = s0.lib, s0.eng.device
        self._host = (_hip.DenseChain * K)()
        self._dev = torch.zeros(ctypes.sizeof(self._host), dtype=torch.uint8, device=self.device)
        self._uploaded = None
        self._sig = None
        self._idx16 = np.zeros((K, _hip.MLP_BATCH_MULTI), dtype=np.uint16)

    def _table(self, batch):
        "per-chain pointers, uploaded when any of them changed (first use, roll-back arrays allocated, ...)"
        sig = [batch]
        for s in self.steppers:
            st = s._by_batch.get(batch)
            if st is None:
                st = s._by_batch[batch] = s._setup(batch)
                s._static_flags = st["A"].flags & ~_hip.CALC_METRICS
            s._bind_grads()
            if s.eng._seg_dirty or s.eng._precond_dirty:
                s.eng.refresh(s.opt._preconditioners())
            sig.append(id(st))
            sig.append(bytes(s.eng.layout))        # (pointers of the arenas / roll-back arrays: 100-odd bytes)
        if sig == self._sig:                       # nothing moved since the table was last built: the usual step
            return
        self._sig = sig
        for c, s in enumerate(self.steppers):
            st = s._by_batch[batch]
            row = self._host[c]
            ctypes.memmove(ctypes.addressof(row.mlp), ctypes.addressof(st["mlp"]), ctypes.sizeof(_hip.MlpArgs))
            ctypes.memmove(ctypes.addressof(row.layout), ctypes.addressof(s.eng.layout), ctypes.sizeof(_hip.Layout))
            row.num_data, row.chain_id = float(s.pot.N), int(s.eng.chain_id)
        raw = bytes(self._host)
        if raw != self._uploaded:
            self._dev.copy_(torch.frombuffer(bytearray(raw), dtype=torch.uint8))     # synchronous: rare
            self._uploaded = raw

    def step(self, idx_list, metrics=False):
        """one leapfrog step of every chain on ITS rows ``idx_list[c]`` (host int arrays of one length <= 128).
        ``metrics``: the transition also updates the temperature estimates / log-prior; returns one
        dict(loss, acc, log_prior, energy, nonfinite) per chain (after one read-back each), else None."""
        K, batch = len(self.steppers), len(idx_list[0])
        if batch > _hip.MLP_BATCH_MULTI or any(len(i) != batch for i in idx_list):
            raise ValueError("one batch size of at most 128 rows for all chains")
        self._table(batch)
        s0 = self.steppers[0]
        st0 = s0._by_batch[batch]
        s0._pp ^= 1
        A = s0._args(st0["App"][s0._pp], metrics)
        A.flags |= _hip.DEFER_FINALIZE
        for s in self.steppers[1:]:
            d = s.eng.next_draw()
            if d != A.draw or s.opt.param_groups[0]["lr"] != s0.opt.param_groups[0]["lr"]:
                raise RuntimeError("chains stepped together must share the schedule and the draw counter")
        for c, idx in enumerate(idx_list):
            self._idx16[c, :batch] = idx
        # the launch finalizes ONE deferred transition for all chains: they must agree on what is pending.  A chain
        # flushed on its own (a state read, a preconditioner refresh) while another was not would otherwise have its
        # bookkeeping run twice -- or the others' dropped: settle every chain first.
        pend = [s.eng.pending for s in self.steppers]
        p0 = pend[0]
        if any((p is None) != (p0 is None) or (p is not None and (p.draw != p0.draw or p.flags != p0.flags))
               for p in pend[1:]):
            for s in self.steppers:
                s.eng.flush()
        pending = s0.eng.pending
        idx16 = np.ascontiguousarray(self._idx16[:, :batch])
        err = self.lib.sgmcmc_dense_step_multi(self._dev.data_ptr(), ctypes.byref(self._host[0]), K, A,
                                               idx16.ctypes.data, pending, s0.eng.stream())
        if err:
            _hip.check(err, "sgmcmc_dense_step_multi")
        for s in self.steppers:
            s.eng.pending = A
            s.eng._state_host = None
            s.eng.energy_ready = True
        if not metrics:
            return None
        for s in self.steppers:
            s.eng.flush()                    # every chain's bookkeeping, now (the rows read its results) ...
            s.eng.metrics_ready = True
        rows = torch.stack([s.eng.report for s in self.steppers]).cpu().numpy()      # ... and ONE read-back for all
        out = []
        for s, v in zip(self.steppers, rows):
            out.append(dict(loss=float(v[4]), acc=float(v[5]), nonfinite=bool(v[1] != 0.0), log_prior=float(v[2]),
                            energy=float(v[3])))
            s.eng._state_host = v[8:].reshape(s.eng.n_seg, -1).copy()
        return out
