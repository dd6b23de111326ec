"""Host side of the HIP leapfrog engine: arena, segment/chunk tables, launches.

One ``Engine`` belongs to one optimizer instance.  It owns the optimizer state
the reference keeps per tensor in ``optimizer.state[p]`` (momentum, square_avg,
the three roll-back copies; mcmc/sgld.py:63-69,170, mcmc/verlet_sgld.py:72-83)
as flat arenas tiled into 4096-element chunks, and the device-resident tables
that let a single kernel launch sweep every parameter tensor of a group
(include/sgmcmc_hip.h).  Parameters and gradients are *not* moved: the segment
table carries their base pointers and is refreshed (one small async H2D copy)
only when a pointer or a preconditioner changed since the previous launch.
"""
import ctypes

import numpy as np
import torch

from .. import _hip

_MASK32 = 0xFFFFFFFF


def philox4x32_10(ctr, key):
    "host statement of the generator in csrc/sgmcmc_hip.hip (for the one M-H uniform)"
    c0, c1, c2, c3 = ctr
    k0, k1 = key
    for _ in range(10):
        p0 = 0xD2511F53 * c0
        p1 = 0xCD9E8D57 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & _MASK32, p1 & _MASK32, \
                         ((p0 >> 32) ^ c3 ^ k1) & _MASK32, p0 & _MASK32
        k0 = (k0 + 0x9E3779B9) & _MASK32
        k1 = (k1 + 0xBB67AE85) & _MASK32
    return c0, c1, c2, c3


def mh_uniform(seed, stream, draw):
    "the Metropolis-Hastings uniform: lane 0 of quad 0, purpose 2 (DESIGN.md 'Noise')"
    ctr = (0, 0, draw & _MASK32, (2 << 28) | ((stream & 0xFFF) << 16) | ((draw >> 32) & 0xFFFF))
    x0 = philox4x32_10(ctr, (seed & _MASK32, (seed >> 32) & _MASK32))[0]
    return (2 * (x0 >> 9) + 1) * 2.0 ** -24


def _dense(t):
    "non-overlapping and dense: sorted by stride, every stride is the product of the smaller extents"
    if t.is_contiguous():
        return True
    expect = 1
    for size, stride in sorted(zip(t.shape, t.stride()), key=lambda z: z[1]):
        if size == 1:
            continue
        if stride != expect:
            return False
        expect *= size
    return True


class Engine:
    _instances = 0

    def __init__(self, param_groups, seed=None, chain_id=0, chunk_elems=None, small_finalize=None):
        self.lib = _hip.lib()
        params = [p for g in param_groups for p in g["params"]]
        if not params:
            raise ValueError("optimizer got an empty parameter list")
        dev, dt = params[0].device, params[0].dtype
        if dev.type != "cuda":
            raise RuntimeError(
                f"bnn_priors_amd samplers run on an MI355X only (parameter on '{dev}'); "
                "there is no CPU path -- use the reference or oracle/ for CPU runs")
        if dt not in (torch.float32, torch.float64):
            raise TypeError(f"unsupported parameter dtype {dt}")
        for p in params:
            if p.device != dev or p.dtype != dt:
                raise TypeError("all parameters must share one device and dtype")
        self.device, self.dtype, self.params = dev, dt, params
        self.n_seg = len(params)
        self.index = {id(p): i for i, p in enumerate(params)}
        if seed is None:
            # reproducible under torch.manual_seed WITHOUT consuming the global generator (the
            # reference draws nothing at optimizer construction; the runners' shuffles must stay
            # aligned with it): the process seed mixed with a per-process instance counter
            Engine._instances += 1
            seed = (torch.initial_seed() * 0x9E3779B97F4A7C15 + Engine._instances * 0xD1B54A32D192ED03)
        self.seed, self.chain_id, self.draw = int(seed) & (2 ** 64 - 1), int(chain_id), 0

        # ---- tables.  Chunk size is a property of the layout: small models get 1024-element
        # chunks (one 16-byte item per thread) so that they still spread over many CUs.
        total_numel = sum(p.numel() for p in params)
        C = self.chunk = _hip.CHUNK_SMALL if total_numel <= (1 << 20) else _hip.CHUNK
        if chunk_elems is not None:      # explicit choice (tests exercise both geometries)
            assert chunk_elems in (_hip.CHUNK, _hip.CHUNK_SMALL)
            C = self.chunk = chunk_elems
        seg = np.zeros(self.n_seg, dtype=_hip.SEGMENT_DTYPE)
        chunk_rows, first, noise = [], 0, 0
        self.group_ranges = []
        s = 0
        for g in param_groups:
            s0, c0 = s, first
            for p in g["params"]:
                n = p.numel()
                nc = max(1, -(-n // C))
                seg[s]["numel"], seg[s]["first_chunk"], seg[s]["noise_base"] = n, first, noise
                seg[s]["M"] = 1.0
                for c in range(nc):
                    chunk_rows.append((s, min(C, n - c * C) if n else 0))
                first += nc
                noise += -(-n // 4) * 4
                s += 1
            self.group_ranges.append((s0, s, c0, first))
        self.n_chunks = first
        self.seg_host = seg
        self._seg_pinned = torch.empty(seg.nbytes, dtype=torch.uint8).pin_memory()
        self._seg_dev = torch.empty(seg.nbytes, dtype=torch.uint8, device=dev)
        chunks = np.array(chunk_rows, dtype=_hip.CHUNK_DTYPE)
        self._chunk_dev = torch.from_numpy(chunks.view(np.uint8).copy()).to(dev)
        self._seg_dirty = True
        self._precond_dirty = True    # set by SegState when a caller assigns a preconditioner
        self._ptr_cache = None
        self._unaligned = False

        # ---- arenas and scratch
        total = self.n_chunks * self.chunk
        self.m = torch.zeros(total, dtype=dt, device=dev)
        self.v = torch.ones(total, dtype=dt, device=dev)
        self.prev_theta = self.prev_g = self.prev_m = None
        self.partials = torch.zeros(self.n_chunks * _hip.PSTRIDE, dtype=torch.float64, device=dev)
        # scalars[8] and the per-segment state array share ONE buffer so a metric step reads
        # everything back with a single copy
        self.report = torch.zeros(8 + self.n_seg * len(_hip.SEG_STATE_FIELDS), dtype=torch.float64,
                                  device=dev)
        self.scalars, self.state_dev = self.report[:8], self.report[8:]
        self.layout = _hip.Layout()
        self._fill_layout()
        self._state_host = None      # cached D2H copy of state_dev
        self.momentum_ready = False
        self.metrics_ready = False
        self.energy_ready = False
        self.hidden_keys = frozenset()   # state keys this sampler family does not have
        # few chunks: one workgroup finalizes all segments and also emits the energy total
        self.small_finalize = len(param_groups) == 1 and self.n_chunks <= 4096
        if small_finalize is not None:
            self.small_finalize = bool(small_finalize) and len(param_groups) == 1

    # ------------------------------------------------------------------ views
    def _view(self, arena, i):
        "the tensor's slice of a state arena, with the parameter's own (dense) strides"
        p = self.params[i]
        off = int(self.seg_host[i]["first_chunk"]) * self.chunk
        flat = arena[off:off + p.numel()]
        return flat.view(p.shape) if p.is_contiguous() else flat.as_strided(p.shape, p.stride())

    def views(self, name):
        """every tensor's slice of the arena ``name`` (m, v, prev_theta, prev_g, prev_m), built ONCE per arena object: a
        slice + view costs ~4 us of host time, and the optimizers handed out 130 - 330 of them per eager transition --
        2.3 ms per Metropolis-Hastings point (final_step, sample_momentum, initial_step) during which the GPU waits for
        the host (3 % of the HMC L = 50 cycle)"""
        arena = getattr(self, name)
        cache = self.__dict__.setdefault("_view_cache", {})
        hit = cache.get(name)
        if hit is None or hit[0] is not arena:
            hit = cache[name] = (arena, [self._view(arena, i) for i in range(len(self.params))])
        return hit[1]

    def momentum_view(self, i):
        return self.views("m")[i]

    def square_avg_view(self, i):
        return self.views("v")[i]

    def ensure_prev(self):
        if self.prev_theta is None:
            self.prev_theta = torch.zeros_like(self.m)
            self.prev_g = torch.zeros_like(self.m)
            self.prev_m = torch.zeros_like(self.m)
            self._fill_layout()

    def _fill_layout(self):
        L = self.layout
        L.dtype = _hip.F32 if self.dtype == torch.float32 else _hip.F64
        L.n_seg, L.n_chunks, L.chunk_elems = self.n_seg, self.n_chunks, self.chunk
        L.segs, L.chunks = self._seg_dev.data_ptr(), self._chunk_dev.data_ptr()
        L.m, L.v = self.m.data_ptr(), self.v.data_ptr()
        for name in ("prev_theta", "prev_g", "prev_m"):
            t = getattr(self, name)
            setattr(L, name, t.data_ptr() if t is not None else self.m.data_ptr())
        L.partials, L.state, L.scalars = (self.partials.data_ptr(), self.state_dev.data_ptr(),
                                          self.scalars.data_ptr())

    # ------------------------------------------------------------------ table refresh
    def stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def refresh(self, preconditioners, need_grad=True, raise_on_no_grad=True, defer_upload=False):
        """Bring the device segment table up to date with the parameters' current
        storage, their gradients' storage and the preconditioners."""
        ptrs = []
        for p in self.params:
            g = p.grad
            if g is None:
                if need_grad and raise_on_no_grad:
                    raise RuntimeError(f"No gradient for parameter with shape {p.shape}")
                ptrs.append((p.data_ptr(), 0))
                continue
            ptrs.append((p.data_ptr(), g.data_ptr()))
        if ptrs != self._ptr_cache:
            unaligned = False
            for i, p in enumerate(self.params):
                # the kernels are element-wise over STORAGE order: any non-overlapping, dense
                # layout works (contiguous, channels_last, ...) provided theta and its gradient
                # share it; the state arenas mirror it through _view
                if not _dense(p):
                    raise RuntimeError("parameters must be dense (non-overlapping) tensors")
                g = p.grad
                if g is not None and (g.stride() != p.stride() or g.dtype != self.dtype
                                      or g.device != self.device):
                    fixed = torch.empty_like(p, memory_format=torch.preserve_format)
                    fixed.copy_(g)
                    p.grad = g = fixed
                    ptrs[i] = (p.data_ptr(), g.data_ptr())
                th, gp = ptrs[i]
                unaligned |= bool(th % 16) or bool(gp % 16)
                self.seg_host[i]["theta"], self.seg_host[i]["g"] = th, gp
            self._unaligned = unaligned
            self._ptr_cache = ptrs
            self._seg_dirty = True
        M = self.seg_host["M"]
        for i, m in enumerate(preconditioners):
            if M[i] != m:
                M[i] = m
                self._seg_dirty = True
                # a transition's pending bookkeeping reads ITS preconditioner from the segment table
                self.flush()
        self._precond_dirty = False
        if self._seg_dirty and not defer_upload:   # deferred: during hipGraph capture
            self._upload_segments()

    def _upload_segments(self):
        # the pinned staging buffer may still be in flight from the previous upload
        if self._seg_inflight():
            self._upload_event.synchronize()
        self._seg_pinned.numpy()[:] = self.seg_host.view(np.uint8)
        self._seg_dev.copy_(self._seg_pinned, non_blocking=True)
        self._upload_event = torch.cuda.Event()
        self._upload_event.record(torch.cuda.current_stream(self.device))
        self._seg_dirty = False

    def _seg_inflight(self):
        ev = getattr(self, "_upload_event", None)
        return ev is not None and not ev.query()

    def set_priors(self, specs, links=None):
        """specs[i] = None or (kind, loc, scale, shape-parameter): enable the in-kernel prior gradient;
        links[i] = index of the hyper segment whose VALUE is segment i's scale (hierarchical priors), or None"""
        self.prior_links = False
        for i, sp in enumerate(specs):
            row = self.seg_host[i]
            row["scale_link"] = 0
            if sp is None:
                row["prior_kind"] = _hip.PRIOR_NONE
            else:
                row["prior_kind"], row["prior_loc"], row["prior_scale"], row["prior_df"] = sp
                if links is not None and links[i] is not None:
                    row["scale_link"] = links[i] + 1
                    self.prior_links = True
        self.prior_max_kind = max([0] + [int(sp[0]) for sp in specs if sp is not None])
        # the layout says which prior code its table needs: lean-only entry points refuse the rest (no silent
        # Student-t evaluation of a generalised normal, no NaN placeholder scale of a linked segment)
        self.layout.prior_flags = self.prior_flags()
        self._seg_dirty = True

    def prior_flags(self):
        return ((_hip.PRIOR_HAS_LINKS if self.prior_links else 0)
                | (_hip.PRIOR_FULL if self.prior_max_kind > _hip.PRIOR_CAUCHY else 0))

    prior_links = False      # some segment takes its scale from a hyper segment
    prior_max_kind = 0

    # ------------------------------------------------------------------ deferred finalize
    # The fused dense step and the captured step leave a transition's per-segment bookkeeping pending (it is
    # not an input of the next gradient evaluation) and run it inside their NEXT launch (the gradient kernel /
    # the batch staging kernel); anything else that touches the partials / state calls flush() first.
    # (The bookkeeping reads a segment's M, numel and whether it HAS a gradient from the segment table, never
    # the pointers: re-binding p.grad to another graph's static tensors does not disturb it.)
    pending = None

    def flush(self):
        A = self.pending
        if A is not None:
            self.pending = None
            _hip.check(self.lib.sgmcmc_finalize(self.layout, A, self.stream()), "sgmcmc_finalize")

    # ------------------------------------------------------------------ launches
    def next_draw(self):
        d = self.draw
        self.draw += 1
        return d

    def _touch(self):
        self._state_host = None

    def make_args(self, gi, kind, flags, draw, *, num_data, b2h2, bh, bhn, mom_decay, grad_v,
                  noise_std, rmsprop_alpha, grad_clamp=0.0):
        s0, s1, c0, c1 = self.group_ranges[gi]
        if self.small_finalize:
            flags |= _hip.SMALL_FINALIZE
        return _hip.StepArgs(kind=kind, flags=flags | (_hip.UNALIGNED if self._unaligned else 0),
                             seg_begin=s0, seg_end=s1, chunk_begin=c0, chunk_end=c1,
                             num_data=num_data, b2h2=b2h2, bh=bh, bhn=bhn, mom_decay=mom_decay,
                             grad_v=grad_v, noise_std=noise_std, rmsprop_alpha=rmsprop_alpha,
                             grad_clamp=grad_clamp, seed=self.seed, draw=draw, stream=self.chain_id)

    def step_indirect(self, A_host, args_dev_ptr):
        "enqueue the fused transition with its scalars read from device memory (graph capture)"
        self.flush()
        _hip.check(self.lib.sgmcmc_step_indirect(ctypes.byref(self.layout), ctypes.byref(A_host),
                                                 ctypes.c_void_p(args_dev_ptr), self.stream()),
                   "sgmcmc_step_indirect")
        self._touch()

    def step(self, gi, kind, flags, draw, **scalars):
        self.flush()
        if flags & _hip.SAVE_STATE:
            self.ensure_prev()
        A = self.make_args(gi, kind, flags, draw, **scalars)
        c0, c1 = A.chunk_begin, A.chunk_end
        if self.kernel_timing:
            e0, e1 = self._event_pair()
            _hip.check(self.lib.sgmcmc_step_timed(ctypes.byref(self.layout), ctypes.byref(A),
                                                  self.stream(), e0, e1), "sgmcmc_step_timed")
            self._timed.append((e0, e1, (c1 - c0), flags))
        else:
            _hip.check(self.lib.sgmcmc_step(ctypes.byref(self.layout), ctypes.byref(A),
                                            self.stream()), "sgmcmc_step")
        self._touch()

    # ---- live kernel timing (bench.py roofline): HIP events around the fused update kernel
    kernel_timing = False

    def _event_pair(self):
        if not hasattr(self, "_event_pool"):
            self._event_pool, self._timed = [], []
        out = []
        for _ in range(2):
            ev = ctypes.c_void_p()
            _hip.check(self.lib.sgmcmc_event_create(ctypes.byref(ev)), "event_create")
            out.append(ev)
            self._event_pool.append(ev)
        return out

    def start_kernel_timing(self):
        self.kernel_timing = True
        self._event_pool, self._timed = getattr(self, "_event_pool", []), []

    def stop_kernel_timing(self):
        """-> list of (milliseconds, chunks, flags) for every fused-kernel launch since start"""
        self.kernel_timing = False
        out = []
        for e0, e1, n_chunks, flags in self._timed:
            ms = ctypes.c_float()
            _hip.check(self.lib.sgmcmc_event_elapsed_ms(e0, e1, ctypes.byref(ms)), "event_elapsed")
            out.append((ms.value, n_chunks, flags))
        for ev in self._event_pool:
            self.lib.sgmcmc_event_destroy(ev)
        self._event_pool, self._timed = [], []
        return out

    def sample_momentum(self, std, keep, draw):
        self.flush()
        _hip.check(self.lib.sgmcmc_sample_momentum(ctypes.byref(self.layout), std, keep, self.seed,
                                                   self.chain_id, draw, self.stream()),
                   "sgmcmc_sample_momentum")
        self.momentum_ready = True

    def restore(self, restore_momentum):
        self.flush()
        _hip.check(self.lib.sgmcmc_restore(ctypes.byref(self.layout), int(restore_momentum), 0,
                                           self.stream()), "sgmcmc_restore")

    def delta_energy_total(self, kind, num_data, b2h2, grad_clamp=0.0):
        self.flush()
        _hip.check(self.lib.sgmcmc_delta_energy(ctypes.byref(self.layout), kind, num_data, b2h2,
                                                grad_clamp, 0, self.stream()), "sgmcmc_delta_energy")
        self._touch()
        return self.scalars[0].item()

    def last_transition_energy(self):
        "scalars[3]: sum_s(delta_energy_s + point_energy_s) left by the last small-finalize launch"
        self.flush()
        return self.scalars[3]

    def segment_sums(self, which):
        self.flush()
        _hip.check(self.lib.sgmcmc_segment_sum(ctypes.byref(self.layout), which, 0, self.stream()),
                   "sgmcmc_segment_sum")
        self._touch()
        return self.fetch_state()[:, _hip.SEG_STATE_FIELDS.index("aux")].copy()

    def prior_grad(self, num_data, calc_log_prob):
        self.flush()
        _hip.check(self.lib.sgmcmc_prior_grad(ctypes.byref(self.layout), float(num_data),
                                              int(bool(calc_log_prob)),
                                              self.prior_flags(),
                                              self.stream()),
                   "sgmcmc_prior_grad")
        if calc_log_prob:
            self._touch()

    def log_prior_total(self):
        "sum of the fused priors' log-densities from the last prior_grad(calc_log_prob=True)"
        self.flush()
        return self.scalars[2]

    def nonfinite_seen(self, reset=True):
        self.flush()
        flag = self.scalars[1].item() != 0.0
        if flag and reset:
            self.scalars[1].zero_()
        return flag

    def fetch_state(self):
        "[n_seg, 12] float64 host copy of the per-segment scalars (one sync, cached)"
        if self._state_host is None:
            self.flush()
            self._state_host = self.state_dev.cpu().numpy().reshape(self.n_seg, -1)
        return self._state_host

    def seg_scalar(self, i, field):
        return float(self.fetch_state()[i, _hip.SEG_STATE_FIELDS.index(field)])

    def mh_uniform(self):
        return mh_uniform(self.seed, self.chain_id, self.next_draw())


class SegState(dict):
    """``optimizer.state[p]`` of the HIP samplers.

    Tensors (``momentum_buffer``, ``square_avg``, ``prev_*``) are views into the
    engine's arenas and ``preconditioner`` is a plain float the caller may
    assign (testing/test_hmc.py:25-26).  The running scalars ``delta_energy``,
    ``prev_new_momentum_delta``, ``est_temperature`` and ``est_config_temp`` live
    on the device; they are fetched (one D2H copy per launch epoch, shared by all
    tensors) the first time somebody reads them.
    """
    _LAZY = {"delta_energy": ("delta_energy", "energy_ready"),
             "prev_new_momentum_delta": ("prev_delta", "energy_ready"),
             "est_temperature": ("est_temperature", "metrics_ready"),
             "est_config_temp": ("est_config_temp", "metrics_ready")}

    def __init__(self, engine, index):
        super().__init__()
        self._engine, self._index = engine, index

    def _lazy_available(self, key):
        if key not in self._LAZY or key in self._engine.hidden_keys:
            return False
        return getattr(self._engine, self._LAZY[key][1])

    def __missing__(self, key):
        if self._lazy_available(key):
            return self._engine.seg_scalar(self._index, self._LAZY[key][0])
        raise KeyError(key)

    def __contains__(self, key):
        return dict.__contains__(self, key) or self._lazy_available(key)

    def __setitem__(self, key, value):
        if key == 'preconditioner':
            self._engine._precond_dirty = True
        dict.__setitem__(self, key, value)

    def setdefault(self, key, default=None):
        if key not in self:
            self[key] = default
        return self[key]

    def get(self, key, default=None):
        try:
            return self[key]
        except KeyError:
            return default
