#!/usr/bin/env python
"""bench.py -- leapfrog steps/sec of VerletSGLDReject on the MI355X path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload googleresnet|convnet|densenet]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot-loop body of the reject runner
(bnn_priors_amd.inference_reject.VerletSGLDRunnerReject.leapfrog, reference
inference_reject.py:86-113) over one synthetic minibatch of 128 that is already
resident in HBM: stochastic gradient of the average potential (forward + backward,
likelihood AND prior), the fused HIP sampler transition (momentum with friction and
in-kernel Philox noise, position, RMSprop statistic, the six energy / temperature
reductions), metric read-back every 10th step, cosine LR schedule.

Default workload: googleresnet on CIFAR-10-shaped batches (BASELINE.json configs[3], the
north-star target workload; it fits one GPU).  ``--workload densenet`` is configs[1],
``--workload convnet`` configs[2].

Timed region: after W untimed warm-up steps, BLOCKS of exactly K steps are run back to back
until at least ``--min-seconds`` (0.5 s) have been timed; the whole region is bracketed by
barrier + ``torch.cuda.synchronize()`` on both sides and every block boundary carries an event
on the launch stream.  ``value`` = K / (median block duration), max over ranks; the wall-clock
rate of the whole region is reported beside it (``wall_steps_per_s``).  With the driver's small
K the figure therefore does not depend on how long the host needs to fill the launch pipeline.

Chains are independent: rank r runs chain r on GPU r with Philox stream r and its own
synthetic data (seed 1234 + r); no collective on the data path ("scaling": "weak").  After the
timed region a multi-rank run performs the one exchange the path has -- the posterior-predictive
ensemble over all chains (two small all-reduces over RCCL) and the sample gather -- on synthetic
tables, checks it against the single-process formula (reference exp_utils.py:300-321) and reports
its time separately (``exchange``).  Rank 0 prints ONE JSON line.  See DESIGN.md "Measurement".
"""
import argparse
import ctypes
import json
import math
import os
import statistics
import sys
import time

# several chains per GPU run on their own HIP streams: give the process more than the default 4 hardware queues to map
# them on (read by the HIP runtime when it starts, hence before torch; multichain.concurrent_streams picks streams that
# really have a queue of their own).  One chain's rate does not depend on it (profiles/r05_chains_per_gpu.txt).
if __name__ == "__main__":        # (not when a test imports this file: the variable would leak into that process and its children)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the BASELINE configurations contain no library compute kernel in a gradient evaluation: make a silent fallback
# (a shape off the kernel tables dispatching to MIOpen / rocBLAS) an error instead of a slower number
if __name__ == "__main__":        # (not when a test imports this file: the variable would leak into that process and its children)
    os.environ.setdefault("SGMCMC_STRICT", "1")

HBM_PEAK_GBS = 8000.0      # MI355X spec (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
MFMA_F32_PEAK_TFLOPS = 157.3   # dense fp32-input MFMA (v_mfma_f32_16x16x4_f32), same guide
BYTES_PER_PARAM = 28       # fp32 intermediate step: read g, m, theta, v + write m, theta, v

WORKLOADS = {
    # name: (model, x shape, N, weight prior)  -- BASELINE.json configs[1], [2], [3]
    "densenet": ("classificationdensenet", (784,), 60000, "gaussian"),
    "convnet": ("classificationconvnet", (784,), 60000, "laplace"),
    "googleresnet": ("googleresnet", (3, 32, 32), 50000, "gaussian"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100, help="K: steps per timed block")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="googleresnet", choices=sorted(WORKLOADS))
    ap.add_argument("--min-seconds", type=float, default=0.5, help="keep timing blocks of K steps until this long")
    ap.add_argument("--max-blocks", type=int, default=400)
    ap.add_argument("--cpu-budget", type=float, default=30.0, help="seconds of CPU baseline, timed as three blocks (0 = skip)")
    ap.add_argument("--sweep-log2", type=int, default=28, help="flat-arena roofline point, log2(elements); 0 = skip")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--eager", action="store_true", help="no hipGraph capture (for comparison)")
    ap.add_argument("--cudnn-benchmark", type=int, default=1,
                    help="MIOpen find mode, as the reference sets it (experiments/train_bnn.py:29-31)")
    ap.add_argument("--channels-last", type=int, default=0)
    ap.add_argument("--samples", type=int, default=10, help="also time K full sample cycles (0 = skip)")
    ap.add_argument("--metrics-skip", type=int, default=10, help="BASELINE configs use 10")
    ap.add_argument("--inference", default="VerletSGLDReject",
                    choices=["VerletSGLDReject", "HMCReject", "SGLDReject"],
                    help="runner, as experiments/train_bnn.py:223-234 names them (headline: VerletSGLDReject)")
    ap.add_argument("--trajectory", type=int, default=0,
                    help="HMCReject only: leapfrog steps per trajectory (BASELINE configs[4]: 50); 0 = one epoch")
    ap.add_argument("--temperature", type=float, default=1.0)
    ap.add_argument("--weight-prior", default=None,
                    help="weight prior family (default: the workload's BASELINE config; HMCReject with --trajectory: "
                         "student-t, configs[4])")
    ap.add_argument("--augment", type=int, default=1,
                    help="googleresnet: every minibatch is gathered from the HBM-resident set through the on-device "
                         "random crop (pad 4) + flip, as configs[3]'s cifar10_augmented (data/CIFAR/cifar.py:136-172)")
    ap.add_argument("--exchange-samples", type=int, default=8, help="synthetic samples per chain in the exchange leg")
    ap.add_argument("--chain-sweep", default="1,2,4,8",
                    help="densenet only: aggregate steps/s of K chains sharing ONE GPU's launches (MultiChainDense); '' = skip")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend of a multi-rank run: nccl = RCCL over xGMI (the product path); gloo = "
                         "the SAME control flow with host collectives, so that two ranks can share one GPU "
                         "(a plumbing check of the multi-rank leg, never a scaling number)")
    ap.add_argument("--other-workloads", type=int, default=None,
                    help="also run BASELINE configs[1], [2], [4] (densenet, convnet, googleresnet HMC L=50 T=0.1) as "
                         "sub-runs and report them in `other_workloads` (default: on for the default one-GPU run)")
    ap.add_argument("--width", type=int, default=50,
                    help="hidden width of densenet / convnet (experiments/train_bnn.py:53-54 makes it a user option; the "
                         "BASELINE configs use 50).  Widths off the kernel tables run on the library path: needs "
                         "SGMCMC_STRICT=0 in the environment, and the line says so in config.step_path")
    ap.add_argument("--eval-rows", type=int, default=10000,
                    help="rows of the synthetic test set of `samples_per_sec_with_eval` (CIFAR-10's test set: 10,000)")
    ap.add_argument("--detail", default=os.path.join(ROOT, "bench_detail.json"),
                    help="side file of the FULL record (per-kernel roofline rows, other workloads, timing, chains per GPU, "
                         "calibration); stdout carries only the compact line (< 4 KB)")
    ap.add_argument("--stream-chains", default=None,
                    help="googleresnet / convnet: aggregate steps/s of K chains on K HIP streams of ONE GPU, e.g. '1,2,3' "
                         "(after the timed region; default: '1,2,4' for a one-GPU googleresnet run, '' = skip)")
    return ap.parse_args()


class PoolSource:
    """Synthetic device-resident data set of N rows with the interface the runner expects from
    its batch source: tensor batches (exact-gradient pass of the generic path), index batches
    (fused dense step), ``x`` / ``y`` (fused exact pass)."""
    fast = True

    def __init__(self, workload, n_rows, device, seed):
        _, xshape, _, _ = WORKLOADS[workload]
        g = torch.Generator(device=device).manual_seed(seed)
        if workload == "googleresnet":
            self.x = torch.randn((n_rows,) + xshape, generator=g, device=device)
        else:
            self.x = torch.rand((n_rows,) + xshape, generator=g, device=device)
        self.y = torch.randint(0, 10, (n_rows,), generator=g, device=device)
        self.n_rows, self.n_batches = n_rows, -(-n_rows // 128)

    def __len__(self):
        return self.n_batches

    # (the three hooks of inference._BatchSource that the exact pass uses: in-place filling of a consumer's buffers,
    # the number of full-size minibatches, a minibatch's shapes)
    _provider = None

    def filling(self, provider):
        import contextlib

        @contextlib.contextmanager
        def scope():
            old, self._provider = self._provider, provider
            try:
                yield self
            finally:
                self._provider = old
        return scope()

    def n_full_batches(self):
        return self.n_rows // 128

    def example(self):
        return self.x[:128], self.y[:128]

    def __iter__(self):
        for b in range(self.n_batches):
            x, y = self.x[128 * b:128 * (b + 1)], self.y[128 * b:128 * (b + 1)]
            dst = self._provider(len(x)) if self._provider is not None else None
            if dst is not None:
                x, y = dst[0].copy_(x), dst[1].copy_(y)
            yield x, y

    def index_batches(self):
        import numpy as np
        from bnn_priors_amd.fused_dense import IndexBatch
        for b in range(self.n_batches):
            idx = np.arange(128 * b, min(128 * (b + 1), self.n_rows), dtype=np.int64)
            yield IndexBatch(idx, self.x, self.y), None


def make_model(workload, device, prior=None, width=50):
    from bnn_priors_amd import models
    name, xshape, _, default_prior = WORKLOADS[workload]
    prior = prior or default_prior
    torch.manual_seed(0)
    x0 = torch.zeros((2,) + xshape)
    net = models.get_model(x0, torch.tensor([0, 9]), name, width=width, depth=3, weight_prior=prior,
                           weight_scale=2 ** .5, bias_prior="gaussian", bias_scale=1.)
    models.he_initialize(net)
    return net.to(device)


class _SyntheticSet(torch.utils.data.Dataset):
    "length-only stand-in: the runner needs len(dataset) = N and len(dataloader) = ceil(N/128)"

    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n


# ------------------------------------------------------------------ live kernel timing
class PacketTimer:
    """Durations of single kernels, measured live: ``sgmcmc_time_next_launch`` makes the next launch of
    the library carry a start / stop event in its dispatch packet, so the elapsed time between the two is
    the kernel's own execution time on the stream it runs on (what rocprofv3 lists per dispatch)."""

    def __init__(self):
        from bnn_priors_amd import _hip
        self._hip, self.lib, self.pairs = _hip, _hip.lib(), []

    def arm(self):
        e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
        self._hip.check(self.lib.sgmcmc_event_create(ctypes.byref(e0)), "event_create")
        self._hip.check(self.lib.sgmcmc_event_create(ctypes.byref(e1)), "event_create")
        self._hip.check(self.lib.sgmcmc_time_next_launch(e0, e1), "time_next_launch")
        self.pairs.append((e0, e1))

    def collect_ms(self):
        out = []
        for e0, e1 in self.pairs:
            ms = ctypes.c_float()
            self._hip.check(self.lib.sgmcmc_event_elapsed_ms(e0, e1, ctypes.byref(ms)), "event_elapsed")
            out.append(ms.value)
            self.lib.sgmcmc_event_destroy(e0)
            self.lib.sgmcmc_event_destroy(e1)
        self.pairs = []
        return out


def conv_rooflines(device, n_img=128, iters=60):
    """The ResNet trunk's convolution kernels against the fp32-MFMA peak: each of the three trunk shapes,
    forward (``conv3x3_kernel`` + batch statistics) and both gradients -- ``conv3x3_bwd_kernel``, and
    ``fused_bwd_kernel`` (BatchNorm backward formed while staging) for the shapes whose blocks take that route
    (resblock.FUSED_BN_BWD) -- launched through the C ABI on synthetic tensors with the same shapes as in the step.
    Algorithmic flops: 2 * N * HW^2 * C^2 * 9 per convolution-shaped contraction (forward: one; backward: two).
    ``launches_per_step``: how often googleresnet's step runs the kernel (6 + 5 + 5 trunk convolutions; 0 = an
    alternative route, timed for comparison)."""
    from bnn_priors_amd import _hip, resblock
    lib = _hip.lib()
    stream = torch.cuda.current_stream(device).cuda_stream
    rows = []
    for c, hw in ((16, 32), (32, 16), (64, 8)):
        g = torch.Generator(device=device).manual_seed(c)
        x = torch.randn((n_img, c, hw, hw), generator=g, device=device)
        dy = torch.randn((n_img, c, hw, hw), generator=g, device=device)
        w = torch.randn((c, c, 3, 3), generator=g, device=device) * (2.0 / (9 * c)) ** .5
        y, dx, dw = torch.empty_like(x), torch.empty_like(x), torch.empty_like(w)
        stats = torch.empty((c, lib.sgmcmc_conv3x3_stat_slices(n_img, c, hw), 2), dtype=torch.float64, device=device)
        scratch = torch.empty(lib.sgmcmc_conv3x3_wrw_scratch_floats(n_img, c, hw), device=device)
        slabs = ctypes.c_int(0)
        flop1 = 2.0 * n_img * hw * hw * c * c * 9
        out = torch.relu(torch.randn((n_img, c, hw, hw), generator=g, device=device))
        sums = torch.zeros(lib.sgmcmc_bn_scratch_doubles(n_img, c, hw * hw, 1), dtype=torch.float64, device=device)
        saved = torch.stack([torch.zeros(c, device=device), torch.ones(c, device=device)])
        gamma, dgb = torch.ones(c, device=device), torch.empty((2, c), device=device)
        n_sums = ctypes.c_int(0)
        _hip.check(lib.sgmcmc_bn_bwd_sums(dy.data_ptr(), out.data_ptr(), y.data_ptr(), saved[0].data_ptr(),
                                          saved[1].data_ptr(), sums.data_ptr(), ctypes.byref(n_sums), n_img, c,
                                          hw * hw, 1, stream), "sgmcmc_bn_bwd_sums")
        A = _hip.ConvBnBwdArgs(dout=dy.data_ptr(), mask_out=out.data_ptr(), y=y.data_ptr(), mean=saved[0].data_ptr(),
                               invstd=saved[1].data_ptr(), gamma=gamma.data_ptr(), sums=sums.data_ptr(),
                               n_sums=n_sums.value, reserved=0, dgamma=dgb[0].data_ptr(), dbeta=dgb[1].data_ptr(),
                               e_dout=0, e_out=0)
        fused_route = (not resblock.EPILOGUE_SUMS) and (c, hw) in resblock.FUSED_BN_BWD
        part = torch.empty((c, lib.sgmcmc_conv3x3_stat_slices(n_img, c, hw), 2), dtype=torch.float64, device=device)
        # (as the step launches them since round 4: dx stored masked, the shortcut's gradient added as it arrives)
        E = _hip.ConvBwdEpilogue(s_y=y.data_ptr(), s_out=out.data_ptr(), s_mean=saved[0].data_ptr(),
                                 s_invstd=saved[1].data_ptr(), s_partial=part.data_ptr(), mask_dx=1)
        n_convs = 6 if c == 16 else 5          # trunk convolutions of this shape in googleresnet (depth 20)

        def fwd():
            _hip.check(lib.sgmcmc_conv3x3(x.data_ptr(), w.data_ptr(), y.data_ptr(), n_img, c, hw, 0,
                                          stats.data_ptr(), stream), "sgmcmc_conv3x3")

        def bwd():
            _hip.check(lib.sgmcmc_conv3x3_bwd(x.data_ptr(), w.data_ptr(), dy.data_ptr(), dx.data_ptr(),
                                              dw.data_ptr(), scratch.data_ptr(), n_img, c, hw,
                                              ctypes.byref(slabs), stream), "sgmcmc_conv3x3_bwd")
        def bwd_sums():     # as the step runs it: the next BatchNorm backward's sums in the data gradient's epilogue
            _hip.check(lib.sgmcmc_conv3x3_bwd_ex(x.data_ptr(), w.data_ptr(), dy.data_ptr(), dx.data_ptr(), ctypes.byref(E),
                                                 dw.data_ptr(), scratch.data_ptr(), n_img, c, hw, ctypes.byref(slabs),
                                                 stream), "sgmcmc_conv3x3_bwd_ex")

        E2 = _hip.ConvBwdEpilogue(e_dout=dy.data_ptr(), e_out=0, s_y=y.data_ptr(), s_out=out.data_ptr(),
                                  s_mean=saved[0].data_ptr(), s_invstd=saved[1].data_ptr(), s_partial=part.data_ptr(),
                                  mask_dx=1)

        def bwd_add_sums():  # an identity block's FIRST convolution: + the shortcut's gradient in the epilogue
            _hip.check(lib.sgmcmc_conv3x3_bwd_ex(x.data_ptr(), w.data_ptr(), dy.data_ptr(), dx.data_ptr(), ctypes.byref(E2),
                                                 dw.data_ptr(), scratch.data_ptr(), n_img, c, hw, ctypes.byref(slabs),
                                                 stream), "sgmcmc_conv3x3_bwd_ex")

        def bn_bwd():
            _hip.check(lib.sgmcmc_conv3x3_bn_bwd(x.data_ptr(), w.data_ptr(), dx.data_ptr(), scratch.data_ptr(),
                                                 ctypes.byref(A), n_img, c, hw, ctypes.byref(slabs), stream),
                       "sgmcmc_conv3x3_bn_bwd")
        # (the down-sampling block's second convolution has no identity-shortcut block around it: plain route)
        n_add = 3 if c == 16 else 2            # identity blocks of this stage: their first convolution's gradient
        table = [(f"conv::conv3x3_kernel<{c},{hw},8,stats>", fwd, flop1, n_convs),
                 (f"conv::conv3x3_bwd_kernel<{c},{hw},8>", bwd, 2 * flop1, 0),
                 (f"conv::conv3x3_bwd_kernel<{c},{hw},8,SUMS>", bwd_sums, 2 * flop1,
                  n_convs - n_add if not fused_route else 1),
                 (f"conv::conv3x3_bwd_kernel<{c},{hw},8,ADD,SUMS>", bwd_add_sums, 2 * flop1, n_add if not fused_route else 0)]
        if _hip.ALTERNATIVES:       # (a measured alternative: only in a library built with SGMCMC_ALTERNATIVES=1)
            table.append((f"conv::fused_bwd_kernel<{c},{hw},8>", bn_bwd, 2 * flop1, n_convs - 1 if fused_route else 0))
        for name, fn, flops, per_step in table:
            for _ in range(5):
                fn()
            torch.cuda.synchronize(device)
            t = PacketTimer()
            for _ in range(iters):
                t.arm()
                fn()
            ms = t.collect_ms()
            avg = sum(ms) / len(ms)
            tf = flops / (avg * 1e-3) / 1e12
            rows.append(dict(kernel=name, bound="mfma", achieved=round(tf, 2), peak=MFMA_F32_PEAK_TFLOPS,
                             unit="TFLOP/s", frac=round(tf / MFMA_F32_PEAK_TFLOPS, 4), traffic=None,
                             algorithmic_flops_per_launch=flops, avg_kernel_us=round(avg * 1e3, 3),
                             min_kernel_us=round(min(ms) * 1e3, 3), launches=len(ms), launches_per_step=per_step,
                             shape=dict(n=n_img, channels=c, hw=hw)))
    return rows


def _stale(committed):
    """True when a committed measurement (profiles/in_step_us.json, pmc_traffic.json) was taken on other kernel sources
    than this tree's: its ``source_sha`` (hash of csrc/* + include/*, bnn_priors_amd._hip.source_sha) differs or is absent."""
    from bnn_priors_amd import _hip
    return committed.get("source_sha") != _hip.library_sha()


def attach_pmc_traffic(rows):
    """`traffic` of the rows that profiles/pmc_traffic.json covers: HBM bytes per launch from a rocprofv3 --pmc pass
    (FETCH_SIZE doubled as the guide prescribes for gfx950, + WRITE_SIZE) over the same kernels at the same shapes.
    Counters cannot be collected from inside this process; the file is a committed measurement (tools/pmc_to_json.py
    names the pass) and `traffic_source` says so.  Rows it does not cover keep traffic = null."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            pmc = json.load(f)
    except (OSError, ValueError):
        return rows
    stale = _stale(pmc)
    for r in rows:
        k = pmc.get("kernels", {}).get(r.get("kernel"))
        if k and r.get("shape", {}).get("n") == 128:
            if stale:           # counters of OTHER kernels than this tree's: not this launch's traffic
                r["traffic_stale"] = True
                continue
            r["traffic"] = k["read_bytes"] + k["write_bytes"]
            r["traffic_unit"] = "bytes per launch (HBM read + write)"
            if "traffic_over_algorithmic" in k:
                r["algorithmic_bytes_per_launch"] = k["algorithmic_bytes"]
                r["traffic_over_algorithmic"] = k["traffic_over_algorithmic"]
            r["traffic_source"] = "profiles/pmc_traffic.json <- " + pmc.get("source", "?")
    return rows


def attach_in_step(rows):
    """`in_step_us` / `frac_in_step` of the rows that profiles/in_step_us.json covers: the kernel's mean duration INSIDE
    the captured step (rocprofv3 --kernel-trace over this script, tools/step_summary.py --json; the rocprof summary it
    was reduced from is named in `in_step_source`) and the roofline fraction that duration gives.  An isolated launch
    (avg_kernel_us, measured live above) finds its operands warm and no neighbour's write-back in flight: 5-15 %
    optimistic against the step.  Rows the file does not cover keep only the live figure."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "in_step_us.json")
    try:
        with open(path) as f:
            prof = json.load(f)
    except (OSError, ValueError):
        return rows
    stale = _stale(prof)
    for r in rows:
        k = prof.get("kernels", {}).get(r.get("kernel"))
        if k and r.get("shape", {}).get("n") == 128:
            if stale:           # durations of OTHER kernels than this tree's: no in-step columns, the live figure stays
                r["stale"] = True
                continue
            r["in_step_us"] = k["in_step_us"]
            r["in_step_launches_per_step"] = k["launches_per_step"]
            work = r.get("algorithmic_flops_per_launch") or r.get("algorithmic_bytes_per_launch")
            if work:
                ach = work / (k["in_step_us"] * 1e-6) / (1e12 if r["unit"] == "TFLOP/s" else 1e9)
                r["achieved_in_step"] = round(ach, 2)
                r["frac_in_step"] = round(ach / r["peak"], 4)
            r["in_step_source"] = "profiles/in_step_us.json <- " + prof.get("source", "?")
    return rows


def bn_rooflines(device, n_img=128, iters=60):
    """The trunk's BatchNorm kernels against HBM: ``bn::apply_kernel`` (normalise + [residual] + ReLU from the
    producing convolution's statistics partials) and ``bn::bwd_dx_kernel`` (the whole BatchNorm backward given the
    sums partials of the upstream epilogue), at the three trunk stages, launched through the C ABI with the
    partials a convolution of the same shape leaves.  Algorithmic bytes per element: apply 8 (read y, write out;
    12 with the residual), backward 12 (read dz, y; write dy -- since round 4 the launch that produces the incoming
    gradient stores it masked, dz = dout * [out > 0], so the step's backward launches are the ReLU-less instantiation
    and do not read `out`: bnlink.PREMASK).  21 + 21 launches per googleresnet step."""
    from bnn_priors_amd import _hip, bn as _bn, conv as _conv
    lib = _hip.lib()
    stream = torch.cuda.current_stream(device).cuda_stream
    rows = []
    for c, hw in ((16, 32), (32, 16), (64, 8)):
        g = torch.Generator(device=device).manual_seed(100 + c)
        x = torch.randn((n_img, c, hw, hw), generator=g, device=device)
        w = torch.randn((c, c, 3, 3), generator=g, device=device) * (2.0 / (9 * c)) ** .5
        res = torch.randn((n_img, c, hw, hw), generator=g, device=device)
        dout = torch.randn((n_img, c, hw, hw), generator=g, device=device)
        y, stats = _conv._run(x, w, False, True)
        slices = stats.shape[1]
        gamma, beta = torch.ones(c, device=device), torch.zeros(c, device=device)
        rm, rv = torch.zeros(c, device=device), torch.ones(c, device=device)
        out, dy = torch.empty_like(y), torch.empty_like(y)
        saved = torch.empty((2, c), dtype=torch.float32, device=device)
        dgb = torch.empty((2, c), dtype=torch.float32, device=device)
        elems = y.numel()
        # the sums partials as the upstream data gradient's epilogue leaves them (same slice count as the statistics)
        partial = torch.zeros((c, slices, 2), dtype=torch.float64, device=device)
        n_convs = 6 if c == 16 else 5

        def apply(residual):
            _hip.check(_bn.train_fwd(lib, y, residual, gamma, beta, rm, rv, 0.1, 1e-5, 1, n_img, c, hw * hw, out, saved,
                                     None, stats, slices, stream), "sgmcmc_bn_train_fwd")

        def bwd_dx():
            _hip.check(lib.sgmcmc_bn_bwd_dx(dout.data_ptr(), 0, y.data_ptr(), gamma.data_ptr(),
                                            saved[0].data_ptr(), saved[1].data_ptr(), 0, n_img, c, hw * hw,
                                            partial.data_ptr(), slices, dy.data_ptr(), 0, dgb.data_ptr(), None, 1,
                                            stream), "sgmcmc_bn_bwd_dx")
        for name, fn, nbytes, per_step in (
                (f"bn::apply_kernel<relu> {c}@{hw}^2", lambda: apply(None), 8 * elems, (n_convs + 1) // 2 + (1 if c == 16 else 0)),
                (f"bn::apply_kernel<relu,residual> {c}@{hw}^2", lambda: apply(res), 12 * elems, n_convs // 2 + (1 if c > 16 else 0)),
                (f"bn::bwd_dx_kernel<> {c}@{hw}^2", bwd_dx, 12 * elems, n_convs + 1 + (1 if c > 16 else 0))):
            for _ in range(5):
                fn()
            torch.cuda.synchronize(device)
            t = PacketTimer()
            for _ in range(iters):
                t.arm()
                fn()
            ms = t.collect_ms()
            avg = sum(ms) / len(ms)
            gbs = nbytes / (avg * 1e-3) / 1e9
            rows.append(dict(kernel=name, bound="hbm", achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                             frac=round(gbs / HBM_PEAK_GBS, 4), traffic=None, algorithmic_bytes_per_launch=nbytes,
                             avg_kernel_us=round(avg * 1e3, 3), min_kernel_us=round(min(ms) * 1e3, 3),
                             launches=len(ms), launches_per_step=per_step,
                             regime="launch-latency bound: the tensor is cache resident and smaller than a launch's "
                                    "fixed cost", shape=dict(n=n_img, channels=c, hw=hw)))
    return rows


def flat_arena_point(log2n, device, iters=20):
    """The same fused kernel on ONE flat segment of 2^log2n fp32 elements (bandwidth-bound
    regime): GB/s of algorithmic traffic (28 B/element), kernel durations measured live."""
    from bnn_priors_amd import mcmc
    n = 1 << log2n
    p = torch.nn.Parameter(torch.zeros(n, device=device))
    p.grad = torch.full((n,), 1e-3, device=device)
    opt = mcmc.VerletSGLD([p], lr=1e-4, num_data=1000, momentum=0.99, temperature=1.0, seed=1)
    opt.sample_momentum()
    opt.initial_step(save_state=False, calc_metrics=False)
    for _ in range(3):
        opt.step(calc_metrics=False)
    torch.cuda.synchronize(device)
    opt.engine.start_kernel_timing()
    for _ in range(iters):
        opt.step(calc_metrics=False)
    times = [ms for ms, _, _ in opt.engine.stop_kernel_timing()]
    avg_ms = sum(times) / len(times)
    gbs = BYTES_PER_PARAM * n / (avg_ms * 1e-3) / 1e9
    return dict(bound="hbm", kernel="step_kernel_stream<float, VERLET>" if log2n >= 24 else "step_kernel<float, VERLET, vec>",
                achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(gbs / HBM_PEAK_GBS, 4),
                traffic=None, algorithmic_bytes_per_launch=BYTES_PER_PARAM * n, elements=n,
                avg_kernel_ms=round(avg_ms, 4), min_kernel_ms=round(min(times), 4), launches=len(times))


# ------------------------------------------------------------------ several chains per GPU (captured-graph nets)
def chains_per_gpu_streams(args, device, rank, ks, steps=200, warmup=30, cycles=0):
    """K independent chains of a captured-graph workload (googleresnet, convnet) on ONE GPU, each on its own HIP
    stream: the chains' dependent launch chains interleave on the GPU (a step is ~90 dependent kernels, each of which
    leaves the GPU partly idle at its boundaries).  Aggregate leapfrog steps/s = K * steps / time, per K."""
    from bnn_priors_amd.inference_reject import runner_class
    from bnn_priors_amd.storage import MemoryMetrics
    name, xshape, N, prior = WORKLOADS[args.workload]
    pool = PoolSource(args.workload, N, device, 4321 + rank)           # read-only: shared by the chains
    batches = [b for b in pool if len(b[0]) == 128]
    from bnn_priors_amd import multichain
    runners, streams, steps_of = [], [], []
    # streams that do not share a hardware queue, at most multichain.MAX_CHAIN_STREAMS of them (chains beyond that share)
    picked = multichain.chain_streams(max(ks), device)
    multichain.reserve(picked, device)                             # (the chains' exact passes take their lanes from the rest)
    for c in range(max(ks)):
        st = picked[c]
        with torch.cuda.stream(st):
            model = make_model(args.workload, device, args.weight_prior)
            loader = torch.utils.data.DataLoader(_SyntheticSet(N), batch_size=128, shuffle=True)
            empty = torch.utils.data.DataLoader(_SyntheticSet(0), batch_size=128)
            r = runner_class("VerletSGLDReject")(
                model=model, dataloader=loader, dataloader_test=empty, epochs_per_cycle=50, warmup_epochs=45,
                sample_epochs=5, learning_rate=0.01, skip=1, metrics_skip=args.metrics_skip, temperature=1.0,
                momentum=0.994, sampling_decay="cosine", cycles=60, precond_update=1, metrics_saver=MemoryMetrics(),
                model_saver=None, reject_samples=True, seed=1234, chain_id=8 * rank + c)
            r._batch_source = pool
            r.use_graph = True
            steps_of.append(r.begin())
        runners.append(r)
        streams.append(st)
    torch.cuda.synchronize(device)
    out = {}
    for K in sorted(ks):
        def run(n):
            for _ in range(n):
                for c in range(K):
                    steps_of[c] += 1
                    x, y = batches[(steps_of[c] + 13 * c) % len(batches)]
                    with torch.cuda.stream(streams[c]):
                        runners[c].leapfrog(steps_of[c], x, y, last_of_epoch=False)
        run(warmup)
        for c in range(K):
            with torch.cuda.stream(streams[c]):
                runners[c]._drain_rows()
        torch.cuda.synchronize(device)
        reps = []
        for _ in range(3):                         # (three blocks, the median reported: a block is only 0.2 - 0.4 s)
            ts = time.perf_counter()
            run(steps)
            t_issued = time.perf_counter() - ts    # the host is done issuing: below the wall time = the GPU is the bound
            for c in range(K):
                with torch.cuda.stream(streams[c]):
                    runners[c]._drain_rows()
            torch.cuda.synchronize(device)
            reps.append((time.perf_counter() - ts, t_issued))
        dt, t_issued = sorted(reps)[1]
        for c in range(K):
            runners[c]._check_finite()
        out[str(K)] = {"aggregate_steps_per_s": round(K * steps / dt, 1), "per_chain_steps_per_s": round(steps / dt, 1),
                       "us_per_lockstep": round(dt / steps * 1e6, 2),
                       "host_issue_us_per_lockstep": round(t_issued / steps * 1e6, 2),
                       "block_us_per_lockstep": [round(r[0] / steps * 1e6, 1) for r in reps]}
        if cycles > 0:
            # stored samples/s of the K chains together: every chain walks ITS epoch (the runner's own batch stream,
            # ragged last minibatch included) in lock-step with the others, then each takes its Metropolis-Hastings point
            # (exact full-data pass on the measured lanes, final_step, test, initial_step) -- what
            # multichain.run_on_streams does per sample cycle (inference_reject.py:86-157 per chain)
            def cycle():
                gens = [runners[c]._hot_batches() for c in range(K)]
                n_b = len(runners[0]._batches())
                accs = [0.0] * K
                for i, row in enumerate(zip(*gens)):
                    for c, (x, y) in enumerate(row):
                        steps_of[c] += 1
                        with torch.cuda.stream(streams[c]):
                            accs[c] = runners[c].leapfrog(steps_of[c], x, y, last_of_epoch=(i == n_b - 1))
                for c in range(K):
                    with torch.cuda.stream(streams[c]):
                        runners[c]._drain_rows()
                        steps_of[c] = runners[c]._mh_point(steps_of[c], accs[c], runners[c]._batches())
            cycle()
            torch.cuda.synchronize(device)
            ts = time.perf_counter()
            for _ in range(cycles):
                cycle()
            torch.cuda.synchronize(device)
            out[str(K)]["aggregate_samples_per_s"] = round(K * cycles / (time.perf_counter() - ts), 3)
    out["distinct_hw_queues"] = len({st.cuda_stream for st in picked})
    out["max_chain_streams"] = multichain.MAX_CHAIN_STREAMS
    return out


# ------------------------------------------------------------------ several chains per GPU (the small nets)
def chains_per_gpu_sweep(args, device, rank, ks, steps=600, warmup=60):
    """K independent chains of the dense classifier on ONE GPU, stepped in lock-step by three launches per step
    (fused_dense.MultiChainDense): aggregate leapfrog steps/s = K * steps / time, per K.  Every chain has its own
    model, synthetic data (seed 1234 + chain), sampler arena and Philox stream; metric read-back every 10th step."""
    from bnn_priors_amd.fused_dense import MultiChainDense
    from bnn_priors_amd.inference_reject import runner_class
    from bnn_priors_amd.storage import MemoryMetrics
    import numpy as np
    name, xshape, N, prior = WORKLOADS["densenet"]
    runners, pools = [], []
    for c in range(max(ks)):
        model = make_model("densenet", device)
        pool = PoolSource("densenet", N, device, 1234 + 8 * rank + c)
        loader = torch.utils.data.DataLoader(_SyntheticSet(N), batch_size=128, shuffle=True)
        empty = torch.utils.data.DataLoader(_SyntheticSet(0), batch_size=128)
        r = runner_class("VerletSGLDReject")(
            model=model, dataloader=loader, dataloader_test=empty, epochs_per_cycle=50, warmup_epochs=45,
            sample_epochs=5, learning_rate=0.01, skip=1, metrics_skip=args.metrics_skip, temperature=1.0,
            momentum=0.994, sampling_decay="cosine", cycles=60, precond_update=1, metrics_saver=MemoryMetrics(),
            model_saver=None, reject_samples=True, seed=1234, chain_id=8 * rank + c)
        r._batch_source = pool
        r.begin()
        runners.append(r)
        pools.append(pool)
    out = {}
    nb = N // 128
    for K in sorted(ks, reverse=True):      # largest first: the chains of a smaller K have advanced together so far
        multi = MultiChainDense([r._fused_dense() for r in runners[:K]])
        idx = [[np.arange(128 * ((t + 7 * c) % nb), 128 * ((t + 7 * c) % nb) + 128, dtype=np.int64) for c in range(K)]
               for t in range(64)]

        def run(n, t0):
            for t in range(t0, t0 + n):
                multi.step(idx[t % 64], metrics=(t % args.metrics_skip == 0))
                for r in runners[:K]:
                    r.scheduler.step()
            return t0 + n
        t = run(warmup, 1)
        torch.cuda.synchronize(device)
        ts = time.perf_counter()
        t = run(steps, t)
        for r in runners[:K]:
            r.optimizer.engine.flush()
        torch.cuda.synchronize(device)
        dt = time.perf_counter() - ts
        out[str(K)] = {"aggregate_steps_per_s": round(K * steps / dt, 1), "per_chain_steps_per_s": round(steps / dt, 1),
                       "us_per_lockstep": round(dt / steps * 1e6, 2)}
    return {k: out[k] for k in sorted(out, key=int)}


# ------------------------------------------------------------------ the one exchange of the multi-chain path
def _exchange_tables(rank, E, N, C, device):
    g = torch.Generator().manual_seed(100 + rank)
    logits = torch.randn(E, N, C, generator=g, dtype=torch.float64) * 3
    acc = logits - logits.logsumexp(-1, keepdim=True)
    y = torch.randint(0, C, (N,), generator=torch.Generator().manual_seed(7))
    lps = acc.gather(-1, y.view(1, N, 1).expand(E, N, 1)).squeeze(-1)
    return lps.to(device), acc.to(device)


def exchange_leg(model, rank, world, device, E, cdev=None, backend="nccl", n_test=10000, classes=10):
    """Posterior-predictive ensemble over ALL chains' samples (two all-reduces, MAX and SUM, over RCCL) and
    the gather of every chain's stored samples to rank 0, on synthetic [E, N_test, C] tables / E copies of
    the model's state; the ensemble is checked on rank 0 against the single-process formula applied to the
    concatenated tables (exp_utils.py:300-321)."""
    import torch.distributed as dist
    from bnn_priors_amd.evaluation import ensemble_across_chains, gather_samples
    cdev = device if cdev is None else cdev
    lps, acc = _exchange_tables(rank, E, n_test, classes, cdev)
    samples = {k: (v.detach().unsqueeze(0).repeat((E,) + (1,) * v.dim()) + float(rank)).to(cdev)
               for k, v in model.state_dict().items() if v.is_floating_point()}
    ensemble_across_chains(lps, acc)                # untimed first use: RCCL sets its rings up lazily
    torch.cuda.synchronize(device)
    dist.barrier()
    t0 = time.perf_counter()
    lp, ens = ensemble_across_chains(lps, acc)
    torch.cuda.synchronize(device)
    t_ens = time.perf_counter() - t0
    dist.barrier()
    t0 = time.perf_counter()
    got = gather_samples(samples)
    torch.cuda.synchronize(device)
    t_gather = time.perf_counter() - t0
    t = torch.tensor([t_ens, t_gather], dtype=torch.float64, device=cdev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    out = None
    if rank == 0:
        parts = [_exchange_tables(r, E, n_test, classes, "cpu") for r in range(world)]
        all_lps, all_acc = torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts])
        ref_lp = all_lps.logsumexp(0) - math.log(all_lps.shape[0])
        ref_ens = all_acc.logsumexp(0) - math.log(all_acc.shape[0])
        err = max((lp.cpu() - ref_lp).abs().max().item(), (ens.cpu() - ref_ens).abs().max().item())
        n_bytes = sum(v.numel() * v.element_size() for v in got.values())
        ok_gather = all(v.shape[0] == E * world for v in got.values())
        k0 = sorted(got)[0]
        ok_gather = ok_gather and all(
            bool((got[k0][r * E:(r + 1) * E] - samples[k0][0:1]).abs().max().item() == float(r)) for r in range(world))
        out = dict(backend="nccl (RCCL)" if backend == "nccl" else "gloo (host collectives: a plumbing check, ranks may share a GPU)",
                   chains=world, samples_per_chain=E, n_test=n_test, classes=classes,
                   ensemble_ms=round(t[0].item() * 1e3, 3), ensemble_max_abs_err=err,
                   ensemble_matches_single_process=bool(err < 1e-9),
                   gather_ms=round(t[1].item() * 1e3, 3), gathered_bytes_rank0=n_bytes,
                   gather_order_checked=bool(ok_gather),
                   collectives="all_reduce(MAX) + all_reduce(SUM) on [N_test, C+1] float64; all_gather per stored tensor")
        if not (out["ensemble_matches_single_process"] and ok_gather):
            raise AssertionError(f"multi-chain exchange disagrees with the single-process formula: {out}")
    return out


def other_workloads(args, only=None):
    """BASELINE configs[1], [2] and [4] as sub-runs of this script (fresh processes: their own captured graphs and
    kernel tables), so that the ONE line the driver records carries all four GPU configurations.  Each sub-run times
    its leapfrog loop exactly as the headline does and K = 10 full sample cycles; no rooflines, no CPU baseline."""
    import subprocess
    subs = {
        "configs[1] densenet VerletSGLDReject": ["--workload", "densenet", "--chain-sweep", ""],
        "configs[2] convnet VerletSGLDReject laplace": ["--workload", "convnet"],
        "configs[4] googleresnet HMCReject L=50 T=0.1 student-t": ["--inference", "HMCReject", "--trajectory", "50",
                                                                  "--temperature", "0.1"],
        "configs[4] googleresnet HMCReject L=50 T=1 student-t": ["--inference", "HMCReject", "--trajectory", "50",
                                                                "--temperature", "1.0"],
        "configs[4] googleresnet HMCReject L=50 T=0.01 student-t": ["--inference", "HMCReject", "--trajectory", "50",
                                                                   "--temperature", "0.01"],
    }
    import tempfile
    out = {}
    for name, flags in subs.items():
        if only is not None and only not in name:
            continue
        with tempfile.NamedTemporaryFile(suffix=".json", prefix="bench_sub_") as side:
            cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", "100", "--warmup", "20",
                   "--cpu-budget", "0", "--sweep-log2", "0", "--no-kernel-timing", "--stream-chains", "", "--samples", "10",
                   "--other-workloads", "0", "--eval-rows", "0", "--metrics-skip", str(args.metrics_skip),
                   "--detail", side.name] + flags
            t0 = time.perf_counter()
            try:
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=300,
                                   env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
                json.loads(r.stdout.strip().splitlines()[-1])           # (the sub-run's own compact line parses)
                with open(side.name) as f:
                    line = json.load(f)                                 # ... and its full record is the side file
                sps = line.get("samples_per_sec") or {}
                out[name] = {"value": line["value"], "unit": line["unit"], "ms_per_step": line["ms_per_step"],
                             "samples_per_sec": sps.get("value"), "leapfrog_steps_per_sample": sps.get("leapfrog_steps_per_sample"),
                             "workload": line["config"]["workload"], "step_path": line["config"]["step_path"],
                             "timed_steps": line["timing"]["timed_steps"], "sub_run_s": round(time.perf_counter() - t0, 1)}
            except Exception as exc:      # a sub-run never takes the headline down
                out[name] = {"error": f"{type(exc).__name__}: {str(exc)[:300]}"}
    return out


def _library_calls():
    from bnn_priors_amd import conv
    return {f"{op}{list(shape)}": n for (op, shape), n in conv.LIBRARY_CALLS.items()}


def per_rank_summary(rates, solo):
    """min / median / max of the ranks' own steps/s in the N-rank region, rank 0's rate when it ran ALONE right before it,
    and their ratio: what one GPU loses to its neighbours' presence (host contention, shared power and cooling) --
    the one-run stand-in for a scaling curve; SCALE_rNN.json's per-N runs are the curve itself."""
    r = sorted(rates)
    out = {"min": round(r[0], 2), "median": round(statistics.median(r), 2), "max": round(r[-1], 2), "ranks": len(r),
           "solo_rank0": None if solo is None else round(solo, 2), "efficiency_vs_solo": None}
    if solo:
        out["efficiency_vs_solo"] = round(statistics.median(r) / solo, 4)
    return out


def cpu_shares(cpus, local_world):
    "the CPUs of every local rank: contiguous, disjoint, equal shares of the sorted list (the remainder stays unused)"
    cpus = sorted(cpus)
    per = max(1, len(cpus) // max(1, local_world))
    return [cpus[r * per:(r + 1) * per] or cpus for r in range(local_world)]


def pin_process(local, local_world):
    "one rank = one contiguous share of the host's CPUs (8 drivers + 8 HIP runtimes must not migrate / collide)"
    try:
        mine = cpu_shares(os.sched_getaffinity(0), local_world)[local]
        os.sched_setaffinity(0, mine)
        torch.set_num_threads(max(1, min(8, len(mine))))
        return len(mine)
    except (AttributeError, OSError):
        return None


# ------------------------------------------------------------------ --gpus N starts N ranks by itself
def self_launch(args):
    """``python bench.py --gpus N`` with N > 1 and no launcher around it: re-execute this command line under
    ``torch.distributed.run --nproc-per-node N`` (one process per chain per GPU, as the reference starts one process per
    replicate chain, experiments/run_experiment.sh:15-34).  Rank 0's one JSON line passes through on stdout.  Returns
    the exit code of the launcher, or None when this process is itself a rank (or N = 1)."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ or "RANK" in os.environ:
        return None
    import socket
    import subprocess
    n_dev = torch.cuda.device_count()
    if args.backend == "nccl" and n_dev < args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {n_dev} GPU(s) visible; RCCL needs one device per rank "
                         "(--backend gloo lets ranks share a GPU: a plumbing check, never a scaling number)")
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: starting", args.gpus, "ranks:", " ".join(cmd), file=sys.stderr)
    return subprocess.call(cmd)


def step_roofline(rows, ms_per_step):
    """The WHOLE step against the fp32-MFMA peak: the trunk's contraction flops per step (every MFMA row x its launches per
    step) / the measured step period, with the launch count and GPU-busy time of the committed in-step profile when it
    belongs to the loaded library (else null)."""
    flops = sum(r["algorithmic_flops_per_launch"] * r["launches_per_step"] for r in rows
                if r.get("bound") == "mfma" and r.get("launches_per_step"))
    # + the contractions the table has no row for (googleresnet at 128 images; forward, data gradient, weight gradient):
    # the 3 -> 16 stem (no data gradient) and the two down-sampling pairs (3x3 / stride 2 + 1x1 shortcut)
    n = 128
    stem = 2 * n * 32 * 32 * 3 * 16 * 9 * 2
    down = sum(2 * n * hw * hw * cin * 2 * cin * (9 + 1) * 3 for cin, hw in ((16, 16), (32, 8)))
    flops += stem + down
    ach = flops / (ms_per_step * 1e-3) / 1e12
    out = dict(flops_per_step=round(flops), achieved_tflops=round(ach, 2), peak=MFMA_F32_PEAK_TFLOPS,
               frac=round(ach / MFMA_F32_PEAK_TFLOPS, 4), launches_per_step=None, gpu_busy_us=None)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "in_step_us.json")
    try:
        with open(path) as f:
            committed = json.load(f)
        if not _stale(committed):
            out["launches_per_step"] = committed.get("launches_per_step")
            out["gpu_busy_us"] = committed.get("gpu_busy_us_per_step")
    except (OSError, ValueError):
        pass
    return out


# ------------------------------------------------------------------ the headline roofline row and the compact line
def _symbol(row):
    "the device symbol a roofline row belongs to: BatchNorm rows of the three stages are ONE kernel, convolutions one per shape"
    return row["kernel"].split(" ")[0]


def headline_rooflines(rows):
    """(top, top_other): the SYMBOL with the largest time per step -- all its rows (shapes) pooled: work summed over
    launches / time summed over launches -- and the largest symbol of the other roofline kind (HBM vs MFMA), so that the
    line names the real largest consumer AND the top contraction."""
    groups = {}
    for r in rows:
        if not r.get("launches_per_step"):
            continue
        groups.setdefault(_symbol(r), []).append(r)

    def pooled(sym, rs):
        def tot(key_us):
            return sum(r.get(key_us, r["avg_kernel_us"]) * r["launches_per_step"] for r in rs)
        work_key = "algorithmic_flops_per_launch" if rs[0]["bound"] == "mfma" else "algorithmic_bytes_per_launch"
        scale = 1e12 if rs[0]["bound"] == "mfma" else 1e9
        n = sum(r["launches_per_step"] for r in rs)
        work = sum(r[work_key] * r["launches_per_step"] for r in rs)
        live_us, step_us = tot("avg_kernel_us"), tot("in_step_us")
        ach_iso = work / (live_us * 1e-6) / scale
        in_step = all("in_step_us" in r for r in rs)
        # achieved / frac are what the kernel does INSIDE the step (rocprofv3 durations of the captured launches,
        # profiles/in_step_us.json) whenever that measurement belongs to the loaded library; the isolated-launch figure
        # (dispatch-packet events, measured live: always a few tenths of a microsecond shorter) stays beside it
        ach = work / (step_us * 1e-6) / scale if in_step else ach_iso
        row = dict(kernel=sym if len(rs) > 1 else rs[0]["kernel"], bound=rs[0]["bound"], achieved=round(ach, 2),
                   peak=rs[0]["peak"], unit=rs[0]["unit"], frac=round(ach / rs[0]["peak"], 4),
                   frac_source="in_step" if in_step else "isolated",
                   achieved_isolated=round(ach_iso, 2), frac_isolated=round(ach_iso / rs[0]["peak"], 4),
                   traffic=(round(sum(r["traffic"] * r["launches_per_step"] for r in rs) / n)
                            if all(r.get("traffic") is not None for r in rs) else None),
                   avg_kernel_us=round(live_us / n, 3), launches_per_step=n, us_per_step=round(step_us, 1),
                   shapes=len(rs))
        row[work_key] = round(work / n)
        if in_step:
            row["frac_in_step"] = row["frac"]
            row["in_step_us"] = round(step_us / n, 3)
        elif any(r.get("stale") for r in rs):
            row["stale"] = True         # profiles/in_step_us.json was measured on other kernel sources
        return row
    ranked = sorted((pooled(k, v) for k, v in groups.items()), key=lambda r: -r["us_per_step"])
    if not ranked:
        return None, None
    top = ranked[0]
    other = next((r for r in ranked[1:] if r["bound"] != top["bound"]), None)
    return top, other


_ROOF_KEYS = ("kernel", "bound", "achieved", "peak", "unit", "frac", "frac_source", "frac_isolated", "stale", "traffic", "in_step_us",
              "algorithmic_bytes_per_launch", "algorithmic_flops_per_launch", "avg_kernel_us", "avg_kernel_ms",
              "launches_per_step", "us_per_step", "elements")


def compact_line(out, detail_path):
    """The ONE stdout line: the fields the driver's record needs, < 4 KB.  Everything else (roofline_kernels,
    other_workloads, chains_per_gpu, timing, calibration, notes) is in the side file ``detail`` and on stderr."""
    def pick(d, keys):
        return None if d is None else {k: d[k] for k in keys if k in d}
    line = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                "scaling", "vs_baseline", "dtype", "data")}
    cfg = out["config"]
    line["config"] = {"workload": cfg["workload"][:300], "params": cfg["params"], "chains": cfg["chains"],
                      "step_path": cfg["step_path"][:200]}
    for k in ("ranks_seen", "backend", "per_rank_steps_per_s"):
        if k in out:
            line[k] = out[k]
    if "devices" in out:
        line["devices"] = out["devices"][:8]
    for k in ("roofline", "roofline_mfma", "roofline_hbm", "roofline_sampler", "roofline_flat_arena"):
        if out.get(k) is not None:
            line[k] = pick(out[k], _ROOF_KEYS)
    if out.get("roofline_step") is not None:
        line["roofline_step"] = out["roofline_step"]
    if out.get("cpu_baseline"):
        line["cpu_baseline"] = pick(out["cpu_baseline"], ("value", "unit", "cores", "host_cpus", "kind", "cpu_model",
                                                          "block_steps_per_s", "threads_calibration_steps_per_s"))
        line["cpu_baseline"]["sample"] = out["cpu_baseline"].get("sample", "")[:160]
        line["speedup_vs_cpu"] = out.get("speedup_vs_cpu")
    if out.get("samples_per_sec"):
        line["samples_per_sec"] = pick(out["samples_per_sec"], ("value", "per_chain", "leapfrog_steps_per_sample",
                                                                "reject_samples"))
    cg = out.get("chains_per_gpu")
    if isinstance(cg, dict):
        line["chains_per_gpu"] = {k: v["aggregate_steps_per_s"] for k, v in cg.items()
                                  if isinstance(v, dict) and "aggregate_steps_per_s" in v}
        sm = {k: v["aggregate_samples_per_s"] for k, v in cg.items() if isinstance(v, dict) and "aggregate_samples_per_s" in v}
        if sm:
            line["chains_per_gpu_samples_per_s"] = sm
        # (the K > 1 figures depend on how many hardware queues the process was given: say so next to them)
        line["chains_per_gpu_queues"] = {"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"),
                                         "distinct_hw_queues": cg.get("distinct_hw_queues")}
    if out.get("exchange"):
        line["exchange"] = pick(out["exchange"], ("chains", "ensemble_ms", "gather_ms", "ensemble_matches_single_process",
                                                  "gather_order_checked"))
    if out.get("other_workloads"):
        line["other_workloads"] = {k.split(" ", 1)[1][:48]: v.get("value", "error") for k, v in out["other_workloads"].items()}
    line["detail"] = os.path.relpath(detail_path, ROOT) if detail_path else None
    text = json.dumps(line, separators=(",", ":"))
    for drop in ("other_workloads", "exchange", "chains_per_gpu", "devices"):      # (never needed at today's sizes)
        if len(text) < 4096:
            break
        line.pop(drop, None)
        text = json.dumps(line, separators=(",", ":"))
    assert len(text) < 4096, len(text)
    return text


def main():
    args = parse()
    rc = self_launch(args)
    if rc is not None:
        raise SystemExit(rc)
    # stdout carries exactly ONE line, the JSON: whatever libraries print while the run is
    # going (RCCL announces its library path on stdout) is sent to stderr at the fd level
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    assert torch.cuda.is_available(), "bench.py measures the MI355X path; no GPU visible"
    # --gpus counts the ranks of ONE node (the launcher's --nproc-per-node = LOCAL_WORLD_SIZE); a multi-node launch has
    # WORLD_SIZE = nodes x that
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    if local_world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} but LOCAL_WORLD_SIZE = {local_world}: the launcher's "
                         "--nproc-per-node and --gpus must agree")
    n_dev = torch.cuda.device_count()
    if local >= n_dev and args.backend != "gloo":
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local} but {n_dev} GPU(s) visible (ranks may share a GPU only with "
                         "--backend gloo: RCCL needs one device per rank)")
    device = torch.device("cuda", local % n_dev)
    torch.cuda.set_device(device)
    distributed = world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ)
    pinned_cpus = None
    if distributed:   # one process per GPU under torch.distributed.run; "nccl" is RCCL on ROCm
        import torch.distributed as dist
        pinned_cpus = pin_process(local, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("gloo")
    # where the control-flow collectives' tensors live: RCCL moves device memory, gloo host memory
    cdev = device if args.backend == "nccl" else torch.device("cpu")

    from bnn_priors_amd.inference_reject import runner_class
    from bnn_priors_amd.storage import MemoryMetrics

    name, xshape, N, prior = WORKLOADS[args.workload]
    hmc = args.inference == "HMCReject"
    if hmc and args.trajectory:
        N = 128 * args.trajectory          # BASELINE configs[4]: a synthetic set of L batches, epoch = trajectory = L steps
        prior = "student-t"                # ... with its weight prior
    prior = args.weight_prior or prior
    L = -(-N // 128)
    torch.backends.cudnn.benchmark = bool(args.cudnn_benchmark)
    model = make_model(args.workload, device, prior, args.width)
    if args.channels_last:
        model = model.to(memory_format=torch.channels_last)
    n_params = sum(p.numel() for p in model.parameters())
    pool = PoolSource(args.workload, N, device, 1234 + rank)   # the whole synthetic data set, in HBM
    # The conv nets read their minibatches through the PRODUCT's batch source (inference._BatchSource over a shuffling
    # DataLoader of the HBM-resident set; googleresnet: augmented on read) -- a fresh permutation per epoch, every
    # minibatch gathered (+ cropped / flipped) inside the captured step's one staging launch (LazyBatch), the exact pass
    # filling its bodies' static inputs in place: what a user's run executes.  The dense net keeps its index batches.
    product_source = args.workload != "densenet"
    augment = None
    if product_source:
        if args.workload == "googleresnet" and args.augment:
            from bnn_priors_amd.augment import AugmentedTensorDataset, RandomCropFlip
            augment = RandomCropFlip(pad=4, flip=True, seed=1234, stream=rank)
            dataset = AugmentedTensorDataset(pool.x, pool.y, augment)
        else:
            dataset = torch.utils.data.TensorDataset(pool.x, pool.y)
        loader = torch.utils.data.DataLoader(dataset, batch_size=128, shuffle=True)
    else:
        loader = torch.utils.data.DataLoader(_SyntheticSet(N), batch_size=128, shuffle=True)
    empty_test = torch.utils.data.DataLoader(_SyntheticSet(0), batch_size=128)
    extra = {}
    if hmc and args.trajectory:
        extra["trajectory_length"] = args.trajectory
    if hmc and args.temperature != 1.0:
        extra["tempered"] = True           # T != 1 extends the reference's HMC (mcmc/hmc.py:39)
    runner = runner_class(args.inference)(
        model=model, dataloader=loader, dataloader_test=empty_test, epochs_per_cycle=50,
        warmup_epochs=50 if hmc else 45, sample_epochs=0 if hmc else 5,
        learning_rate=0.01 if not hmc else 1e-4 * args.temperature,
        skip=1, metrics_skip=args.metrics_skip, temperature=args.temperature, momentum=1.0 if hmc else 0.994,
        sampling_decay="cosine", cycles=60, precond_update=1, metrics_saver=MemoryMetrics(),
        model_saver=None, reject_samples=args.inference != "SGLDReject", seed=1234, chain_id=rank, **extra)
    # the exact initial gradient over the synthetic pool stands in for the full-data pass
    if not product_source:
        runner._batch_source = pool
    runner.use_graph = not args.eager
    step = runner.begin()
    eng = runner.optimizer.engine
    fused = runner._fused_dense() is not None
    source = runner._batches()
    path = ("fused dense step: mlp_fwdbwd + sampler(+slice-sum, prior) + finalize, 3 direct launches" if fused else
            "eager" if args.eager else "hipGraph replay (hand-written conv / BN fwd+bwd kernels, fused sampler) + 1 "
            "staging launch (minibatch gather, augmentation, argument block)")

    def full_batches():
        "the runner's own hot-loop batch stream, epoch after epoch, without the ragged last minibatch (N % 128)"
        while True:
            for xb, yb in runner._hot_batches():
                if len(xb) == 128:
                    yield xb, yb
    stream_of_batches = full_batches()

    def run(k, step):
        for _ in range(k):
            step += 1
            x, y = next(stream_of_batches)
            runner.leapfrog(step, x, y, last_of_epoch=False)
        return step

    step = run(args.warmup, step)
    runner._drain_rows()
    stream = torch.cuda.current_stream(device)
    K = args.steps
    # how many blocks of K steps make up >= min-seconds: one untimed calibration block, rank 0's clock decides
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    step = run(K, step)
    runner._drain_rows()
    torch.cuda.synchronize(device)
    n_blocks = max(1, min(args.max_blocks, math.ceil(args.min_seconds / max(time.perf_counter() - t0, 1e-6))))
    if distributed:
        nb = torch.tensor([n_blocks], device=cdev)
        dist.broadcast(nb, src=0)
        n_blocks = int(nb.item())
        dist.barrier()
    # Multi-rank runs: rank 0 first times the same blocks ALONE (the other ranks wait at a barrier, their GPUs idle) -- the
    # per-GPU rate that the N-rank region's per-rank rates are set against (efficiency_vs_solo in the line; the driver
    # computes its own scaling efficiency from the per-N values of separate runs)
    solo_ms = None
    if distributed and world > 1:
        if rank == 0:
            torch.cuda.synchronize(device)
            sm = [torch.cuda.Event(enable_timing=True)]
            sm[0].record(stream)
            for _ in range(n_blocks):
                step = run(K, step)
                ev = torch.cuda.Event(enable_timing=True)
                ev.record(stream)
                sm.append(ev)
            runner._drain_rows()
            torch.cuda.synchronize(device)
            solo_ms = statistics.median(a.elapsed_time(b) for a, b in zip(sm[:-1], sm[1:]))
        dist.barrier()
    torch.cuda.synchronize(device)
    marks = [torch.cuda.Event(enable_timing=True)]
    marks[0].record(stream)
    t0 = time.perf_counter()
    for _ in range(n_blocks):
        step = run(K, step)
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(stream)
        marks.append(ev)
    runner._drain_rows()          # every metric row of these steps has been logged
    torch.cuda.synchronize(device)
    if distributed:
        dist.barrier()
    wall = time.perf_counter() - t0
    block_ms = [a.elapsed_time(b) for a, b in zip(marks[:-1], marks[1:])]
    med_ms = statistics.median(block_ms)
    per_rank = None
    if distributed:
        mine = torch.tensor([med_ms], device=cdev, dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = [K / (float(t.item()) * 1e-3) for t in every]          # steps/s of every rank's median block
        t = torch.tensor([med_ms, wall], device=cdev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        med_ms, wall = t[0].item(), t[1].item()
    dt_block = med_ms * 1e-3

    runner._check_finite()
    # Live duration of the fused sampler kernel: a node inside a graph replay cannot carry events, so the same
    # launches (same arena, same arguments) are issued right here, each with a start / stop event in its
    # dispatch packet.  rocprofv3's per-dispatch durations of the in-graph launches are in profiles/.
    ktimes = []
    if not args.no_kernel_timing:
        opt = runner.optimizer
        if any(p.grad is None for p in eng.params):
            x, y = next(iter(source))
            runner._model_potential_and_grad(x, y, False)
        eng.start_kernel_timing()
        for _ in range(200):
            opt.step(calc_metrics=False)
        ktimes = eng.stop_kernel_timing()
    samples = samples_noreject = samples_eval = None
    if args.samples > 0:
        # one stored sample = L leapfrog steps + the runner's own M-H point (inference_reject.py:115-157: exact
        # full-data gradient, final_step, energy difference, M-H test, metrics row, momentum refresh for HMC,
        # initial_step) -- timed end to end, K times, through the SAME method the runner's loop calls
        n_b = len(source)

        def one_sample(step, save=None):
            acc = 0.0
            for i, (x, y) in enumerate(runner._hot_batches()):        # one epoch of the runner's own batch stream
                step += 1
                acc = runner.leapfrog(step, x, y, last_of_epoch=(i == n_b - 1))
            runner._drain_rows()
            return runner._mh_point(step, acc, source, save=save)

        def timed_samples(save=None):
            s = one_sample(step, save)          # untimed: first use of this variant's launches
            torch.cuda.synchronize(device)
            ts = time.perf_counter()
            for _ in range(args.samples):
                s = one_sample(s, save)
            torch.cuda.synchronize(device)
            return args.samples / (time.perf_counter() - ts), s

        samples, step = timed_samples()
        # the paper's default, reject_samples=False: no state snapshot at initial_step, no M-H test
        if runner.reject_samples:
            runner.reject_samples = False
            samples_noreject, step = timed_samples()
            runner.reject_samples = True
        # ... and what the reference's sample cycle also contains (inference_reject.py:142-144): the posterior-
        # predictive evaluation of the new sample over the test set and the sample written to the HDF5 store
        if args.eval_rows > 0 and world == 1:      # (one process: a rank that evaluates would keep the others waiting in the exchange)
            import tempfile
            from bnn_priors_amd import _h5, storage
            g = torch.Generator(device=device).manual_seed(99)
            xt = (torch.randn if args.workload == "googleresnet" else torch.rand)((args.eval_rows,) + xshape, generator=g, device=device)
            yt = torch.randint(0, 10, (args.eval_rows,), generator=g, device=device)
            runner.dataloader_test = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(xt, yt), batch_size=128)
            tmp = tempfile.mkdtemp(prefix="sgmcmc_bench_")
            runner.model_saver = (storage.HDF5ModelSaver(os.path.join(tmp, "samples.h5"), "w").__enter__()
                                  if _h5.available() else storage.MemoryModelSaver())
            store = type(runner.model_saver).__name__
            t_eval0 = time.perf_counter()
            runner._evaluate_model(runner.model.state_dict(), step)
            torch.cuda.synchronize(device)
            t_eval = time.perf_counter() - t_eval0          # (first call: includes any warm-up)
            rate, step = timed_samples(save=(0, runner.epochs_per_cycle - 1))
            t_eval0 = time.perf_counter()
            runner._evaluate_model(runner.model.state_dict(), step)
            torch.cuda.synchronize(device)
            samples_eval = dict(per_chain=round(rate, 3), test_rows=args.eval_rows, store=store,
                                evaluate_model_ms=round((time.perf_counter() - t_eval0) * 1e3, 2),
                                first_evaluate_model_ms=round(t_eval * 1e3, 2))
            if hasattr(runner.model_saver, "__exit__"):
                runner.model_saver.__exit__(None, None, None)
            runner.dataloader_test, runner.model_saver = empty_test, None
            import shutil
            shutil.rmtree(tmp, ignore_errors=True)
    exchange = exchange_leg(model, rank, world, device, args.exchange_samples, cdev, args.backend) if distributed else None

    ranks_seen, devices = 1, [{"rank": 0, "local_rank": local, "device": torch.cuda.get_device_name(device),
                               "index": device.index}]
    if distributed:
        ranks_seen = dist.get_world_size()
        devices = [None] * ranks_seen
        dist.all_gather_object(devices, {"rank": rank, "local_rank": local, "index": device.index,
                                         "device": torch.cuda.get_device_name(device)})
    value = world * K / dt_block
    out = {
        "metric": f"leapfrog steps/sec, {args.inference}", "value": round(value, 2),
        "unit": "steps/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
        "ms_per_step": round(dt_block / K * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "ranks_seen": ranks_seen, "backend": ("nccl (RCCL)" if args.backend == "nccl" else "gloo") if distributed else None,
        "devices": devices,
        "config": {"workload": f"{name} {args.inference} batch=128 N={N} (L={L} steps/epoch) "
                               f"lr={'1e-4' if hmc else '0.01'} cosine momentum={'1' if hmc else '0.994'} "
                               f"T={args.temperature:g} metrics_skip={args.metrics_skip} prior={prior}"
                               + (" augment=crop(pad 4)+flip on device" if augment is not None else "")
                               + (f" trajectory={args.trajectory}" if hmc and args.trajectory else ""),
                   "params": n_params, "tensors": len(list(model.parameters())),
                   "chains": world, "parallelism": f"{world} independent chain(s), one per GPU"
                   if args.backend == "nccl" else
                   f"{world} ranks over gloo, sharing GPUs: a plumbing check of the multi-rank leg, NOT a scaling number",
                   "step_path": path + (f" [width {args.width}: off the kernel tables, library calls in the step: "
                                        f"{_library_calls()}]" if args.width != 50 else "")},
        "timing": {"blocks": n_blocks, "steps_per_block": K, "timed_steps": n_blocks * K,
                   "median_block_ms": round(med_ms, 4), "min_block_ms": round(min(block_ms), 4),
                   "max_block_ms": round(max(block_ms), 4), "region_wall_s": round(wall, 4),
                   "wall_steps_per_s": round(world * n_blocks * K / wall, 2),
                   "method": "blocks of K steps back to back, event on the launch stream at every block "
                             "boundary, region bracketed by barrier + synchronize; value = K / median block "
                             "(max over ranks)"},
    }
    if pinned_cpus is not None:
        out["config"]["cpus_per_rank"] = pinned_cpus
    if per_rank is not None:
        out["per_rank_steps_per_s"] = per_rank_summary(per_rank, None if solo_ms is None else K / (solo_ms * 1e-3))
    if samples is not None and args.samples >= 10:
        out["samples_per_sec"] = {"value": round(world * samples, 3), "per_chain": round(samples, 3),
                                  "cycles_timed": args.samples,
                                  "leapfrog_steps_per_sample": L, "reject_samples": runner.reject_samples,
                                  "per_chain_reject_samples_false":
                                      None if samples_noreject is None else round(samples_noreject, 3),
                                  "includes": "L leapfrog steps, then the runner's own M-H point (_mh_point): exact "
                                              "full-data gradient pass (N rows), final_step, M-H test, metrics row, "
                                              "initial_step"}
        if samples_eval is not None:
            out["samples_per_sec_with_eval"] = dict(
                samples_eval, value=round(world * samples_eval["per_chain"], 3),
                includes="the same cycle + _evaluate_model over a synthetic test set (model.eval(), E = 1) + "
                         "_save_sample into the sample store, as inference_reject.py:142-144")
    if exchange is not None:
        out["exchange"] = exchange
    if rank == 0:
        sampler_line = None
        if ktimes:
            chunks = ktimes[0][1]
            avg_ms = sum(t for t, _, _ in ktimes) / len(ktimes)
            algo_bytes = BYTES_PER_PARAM * n_params
            ach = algo_bytes / (avg_ms * 1e-3) / 1e9
            sampler_line = {
                "bound": "hbm", "kernel": "step_kernel<float, VERLET, vec>",
                "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": None,
                "algorithmic_bytes_per_launch": algo_bytes, "avg_kernel_us": round(avg_ms * 1e3, 3),
                "launches": len(ktimes), "chunks_per_launch": chunks,
                "regime": "launch-latency-bound: the whole sampler state of this net "
                          f"({algo_bytes / 1e6:.2f} MB/launch) is below one launch's fixed cost; "
                          "see roofline_flat_arena for the same kernel in its bandwidth-bound regime"}
        if args.workload == "googleresnet" and not args.no_kernel_timing:
            # the step is dominated by the trunk's convolution gradients (profiles/): the headline roofline is
            # the kernel with the largest share of the step, the sampler's HBM line is reported beside it
            convs = attach_pmc_traffic(conv_rooflines(device))
            try:
                bns = attach_pmc_traffic(bn_rooflines(device))
            except Exception as exc:       # (an extra table: never takes the bench line down)
                bns = []
                out["roofline_bn_error"] = f"{type(exc).__name__}: {exc}"
            rows_all_k = attach_in_step(convs + bns)
            # the headline row: the SYMBOL with the largest time per step (BatchNorm's backward is one symbol over
            # three shapes), and beside it the largest symbol of the other roofline kind
            top, other = headline_rooflines(rows_all_k)
            out["roofline"] = top
            if other is not None:
                out["roofline_mfma" if other["bound"] == "mfma" else "roofline_hbm"] = other
            out["roofline_note"] = ("roofline = the device symbol with the largest (launches per step x duration), its shapes "
                                    "pooled; achieved / frac from live isolated launches (HIP events in the dispatch packet), "
                                    "frac_in_step from profiles/in_step_us.json when its source_sha matches this tree "
                                    "(else stale: true and no in-step figure); traffic from profiles/pmc_traffic.json "
                                    "under the same rule")
            out["roofline_kernels"] = rows_all_k
            out["roofline_step"] = step_roofline(rows_all_k, out["ms_per_step"])
            if sampler_line:
                out["roofline_sampler"] = sampler_line
        elif sampler_line:
            out["roofline"] = sampler_line
        if args.sweep_log2:
            out["roofline_flat_arena"] = flat_arena_point(args.sweep_log2, device)
        if args.workload == "densenet" and args.chain_sweep and args.inference == "VerletSGLDReject":
            out["chains_per_gpu"] = chains_per_gpu_sweep(args, device, rank, [int(k) for k in args.chain_sweep.split(",")])
        stream_chains = args.stream_chains
        if stream_chains is None:
            stream_chains = "1,2,4,8" if (args.workload == "googleresnet" and world == 1 and not args.eager) else ""
        if args.workload != "densenet" and stream_chains and args.inference == "VerletSGLDReject":
            try:
                out["chains_per_gpu"] = dict(
                    chains_per_gpu_streams(args, device, rank, [int(k) for k in stream_chains.split(",")],
                                           cycles=2 if args.samples >= 10 else 0),
                    method="K runners with their own captured steps on min(K, 4) HIP streams that own a hardware queue each "
                           "(chain c on stream c mod 4; no augmentation gather)")
            except Exception as exc:       # an extension after the timed region: never takes the bench line down
                out["chains_per_gpu"] = {"error": f"{type(exc).__name__}: {exc}"}
        other = args.other_workloads
        if other is None:
            other = int(world == 1 and args.workload == "googleresnet" and args.inference == "VerletSGLDReject"
                        and not args.eager)
        if other:
            out["other_workloads"] = other_workloads(args)
        if world == 1 and args.cpu_budget > 0:
            from oracle import nets as oracle_nets
            from oracle.runner import time_cpu_baseline
            cpu_batches = [(x.cpu(), y.cpu()) for x, y in list(pool)[:16]]

            def make_cpu_model():      # the oracle's own plain-torch restatement of the net: nothing of the product
                torch.manual_seed(0)
                m = oracle_nets.BUILDERS[name](weight_prior=prior)
                oracle_nets.he_initialize(m)
                return m
            res = time_cpu_baseline(make_cpu_model, cpu_batches,
                                    num_data=float(N), lr=0.01, momentum=0.994, temperature=1.0,
                                    steps_per_cycle=L * 50, budget_s=args.cpu_budget)
            out["cpu_baseline"] = {
                "value": round(res["steps_per_s"], 2), "unit": "steps/s", "cores": res["cores"],
                "kind": "port", "host_cpus": res["host_cpus"], "cpu_model": res["cpu_model"],
                "threads_calibration_steps_per_s": res["calibration"], "block_steps_per_s": res["block_steps_per_s"],
                "sample": f"median of {len(res['block_steps_per_s'])} blocks: {res['steps']} leapfrog steps in {res['seconds']:.1f} s of the same workload "
                          "(oracle/: reference-op-order torch-CPU loop on oracle/nets.py's plain-torch restatement of "
                          "the net, per-tensor sampler)"}
            out["speedup_vs_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 2)
        from bnn_priors_amd import _hip
        out["source_sha"] = _hip.library_sha()
        detail_path = args.detail or None
        if detail_path:
            try:
                with open(detail_path, "w") as f:
                    json.dump(out, f, indent=1)
            except OSError as exc:
                print(f"bench.py: cannot write {detail_path}: {exc}", file=sys.stderr)
                detail_path = None
        print(json.dumps(out), file=sys.stderr)        # the full record, for logs
        sys.stderr.flush()
        sys.stdout.flush()
        os.write(json_fd, (compact_line(out, detail_path) + "\n").encode())
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
