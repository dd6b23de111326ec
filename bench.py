#!/usr/bin/env python
"""bench.py -- leapfrog steps/sec of VerletSGLDReject on the MI355X path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload densenet|convnet|googleresnet]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot-loop body of the reject runner
(bnn_priors_amd.inference_reject.VerletSGLDRunnerReject.leapfrog, reference
inference_reject.py:86-113) over one synthetic minibatch of 128 that is already
resident in HBM: stochastic gradient of the average potential (forward + backward,
likelihood AND prior), the fused HIP sampler transition (momentum with friction and
in-kernel Philox noise, position, RMSprop statistic, the six energy / temperature
reductions), metric read-back every 10th step, cosine LR schedule.

Chains are independent: rank r runs chain r on GPU r with Philox stream r and its own
synthetic data (seed 1234 + r); no collective on the data path ("scaling": "weak").
Rank 0 prints ONE JSON line.  See DESIGN.md "Measurement" for the roofline accounting.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X spec (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
BYTES_PER_PARAM = 28    # fp32 intermediate step: read g, m, theta, v + write m, theta, v

WORKLOADS = {
    # name: (model, x shape, N, weight prior)  -- BASELINE.json configs[1], [2], [3]
    "densenet": ("classificationdensenet", (784,), 60000, "gaussian"),
    "convnet": ("classificationconvnet", (784,), 60000, "laplace"),
    "googleresnet": ("googleresnet", (3, 32, 32), 50000, "gaussian"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--workload", default="densenet", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-budget", type=float, default=12.0, help="seconds of CPU baseline (0 = skip)")
    ap.add_argument("--sweep-log2", type=int, default=28, help="flat-arena roofline point, log2(elements); 0 = skip")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--eager", action="store_true", help="no hipGraph capture (for comparison)")
    ap.add_argument("--cudnn-benchmark", type=int, default=1,
                    help="MIOpen find mode, as the reference sets it (experiments/train_bnn.py:29-31)")
    ap.add_argument("--channels-last", type=int, default=0)
    ap.add_argument("--samples", type=int, default=3, help="also time K full sample cycles (0 = skip)")
    ap.add_argument("--metrics-skip", type=int, default=10, help="BASELINE configs use 10")
    ap.add_argument("--inference", default="VerletSGLDReject",
                    choices=["VerletSGLDReject", "HMCReject", "SGLDReject"],
                    help="runner, as experiments/train_bnn.py:223-234 names them (headline: VerletSGLDReject)")
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

    def __iter__(self):
        for b in range(self.n_batches):
            yield self.x[128 * b:128 * (b + 1)], self.y[128 * b:128 * (b + 1)]

    def index_batches(self):
        import numpy as np
        from bnn_priors_amd.fused_dense import IndexBatch
        for b in range(self.n_batches):
            idx = np.arange(128 * b, min(128 * (b + 1), self.n_rows), dtype=np.int64)
            yield IndexBatch(idx, self.x, self.y), None


def make_model(workload, device):
    from bnn_priors_amd import models
    name, xshape, _, prior = WORKLOADS[workload]
    torch.manual_seed(0)
    x0 = torch.zeros((2,) + xshape)
    net = models.get_model(x0, torch.tensor([0, 9]), name, width=50, depth=3, weight_prior=prior,
                           weight_scale=2 ** .5, bias_prior="gaussian", bias_scale=1.)
    models.he_initialize(net)
    return net.to(device)


class _SyntheticSet(torch.utils.data.Dataset):
    "length-only stand-in: the runner needs len(dataset) = N and len(dataloader) = ceil(N/128)"

    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n


def flat_arena_point(log2n, device, iters=20):
    """The same fused kernel on ONE flat segment of 2^log2n fp32 elements (bandwidth-bound
    regime): GB/s of algorithmic traffic (28 B/element) from HIP events."""
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
    out = dict(bound="hbm", kernel="step_kernel_stream<float, VERLET>" if log2n >= 24 else "step_kernel<float, VERLET, vec>",
               achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(gbs / HBM_PEAK_GBS, 4),
               traffic=None, algorithmic_bytes_per_launch=BYTES_PER_PARAM * n, elements=n,
               avg_kernel_ms=round(avg_ms, 4), min_kernel_ms=round(min(times), 4), launches=len(times))
    if log2n == 28:
        # HBM bytes per launch from the PMC counters, collected by separate rocprofv3 --pmc passes of this same
        # launch (FETCH_SIZE x2 for gfx950's wide coalesced reads + WRITE_SIZE): not re-measured in this run
        out["traffic"] = 7.53e9
        out["traffic_source"] = "profiles/r01_flat_arena_2p28_pmc_{FETCH,WRITE}_SIZE.txt (ratio to algorithmic 1.002)"
    return out


def main():
    args = parse()
    # stdout carries exactly ONE line, the JSON: whatever libraries print while the run is
    # going (RCCL announces its library path on stdout) is sent to stderr at the fd level
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    assert torch.cuda.is_available(), "bench.py measures the MI355X path; no GPU visible"
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    distributed = world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ)
    if distributed:   # one process per GPU under torch.distributed.run; "nccl" is RCCL on ROCm
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=device)

    from bnn_priors_amd.inference_reject import runner_class
    from bnn_priors_amd.storage import MemoryMetrics

    name, xshape, N, prior = WORKLOADS[args.workload]
    L = -(-N // 128)
    torch.backends.cudnn.benchmark = bool(args.cudnn_benchmark)
    model = make_model(args.workload, device)
    if args.channels_last:
        model = model.to(memory_format=torch.channels_last)
    n_params = sum(p.numel() for p in model.parameters())
    pool = PoolSource(args.workload, N, device, 1234 + rank)   # the whole synthetic data set, in HBM
    loader = torch.utils.data.DataLoader(_SyntheticSet(N), batch_size=128, shuffle=True)
    empty_test = torch.utils.data.DataLoader(_SyntheticSet(0), batch_size=128)
    hmc = args.inference == "HMCReject"
    runner = runner_class(args.inference)(
        model=model, dataloader=loader, dataloader_test=empty_test, epochs_per_cycle=50,
        warmup_epochs=50 if hmc else 45, sample_epochs=0 if hmc else 5, learning_rate=0.01 if not hmc else 1e-4,
        skip=1, metrics_skip=args.metrics_skip, temperature=1.0, momentum=1.0 if hmc else 0.994,
        sampling_decay="cosine", cycles=60, precond_update=1, metrics_saver=MemoryMetrics(),
        model_saver=None, reject_samples=args.inference != "SGLDReject", seed=1234, chain_id=rank)
    # the exact initial gradient over the synthetic pool stands in for the full-data pass
    runner._batch_source = pool
    runner.use_graph = not args.eager
    step = runner.begin()
    eng = runner.optimizer.engine
    fused = runner._fused_dense() is not None
    batches = list(pool.index_batches()) if fused else list(pool)
    batches = [b for b in batches if len(b[0]) == 128]       # the L-th minibatch is ragged (N % 128)
    path = ("fused dense step: mlp_fwdbwd + sampler(+slice-sum, prior) + finalize, 3 direct launches" if fused else
            "eager" if args.eager else "hipGraph of autograd fwd/bwd + fused sampler")

    def run(k, step):
        for _ in range(k):
            step += 1
            x, y = batches[step % len(batches)]
            runner.leapfrog(step, x, y, last_of_epoch=False)
        runner._drain_rows()          # every metric row of these steps has been logged
        return step

    step = run(args.warmup, step)
    torch.cuda.synchronize(device)
    if distributed:
        dist.barrier()
    t0 = time.perf_counter()
    step = run(args.steps, step)
    torch.cuda.synchronize(device)
    if distributed:
        dist.barrier()
    dt = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()

    runner._check_finite()
    # Live duration of the fused sampler kernel: HIP events cannot bracket a node inside a graph
    # replay, so the same launches (same arena, same arguments) are issued eagerly right here,
    # each bracketed by an event pair on its stream.  rocprofv3's per-dispatch durations of the
    # in-graph launches are in profiles/ and agree.
    ktimes = []
    if not args.no_kernel_timing:
        opt = runner.optimizer
        if any(p.grad is None for p in eng.params):
            x, y = next(iter(pool))
            runner._model_potential_and_grad(x, y, False)
        eng.start_kernel_timing()
        for _ in range(200):
            opt.step(calc_metrics=False)
        ktimes = eng.stop_kernel_timing()
    samples = None
    if args.samples > 0:
        # one stored sample = L leapfrog steps + the exact full-data gradient + final_step + M-H test
        # + initial_step (inference_reject.py:86-157); timed end to end, K times
        all_b = list(pool.index_batches()) if fused else list(pool)
        opt = runner.optimizer

        def one_sample(step, reject=runner.reject_samples):
            for i, (x, y) in enumerate(all_b):
                step += 1
                runner.leapfrog(step, x, y, last_of_epoch=(i == len(all_b) - 1))
            step += 1
            loss, log_prior, potential = runner._exact_model_potential_and_grad(pool)
            opt.final_step(calc_metrics=True)
            de = runner._delta_energy(potential)
            de = de.item() if isinstance(de, torch.Tensor) else de
            runner._initial_potential = potential.item()
            if reject:
                opt.maybe_reject(de)
            runner.scheduler.step()
            opt.initial_step(calc_metrics=False, save_state=reject)
            return step

        step = one_sample(step)          # untimed: first use of the ragged last minibatch etc.
        torch.cuda.synchronize(device)
        ts = time.perf_counter()
        for _ in range(args.samples):
            step = one_sample(step)
        torch.cuda.synchronize(device)
        samples = args.samples / (time.perf_counter() - ts)
        # the paper's default, reject_samples=False: no state snapshot at initial_step, no M-H test
        samples_noreject = None
        if runner.reject_samples:
            ts = time.perf_counter()
            for _ in range(args.samples):
                step = one_sample(step, reject=False)
            torch.cuda.synchronize(device)
            samples_noreject = args.samples / (time.perf_counter() - ts)
    out = {
        "metric": f"leapfrog steps/sec, {args.inference}", "value": round(world * args.steps / dt, 2),
        "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{name} {args.inference} batch=128 N={N} (L={L} steps/epoch) "
                               f"lr=0.01 cosine momentum=0.994 T=1 metrics_skip={args.metrics_skip} prior={prior}",
                   "params": n_params, "tensors": len(list(model.parameters())),
                   "chains": world, "parallelism": f"{world} independent chain(s), one per GPU",
                   "step_path": path},
    }
    if samples is not None:
        out["samples_per_sec"] = {"value": round(world * samples, 3), "per_chain": round(samples, 3),
                                  "leapfrog_steps_per_sample": L, "reject_samples": runner.reject_samples,
                                  "per_chain_reject_samples_false":
                                      None if samples_noreject is None else round(samples_noreject, 3),
                                  "includes": "L leapfrog steps, exact full-data gradient pass "
                                              "(N rows), final_step, M-H test, initial_step"}
    if rank == 0:
        if ktimes:
            chunks = ktimes[0][1]
            avg_ms = sum(t for t, _, _ in ktimes) / len(ktimes)
            algo_bytes = BYTES_PER_PARAM * n_params
            ach = algo_bytes / (avg_ms * 1e-3) / 1e9
            out["roofline"] = {
                "bound": "hbm", "kernel": "step_kernel<float, VERLET, vec>",
                "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": None,
                "algorithmic_bytes_per_launch": algo_bytes, "avg_kernel_us": round(avg_ms * 1e3, 3),
                "launches": len(ktimes), "chunks_per_launch": chunks,
                "regime": "launch-latency-bound: the whole sampler state of this net "
                          f"({algo_bytes / 1e6:.2f} MB/launch) is below one launch's fixed cost; "
                          "see roofline_flat_arena for the same kernel in its bandwidth-bound regime"}
        if args.sweep_log2:
            out["roofline_flat_arena"] = flat_arena_point(args.sweep_log2, device)
        if world == 1 and args.cpu_budget > 0:
            from oracle.runner import time_cpu_baseline
            cpu_batches = [(x.cpu(), y.cpu()) for x, y in list(pool)[:16]]
            res = time_cpu_baseline(lambda: make_model(args.workload, "cpu"), cpu_batches,
                                    num_data=float(N), lr=0.01, momentum=0.994, temperature=1.0,
                                    steps_per_cycle=L * 50, budget_s=args.cpu_budget)
            out["cpu_baseline"] = {
                "value": round(res["steps_per_s"], 2), "unit": "steps/s", "cores": res["cores"],
                "kind": "port", "host_cpus": res["host_cpus"],
                "threads_calibration_steps_per_s": res["calibration"],
                "sample": f"{res['steps']} leapfrog steps in {res['seconds']:.1f} s of the same workload "
                          "(oracle/: reference-op-order torch-CPU loop, per-tensor sampler)"}
            out["speedup_vs_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 2)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
