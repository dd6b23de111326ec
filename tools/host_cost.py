"""Host-side cost of one captured leapfrog step: time from calling runner.leapfrog() to its return with an EMPTY GPU
queue (synchronize before every call), against the GPU time of the step.  python tools/host_cost.py [--workload ...]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="googleresnet")
ap.add_argument("--steps", type=int, default=300)
a = ap.parse_args()
sys.argv = [sys.argv[0]]
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
from bnn_priors_amd.inference_reject import runner_class
from bnn_priors_amd.storage import MemoryMetrics
name, xshape, N, prior = bench.WORKLOADS[a.workload]
model = bench.make_model(a.workload, dev)
pool = bench.PoolSource(a.workload, N, dev, 1234)
loader = torch.utils.data.DataLoader(bench._SyntheticSet(N), batch_size=128, shuffle=True)
empty = torch.utils.data.DataLoader(bench._SyntheticSet(0), batch_size=128)
runner = runner_class("VerletSGLDReject")(
    model=model, dataloader=loader, dataloader_test=empty, epochs_per_cycle=50, warmup_epochs=45, sample_epochs=5,
    learning_rate=0.01, skip=1, metrics_skip=10, temperature=1.0, momentum=0.994, sampling_decay="cosine", cycles=60,
    precond_update=1, metrics_saver=MemoryMetrics(), model_saver=None, reject_samples=True, seed=1234, chain_id=0)
runner._batch_source = pool
runner.use_graph = True
step = runner.begin()
fused = runner._fused_dense() is not None
batches = list(pool.index_batches()) if fused else list(pool)
batches = [b for b in batches if len(b[0]) == 128]
for _ in range(40):
    step += 1
    x, y = batches[step % len(batches)]
    runner.leapfrog(step, x, y, last_of_epoch=False)
runner._drain_rows()
host, gpu = [], []
for _ in range(a.steps):
    step += 1
    x, y = batches[step % len(batches)]
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    runner.leapfrog(step, x, y, last_of_epoch=False)
    e1.record()
    host.append(time.perf_counter() - t0)
    torch.cuda.synchronize(dev)
    gpu.append(e0.elapsed_time(e1) * 1e-3)
host.sort(); gpu.sort()
print(f"{a.workload}: host per step median {1e6 * host[len(host) // 2]:.0f} us (p90 {1e6 * host[int(.9 * len(host))]:.0f}); "
      f"GPU per step (empty queue) median {1e6 * gpu[len(gpu) // 2]:.0f} us")
