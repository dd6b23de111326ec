"""The exact full-data gradient pass of the googleresnet / convnet reject runner on its own (inference_reject.py:18-33):
wall time per pass for a given number of lanes and minibatches per launch chain.
    SGMCMC_EXACT_LANES=2 SGMCMC_EXACT_GROUP=4 python tools/exact_pass_probe.py [--workload googleresnet] [--passes 5]
Under `rocprofv3 --kernel-trace --stats` the per-kernel totals of the passes are what the pass costs on the GPU."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SGMCMC_STRICT", "1")
import torch
import bench
from bnn_priors_amd import graphed
from bnn_priors_amd.inference_reject import VerletSGLDRunnerReject
from bnn_priors_amd.storage import MemoryMetrics

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="googleresnet")
ap.add_argument("--passes", type=int, default=5)
ap.add_argument("--rows", type=int, default=0, help="data-set rows (default: the workload's N)")
ap.add_argument("--product-source", type=int, default=1,
                help="1: the runner's own batch source over a shuffling DataLoader of the HBM-resident set (what bench.py "
                     "runs); 0: bench.PoolSource (sequential slices copied in)")
a = ap.parse_args()
device = torch.device("cuda", 0)
name, xshape, N, prior = bench.WORKLOADS[a.workload]
N = a.rows or N
model = bench.make_model(a.workload, device)
pool = bench.PoolSource(a.workload, N, device, 1234)
if a.product_source:
    if a.workload == "googleresnet":
        from bnn_priors_amd.augment import AugmentedTensorDataset, RandomCropFlip
        ds = AugmentedTensorDataset(pool.x, pool.y, RandomCropFlip(pad=4, flip=True, seed=1234, stream=0))
    else:
        ds = torch.utils.data.TensorDataset(pool.x, pool.y)
    loader = torch.utils.data.DataLoader(ds, batch_size=128, shuffle=True)
else:
    loader = torch.utils.data.DataLoader(bench._SyntheticSet(N), batch_size=128, shuffle=True)
empty = torch.utils.data.DataLoader(bench._SyntheticSet(0), batch_size=128)
r = VerletSGLDRunnerReject(model=model, dataloader=loader, dataloader_test=empty, epochs_per_cycle=50, warmup_epochs=45,
                           sample_epochs=5, learning_rate=0.01, metrics_skip=10, momentum=0.994, cycles=60, precond_update=1,
                           metrics_saver=MemoryMetrics(), reject_samples=True, seed=1234)
if not a.product_source:
    r._batch_source = pool
r.begin()
pool = r._batches()
torch.cuda.synchronize()
import gc
if os.environ.get("PROBE_GC") == "freeze":
    gc.collect(); gc.freeze()
elif os.environ.get("PROBE_GC") == "off":
    gc.disable()
ts, hs = [], []
for k in range(a.passes):
    t0 = time.perf_counter()
    loss, lp, pot = r._exact_model_potential_and_grad(pool)
    hs.append(1e3 * (time.perf_counter() - t0))          # host time to enqueue the pass
    torch.cuda.synchronize()
    ts.append(1e3 * (time.perf_counter() - t0))
acc = r._potential()._exact_acc
print(f"{a.workload}: N={N} lanes={graphed.EXACT_LANES} group={getattr(acc, 'group', 1)} "
      f"pass ms: {' '.join(f'{t:.1f}' for t in ts)}  (min {min(ts):.1f}; {min(ts) / -(-N // 128) * 1e3:.0f} us per minibatch)  "
      f"host enqueue ms: {min(hs):.1f}  potential {pot.item():.6f}", flush=True)
