"""End-of-run posterior-predictive evaluation (exp_utils.py:250-340) of E stored samples over a synthetic test set: the
grouped (vmap, library layers) path against the sample-by-sample path on this package's kernels.
    python tools/eval_probe.py [--workload googleresnet] [--samples 10] [--rows 10000]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from bnn_priors_amd import evaluation as ev

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="googleresnet")
ap.add_argument("--samples", type=int, default=10)
ap.add_argument("--rows", type=int, default=10000)
a = ap.parse_args()
dev = torch.device("cuda", 0)
name, xshape, N, prior = bench.WORKLOADS[a.workload]
model = bench.make_model(a.workload, dev).eval()
g = torch.Generator(device=dev).manual_seed(3)
x = torch.randn((a.rows,) + xshape, generator=g, device=dev)
y = torch.randint(0, 10, (a.rows,), generator=g, device=dev)
loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(x, y), batch_size=128)
samples = {}
for k, v in model.state_dict().items():
    if v.is_floating_point():
        s = v.unsqueeze(0) + 0.02 * v.abs().mean() * torch.randn((a.samples,) + tuple(v.shape), generator=g, device=dev)
        samples[k] = s.abs() + 0.5 if k.endswith("running_var") else s
    else:
        samples[k] = v.unsqueeze(0).repeat((a.samples,) + (1,) * v.dim())
out = {}
for batched in (True, False):
    ev.BATCHED = batched
    res = None
    ts = []
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = ev.evaluate_model(model, loader, samples)
        torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    out[batched] = res
    print(f"{a.workload} E={a.samples} rows={a.rows} {'grouped (vmap, library layers)' if batched else 'sample by sample (own kernels)'}: "
          f"ms {' '.join(f'{t:.0f}' for t in ts)}  lp_ensemble {res['lp_ensemble']:.5f} acc {res['acc_ensemble']:.4f}", flush=True)
