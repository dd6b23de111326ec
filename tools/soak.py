"""A short END-TO-END run of the reject runners at BASELINE sizes through `runner.run()` itself (not bench.py's pieces):
synthetic CIFAR-10 / MNIST-shaped sets resident in HBM, the product's DataLoader-backed batch source (lazy minibatches,
grouped exact pass), per-epoch evaluation on a 10,000-row test set, HDF5 sample + metrics stores.
    python tools/soak.py [--workload googleresnet|convnet|densenet] [--hmc 0|1] [--cycles 2]"""
import argparse
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SGMCMC_STRICT", "1")
import numpy as np
import torch
import bench
from bnn_priors_amd.inference_reject import runner_class
from bnn_priors_amd.storage import HDF5Metrics, HDF5ModelSaver, load_samples
from bnn_priors_amd.evaluation import evaluate_model

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="googleresnet")
ap.add_argument("--hmc", type=int, default=0)
ap.add_argument("--cycles", type=int, default=2)
a = ap.parse_args()
dev = torch.device("cuda", 0)
name, xshape, N, prior = bench.WORKLOADS[a.workload]
model = bench.make_model(a.workload, dev, "student-t" if a.hmc else None)
pool = bench.PoolSource(a.workload, N, dev, 1234)
test = bench.PoolSource(a.workload, 10000, dev, 99)
if a.workload == "googleresnet":
    from bnn_priors_amd.augment import AugmentedTensorDataset, RandomCropFlip
    ds = AugmentedTensorDataset(pool.x, pool.y, RandomCropFlip(pad=4, flip=True, seed=1234, stream=0))
else:
    ds = torch.utils.data.TensorDataset(pool.x, pool.y)
train = torch.utils.data.DataLoader(ds, batch_size=128, shuffle=True)
test_loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(test.x, test.y), batch_size=128)
tmp = tempfile.mkdtemp()
extra = dict(trajectory_length=50, tempered=True) if a.hmc else {}
with HDF5Metrics(os.path.join(tmp, "metrics.h5"), "w") as metrics, \
        HDF5ModelSaver(os.path.join(tmp, "samples.h5"), "w") as saver:
    r = runner_class("HMCReject" if a.hmc else "VerletSGLDReject")(
        model=model, dataloader=train, dataloader_test=test_loader, epochs_per_cycle=2, warmup_epochs=1, sample_epochs=1,
        learning_rate=1e-5 if a.hmc else 0.01 / 50, skip=1, metrics_skip=10, temperature=0.1 if a.hmc else 1.0,
        momentum=1.0 if a.hmc else 0.994, sampling_decay="cosine", cycles=a.cycles, precond_update=1,
        metrics_saver=metrics, model_saver=saver, reject_samples=True, seed=1234, chain_id=0, **extra)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r.run()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
samples = load_samples(os.path.join(tmp, "samples.h5"), keep_steps=False)
E = len(next(iter(samples.values())))
ev = evaluate_model(model, test_loader, {k: v.to(dev) for k, v in samples.items()})
steps = a.cycles * 2 * len(train)
print(f"{a.workload}{' HMC L=50' if a.hmc else ''}: run() {dt:.2f} s for {steps} leapfrog steps + {a.cycles} stored samples "
      f"({steps / dt:.0f} steps/s end to end incl. exact passes, evaluation, stores); samples on disk {E}; "
      f"ensemble lp {ev['lp_ensemble']:.4f} acc {ev['acc_ensemble']:.4f}; peak memory {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB")
