"""Where an M-H point's time goes (reject runners, inference_reject.py:115-157): the exact pass, final_step + energy
read-back, the M-H test, the metrics row, momentum refresh + initial_step -- each bracketed by a device synchronize.
    python tools/mh_point_probe.py [--hmc 1] [--cycles 6]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SGMCMC_STRICT", "1")
import torch
import bench
from bnn_priors_amd.inference_reject import runner_class, _f
from bnn_priors_amd.inference import _is_hmc
from bnn_priors_amd.storage import MemoryMetrics

ap = argparse.ArgumentParser()
ap.add_argument("--hmc", type=int, default=1)
ap.add_argument("--cycles", type=int, default=6)
a = ap.parse_args()
dev = torch.device("cuda", 0)
N = 6400 if a.hmc else 50000
model = bench.make_model("googleresnet", dev, "student-t" if a.hmc else None)
pool = bench.PoolSource("googleresnet", N, dev, 1234)
from bnn_priors_amd.augment import AugmentedTensorDataset, RandomCropFlip
ds = AugmentedTensorDataset(pool.x, pool.y, RandomCropFlip(pad=4, flip=True, seed=1234, stream=0))
loader = torch.utils.data.DataLoader(ds, batch_size=128, shuffle=True)
empty = torch.utils.data.DataLoader(bench._SyntheticSet(0), batch_size=128)
extra = dict(trajectory_length=50, tempered=True) if a.hmc else {}
r = runner_class("HMCReject" if a.hmc else "VerletSGLDReject")(
    model=model, dataloader=loader, dataloader_test=empty, epochs_per_cycle=50, warmup_epochs=50 if a.hmc else 45,
    sample_epochs=0 if a.hmc else 5, learning_rate=1e-5 if a.hmc else 0.01, skip=1, metrics_skip=10,
    temperature=0.1 if a.hmc else 1.0, momentum=1.0 if a.hmc else 0.994, sampling_decay="cosine", cycles=60,
    precond_update=1, metrics_saver=MemoryMetrics(), model_saver=None, reject_samples=True, seed=1234, chain_id=0, **extra)
step = r.begin()
src = r._batches()
sync = lambda: torch.cuda.synchronize()
rows = []
for c in range(a.cycles):
    sync(); t = [time.perf_counter()]
    mark = lambda: (sync(), t.append(time.perf_counter()))
    acc = 0.0
    n_b = len(src)
    for i, (x, y) in enumerate(r._hot_batches()):
        step += 1
        acc = r.leapfrog(step, x, y, last_of_epoch=(i == n_b - 1))
    r._drain_rows(); mark()
    opt = r.optimizer
    step += 1
    loss, log_prior, potential = r._exact_model_potential_and_grad(src); mark()
    opt.final_step(calc_metrics=True); de = _f(r._delta_energy(potential)); mark()
    r._total_energy += de; r._initial_potential = potential.item()
    rejected, _ = opt.maybe_reject(de); r._check_finite(); mark()
    r.store_metrics(i=step, loss=loss.item(), log_prior=log_prior.item(), potential=potential.item(), acc=_f(acc),
                    lr=opt.param_groups[0]["lr"], corresponds_to_sample=False, delta_energy=de,
                    total_energy=r._total_energy, rejected=rejected); mark()
    r.scheduler.step()
    if _is_hmc(opt):
        opt.sample_momentum()
    opt.initial_step(calc_metrics=False, save_state=True); mark()
    rows.append([1e3 * (b - a_) for a_, b in zip(t[:-1], t[1:])])
names = ["leapfrog", "exact pass", "final_step+dE", "M-H test", "metrics row", "refresh+initial_step"]
for k, nm in enumerate(names):
    print(f"{nm:22s} ms: " + " ".join(f"{r_[k]:7.2f}" for r_ in rows))
print("cycle total ms:        " + " ".join(f"{sum(r_):7.2f}" for r_ in rows))
