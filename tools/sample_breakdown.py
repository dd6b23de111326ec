import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py", "--steps", "200", "--warmup", "50", "--cpu-budget", "0", "--sweep-log2", "0", "--samples", "0", "--no-kernel-timing"]
import torch, bench
# reuse bench's setup by monkeypatching: run main pieces manually
args = bench.parse()
device = torch.device("cuda", 0)
from bnn_priors_amd.inference_reject import VerletSGLDRunnerReject
from bnn_priors_amd.storage import MemoryMetrics
name, xshape, N, prior = bench.WORKLOADS["densenet"]
model = bench.make_model("densenet", device)
pool = bench.PoolSource("densenet", N, device, 1234)
loader = torch.utils.data.DataLoader(bench._SyntheticSet(N), batch_size=128, shuffle=True)
empty = torch.utils.data.DataLoader(bench._SyntheticSet(0), batch_size=128)
r = VerletSGLDRunnerReject(model=model, dataloader=loader, dataloader_test=empty, epochs_per_cycle=50, warmup_epochs=45,
    sample_epochs=5, learning_rate=0.01, metrics_skip=10, momentum=0.994, cycles=60, precond_update=1,
    metrics_saver=MemoryMetrics(), reject_samples=True, seed=1234)
r._batch_source = pool
step = r.begin()
all_b = list(pool.index_batches())
opt = r.optimizer
def sync(): torch.cuda.synchronize()
for k in range(4):
    sync(); t0 = time.perf_counter()
    for i, (x, y) in enumerate(all_b):
        step += 1
        r.leapfrog(step, x, y, last_of_epoch=(i == len(all_b) - 1))
    sync(); t1 = time.perf_counter()
    loss, lp, pot = r._exact_model_potential_and_grad(pool)
    sync(); t2 = time.perf_counter()
    opt.final_step(calc_metrics=True)
    de = r._delta_energy(pot); de = de.item() if isinstance(de, torch.Tensor) else de
    r._initial_potential = pot.item()
    rej = opt.maybe_reject(de)
    r.scheduler.step()
    opt.initial_step(calc_metrics=False, save_state=True)
    sync(); t3 = time.perf_counter()
    print(f"sample {k}: leapfrog {1e3*(t1-t0):.1f} ms, exact {1e3*(t2-t1):.1f} ms, M-H {1e3*(t3-t2):.1f} ms, rejected={rej[0]}, dE={de:.3f}")
