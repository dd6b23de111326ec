"""K chains on K streams of one GPU, issued (a) round-robin from ONE host thread (bench.chains_per_gpu_streams),
(b) from K host threads, each free-running its own chain.  Tells a host-side limit from a GPU-side one.
    python tools/lab/chains_threads.py [--workload googleresnet] [--ks 1,2,3,4] [--steps 60]"""
import argparse, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="googleresnet")
ap.add_argument("--ks", default="1,2,3,4")
ap.add_argument("--steps", type=int, default=60)
a = ap.parse_args()
sys.argv = [sys.argv[0], "--workload", a.workload]
args = bench.parse()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
from bnn_priors_amd.inference_reject import runner_class
from bnn_priors_amd.storage import MemoryMetrics
name, xshape, N, prior = bench.WORKLOADS[a.workload]
pool = bench.PoolSource(a.workload, N, dev, 4321)
batches = [b for b in pool if len(b[0]) == 128]
ks = [int(k) for k in a.ks.split(",")]
runners, streams, steps_of = [], [], []
from bnn_priors_amd import multichain
picked = multichain.concurrent_streams(max(ks), dev) if os.environ.get("PICK", "1") == "1" else [torch.cuda.Stream(device=dev) for _ in range(max(ks))]
print("streams with a hardware queue of their own:", len(picked), flush=True)
for c in range(max(ks)):
    st = picked[c % len(picked)]
    with torch.cuda.stream(st):
        model = bench.make_model(a.workload, dev, args.weight_prior)
        loader = torch.utils.data.DataLoader(bench._SyntheticSet(N), batch_size=128, shuffle=True)
        empty = torch.utils.data.DataLoader(bench._SyntheticSet(0), batch_size=128)
        r = runner_class("VerletSGLDReject")(
            model=model, dataloader=loader, dataloader_test=empty, epochs_per_cycle=50, warmup_epochs=45,
            sample_epochs=5, learning_rate=0.01, skip=1, metrics_skip=10, temperature=1.0,
            momentum=0.994, sampling_decay="cosine", cycles=60, precond_update=1, metrics_saver=MemoryMetrics(),
            model_saver=None, reject_samples=True, seed=1234, chain_id=c)
        r._batch_source = pool
        r.use_graph = True
        steps_of.append(r.begin())
    runners.append(r); streams.append(st)
torch.cuda.synchronize(dev)


def chain_loop(c, n):
    torch.cuda.set_device(dev)
    with torch.cuda.stream(streams[c]):
        for _ in range(n):
            steps_of[c] += 1
            x, y = batches[(steps_of[c] + 13 * c) % len(batches)]
            runners[c].leapfrog(steps_of[c], x, y, last_of_epoch=False)
        runners[c]._drain_rows()


for K in ks:
    for mode in ("one thread", "K threads"):
        def go(n):
            if mode == "one thread":
                for _ in range(n):
                    for c in range(K):
                        steps_of[c] += 1
                        x, y = batches[(steps_of[c] + 13 * c) % len(batches)]
                        with torch.cuda.stream(streams[c]):
                            runners[c].leapfrog(steps_of[c], x, y, last_of_epoch=False)
                for c in range(K):
                    with torch.cuda.stream(streams[c]):
                        runners[c]._drain_rows()
            else:
                th = [threading.Thread(target=chain_loop, args=(c, n)) for c in range(K)]
                [t.start() for t in th]; [t.join() for t in th]
        go(15)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        go(a.steps)
        t1 = time.perf_counter()
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        print(f"{a.workload} K={K} {mode:10s}: aggregate {K * a.steps / dt:8.1f} steps/s  lockstep {dt / a.steps * 1e6:8.1f} us  "
              f"(host done issuing after {100 * (t1 - t0) / dt:.0f}% of the wall time)", flush=True)
