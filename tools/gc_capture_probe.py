"""Probe: does a cyclic garbage collection INSIDE a stream capture (with dead runners -- and their graphs -- waiting
in reference cycles) bring the process down?  It did (~CUDAGraph throws while a stream captures => terminate) until
every capture went through bnn_priors_amd/_capture.py; this now prints "ok".  `python tools/gc_capture_probe.py <case>`"""
import faulthandler, gc, os, sys
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import runner_cases as RC
from bnn_priors_amd import conv as _conv, inference_reject, inference, models
from test_runners import _runner_class, MemoryMetrics

name = sys.argv[1]
cfg = RC.CASES[name]
dev = "cuda:0"


def make(chain):
    train, test, (x, y) = RC.make_data(dev, cfg=cfg)
    model = RC.make_net(models, x, y, device=dev, cfg=cfg)
    metrics = MemoryMetrics()
    runner = _runner_class(name)(
        model=model, dataloader=train, dataloader_test=test, learning_rate=cfg["lr"],
        temperature=cfg["temperature"], momentum=cfg["momentum"], reject_samples=cfg["reject_samples"],
        metrics_saver=metrics, model_saver=None, seed=RC.SEED, chain_id=chain, cycle_seed=RC.CYCLE_SEED, **RC.RUN_KW)
    begin = runner.begin
    runner.begin = lambda: (torch.manual_seed(RC.SEED + chain), begin())[1]      # a cycle: runner -> lambda -> runner
    return runner, metrics


gc.disable()
for chain in (0, 1):
    runner, metrics = make(chain)
    runner.run()
del runner, metrics                              # dead, but in cycles: only the collector frees them
hits = [0]
real = _conv._stream


def noisy():
    if torch.cuda.is_current_stream_capturing():
        hits[0] += 1
        junk = [[] for _ in range(20000)]         # trips the automatic collector if it is enabled
    return real()


gc.enable()
_conv._stream = noisy
r, _ = make(0)
r.run()
torch.cuda.synchronize()
print("ok", name, "allocation bursts inside captures:", hits[0], flush=True)
