"""GPU-side cost of a chain of tiny dependent kernels: launched on a stream vs replayed from a captured hipGraph.
python tools/launch_floor.py"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bnn_priors_amd import _hip

dev = torch.device("cuda", 0)
lib = _hip.lib()
x = torch.zeros(256, device=dev)
y = torch.zeros(256, device=dev)
N = 200


def chain_aten():
    for _ in range(N):
        x.add_(1.0)


src = (ctypes.c_void_p * 1)(x.data_ptr())
dst = (ctypes.c_void_p * 1)(y.data_ptr())
nb = (ctypes.c_int64 * 1)(1024)


def chain_own():
    s = torch.cuda.current_stream(dev).cuda_stream
    for _ in range(N):
        lib.sgmcmc_stage_batch(src, dst, nb, 1, None, None, s)


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize(dev)
    return e0.elapsed_time(e1) * 1e3 / (reps * N), (time.perf_counter() - t0) * 1e6 / (reps * N)


for name, chain in (("ATen add_", chain_aten), ("own copy kernel (C ABI)", chain_own)):
    gpu, wall = timed(chain)
    print(f"{name:26s} stream launches : {gpu:6.2f} us per kernel between events (wall {wall:6.2f})")
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        chain()
    torch.cuda.current_stream(dev).wait_stream(side)
    with torch.cuda.graph(g):
        chain()
    gpu, wall = timed(g.replay)
    print(f"{name:26s} graph replay    : {gpu:6.2f} us per kernel between events (wall {wall:6.2f})")
