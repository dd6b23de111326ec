"""What a stream-level transition costs around a captured graph (MI355X / ROCm 7.2 / torch CUDAGraph): per iteration,
(a) graph alone, (b) graph + one small kernel launch between replays, (c) the kernel alone -- wall time per iteration over
400 iterations with the GPU kept busy (no host stall).  (b) - (a) - kernel time = what the launch between two replays costs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device("cuda", 0)
x = torch.randn(1 << 20, device=dev)
y = torch.empty_like(x)
z = torch.zeros(1024, device=dev)

def body():
    for _ in range(8):          # eight dependent ~3-5 us kernels
        torch.mul(x, 1.0001, out=y)
        torch.add(y, 1.0, out=x)

s = torch.cuda.Stream()
with torch.cuda.stream(s):
    body()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    body()

def timed(fn, n=400):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

a = timed(lambda: g.replay())
b = timed(lambda: (z.add_(1.0), g.replay()))
c = timed(lambda: z.add_(1.0))
print(f"graph alone {a:.2f} us/iter; kernel + graph {b:.2f}; kernel alone {c:.2f}; the launch between two replays costs "
      f"{b - a:.2f} us of which the kernel itself ~{c:.2f}")
