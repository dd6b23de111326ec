"""How many HIP streams of one process does this stack run CONCURRENTLY?  K streams each get a chain of spin kernels
(torch.cuda._sleep: one thread, no memory, no LDS -- nothing but a queue slot), as plain launches and as replays of a
captured graph; wall time of all K chains / time of one chain = 1 when they overlap completely, K when they serialise.
    python tools/stream_concurrency_probe.py"""
import os
import sys
import time

import torch

dev = torch.device("cuda", 0)
CYCLES, LINKS = 2_000_000, 8        # ~1 ms per kernel at 2 GHz


def chain():
    for _ in range(LINKS):
        torch.cuda._sleep(CYCLES)


def run(K, graphs=None, streams=None):
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for k in range(K):
        with torch.cuda.stream(streams[k]):
            if graphs is None:
                chain()
            else:
                graphs[k].replay()
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) * 1e3


def main():
    KMAX = 8
    streams = [torch.cuda.Stream(device=dev) for _ in range(KMAX)]
    graphs = []
    for s in streams:
        with torch.cuda.stream(s):
            chain()
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            chain()
        graphs.append(g)
    print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"), " torch streams:", [hex(s.cuda_stream) for s in streams])
    for mode, gr in (("plain launches", None), ("graph replays", graphs)):
        base = min(run(1, gr, streams) for _ in range(3))
        row = []
        for K in (1, 2, 3, 4, 6, 8):
            t = min(run(K, gr, streams) for _ in range(3))
            row.append(f"K={K}: {t:.2f} ms ({t / base:.2f}x)")
        print(f"{mode:15s} one chain {base:.2f} ms | " + "  ".join(row))


if __name__ == "__main__":
    main()
