"""Do K streams run K captured CHAINS OF MANY SHORT KERNELS concurrently?  Each graph = NODES dependent spin kernels of
~US microseconds (one thread each: nothing but a queue slot), replayed REPS times per stream.
    python tools/lab/graph_chain_concurrency.py"""
import os, time, torch
dev = torch.device("cuda", 0)
NODES, REPS = 81, 20
for US in (8, 40):
    CYC = int(US * 100)          # torch.cuda._sleep counts cycles of a 100 MHz-ish counter on this stack? calibrate below
    streams = [torch.cuda.Stream(device=dev) for _ in range(6)]
    graphs = []
    for s in streams:
        with torch.cuda.stream(s):
            for _ in range(3): torch.cuda._sleep(CYC)
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(NODES): torch.cuda._sleep(CYC)
        graphs.append(g)
    def run(K):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(REPS):
            for k in range(K):
                with torch.cuda.stream(streams[k]):
                    graphs[k].replay()
        t1 = time.perf_counter()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) * 1e3 / REPS, (t1 - t0) * 1e3 / REPS
    base = min(run(1)[0] for _ in range(3))
    row = []
    for K in (1, 2, 3, 4, 6):
        t, h = min(run(K) for _ in range(3))
        row.append(f"K={K}: {t:.3f} ms ({t / base:.2f}x, host {h:.3f})")
    print(f"GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')} {NODES}-node graphs, sleep({CYC}): one chain {base:.3f} ms per replay = {1e3 * base / NODES:.1f} us per node | " + "  ".join(row), flush=True)
