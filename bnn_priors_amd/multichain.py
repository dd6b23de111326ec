"""Several chains on ONE GPU from one process: every runner on its own HIP stream, their steps interleaved.

A captured step of the convolutional nets is a chain of ~90 dependent launches, each of which leaves the GPU partly idle at
its boundaries; independent chains fill those gaps (googleresnet: 1.36x one chain's throughput with two chains on two
streams, profiles/r02_bench_googleresnet_stream_chains.json).  The reference runs one chain per process
(experiments/run_experiment.sh:15-34); this is the same set of independent Markov chains -- own model, own data order,
Philox stream = ``chain_id`` -- scheduled differently.  For the dense classifier use ``fused_dense.MultiChainDense``
(chains as a grid dimension of the step's kernels) instead.

    runners = [runner_class("VerletSGLDReject")(model=make_model(), ..., seed=1234, chain_id=c) for c in range(2)]
    multichain.run_on_streams(runners)         # == r.run() for every r, interleaved step by step
"""
import time

import torch


def _spin_time(streams, device, cycles, links):
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for s in streams:
        with torch.cuda.stream(s):
            for _ in range(links):
                torch.cuda._sleep(cycles)
    torch.cuda.synchronize(device)
    return time.perf_counter() - t0


_distinct = {}      # device index -> streams measured to own a hardware queue each (once per process and device)


def concurrent_streams(k, device, exclude=(), candidates=16, cycles=400_000, links=3):
    """Up to ``k`` HIP streams that the GPU really runs side by side (none of them in ``exclude``).

    HIP multiplexes a process's streams onto a few hardware queues (``GPU_MAX_HW_QUEUES``, 4 unless the variable is set
    before the runtime starts) and two streams that share a queue run their work back to back: measured in round 5,
    chains 3 and 4 of ``run_on_streams`` landed on the queues of chains 1 and 2 and the aggregate fell back to two
    chains' throughput (profiles/r05_chains_per_gpu.txt).  Which stream shares which queue is not an API property, so
    it is measured, once per device: every candidate is raced against the streams already chosen with chains of
    one-thread spin kernels (a queue slot and nothing else) and kept when the race takes one chain's time rather than
    two -- about 20 ms.  Returns fewer than ``k`` streams if the process does not have that many queues."""
    device = torch.device(device)
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    chosen = _distinct.get(device.index)
    if chosen is None:
        chosen = []
        pool = [torch.cuda.Stream(device=device) for _ in range(candidates)]
        for s in pool:                                   # first use of a stream creates its queue: not inside the race
            _spin_time([s], device, 1000, 1)
        alone = min(_spin_time([pool[0]], device, cycles, links) for _ in range(3))
        for s in pool:
            if len(chosen) == 8:
                break
            if chosen:
                raced = min(_spin_time(chosen + [s], device, cycles, links) for _ in range(2))
                if raced >= 1.5 * alone:                 # (a shared queue gives >= 2.0, distinct queues ~1.0)
                    continue
            chosen.append(s)
        _distinct[device.index] = chosen
    skip = {e.cuda_stream for e in exclude}
    return [s for s in chosen if s.cuda_stream not in skip][:k]


def spread(streams, n):
    "``n`` streams out of ``streams``, cycling when there are fewer (work on a shared stream simply runs back to back)"
    return [streams[i % len(streams)] for i in range(n)]


def run_on_streams(runners, streams=None):
    """``runner.run()`` for every runner, one HIP stream each, advanced round-robin one minibatch step at a time
    (``run_iter``).  Every chain's results are those of running it alone (the chains share nothing but the GPU)."""
    runners = list(runners)
    if not runners:
        return
    device = next(runners[0].model.parameters()).device
    if streams is None:
        streams = spread(concurrent_streams(len(runners), device), len(runners))
    main = torch.cuda.current_stream(device)
    for s in streams:
        s.wait_stream(main)
    gens = [r.run_iter() for r in runners]
    alive = list(range(len(runners)))
    while alive:
        for k in list(alive):
            with torch.cuda.stream(streams[k]):
                try:
                    next(gens[k])
                except StopIteration:
                    alive.remove(k)
    for s in streams:
        main.wait_stream(s)
