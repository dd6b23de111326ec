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


_distinct = {}      # device index -> streams measured to own a hardware queue each
_probes = {}        # device index -> how often the measurement ran (a busy GPU makes it reject good streams: it is repeated)
_reserved = {}      # device index -> handles of the streams that carry a chain's own launches (never handed out as lanes)
MAX_PROBES = 3
# How many streams carry CHAINS at once.  The spin-kernel race finds up to 8 queues that run trivial kernels side by side,
# but with real work the aggregate of K chains on K streams collapses beyond four (googleresnet, round 6,
# profiles/r06_chains_per_gpu.txt: lock-step of 4 chains 1.93 ms, of 5 chains 3.35 ms, of 8 chains 4.93 ms; the same with
# GPU_MAX_HW_QUEUES=16) -- four queues make progress together, a fifth makes the hardware take turns.  More chains than this
# share the four streams (two chains on one stream run back to back, which costs nothing against taking turns): 8 chains
# then run at the aggregate of 4 instead of 0.79 of it.
MAX_CHAIN_STREAMS = 4


def _measure(device, candidates, cycles, links):
    pool = [torch.cuda.Stream(device=device) for _ in range(candidates)]
    for s in pool:                                   # first use of a stream creates its queue: not inside the race
        _spin_time([s], device, 1000, 1)
    alone = min(_spin_time([pool[0]], device, cycles, links) for _ in range(3))
    chosen = []
    for s in pool:
        if len(chosen) == 8:
            break
        if chosen:
            raced = min(_spin_time(chosen + [s], device, cycles, links) for _ in range(2))
            if raced >= 1.5 * alone:                 # (a shared queue gives >= 2.0, distinct queues ~1.0)
                continue
        chosen.append(s)
    return chosen


def concurrent_streams(k, device, exclude=(), candidates=16, cycles=400_000, links=3):
    """Up to ``k`` HIP streams that the GPU really runs side by side (none of them in ``exclude``).

    HIP multiplexes a process's streams onto a few hardware queues (``GPU_MAX_HW_QUEUES``, 4 unless the variable is set
    before the runtime starts -- bench.py sets 8) and two streams that share a queue run their work back
    to back: measured in round 5, chains 3 and 4 of ``run_on_streams`` landed on the queues of chains 1 and 2 and the
    aggregate fell back to two chains' throughput (profiles/r05_chains_per_gpu.txt).  Which stream shares which queue is
    not an API property, so it is measured: every candidate is raced against the streams already chosen with chains of
    one-thread spin kernels (a queue slot and nothing else) and kept when the race takes one chain's time rather than
    two -- about 20 ms.  The race is a wall-clock measurement: with other work on the GPU (ranks sharing it over gloo,
    another process, launches in flight) it rejects streams that are fine.  Hence: a result with fewer than ``k``
    streams is measured again (up to MAX_PROBES times per device, the largest set kept), one with a single stream is
    never kept for the process, and falling short is said out loud.  ``SGMCMC_STREAM_PROBE=0`` skips the measurement:
    ``k`` fresh streams are returned as they are (what round 4 did)."""
    import os
    import warnings
    device = torch.device(device)
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    skip = {e.cuda_stream for e in exclude}
    if os.environ.get("SGMCMC_STREAM_PROBE", "1") == "0":
        return [torch.cuda.Stream(device=device) for _ in range(k)]

    def usable(chosen):
        return [s for s in chosen if s.cuda_stream not in skip]
    chosen = _distinct.get(device.index, [])
    while len(usable(chosen)) < k and len(chosen) < 8 and _probes.get(device.index, 0) < MAX_PROBES:
        _probes[device.index] = _probes.get(device.index, 0) + 1
        again = _measure(device, candidates, cycles, links)
        if len(again) > len(chosen):
            chosen = again
        if len(chosen) > 1:
            _distinct[device.index] = chosen
    out = usable(chosen)[:k]
    if len(out) < k:
        warnings.warn(f"multichain: {len(out)} concurrent HIP stream(s) found where {k} were asked for (GPU_MAX_HW_QUEUES="
                      f"{os.environ.get('GPU_MAX_HW_QUEUES', 'unset: 4')}; a GPU that is busy during the 20 ms probe also "
                      "hides queues): chains / lanes beyond that share streams and run back to back", stacklevel=2)
    return out


def reserve(streams, device):
    "these streams carry a chain's own launches from now on: ``lanes`` never hands them out"
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    _reserved.setdefault(idx, set()).update(s.cuda_stream for s in streams)


def lanes(k, device, exclude=()):
    """``k`` DISTINCT streams for helper lanes (the exact pass's ``ConcurrentAccumulate``): measured-concurrent streams
    that are nobody's main stream (``reserve``) and not in ``exclude``; when the measured set has fewer, FRESH streams make
    up the number -- a fresh stream may share a hardware queue with another one, but no lane is the same stream twice and
    no lane queues behind another chain's steps (round 5 took the lanes from the chains' own pool: with K chains a
    chain's exact pass ran on the other chains' main streams and the K > 1 samples/s figures contained that
    serialisation)."""
    import warnings
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    taken = _reserved.get(idx, set()) | {e.cuda_stream for e in exclude}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        pool = [s for s in concurrent_streams(8, device) if s.cuda_stream not in taken]
    out = pool[:k]
    out += [torch.cuda.Stream(device=device) for _ in range(k - len(out))]
    return out


def chain_streams(k, device):
    "one stream per chain for ``k`` chains: at most MAX_CHAIN_STREAMS distinct ones, dealt round-robin (chain c on stream c mod 4)"
    own = concurrent_streams(min(k, MAX_CHAIN_STREAMS), device) or [torch.cuda.Stream(device=device)]
    return spread(own, k)


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
        streams = chain_streams(len(runners), device)
    reserve(streams, device)        # (the chains' exact passes take their lanes from the rest)
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
