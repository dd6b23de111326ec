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
import torch


def run_on_streams(runners, streams=None):
    """``runner.run()`` for every runner, one HIP stream each, advanced round-robin one minibatch step at a time
    (``run_iter``).  Every chain's results are those of running it alone (the chains share nothing but the GPU)."""
    runners = list(runners)
    if not runners:
        return
    device = next(runners[0].model.parameters()).device
    if streams is None:
        streams = [torch.cuda.Stream(device=device) for _ in runners]
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
