"""round 6 lab: how much of the weight-gradient work can hide beside the dependent chain?  SIDE_MODE=merged: the product's
merged backward launch; =fork: conv.SIDE_STREAM as it is (one fork edge per convolution); =once: ONE fork per backward
pass -- the side branch runs ahead of its operands, so the RESULTS ARE WRONG and only the timing means something: an
upper bound of what an edge-free hand-off (stream memory operations) could reach."""
import os
import sys

mode = os.environ.get("SIDE_MODE", "once")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ["SGMCMC_STRICT"] = "1"
os.environ["SGMCMC_ALTERNATIVES"] = "1"
if mode != "merged":
    os.environ["SGMCMC_CONV_SIDE_STREAM"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bnn_priors_amd import conv  # noqa: E402

if mode == "once":
    fork = conv.side_stream_for

    def once(t):
        if conv._side["forked"] is not None:
            return conv._side["forked"][1]
        return fork(t)

    conv.side_stream_for = once
if mode != "merged":
    # the exact pass's grouped launches (several minibatches per launch) keep the merged route
    import contextlib
    from bnn_priors_amd import bn
    plain_grouped = bn.grouped

    @contextlib.contextmanager
    def grouped(G):
        old, conv.SIDE_STREAM = conv.SIDE_STREAM, conv.SIDE_STREAM and G <= 1
        try:
            with plain_grouped(G):
                yield
        finally:
            conv.SIDE_STREAM = old

    bn.grouped = grouped
import bench  # noqa: E402

sys.argv = ["bench.py", "--steps", "200", "--warmup", "30", "--samples", "0", "--cpu-budget", "0", "--sweep-log2", "0",
            "--no-kernel-timing", "--other-workloads", "0", "--stream-chains", ""] + sys.argv[1:]
bench.main()
