"""Stream capture that a garbage collection cannot bring down.

On ROCm ``at::cuda::CUDAGraph::~CUDAGraph`` synchronises, and a synchronising call made while ANY stream of the
process is capturing fails with hipErrorStreamCaptureUnsupported -- thrown from a destructor, i.e. ``terminate``.
A runner that went out of use usually sits in a reference cycle (bound methods, closures over itself), so its graphs
are freed by the cyclic collector, whenever an allocation count happens to trip it: if that is inside a later runner's
capture, the process aborts (seen as a once-in-a-few-runs abort of the GPU test suite; tools/gc_capture_probe.py
reproduces it at will).  Every capture of this package therefore goes through ``capture``: dead cycles are collected
BEFORE the capture begins, and the automatic collector is off until it has ended."""
import contextlib
import gc

import torch


@contextlib.contextmanager
def capture(graph, **kwargs):
    gc.collect()
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        with torch.cuda.graph(graph, **kwargs):
            yield
    finally:
        if was_enabled:
            gc.enable()
