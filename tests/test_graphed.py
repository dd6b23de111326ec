"""Stream capture that survives the garbage collector (bnn_priors_amd/_capture.py): every graph capture of the
package -- GraphedLeapfrog, GraphedAccumulate, the evaluation's logits graphs -- goes through it."""
import pytest
import torch


@pytest.mark.gpu
def test_a_dead_runners_graphs_are_collected_before_a_capture_not_inside_it():
    """ROCm's ~CUDAGraph throws (=> terminate) when any stream is capturing: a dead reference cycle that owns a graph
    must be collected before the capture begins, and the automatic collector stays off until it ends."""
    import gc
    import weakref
    from bnn_priors_amd import _capture

    class Holder:
        pass

    was = gc.isenabled()
    gc.disable()
    try:
        h = Holder()
        h.me = h
        h.buf = torch.zeros(8, device="cuda:0")
        h.graph = torch.cuda.CUDAGraph()
        with _capture.capture(h.graph):
            h.buf += 1
        alive = weakref.ref(h)
        del h
        assert alive() is not None                                  # only the collector can free it
        gc.enable()
        buf = torch.zeros(8, device="cuda:0")
        g = torch.cuda.CUDAGraph()
        with _capture.capture(g):
            assert alive() is None and not gc.isenabled()
            junk = [[] for _ in range(5000)]                        # would trip the generation-0 threshold
            buf += 2
        assert gc.isenabled()
        g.replay()
        torch.cuda.synchronize()
        assert buf.tolist() == [2.0] * 8
    finally:
        gc.enable() if was else gc.disable()
