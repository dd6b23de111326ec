"""Is the captured googleresnet / convnet step a CHAIN?  Node count by type, edges, roots, nodes with several
predecessors / successors of the step's hipGraph (torch CUDAGraph(keep_graph=True).raw_cuda_graph(), hipGraphGetNodes /
hipGraphGetEdges through ctypes).    python tools/graph_shape_probe.py [--workload convnet]"""
import argparse, collections, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SGMCMC_STRICT", "1")
import torch
import bench
from bnn_priors_amd import graphed
from bnn_priors_amd.inference_reject import runner_class
from bnn_priors_amd.storage import MemoryMetrics

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="googleresnet")
a = ap.parse_args()
orig = torch.cuda.CUDAGraph
class _Cuda:          # (a stand-in for torch.cuda inside graphed.py whose CUDAGraph keeps its hipGraph_t)
    def __getattr__(self, k):
        return getattr(torch.cuda, k)
    @staticmethod
    def CUDAGraph(*a, **kw):
        return orig(keep_graph=True)
class _Torch:
    cuda = _Cuda()
    def __getattr__(self, k):
        return getattr(torch, k)
graphed.torch = _Torch()
dev = torch.device("cuda", 0)
name, xshape, N, prior = bench.WORKLOADS[a.workload]
model = bench.make_model(a.workload, dev)
pool = bench.PoolSource(a.workload, 1280, dev, 1234)
ds = torch.utils.data.TensorDataset(pool.x, pool.y)
loader = torch.utils.data.DataLoader(ds, batch_size=128, shuffle=True)
empty = torch.utils.data.DataLoader(bench._SyntheticSet(0), batch_size=128)
r = runner_class("VerletSGLDReject")(model=model, dataloader=loader, dataloader_test=empty, epochs_per_cycle=2, warmup_epochs=1,
                                     sample_epochs=1, learning_rate=0.01, skip=1, metrics_skip=10, temperature=1.0,
                                     momentum=0.994, sampling_decay="cosine", cycles=1, precond_update=1,
                                     metrics_saver=MemoryMetrics(), model_saver=None, reject_samples=True, seed=1, chain_id=0)
step = r.begin()
for i, (x, y) in enumerate(r._hot_batches()):
    step += 1
    r.leapfrog(step, x, y, last_of_epoch=False)
    if i > 3:
        break
torch.cuda.synchronize()
hip = ctypes.CDLL("libamdhip64.so.7")
g = r._graphed
for variant, cg in g.graphs.items():
    graph = cg.raw_cuda_graph()
    n = ctypes.c_size_t(0)
    hip.hipGraphGetNodes(ctypes.c_void_p(graph), None, ctypes.byref(n))
    nodes = (ctypes.c_void_p * n.value)()
    hip.hipGraphGetNodes(ctypes.c_void_p(graph), nodes, ctypes.byref(n))
    types = collections.Counter()
    for nd in nodes:
        t = ctypes.c_int(0)
        hip.hipGraphNodeGetType(ctypes.c_void_p(nd), ctypes.byref(t))
        types[t.value] += 1
    ne = ctypes.c_size_t(0)
    hip.hipGraphGetEdges(ctypes.c_void_p(graph), None, None, ctypes.byref(ne))
    fr, to = (ctypes.c_void_p * max(ne.value, 1))(), (ctypes.c_void_p * max(ne.value, 1))()
    hip.hipGraphGetEdges(ctypes.c_void_p(graph), fr, to, ctypes.byref(ne))
    succ, pred = collections.Counter(), collections.Counter()
    for i in range(ne.value):
        succ[fr[i]] += 1
        pred[to[i]] += 1
    nr = ctypes.c_size_t(0)
    hip.hipGraphGetRootNodes(ctypes.c_void_p(graph), None, ctypes.byref(nr))
    print(f"{a.workload} variant metrics={variant}: {n.value} nodes by type {dict(types)} (0 kernel, 1 memcpy, 2 memset, 4 empty, "
          f"...), {ne.value} edges, {nr.value} roots, nodes with >1 successors {sum(1 for v in succ.values() if v > 1)}, "
          f">1 predecessors {sum(1 for v in pred.values() if v > 1)}, leaves {n.value - len(succ)}")
