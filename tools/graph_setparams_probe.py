"""Feasibility probe: a kernel node of a captured torch graph re-parameterised per replay with hipGraphExecKernelNodeSetParams
(torch.cuda.CUDAGraph(keep_graph=True).raw_cuda_graph() / raw_cuda_graph_exec()), through ctypes on libamdhip64."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bnn_priors_amd import _hip

hip = ctypes.CDLL("libamdhip64.so", mode=ctypes.RTLD_GLOBAL) if False else None
for name in ("libamdhip64.so.7", "libamdhip64.so"):
    try:
        hip = ctypes.CDLL(name)
        break
    except OSError:
        pass
print("hip lib", hip)

class dim3(ctypes.Structure):
    _fields_ = [("x", ctypes.c_uint32), ("y", ctypes.c_uint32), ("z", ctypes.c_uint32)]
class KParams(ctypes.Structure):
    _fields_ = [("blockDim", dim3), ("extra", ctypes.c_void_p), ("func", ctypes.c_void_p), ("gridDim", dim3),
                ("kernelParams", ctypes.c_void_p), ("sharedMemBytes", ctypes.c_uint32)]

dev = torch.device("cuda", 0)
lib = _hip.lib()
data = torch.arange(40, dtype=torch.float32, device=dev).view(10, 1, 1, 4)
labels = torch.arange(10, dtype=torch.int64, device=dev)
idxA = torch.tensor([0, 1, 2], dtype=torch.int64, device=dev)
idxB = torch.tensor([7, 8, 9], dtype=torch.int64, device=dev)
out = torch.zeros(3, 1, 1, 4, device=dev)
lout = torch.zeros(3, dtype=torch.int64, device=dev)
res = torch.zeros(3, 1, 1, 4, device=dev)

def gather_args(idx):
    return _hip.Gather(data=data.data_ptr(), labels=labels.data_ptr(), idx=idx.data_ptr(), out=out.data_ptr(),
                       labels_out=lout.data_ptr(), fill=0, batch=3, channels=1, height=1, width=4, pad=0, flip=0, seed=0,
                       draw=0, stream=0, reserved=0)

def launch(idx):
    G = gather_args(idx)
    _hip.check(lib.sgmcmc_gather_stage(ctypes.byref(G), None, None, None, 0, None, None,
                                       torch.cuda.current_stream(dev).cuda_stream), "gather_stage")

s = torch.cuda.Stream()
with torch.cuda.stream(s):
    launch(idxA); res.copy_(out * 2)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph(keep_graph=True)
with torch.cuda.graph(g):
    launch(idxA)
    res.copy_(out * 2)
try:
    g.instantiate()
except Exception as e:
    print("instantiate:", e)
g.replay(); torch.cuda.synchronize()
print("replay A:", res.flatten().tolist())
graph, gexec = g.raw_cuda_graph(), g.raw_cuda_graph_exec()
print("handles", hex(graph), hex(gexec))
n = ctypes.c_size_t(0)
print("getnodes", hip.hipGraphGetNodes(ctypes.c_void_p(graph), None, ctypes.byref(n)), n.value)
nodes = (ctypes.c_void_p * n.value)()
hip.hipGraphGetNodes(ctypes.c_void_p(graph), nodes, ctypes.byref(n))
knode = None
for nd in nodes:
    t = ctypes.c_int(0)
    hip.hipGraphNodeGetType(ctypes.c_void_p(nd), ctypes.byref(t))
    if t.value == 0 and knode is None:      # hipGraphNodeTypeKernel
        p = KParams()
        e = hip.hipGraphKernelNodeGetParams(ctypes.c_void_p(nd), ctypes.byref(p))
        print("kernel node", hex(nd), "err", e, "grid", p.gridDim.x, "block", p.blockDim.x, "func", hex(p.func or 0))
        knode, kp = nd, p
# new kernel params: the kernel takes (sgmcmc_gather G, int n_parts, Copies Cp, sgmcmc_layout L, sgmcmc_step_args A, int n_fin)
# -> reuse the captured argument pointers except the first (G): kernelParams is an array of pointers to argument storage
old = ctypes.cast(kp.kernelParams, ctypes.POINTER(ctypes.c_void_p))
GB = gather_args(idxB)
newp = (ctypes.c_void_p * 6)(ctypes.addressof(GB), old[1], old[2], old[3], old[4], old[5])
kp2 = KParams(blockDim=kp.blockDim, extra=None, func=kp.func, gridDim=kp.gridDim,
              kernelParams=ctypes.cast(newp, ctypes.c_void_p), sharedMemBytes=kp.sharedMemBytes)
e = hip.hipGraphExecKernelNodeSetParams(ctypes.c_void_p(gexec), ctypes.c_void_p(knode), ctypes.byref(kp2))
print("setparams err", e)
g.replay(); torch.cuda.synchronize()
print("replay B:", res.flatten().tolist(), "labels", lout.tolist())
t0 = time.perf_counter()
for _ in range(1000):
    hip.hipGraphExecKernelNodeSetParams(ctypes.c_void_p(gexec), ctypes.c_void_p(knode), ctypes.byref(kp2))
print("setparams host us", (time.perf_counter() - t0) * 1e3)
