"""Phase timing of mlp_fwdbwd_kernel (clock64 stamps of workgroup 0)."""
import ctypes, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
from bnn_priors_amd import _hip
import test_fused_dense as T
g = torch.Generator().manual_seed(3)
X = torch.randn(2048, 784, generator=g).cuda(); Y = torch.randint(0, 10, (2048,), generator=g).cuda()
idx = torch.randperm(2048, generator=g)[:128].cuda()
mk = lambda *s: (torch.randn(*s, generator=g) * 0.05).cuda()
Ws = [mk(50, 784), mk(50), mk(50, 50), mk(50), mk(10, 50), mk(10)]
trace = torch.zeros(16, dtype=torch.int64, device="cuda")
orig = _hip.MlpArgs
class A(orig):
    def __init__(self, **kw):
        super().__init__(trace=trace.data_ptr(), **kw)
_hip.MlpArgs = A
names = ["phase0", "f1", "f2", "f3", "softmax", "b3", "b2", "b1"]
for rep in range(3):
    T._run_kernel(X, Y, idx, Ws, 128)
    t = trace.cpu().numpy()[:9]
    d = (t[1:] - t[:-1])
    print(" ".join(f"{n}={int(v)}" for n, v in zip(names, d)), "total", int(t[8] - t[0]), "cycles")
# whole-kernel time via events
_hip.MlpArgs = orig
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(5): T._run_kernel(X, Y, idx, Ws, 128)
