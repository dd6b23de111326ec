"""How much cheaper per image is the gradient evaluation of googleresnet at 256 / 512 rows per launch set than at 128?
(What a mega-batch exact pass -- several minibatches per launch, BatchNorm statistics per minibatch -- could gain over
the two-lane pass: here with ONE BatchNorm group per launch set, i.e. timing only.)  python tools/batch_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from bnn_priors_amd import _capture, conv, models, pool

dev = "cuda:0"
torch.manual_seed(0)
x0, y0 = torch.rand(16, 3, 32, 32), torch.arange(16) % 10
net = models.get_model(x0, y0, "googleresnet", width=50, depth=20, weight_prior="gaussian", weight_loc=0.,
                       weight_scale=2 ** .5, bias_prior="gaussian", bias_loc=0., bias_scale=1., batchnorm=True,
                       weight_prior_params={}, bias_prior_params={}).to(dev).train()
params = [p for p in net.parameters() if p.requires_grad]
for n in (128, 256, 512, 1024):
    xb = torch.rand(n, 3, 32, 32, device=dev)
    yb = (torch.arange(n, device=dev) % 10)

    def body():
        for p in params:
            p.grad = None
        with conv.deferring(net):
            with pool.head_loss(yb, "sum", 50000):
                f = net.net(xb)
            pool.cross_entropy_backward(f, yb, reduction="sum", divide_by=50000)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            body()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with _capture.capture(g):
        body()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 40
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"rows {n:5d}: {ms * 1e3:8.1f} us per gradient evaluation, {ms * 1e3 / (n / 128):7.1f} us per 128 rows", flush=True)
