"""Check and time the fp32-MFMA 3x3 convolution kernels against ATen/MIOpen (needs a GPU)."""
import ctypes
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from bnn_priors_amd import _hip

if len(sys.argv) > 1:
    _hip.LIB_PATH = sys.argv[1]      # an experimental build
lib = _hip.lib()
dev = "cuda:0"
torch.backends.cudnn.benchmark = True


def timeit(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def conv(x, w, transpose):
    y = torch.empty_like(x)
    err = lib.sgmcmc_conv3x3(x.data_ptr(), w.data_ptr(), y.data_ptr(), x.shape[0], x.shape[1], x.shape[2],
                             int(transpose), 0, torch.cuda.current_stream().cuda_stream)
    assert err == 0, err
    return y


def wrw(x, dy):
    C = x.shape[1]
    n = lib.sgmcmc_conv3x3_wrw_scratch_floats(x.shape[0], C, x.shape[2])
    scratch = torch.empty(n, device=x.device)
    dw = torch.empty(C, C, 3, 3, device=x.device)
    err = lib.sgmcmc_conv3x3_wrw(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), scratch.data_ptr(), x.shape[0], C,
                                 x.shape[2], torch.cuda.current_stream().cuda_stream)
    assert err == 0, err
    return dw


for C, HW in ((16, 32), (32, 16), (64, 8)):
    g = torch.Generator(device=dev).manual_seed(C)
    x = torch.randn(128, C, HW, HW, device=dev, generator=g)
    w = torch.randn(C, C, 3, 3, device=dev, generator=g) * 0.1
    dy = torch.randn(128, C, HW, HW, device=dev, generator=g)
    ref = F.conv2d(x.double(), w.double(), padding=1)
    got = conv(x, w, False)
    lib_y = F.conv2d(x, w, padding=1)
    print(f"C={C} HW={HW} fwd: max|err| ours {(got.double() - ref).abs().max().item():.3e}  "
          f"MIOpen {(lib_y.double() - ref).abs().max().item():.3e}  (|y| max {ref.abs().max().item():.2f})")
    ref_dx = torch.nn.grad.conv2d_input(x.shape, w.double(), dy.double(), padding=1)
    got_dx = conv(dy, w, True)
    lib_dx = torch.nn.grad.conv2d_input(x.shape, w, dy, padding=1)
    print(f"          bwd-data: ours {(got_dx.double() - ref_dx).abs().max().item():.3e}  "
          f"MIOpen {(lib_dx.double() - ref_dx).abs().max().item():.3e}")
    t_f = timeit(lambda: conv(x, w, False))
    t_fl = timeit(lambda: F.conv2d(x, w, padding=1))
    t_b = timeit(lambda: conv(dy, w, True))
    t_bl = timeit(lambda: torch.nn.grad.conv2d_input(x.shape, w, dy, padding=1))
    flop = 2 * 128 * HW * HW * C * C * 9
    print(f"          fwd {t_f:.1f} us ({flop / t_f / 1e6:.1f} TFLOP/s) vs MIOpen {t_fl:.1f} us;  "
          f"bwd-data {t_b:.1f} us vs MIOpen {t_bl:.1f} us")
    ref_dw = torch.nn.grad.conv2d_weight(x.double(), w.shape, dy.double(), padding=1)
    got_dw = wrw(x, dy)
    lib_dw = torch.nn.grad.conv2d_weight(x, w.shape, dy, padding=1)
    scale = ref_dw.abs().max().item()
    print(f"          wrw: ours {(got_dw.double() - ref_dw).abs().max().item():.3e}  "
          f"MIOpen {(lib_dw.double() - ref_dw).abs().max().item():.3e}  (|dw| max {scale:.1f}); "
          f"deterministic: {torch.equal(got_dw, wrw(x, dy))}")
    t_w = timeit(lambda: wrw(x, dy))
    t_wl = timeit(lambda: torch.nn.grad.conv2d_weight(x, w.shape, dy, padding=1))
    print(f"          wrw {t_w:.1f} us vs MIOpen {t_wl:.1f} us")
