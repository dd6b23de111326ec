"""Timeline of the merged backward launch (and the forward one) from in-kernel wall-clock stamps of a -DSGMCMC_STAMPS build:
    python tools/lab/bwd_timeline.py tools/_ab/stamps.so
per role (weight-gradient / data-gradient workgroups): mean and max of every stamp relative to the launch's first start,
and how the workgroups were placed on the CUs."""
import ctypes, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from bnn_priors_amd import _hip
_hip.LIB_PATH = os.path.abspath(sys.argv[1])
lib, dev = _hip.lib(), torch.device("cuda", 0)
raw = ctypes.CDLL(_hip.LIB_PATH)
raw.sgmcmc_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
s = torch.cuda.current_stream(dev).cuda_stream
n = 128


def stamps(n_blocks):
    torch.cuda.synchronize()
    buf = np.zeros((n_blocks, 8), dtype=np.uint64)
    assert raw.sgmcmc_debug_stamps(buf.ctypes.data, n_blocks) == 0
    return buf


def report(title, buf, roles):
    t0 = buf[:, 0].min()
    print(f"== {title}: {len(buf)} workgroups, first start -> last end {(buf[:, 6].max() - t0) / 100:.2f} us")
    for name, sel, labels in roles:
        b = buf[sel]
        cols = "  ".join(f"{lab} {np.mean((b[:, k] - t0) / 100.0):5.2f} (max {np.max((b[:, k].astype(np.int64) - int(t0)) / 100.0):5.2f})" for k, lab in labels)
        print(f"  {name:6s} x{len(b):4d}: {cols}")
    hw = buf[:, 7]
    xcc, hwid = (hw >> np.uint64(32)).astype(np.int64), (hw & np.uint64(0xffffffff)).astype(np.int64)
    cu = ((hwid >> 8) & 0xf) | (((hwid >> 13) & 0x7) << 4) | (xcc << 8)         # (cu_id, se_id/sh bits, xcc): a label, not a decode
    per = collections.Counter(cu.tolist())
    print(f"  placement: {len(per)} distinct (xcc, se, cu) labels; workgroups per label: {dict(collections.Counter(per.values()))}")
    for name, sel, _ in roles:
        c = collections.Counter(cu[sel].tolist())
        print(f"    {name}: per label {dict(collections.Counter(c.values()))}")


for c, hw in ((16, 32), (32, 16), (64, 8)):
    g = torch.Generator(device=dev).manual_seed(c)
    x = torch.randn((n, c, hw, hw), generator=g, device=dev)
    dy = torch.randn((n, c, hw, hw), generator=g, device=dev)
    w = torch.randn((c, c, 3, 3), generator=g, device=dev) * (2.0 / (9 * c)) ** .5
    y, dx, out = torch.zeros_like(x), torch.zeros_like(x), torch.relu(torch.randn((n, c, hw, hw), generator=g, device=dev))
    slices = lib.sgmcmc_conv3x3_stat_slices(n, c, hw)
    st = torch.zeros((c, slices * 2, 2), dtype=torch.float64, device=dev)      # (room for a lab build with 4-row bands)
    scratch = torch.zeros(lib.sgmcmc_conv3x3_wrw_scratch_floats(n, c, hw), device=dev)
    saved = torch.stack([torch.zeros(c, device=dev), torch.ones(c, device=dev)])
    part = torch.zeros((c, slices, 2), dtype=torch.float64, device=dev)
    E = _hip.ConvBwdEpilogue()
    E.s_y, E.s_out, E.s_mean, E.s_invstd, E.s_partial = y.data_ptr(), out.data_ptr(), saved[0].data_ptr(), saved[1].data_ptr(), part.data_ptr()
    E.mask_dx = 1
    slabs = ctypes.c_int(0)
    flush = torch.empty(64 << 20, device=dev)
    for rep in range(3):
        flush.normal_()          # something else in the caches, as in the step
        _hip.check(lib.sgmcmc_conv3x3(x.data_ptr(), w.data_ptr(), y.data_ptr(), n, c, hw, 0, st.data_ptr(), s), "fwd")
    nb = n * (hw // int(os.environ.get("FWD_ROWS", "8"))) * (c // 16) if c < 64 else n * (hw // 8) * (c // 16)
    f = stamps(nb)
    nb_fwd, nb = nb, n * (hw // 8) * (c // 16)
    report(f"forward {c}@{hw}", f, [("fwd", slice(0, nb_fwd), ((1, "half 0 in LDS"), (4, "B in registers"), (5, "half 0 MFMAs issued"), (2, "half 1 in LDS"), (3, "MFMAs + stores issued"), (6, "end")))])
    for rep in range(3):
        flush.normal_()
        _hip.check(lib.sgmcmc_conv3x3(dy.data_ptr(), w.data_ptr(), dx.data_ptr(), n, c, hw, 1, 0, s), "dgrad")
    f = stamps(nb)
    report(f"data gradient alone {c}@{hw} (transposed weights, no epilogue)", f, [("dgrad", slice(0, nb), ((1, "half 0 in LDS"), (4, "B in registers"), (5, "half 0 MFMAs issued"), (2, "half 1 in LDS"), (3, "MFMAs + stores issued"), (6, "end")))])
    for rep in range(3):
        flush.normal_()
        _hip.check(lib.sgmcmc_conv3x3_wrw(x.data_ptr(), dy.data_ptr(), w.data_ptr() * 0 + torch.empty_like(w).data_ptr(), scratch.data_ptr(), n, c, hw, s), "wrw")
    n_w = (n * (hw // 8) // (4 if c >= 64 else 2)) * (c // 16)
    f = stamps(n_w)
    report(f"weight gradient alone {c}@{hw}", f, [("wrw", slice(0, n_w), ((0, "start"), (1, "item A in LDS"), (2, "A MFMAs done"), (3, "item B in LDS"), (4, "B MFMAs done"), (6, "end")))])
    for rep in range(3):
        flush.normal_()
        _hip.check(lib.sgmcmc_conv3x3_bwd_ex(x.data_ptr(), w.data_ptr(), dy.data_ptr(), dx.data_ptr(), ctypes.byref(E), 0,
                                             scratch.data_ptr(), n, c, hw, ctypes.byref(slabs), s), "bwd")
    n_wrw = slabs.value * (c // 16)
    b = stamps(n_wrw + nb)
    report(f"merged backward {c}@{hw} ({n_wrw} weight-gradient + {nb} data-gradient workgroups)", b,
           [("wrw", slice(0, n_wrw), ((0, "start"), (1, "item A in LDS"), (2, "A MFMAs done"), (3, "item B in LDS"), (4, "B MFMAs done"), (6, "end"))),
            ("dgrad", slice(n_wrw, n_wrw + nb), ((0, "start"), (1, "half 0 in LDS"), (2, "half 1 in LDS"), (3, "MFMAs + stores issued"), (6, "end")))])
