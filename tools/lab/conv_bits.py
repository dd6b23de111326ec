"""sha1 of the trunk convolutions' outputs on fixed inputs (forward + statistics, merged backward with every epilogue):
two library builds that print the same lines compute the same bits.    python tools/lab/conv_bits.py"""
import ctypes, hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bnn_priors_amd import _hip
lib, dev = _hip.lib(), torch.device("cuda", 0)
s = torch.cuda.current_stream(dev).cuda_stream
h = lambda *ts: hashlib.sha1(b"".join(t.detach().cpu().contiguous().numpy().tobytes() for t in ts)).hexdigest()[:12]
for n in (128, 5):
    for c, hw in ((16, 32), (32, 16), (64, 8)):
        g = torch.Generator(device=dev).manual_seed(c + n)
        x = torch.randn((n, c, hw, hw), generator=g, device=dev)
        dy = torch.randn((n, c, hw, hw), generator=g, device=dev)
        w = torch.randn((c, c, 3, 3), generator=g, device=dev) * (2.0 / (9 * c)) ** .5
        y, dx, out = torch.zeros_like(x), torch.zeros_like(x), torch.relu(torch.randn((n, c, hw, hw), generator=g, device=dev))
        slices = lib.sgmcmc_conv3x3_stat_slices(n, c, hw)
        stats = torch.zeros((c, slices, 2), dtype=torch.float64, device=dev)
        scratch = torch.zeros(lib.sgmcmc_conv3x3_wrw_scratch_floats(n, c, hw), device=dev)
        saved = torch.stack([torch.randn(c, generator=g, device=dev) * .1, torch.rand(c, generator=g, device=dev) + .5])
        part = torch.zeros((c, slices, 2), dtype=torch.float64, device=dev)
        dw = torch.zeros_like(w)
        _hip.check(lib.sgmcmc_conv3x3(x.data_ptr(), w.data_ptr(), y.data_ptr(), n, c, hw, 0, stats.data_ptr(), s), "fwd")
        line = [f"n={n} c={c}: fwd {h(y, stats)}"]
        for name, add, sums, mask in (("plain", 0, 0, 0), ("add", 1, 0, 0), ("sums", 0, 1, 0), ("add+sums+mask", 1, 1, 1)):
            E = _hip.ConvBwdEpilogue()
            if sums:
                E.s_y, E.s_out, E.s_mean, E.s_invstd, E.s_partial = y.data_ptr(), out.data_ptr(), saved[0].data_ptr(), saved[1].data_ptr(), part.data_ptr()
                E.mask_dx = mask
            if add:
                E.e_dout, E.e_out = x.data_ptr(), out.data_ptr()
            part.zero_(); dx.zero_(); dw.zero_()
            _hip.check(lib.sgmcmc_conv3x3_bwd_ex(x.data_ptr(), w.data_ptr(), dy.data_ptr(), dx.data_ptr(), ctypes.byref(E), dw.data_ptr(),
                                                 scratch.data_ptr(), n, c, hw, None, s), name)
            line.append(f"{name} dx {h(dx)} dw {h(dw)} part {h(part)}")
        torch.cuda.synchronize()
        print("  ".join(line))
