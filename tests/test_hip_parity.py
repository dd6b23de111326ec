"""Parity of the HIP path (through the C ABI) with the oracle and the goldens.

Bars: bit-exact for integers and flags (Philox words, accept/reject, step indices)
and -- because the arithmetic is specified operation by operation -- also for the
element-wise state theta / m / v against the C oracle; fp64-accumulated dot products
within 1e-12 relative of the oracle's serial fp64 sums (only the summation order
differs); golden trajectories from the reference within the north star's floating
point tolerance (written next to each assertion).
"""
import ctypes
import math

import numpy as np
import pytest
import torch

import scenarios as S
from helpers import PlainHooks, compare, default_dtype, golden
from oracle import noise
from oracle.flat import FLAG_FINAL, FLAG_INITIAL, FLAG_SAVE, FlatArena

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _mcmc():
    from bnn_priors_amd import mcmc
    return mcmc


# ------------------------------------------------------------------ noise
@pytest.mark.parametrize("start,n,seed,stream,draw,purpose", [
    (0, 1 << 16, 1234, 0, 0, 0), (3, 4099, 2 ** 63 + 5, 7, 2 ** 35 + 1, 1),
    (2 ** 34 + 6, 1000, 99, 4095, 17, 2), (1, 1, 5, 1, 1, 0)])
def test_device_normals_bit_exact(start, n, seed, stream, draw, purpose):
    from bnn_priors_amd import _hip
    out = torch.empty(n, dtype=torch.float32, device=DEV)
    _hip.check(_hip.lib().sgmcmc_debug_normals(out.data_ptr(), start, n, seed, stream, draw, purpose,
                                               ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
               "debug_normals")
    ref = noise.normals(seed, stream, draw, purpose, start, n)
    got = out.cpu().numpy()
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_single_hip_runtime_in_process():
    _mcmc()
    from bnn_priors_amd import _hip
    _hip.lib()
    libs = {line.split()[-1] for line in open("/proc/self/maps") if "libamdhip64" in line}
    assert len(libs) == 1, libs


# ------------------------------------------------------------------ flat step vs C oracle
NUMELS = [5, 4096, 1, 4097, 130, 12288, 64]


def _setup(kind, dtype, a, T, numels=NUMELS, seed=4242, stream=2, **engine_options):
    mcmc = _mcmc()
    g = torch.Generator().manual_seed(11)
    params = [torch.nn.Parameter(torch.randn(n, generator=g, dtype=torch.float64).to(dtype).to(DEV))
              for n in numels]
    if kind == "hmc":
        opt = mcmc.HMC(params, lr=0.02, num_data=7, raise_on_nan=False, seed=seed, chain_id=stream,
                       **engine_options)
    else:
        cls = mcmc.VerletSGLD if kind == "verlet" else mcmc.SGLD
        opt = cls(params, lr=0.02, num_data=7, momentum=a, temperature=T, seed=seed, chain_id=stream,
                  **engine_options)
    npdt = np.float32 if dtype == torch.float32 else np.float64
    fa = FlatArena(numels, npdt)
    for s, p in enumerate(params):
        opt.state[p]['preconditioner'] = 0.3 + 0.1 * s
        v0 = (torch.rand(p.shape, generator=g, dtype=torch.float64) + 0.5).to(dtype)
        opt.state[p]['square_avg'].copy_(v0)
        fa.seg(fa.v, s)[:] = v0.numpy()
        fa.seg(fa.theta, s)[:] = p.detach().cpu().numpy()
        fa.M[s] = 0.3 + 0.1 * s
    return params, opt, fa, g


def _set_grads(params, fa, g, dtype, scale=1.0):
    for s, p in enumerate(params):
        gr = (scale * torch.randn(p.shape, generator=g, dtype=torch.float64)).to(dtype)
        p.grad = gr.to(DEV)
        fa.seg(fa.g, s)[:] = gr.numpy()


def _assert_state_bit_exact(params, opt, fa, check_m=True, what=""):
    for s, p in enumerate(params):
        assert np.array_equal(p.detach().cpu().numpy(), fa.seg(fa.theta, s)), f"theta seg {s} {what}"
        assert np.array_equal(opt.state[p]['square_avg'].cpu().numpy(), fa.seg(fa.v, s)), f"v seg {s} {what}"
        if check_m:
            assert np.array_equal(opt.state[p]['momentum_buffer'].cpu().numpy(), fa.seg(fa.m, s)), \
                f"m seg {s} {what}"


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("kind,a,T", [("verlet", 0.9, 0.75), ("verlet", 0.0, 1.0), ("verlet", 1.0, 1.0),
                                      ("verlet", 0.994, 0.0), ("hmc", 1.0, 1.0), ("sgld", 0.9, 0.75),
                                      ("sgld", 0.0, 0.75), ("sgld", 0.9, 0.0)])
def test_step_bit_exact_vs_c_oracle(kind, a, T, dtype):
    params, opt, fa, g = _setup(kind, dtype, a, T)
    eng = opt.engine
    opt.sample_momentum()
    fa.sample_momentum(math.sqrt(T if kind != "hmc" else 1.0), 0.0, 4242, 0, stream=2)
    _assert_state_bit_exact(params, opt, fa, what="after sample_momentum")
    calls = [("initial", FLAG_INITIAL | FLAG_SAVE), ("middle", 0), ("middle", 0), ("final", FLAG_FINAL)]
    for step_i, (which, flags) in enumerate(calls):
        _set_grads(params, fa, g, dtype)
        if kind == "sgld":
            if which == "final":
                opt.final_step(calc_metrics=a > 0)
            else:
                opt.step()
            grp = opt.param_groups[0]
            sums = fa.step("sgld", grad_v=1.0, bhn=grp['hn'], bh=grp['h'], mom_decay=a,
                           noise_std=grp['noise_std'] if T > 0 else 0.0, alpha=0.99, seed=4242,
                           draw=step_i + 1, stream=2, flags=flags & ~FLAG_SAVE).copy()
        else:
            if which == "initial":
                opt.initial_step(save_state=True)
            elif which == "middle":
                opt.step()
            else:
                opt.final_step()
            grp = opt.param_groups[0]
            sums = fa.step(kind, grad_v=grp['grad_v'], bhn=grp['bhn'], bh=grp['bh'],
                           mom_decay=grp['mom_decay'], noise_std=grp['noise_std'], alpha=0.99,
                           seed=4242, draw=step_i + 1, stream=2, flags=flags).copy()
        _assert_state_bit_exact(params, opt, fa, check_m=not (kind == "sgld" and a == 0),
                                what=f"after {which} #{step_i}")
        dev_sums = eng.fetch_state()[:, :6]
        np.testing.assert_allclose(dev_sums, sums, rtol=1e-12, atol=1e-300)
        if flags & FLAG_SAVE and kind != "sgld":
            for s, p in enumerate(params):
                st = opt.state[p]
                assert np.array_equal(st['prev_parameter'].cpu().numpy(), fa.seg(fa.prev_theta, s))
                assert np.array_equal(st['prev_grad'].cpu().numpy(), fa.seg(fa.prev_g, s))
                assert np.array_equal(st['prev_momentum_buffer'].cpu().numpy(), fa.seg(fa.prev_m, s))
    # momentum refresh with keep != 0 (sgld.py:69)
    if kind != "sgld" or a > 0:
        opt.sample_momentum(keep=0.3)
        fa.sample_momentum(math.sqrt((T if kind != "hmc" else 1.0) * 0.7), 0.3, 4242, len(calls) + 1, stream=2)
        _assert_state_bit_exact(params, opt, fa, what="after partial refresh")


def _baseline_numels(workload):
    "tensor sizes of BASELINE.json's nets (configs[1], [2], [3]), in parameter order"
    from bnn_priors_amd import models
    name, xshape = {"densenet": ("classificationdensenet", (784,)), "convnet": ("classificationconvnet", (784,)),
                    "googleresnet": ("googleresnet", (3, 32, 32))}[workload]
    net = models.get_model(torch.zeros((2,) + xshape), torch.tensor([0, 9]), name, width=50, depth=3,
                           weight_prior="gaussian", weight_scale=2 ** .5, bias_prior="gaussian", bias_scale=1.)
    return [p.numel() for p in net.parameters()]


@pytest.mark.parametrize("kind,a,T", [("verlet", 0.994, 1.0), ("hmc", 1.0, 1.0)])
@pytest.mark.parametrize("workload,total", [("densenet", 42310), ("convnet", None), ("googleresnet", 272474),
                                            ("flat_2p24", (1 << 24) + 3)])
def test_step_bit_exact_at_baseline_sizes(workload, total, kind, a, T):
    """The sampler transition at BASELINE.json's FULL sizes -- the three nets' real tensor lists, and one 2^24 + 3
    element segment (the non-temporal streaming variant of the kernel) -- against the C oracle, bit for bit:
    momentum refresh, initial / ordinary / final step with saved state, the six fp64 sums to 1e-12."""
    numels = [total] if workload.startswith("flat") else _baseline_numels(workload)
    if total is not None:
        assert sum(numels) == total
    dtype = torch.float32
    params, opt, fa, g = _setup(kind, dtype, a, T, numels=numels)
    opt.sample_momentum()
    fa.sample_momentum(math.sqrt(T if kind != "hmc" else 1.0), 0.0, 4242, 0, stream=2)
    _assert_state_bit_exact(params, opt, fa, what="after sample_momentum")
    calls = [("initial", FLAG_INITIAL | FLAG_SAVE), ("middle", 0), ("final", FLAG_FINAL)]
    for step_i, (which, flags) in enumerate(calls):
        _set_grads(params, fa, g, dtype)
        if which == "initial":
            opt.initial_step(save_state=True)
        elif which == "middle":
            opt.step()
        else:
            opt.final_step()
        grp = opt.param_groups[0]
        sums = fa.step(kind, grad_v=grp['grad_v'], bhn=grp['bhn'], bh=grp['bh'], mom_decay=grp['mom_decay'],
                       noise_std=grp['noise_std'], alpha=0.99, seed=4242, draw=step_i + 1, stream=2, flags=flags).copy()
        _assert_state_bit_exact(params, opt, fa, what=f"after {which} #{step_i}")
        np.testing.assert_allclose(opt.engine.fetch_state()[:, :6], sums, rtol=1e-12, atol=1e-300)
    for s_, p in enumerate(params):        # the saved M-H state of the initial step
        assert np.array_equal(opt.state[p]['prev_parameter'].cpu().numpy(), fa.seg(fa.prev_theta, s_))
        assert np.array_equal(opt.state[p]['prev_grad'].cpu().numpy(), fa.seg(fa.prev_g, s_))



def test_unaligned_pointers_take_scalar_path_and_agree():
    "parameters that are views at odd offsets of a bigger buffer (4-byte aligned only)"
    mcmc = _mcmc()
    dtype = torch.float32
    numels = [4099, 33]
    gcpu = torch.Generator().manual_seed(5)
    buf = torch.randn(sum(numels) + 8, generator=gcpu).to(DEV)
    gbuf = torch.randn(sum(numels) + 8, generator=gcpu).to(DEV)
    params, off = [], 1
    for n in numels:
        p = torch.nn.Parameter(torch.empty(0, device=DEV))
        p.data = buf[off:off + n]
        p.grad = gbuf[off:off + n]
        params.append(p)
        off += n + 2
    assert any(p.data_ptr() % 16 for p in params)
    opt = mcmc.VerletSGLD(params, lr=0.02, num_data=7, momentum=0.9, temperature=1.0, seed=1, chain_id=0)
    fa = FlatArena(numels, np.float32)
    for s, p in enumerate(params):
        fa.seg(fa.theta, s)[:] = p.detach().cpu().numpy()
        fa.seg(fa.g, s)[:] = p.grad.cpu().numpy()
        fa.seg(fa.v, s)[:] = 1.0
    opt.sample_momentum()
    fa.sample_momentum(1.0, 0.0, 1, 0)
    opt.initial_step(save_state=False)
    assert opt.engine._unaligned
    grp = opt.param_groups[0]
    fa.step("verlet", grad_v=grp['grad_v'], bhn=grp['bhn'], bh=grp['bh'], mom_decay=grp['mom_decay'],
            noise_std=grp['noise_std'], alpha=0.99, seed=1, draw=1, flags=FLAG_INITIAL)
    _assert_state_bit_exact(params, opt, fa)


def test_gradient_clamp_in_flight():
    "grad_clamp reproduces p.grad.clamp_(+-c) before the step (inference.py:219-220)"
    params, opt, fa, g = _setup("verlet", torch.float32, 0.9, 1.0)
    opt.grad_clamp = 0.5
    opt.sample_momentum()
    fa.sample_momentum(1.0, 0.0, 4242, 0, stream=2)
    _set_grads(params, fa, g, torch.float32, scale=1.0)
    np.clip(fa.g, -0.5, 0.5, out=fa.g)
    opt.initial_step(save_state=True)
    grp = opt.param_groups[0]
    sums = fa.step("verlet", grad_v=grp['grad_v'], bhn=grp['bhn'], bh=grp['bh'],
                   mom_decay=grp['mom_decay'], noise_std=grp['noise_std'], alpha=0.99, seed=4242,
                   draw=1, stream=2, flags=FLAG_INITIAL | FLAG_SAVE).copy()
    _assert_state_bit_exact(params, opt, fa)
    np.testing.assert_allclose(opt.engine.fetch_state()[:, :6], sums, rtol=1e-12)
    for s, p in enumerate(params):
        assert np.array_equal(opt.state[p]['prev_grad'].cpu().numpy(), fa.seg(fa.prev_g, s))


def test_determinism_same_seed_same_bits():
    outs = []
    for _ in range(2):
        params, opt, fa, g = _setup("verlet", torch.float32, 0.9, 1.0, numels=[100000, 37, 8192])
        opt.sample_momentum()
        for k in range(5):
            _set_grads(params, fa, g, torch.float32)
            (opt.initial_step if k == 0 else opt.step)()
        de = opt.delta_energy(0.0, 0.0)
        outs.append((torch.cat([p.detach().reshape(-1) for p in params]).cpu().numpy().copy(),
                     opt.engine.fetch_state().copy(), de))
    assert np.array_equal(outs[0][0].view(np.uint32), outs[1][0].view(np.uint32))
    assert np.array_equal(outs[0][1].view(np.uint64), outs[1][1].view(np.uint64))
    assert outs[0][2] == outs[1][2]


# ------------------------------------------------------------------ goldens from the reference
def run_hip(name, dtype_name):
    import bnn_priors_amd.models as models
    mcmc = _mcmc()
    cfg = S.SCENARIOS[name]
    dtype = getattr(torch, dtype_name)
    with default_dtype(dtype):
        torch.manual_seed(0)
        model, closure = S.make_model(cfg["model"], models, dtype, device=DEV)
        classes = dict(sgld=mcmc.SGLD, verlet=mcmc.VerletSGLD, hmc=mcmc.HMC)
        opt = S.build_optimizer(classes, model.parameters(), cfg, seed=S.SEED, chain_id=0)
        S.preset(model, opt, cfg, dtype, lambda p: opt.state[p])
        return S.drive(opt, model, closure, cfg, PlainHooks(opt))


@pytest.mark.parametrize("dtype_name", S.DTYPES)
@pytest.mark.parametrize("name", sorted(S.SCENARIOS))
def test_hip_reproduces_reference_goldens(name, dtype_name):
    """Same Philox key as the golden run of the imported reference.  Accept/reject
    flags, M-H step indices and the LR stream must be identical; floats: the
    gradients come from device autograd (different libm / reduction order than the
    CPU), so trajectories agree to ~1e-6 relative per step in fp32 and ~1e-13 in fp64;
    tolerances below leave room for 24 steps of growth."""
    rec = run_hip(name, dtype_name)
    if dtype_name == "float32":
        big = "biglr" in name  # lr = 40..120: errors amplify ~10x per M-H block
        compare(rec, golden(dtype_name), name, rtol=2e-3 if big else 2e-4, atol=2e-3 if big else 2e-5,
                u_eps=2.0 ** -23)
    else:
        compare(rec, golden(dtype_name), name, rtol=1e-9, atol=1e-10)


def test_small_finalize_energy_total_equals_delta_energy():
    "scalars[3] of the fused launch == delta_energy(0, 0) recomputed from the current gradient"
    for kind in ("verlet", "hmc"):
        params, opt, fa, g = _setup(kind, torch.float32, 0.9 if kind == "verlet" else 1.0, 1.0)
        assert opt.engine.small_finalize
        opt.sample_momentum()
        for k, call in enumerate(("initial_step", "step", "step", "final_step")):
            _set_grads(params, fa, g, torch.float32)
            getattr(opt, call)()
            fast = opt.delta_energy_of_last_transition(0.25, 0.5)
            slow = opt.delta_energy(0.25, 0.5)
            assert fast == pytest.approx(slow, rel=1e-12, abs=1e-12), (kind, call)


@pytest.mark.parametrize("chunk,small", [(4096, True), (4096, False), (1024, False)])
@pytest.mark.parametrize("kind", ["verlet", "hmc"])
def test_both_chunk_geometries_and_both_finalize_paths(kind, chunk, small):
    """the 4096-element chunk layout (4 items per thread) and the multi-workgroup finalize +
    separate delta_energy reduction are what big arenas use; force them on a small problem"""
    a = 0.9 if kind == "verlet" else 1.0
    params, opt, fa, g = _setup(kind, torch.float32, a, 1.0, numels=[5, 9000, 1, 4097, 12288],
                                chunk_elems=chunk, small_finalize=small)
    assert opt.engine.chunk == chunk and opt.engine.small_finalize == small
    opt.sample_momentum()
    fa.sample_momentum(1.0, 0.0, 4242, 0, stream=2)
    _assert_state_bit_exact(params, opt, fa, what="after sample_momentum")
    for step_i, (call, flags) in enumerate((("initial_step", FLAG_INITIAL | FLAG_SAVE), ("step", 0),
                                            ("final_step", FLAG_FINAL))):
        _set_grads(params, fa, g, torch.float32)
        getattr(opt, call)()
        grp = opt.param_groups[0]
        sums = fa.step(kind, grad_v=grp['grad_v'], bhn=grp['bhn'], bh=grp['bh'],
                       mom_decay=grp['mom_decay'], noise_std=grp['noise_std'], alpha=0.99, seed=4242,
                       draw=step_i + 1, stream=2, flags=flags).copy()
        _assert_state_bit_exact(params, opt, fa, what=call)
        np.testing.assert_allclose(opt.engine.fetch_state()[:, :6], sums, rtol=1e-12, atol=1e-300)
        # energy: per-tensor oracle formula from the sums
        de = opt.delta_energy(0.5, 0.75)
        st = opt.engine.fetch_state()
        if kind == "verlet":
            curv = np.array([(0.3 + 0.1 * s) ** 2 * 7 ** 2 * grp['b^2h^2'] / 8 for s in range(len(params))])
            want = float(np.sum(st[:, 6] + curv * sums[:, 0])) + 0.25 * 7
        else:
            want = float(np.sum(st[:, 6] + 0.5 * sums[:, 4])) + 0.25 * 7
        assert de == pytest.approx(want, rel=1e-11, abs=1e-11)


def test_two_parameter_groups_match_per_tensor_oracle():
    "param groups with different learning rates / momenta: one fused launch per group"
    from oracle.noise import NoiseSource
    from oracle.samplers import RefVerletSGLD
    mcmc = _mcmc()
    g = torch.Generator().manual_seed(21)
    numels = [300, 5000, 17, 2048]
    cpu = [torch.nn.Parameter(torch.randn(n, generator=g)) for n in numels]
    dev = [torch.nn.Parameter(p.detach().clone().to(DEV)) for p in cpu]

    def groups(ps):
        return [dict(params=ps[:2], lr=0.02, momentum=0.9), dict(params=ps[2:], lr=0.005, momentum=0.5)]
    ref = RefVerletSGLD(groups(cpu), lr=0.01, num_data=11, momentum=0.7, temperature=1.0,
                        noise=NoiseSource(99, numels, stream=1))
    hip = mcmc.VerletSGLD(groups(dev), lr=0.01, num_data=11, momentum=0.7, temperature=1.0, seed=99,
                          chain_id=1)
    assert not hip.engine.small_finalize
    for i, (p, q) in enumerate(zip(cpu, dev)):
        ref.state[p]['preconditioner'] = hip.state[q]['preconditioner'] = 0.4 + 0.2 * i
    ref.sample_momentum()
    hip.sample_momentum()
    for call in ("initial_step", "step", "step", "final_step"):
        for p, q in zip(cpu, dev):
            p.grad = torch.randn(p.shape, generator=g)
            q.grad = p.grad.to(DEV)
        getattr(ref, call)()
        getattr(hip, call)()
        for p, q in zip(cpu, dev):
            torch.testing.assert_close(q.detach().cpu(), p.detach(), rtol=2e-6, atol=2e-7)
            torch.testing.assert_close(hip.state[q]['momentum_buffer'].cpu(), ref.state[p]['momentum_buffer'],
                                       rtol=2e-6, atol=2e-7)
            for k in ("delta_energy", "prev_new_momentum_delta", "est_temperature", "est_config_temp"):
                assert hip.state[q][k] == pytest.approx(ref.state[p][k], rel=2e-5, abs=1e-6), (call, k)
    assert hip.delta_energy(0.1, 0.3) == pytest.approx(ref.delta_energy(0.1, 0.3), rel=2e-5)


def test_channels_last_parameters_are_supported():
    "dense non-contiguous layouts: theta, grad and the state views share the storage order"
    mcmc = _mcmc()
    g = torch.Generator().manual_seed(8)
    w = torch.randn(16, 8, 3, 3, generator=g)
    p_cl = torch.nn.Parameter(w.clone().to(DEV).contiguous(memory_format=torch.channels_last))
    p_ct = torch.nn.Parameter(w.clone().to(DEV))
    assert not p_cl.is_contiguous()
    outs = []
    for p in (p_cl, p_ct):
        opt = mcmc.HMC([p], lr=0.01, num_data=10, raise_on_nan=False, seed=3)
        m0 = torch.randn(w.shape, generator=torch.Generator().manual_seed(1)).to(DEV)
        opt.sample_momentum()
        opt.state[p]['momentum_buffer'].copy_(m0)       # logical values, whatever the layout
        gr = torch.randn(w.shape, generator=torch.Generator().manual_seed(2)).to(DEV)
        p.grad = gr.contiguous(memory_format=torch.channels_last) if p is p_cl else gr
        opt.initial_step(save_state=True)
        assert opt.state[p]['momentum_buffer'].stride() == p.stride()
        outs.append((p.detach().clone().contiguous(), opt.state[p]['momentum_buffer'].clone().contiguous(),
                     opt.state[p]['square_avg'].clone().contiguous(), opt.delta_energy(0., 0.)))
    for a, b in zip(outs[0][:3], outs[1][:3]):
        assert torch.equal(a, b)            # HMC draws no noise: layouts must agree bit for bit
    assert outs[0][3] == pytest.approx(outs[1][3], rel=1e-12)


# ------------------------------------------------------------------ round-2 additions
@pytest.mark.parametrize("small", [True, False])
def test_tensor_without_gradient_is_left_untouched(small):
    """raise_on_no_grad=False: the reference skips tensors whose grad is None -- parameter, momentum, noise,
    square_avg and the running scalars all stay (mcmc/sgld.py:96-100); the others step as if alone."""
    from oracle.noise import NoiseSource
    from oracle.samplers import RefVerletSGLD
    mcmc = _mcmc()
    g = torch.Generator().manual_seed(21)
    numels = [700, 33, 5000]
    cpu = [torch.nn.Parameter(torch.randn(n, generator=g)) for n in numels]
    dev = [torch.nn.Parameter(p.detach().clone().to(DEV)) for p in cpu]
    ref = RefVerletSGLD(cpu, lr=0.01, num_data=11, momentum=0.7, temperature=1.0, raise_on_no_grad=False,
                        noise=NoiseSource(5, numels, stream=0))
    hip = mcmc.VerletSGLD(dev, lr=0.01, num_data=11, momentum=0.7, temperature=1.0, raise_on_no_grad=False,
                          seed=5, chain_id=0, small_finalize=small)
    ref.sample_momentum()
    hip.sample_momentum()
    frozen = 1
    for call in ("initial_step", "step", "step"):
        for i, (p, q) in enumerate(zip(cpu, dev)):
            p.grad = None if i == frozen else torch.randn(p.shape, generator=g)
            q.grad = None if i == frozen else p.grad.to(DEV)
        th0 = dev[frozen].detach().clone()
        m0 = hip.state[dev[frozen]]['momentum_buffer'].clone()
        v0 = hip.state[dev[frozen]]['square_avg'].clone()
        getattr(ref, call)()
        getattr(hip, call)()
        assert dev[frozen].grad is None                      # no gradient is fabricated
        assert torch.equal(dev[frozen].detach(), th0)
        assert torch.equal(hip.state[dev[frozen]]['momentum_buffer'], m0)
        assert torch.equal(hip.state[dev[frozen]]['square_avg'], v0)
        for i, (p, q) in enumerate(zip(cpu, dev)):
            torch.testing.assert_close(q.detach().cpu(), p.detach(), rtol=2e-6, atol=2e-7)
            torch.testing.assert_close(hip.state[q]['momentum_buffer'].cpu(), ref.state[p]['momentum_buffer'],
                                       rtol=2e-6, atol=2e-7)
            if i != frozen:
                for k in ("delta_energy", "est_temperature", "est_config_temp"):
                    assert hip.state[q][k] == pytest.approx(ref.state[p][k], rel=2e-5, abs=1e-6), (call, k)


def test_state_dict_round_trip_continues_the_chain_bit_for_bit():
    "optimizer.state_dict() / load_state_dict(): arenas, device-resident scalars and the Philox position survive"
    mcmc = _mcmc()

    def make():
        g = torch.Generator().manual_seed(3)
        ps = [torch.nn.Parameter(torch.randn(n, generator=g).to(DEV)) for n in (1500, 40, 4100)]
        return ps, mcmc.VerletSGLD(ps, lr=0.01, num_data=50, momentum=0.9, temperature=1.0, seed=77, chain_id=3)

    def grads(ps, k):
        g = torch.Generator().manual_seed(100 + k)
        for p in ps:
            p.grad = torch.randn(p.shape, generator=g).to(DEV)

    pa, a = make()
    a.sample_momentum()
    for k, call in enumerate(("initial_step", "step", "step")):
        grads(pa, k)
        getattr(a, call)()
    a.update_preconditioner()
    sd = a.state_dict()
    theta = [p.detach().clone() for p in pa]
    pb, b = make()
    with torch.no_grad():
        for p, t in zip(pb, theta):
            p.copy_(t)
    b.load_state_dict(sd)
    for p, q in zip(pa, pb):
        for key in ("delta_energy", "prev_new_momentum_delta", "est_temperature", "est_config_temp", "preconditioner"):
            assert a.state[p][key] == b.state[q][key], key
        assert b.state[q]['momentum_buffer'].data_ptr() == b.engine.momentum_view(b.engine.index[id(q)]).data_ptr()
    for k, call in enumerate(("step", "step", "final_step")):
        grads(pa, 10 + k)
        grads(pb, 10 + k)
        getattr(a, call)()
        getattr(b, call)()
    for p, q in zip(pa, pb):
        assert torch.equal(p.detach(), q.detach())
        assert torch.equal(a.state[p]['momentum_buffer'], b.state[q]['momentum_buffer'])
        assert torch.equal(a.state[p]['square_avg'], b.state[q]['square_avg'])
        assert a.state[p]['delta_energy'] == b.state[q]['delta_energy']
    assert a.delta_energy(0.2, 0.1) == b.delta_energy(0.2, 0.1)
    assert a.engine.mh_uniform() == b.engine.mh_uniform()


# ------------------------------------------------------------------ BASELINE configs[4]: L = 50, T in {1, 0.1, 0.01}
@pytest.mark.parametrize("name", sorted(S.TEMPERED))
def test_tempered_hmc_trajectories_match_the_oracle(name):
    """HMC with trajectories of 50 leapfrog steps at temperature T (an extension beyond mcmc/hmc.py:39, which asserts
    T == 1): momentum refresh N(0, T), leapfrog, M-H with exp(-dH/T).  No reference golden can exist; the HIP path is
    compared with the oracle's per-tensor restatement driven by the same Philox key -- accept/reject flags identical,
    trajectories within the fp32 tolerance of 100 steps, kinetic temperature ~ T."""
    from oracle.noise import NoiseSource
    from oracle.samplers import RefHMC
    import bnn_priors_amd.models as models
    cfg = S.TEMPERED[name]
    S.SCENARIOS[name] = cfg                       # (helpers.compare looks N and T up by name)
    try:
        with default_dtype(torch.float32):
            torch.manual_seed(0)
            model, closure = S.make_model(cfg["model"], models, torch.float32)
            noise = NoiseSource(S.SEED, [p.numel() for p in model.parameters()])
            ref = S.build_optimizer(dict(hmc=RefHMC), model.parameters(), cfg, noise=noise)
            S.preset(model, ref, cfg, torch.float32, lambda p: ref.state[p])
            want = S.drive(ref, model, closure, cfg, PlainHooks(ref), record_every=10)
            torch.manual_seed(0)
            model, closure = S.make_model(cfg["model"], models, torch.float32, device=DEV)
            opt = S.build_optimizer(dict(hmc=_mcmc().HMC), model.parameters(), cfg, seed=S.SEED, chain_id=0)
            S.preset(model, opt, cfg, torch.float32, lambda p: opt.state[p])
            got = S.drive(opt, model, closure, cfg, PlainHooks(opt), record_every=10)
        gold = {f"{name}/{k}": v for k, v in want.items()}
        # log_acc = -dE / T: the fp32 noise of dE (a difference of O(N * U) terms, a few N * ulp(U)) is divided by T,
        # so that stream is compared with the dE tolerance scaled by 1 / T and then taken out of the generic check
        noise = 5e-5 + 4 * cfg["N"] * 2.0 ** -23 * float(np.abs(want["loss"]).max())
        np.testing.assert_allclose(got["mh_log_acc"], want["mh_log_acc"], rtol=5e-4, atol=noise / cfg["T"])
        got["mh_log_acc"] = want["mh_log_acc"].copy()
        compare(got, gold, name, rtol=5e-4, atol=5e-5, u_eps=2.0 ** -23)
        assert len(got["mh_step"]) == 2 and list(got["mh_step"]) == [50, 100]
        # the momentum refresh draws N(0, T): kinetic temperature right after it is T within sampling error (d = 10..160)
        t_est = np.asarray(got["est_temp"])[1]          # the initial_step's estimate, per tensor
        assert np.all(np.abs(t_est / cfg["T"] - 1) < 1.2), (t_est, cfg["T"])
    finally:
        S.SCENARIOS.pop(name, None)
