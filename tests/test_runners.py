"""Runner-level parity: metric streams of
this build's runners against goldens captured from the reference's runners.

* CPU (not gpu): the runner HOST LOGIC (step numbering, LR / temperature phases,
  M-H bookkeeping, batch order, metric keys) is exercised with the oracle's samplers
  plugged into ``_make_optimizer`` -- the product samplers have no CPU path.
* GPU: the real thing end to end.
"""
import numpy as np
import pytest
import torch

import runner_cases as RC
from bnn_priors_amd import inference, inference_reject, models
from bnn_priors_amd.storage import MemoryMetrics

GOLD = None


def gold():
    global GOLD
    if GOLD is None:
        import os
        GOLD = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "runners.npz")))
    return GOLD


def _runner_class(name):
    return inference_reject.runner_class(RC.CASES[name].get("runner", name))


def _check(name, metrics, runner, rtol, atol, de_atol, acc_atol=0.0, cfg_rtol=None, cfg_atol=0.0):
    cfg_rtol = rtol if cfg_rtol is None else cfg_rtol
    g = gold()
    got = RC.streams_of(metrics)
    want_keys = sorted({k.split("/", 1)[1].rsplit("/", 1)[0] for k in g
                        if k.startswith(name + "/") and k.endswith("/steps")})
    # all 3 * n_tensors + 12 keys of store_metrics (inference.py:262-294), the per-parameter ones included
    assert sorted(got) == want_keys
    assert sum(k.startswith("preconditioner/") for k in want_keys) == len(runner.param_names)
    for k in want_keys:
        s, v = got[k]
        gs, gv = g[f"{name}/{k}/steps"], g[f"{name}/{k}/values"]
        assert np.array_equal(s, gs), (name, k, s, gs)          # step indices: bit-exact
        if k in RC.STREAMS_EXACT:
            assert np.array_equal(v, gv), (name, k, v, gv)      # flags, lr, temperature: bit-exact
        else:
            fin = np.isfinite(gv)
            assert np.array_equal(np.isfinite(v), fin), (name, k)
            at = de_atol if k in ("delta_energy", "total_energy") else atol
            rt = rtol
            if k in ("acc", "test/acc"):
                at = max(at, acc_atol)    # discrete: one near-tie argmax flips 1/128 (1/256)
            if k.startswith("est_config_temp/") and k != "est_config_temp/all":
                # (theta . g) N / d of ONE tensor: a sum of d signed terms that cancels to a small multiple of
                # sqrt(d) of them (biases: d = 10..64), scaled by N -- gradient rounding of the other device's
                # reduction order shows up relative to the terms' magnitude, not to the cancelled result
                rt, at = cfg_rtol, max(at, cfg_atol * max(1.0, float(np.abs(gv[fin]).max())))
            np.testing.assert_allclose(v[fin], gv[fin], rtol=rt, atol=at, err_msg=f"{name}:{k}")
    samples = runner.get_samples()
    first = next(iter(k for k in samples if k.endswith("weight_prior.p")))
    np.testing.assert_allclose(samples[first].reshape(samples[first].shape[0], -1)[:, :16].cpu().numpy(),
                               g[f"{name}/sample_first_weight"], rtol=rtol * 10, atol=atol)
    # posterior-predictive ensemble of the stored samples vs the reference's evaluate_model
    from bnn_priors_amd.evaluation import evaluate_model
    dev = runner._device
    ev = evaluate_model(runner.model, runner.dataloader_test, {k: v.to(dev) for k, v in samples.items()})
    got = np.array([ev["lp_ensemble"], ev["lp_last"], ev["acc_ensemble"], ev["acc_last"]])
    np.testing.assert_allclose(got, g[f"{name}/evaluate_model"], rtol=max(rtol, 1e-5),
                               atol=max(atol, acc_atol), err_msg=f"{name}:evaluate_model")


# ------------------------------------------------------------------ CPU: host logic
def _with_oracle_sampler(base):
    from oracle.noise import NoiseSource
    from oracle.samplers import RefHMC, RefSGLD, RefVerletSGLD

    class HostLogicOnly(base):
        def _make_optimizer(self, params):
            noise = NoiseSource(RC.SEED, [p.numel() for p in params])
            name = base.__name__
            if "HMC" in name:
                opt = RefHMC(params, lr=self.learning_rate, num_data=self.eff_num_data, noise=noise)
                opt.is_hmc = True
            else:
                cls = RefSGLD if name.startswith("SGLD") else RefVerletSGLD
                opt = cls(params, lr=self.learning_rate, num_data=self.eff_num_data,
                          momentum=self.momentum, temperature=self.temperature, noise=noise)
            return opt

        def _model_potential_and_grad(self, x, y, want_metrics=True):
            out = super()._model_potential_and_grad(x, y, want_metrics)
            for p in self.optimizer.param_groups[0]["params"]:     # inference.py:219-220
                p.grad.clamp_(min=-self.grad_max, max=self.grad_max)
            return out

        def _check_finite(self):
            pass
    return HostLogicOnly


@pytest.mark.parametrize("name", sorted(RC.CASES))
def test_runner_host_logic_matches_reference_goldens(name):
    cfg = RC.CASES[name]
    train, test, (x, y) = RC.make_data(cfg=cfg)
    model = RC.make_net(models, x, y, cfg=cfg)
    metrics = MemoryMetrics()
    torch.manual_seed(RC.SEED)
    runner = _with_oracle_sampler(_runner_class(name))(
        model=model, dataloader=train, dataloader_test=test, learning_rate=cfg["lr"],
        temperature=cfg["temperature"], momentum=cfg["momentum"],
        reject_samples=cfg["reject_samples"], metrics_saver=metrics, model_saver=None,
        **RC.run_kw(cfg), **({"cycle_seed": RC.CYCLE_SEED} if "Reject" in name else {}))
    runner.run()
    # same torch-CPU ops, same noise: near bit equality on the host that generated the goldens; float32 CPU kernels
    # round differently on another instruction set (AVX-512 vs AVX2), so the tolerance is one that holds across hosts
    # (per-tensor configurational temperatures are sums that cancel: up to 1.5e-2 between an AVX2 and an AVX-512 host
    # after two epochs -- the allowance of the GPU test below)
    _check(name, metrics, runner, rtol=1e-3, atol=1e-4, de_atol=5e-2, cfg_rtol=0.1, cfg_atol=0.05)


# ------------------------------------------------------------------ GPU: end to end
@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(RC.CASES))
def test_runner_on_gpu_matches_reference_goldens(name):
    cfg = RC.CASES[name]
    dev = "cuda:0"
    train, test, (x, y) = RC.make_data(dev, cfg=cfg)
    model = RC.make_net(models, x, y, device=dev, cfg=cfg)
    metrics = MemoryMetrics()
    torch.manual_seed(RC.SEED)
    runner = _runner_class(name)(
        model=model, dataloader=train, dataloader_test=test, learning_rate=cfg["lr"],
        temperature=cfg["temperature"], momentum=cfg["momentum"],
        reject_samples=cfg["reject_samples"], metrics_saver=metrics, model_saver=None,
        seed=RC.SEED, chain_id=0, **RC.run_kw(cfg),
        **({"cycle_seed": RC.CYCLE_SEED} if "Reject" in name else {}))
    runner.run()
    # accept/reject flags, step indices, lr and temperature streams are compared exactly
    # inside _check.  Floats: N = 1024 multiplies one fp32 ulp of the potential (U ~ 60) into
    # ~8e-3 of delta_energy (inherent to the reference's formula); the
    # GPU's GEMM / reduction order differs from the CPU's, so allow a few of those.
    _check(name, metrics, runner, rtol=2e-3, atol=2e-4, de_atol=0.5, acc_atol=2.5 / 128,
           # per-tensor configurational temperatures at model scale: the ResNet's gradient moves by per cents under
           # one-ulp changes, and a BatchNorm scale has 16 elements to average over
           cfg_rtol=0.1, cfg_atol=0.05)


@pytest.mark.gpu
def test_graph_replay_is_bit_identical_to_eager():
    "hipGraph replay of ordinary steps changes nothing: same metric streams, same samples, same bits"
    outs = []
    for use_graph in (False, True):
        cfg = RC.CASES["VerletSGLDReject"]
        dev = "cuda:0"
        train, test, (x, y) = RC.make_data(dev)
        model = RC.make_net(models, x, y, device=dev)
        metrics = MemoryMetrics()
        torch.manual_seed(RC.SEED)
        runner = inference_reject.VerletSGLDRunnerReject(
            model=model, dataloader=train, dataloader_test=test, learning_rate=cfg["lr"],
            temperature=cfg["temperature"], momentum=cfg["momentum"], reject_samples=True,
            metrics_saver=metrics, model_saver=None, seed=RC.SEED, chain_id=0,
            cycle_seed=RC.CYCLE_SEED, use_graph=use_graph, **RC.RUN_KW)
        runner._fused = False      # this test is about the generic captured-autograd path
        runner.run()
        assert (runner._graphed not in (None, False)) == use_graph
        outs.append((RC.streams_of(metrics), {k: v.clone() for k, v in runner.get_samples().items()}))
    (s0, p0), (s1, p1) = outs
    assert sorted(s0) == sorted(s1)
    for k in s0:
        if k in ("timestamps",):
            continue
        assert np.array_equal(s0[k][0], s1[k][0]), k
        assert np.array_equal(s0[k][1], s1[k][1]), (k, s0[k][1], s1[k][1])
    for k in p0:
        assert torch.equal(p0[k], p1[k]), k


@pytest.mark.gpu
def test_fused_dense_step_agrees_with_autograd_graph_path():
    """the 3-kernel fused dense step vs the generic captured-autograd path: same flags and step
    indices, floats equal up to fp32 summation order in the gradient"""
    outs = []
    for fused in (False, True):
        cfg = RC.CASES["VerletSGLDReject"]
        dev = "cuda:0"
        train, test, (x, y) = RC.make_data(dev)
        model = RC.make_net(models, x, y, device=dev)
        metrics = MemoryMetrics()
        torch.manual_seed(RC.SEED)
        runner = inference_reject.VerletSGLDRunnerReject(
            model=model, dataloader=train, dataloader_test=test, learning_rate=cfg["lr"],
            temperature=cfg["temperature"], momentum=cfg["momentum"], reject_samples=True,
            metrics_saver=metrics, model_saver=None, seed=RC.SEED, chain_id=0,
            cycle_seed=RC.CYCLE_SEED, **RC.RUN_KW)
        if not fused:
            runner._fused = False
        runner.run()
        assert (runner._fused not in (None, False)) == fused
        outs.append((RC.streams_of(metrics), {k: v.clone() for k, v in runner.get_samples().items()}))
    (s0, p0), (s1, p1) = outs
    assert sorted(s0) == sorted(s1)
    for k in s0:
        assert np.array_equal(s0[k][0], s1[k][0]), k
        if k in RC.STREAMS_EXACT:
            assert np.array_equal(s0[k][1], s1[k][1]), k
        else:
            at = 0.05 if k in ("delta_energy", "total_energy") else (2.5 / 128 if "acc" in k else 1e-4)
            rt = 1e-4
            if k.startswith("est_config_temp/"):
                # (theta . g) N / d cancels heavily for small tensors: ulp-level differences of g
                # (summation order of the fused kernel vs rocBLAS) show up at the 1e-3 level
                rt, at = 1e-2, 2e-3
            np.testing.assert_allclose(s0[k][1], s1[k][1], rtol=rt, atol=at, err_msg=k)
    for k in p0:
        torch.testing.assert_close(p0[k], p1[k], rtol=1e-3, atol=1e-5)


@pytest.mark.gpu
def test_dense_step_direct_and_graph_replica_modes_are_bit_identical(monkeypatch):
    "three by-value launches vs replaying the captured per-slot graph replicas: same kernels, same bits"
    outs = []
    for direct in ("1", "0"):
        monkeypatch.setenv("SGMCMC_DENSE_DIRECT", direct)
        cfg = RC.CASES["VerletSGLDReject"]
        dev = "cuda:0"
        train, test, (x, y) = RC.make_data(dev)
        model = RC.make_net(models, x, y, device=dev)
        metrics = MemoryMetrics()
        torch.manual_seed(RC.SEED)
        runner = inference_reject.VerletSGLDRunnerReject(
            model=model, dataloader=train, dataloader_test=test, learning_rate=cfg["lr"],
            temperature=cfg["temperature"], momentum=cfg["momentum"], reject_samples=True,
            metrics_saver=metrics, model_saver=None, seed=RC.SEED, chain_id=0,
            cycle_seed=RC.CYCLE_SEED, **RC.RUN_KW)
        runner.run()
        assert runner._fused.direct == (direct == "1")
        outs.append((RC.streams_of(metrics), {k: v.clone() for k, v in runner.get_samples().items()}))
    (s0, p0), (s1, p1) = outs
    for k in s0:
        assert np.array_equal(s0[k][0], s1[k][0]) and np.array_equal(s0[k][1], s1[k][1]), k
    for k in p0:
        assert torch.equal(p0[k], p1[k]), k


# ------------------------------------------------------------------ BASELINE configs[4]: L-step trajectories, T != 1
def _trajectory_runner(base, T, L, device="cpu"):
    cfg = RC.CASES["HMCReject"]
    train, test, (x, y) = RC.make_data(device, cfg=cfg)
    model = RC.make_net(models, x, y, device=device, cfg=cfg)
    metrics = MemoryMetrics()
    torch.manual_seed(RC.SEED)
    runner = base(model=model, dataloader=train, dataloader_test=test, learning_rate=cfg["lr"] * T,
                  temperature=T, momentum=1.0, reject_samples=True, metrics_saver=metrics, model_saver=None,
                  trajectory_length=L, tempered=(T != 1.0), cycle_seed=RC.CYCLE_SEED, **RC.RUN_KW)
    return runner, metrics


def _check_trajectory_streams(metrics, runner, L):
    got = RC.streams_of(metrics)
    steps, rej = got["acceptance/rejected"]
    _, is_sample = got["acceptance/is_sample"]
    s_all, _ = got["acceptance/is_sample"]
    # 2 cycles x 2 epochs x 8 minibatches, L = 3: inside a cycle a trajectory ends after leapfrog steps 3, 6, 9, 12, 15
    # (the count runs across the non-sampling epoch's end) and at the end of the sampling epoch (step 16), where the
    # count restarts; every M-H point consumes a step index (quirk 6): rows at 0 | 4 8 12 16 20 22 | 26 30 34 38 42 44
    assert steps.tolist() == [0, 4, 8, 12, 16, 20, 22, 26, 30, 34, 38, 42, 44] and set(rej.tolist()) <= {0, 1}
    sample_rows = s_all[is_sample == 1]
    assert len(sample_rows) == 1 + 2                       # the begin() row + one per sampling epoch
    samples = runner.get_samples()
    first = next(iter(k for k in samples if k.endswith("weight_prior.p")))
    assert samples[first].shape[0] == 2
    de_s, de = got["delta_energy"]
    assert np.isfinite(de).all()


def test_hmc_trajectory_length_host_logic():
    "HMCRunnerReject(trajectory_length=3): the M-H schedule, with the oracle's sampler plugged in (CPU)"
    runner, metrics = _trajectory_runner(_with_oracle_sampler(inference_reject.HMCRunnerReject), 1.0, 3)
    runner.run()
    _check_trajectory_streams(metrics, runner, 3)


def test_hmc_trajectory_ending_on_a_warmup_epochs_last_step_and_the_lr_schedule():
    """L = 4 divides the 8 minibatches of an epoch: a trajectory ends exactly on the last step of the warm-up epoch
    (no sample is stored there, so the M-H point is the trajectory's own) -- every trajectory has exactly L steps --
    and the learning rate advances once per minibatch whatever L is: the lr logged at a stored sample equals the lr
    of a run without intra-epoch trajectories."""
    cls = _with_oracle_sampler(inference_reject.HMCRunnerReject)
    runner, metrics = _trajectory_runner(cls, 1.0, 4)
    runner.run()
    got = RC.streams_of(metrics)
    steps, _ = got["acceptance/rejected"]
    # leapfrog 1-4 | M-H 5 | 6-9 (9 = last of the warm-up epoch) | M-H 10 | 11-14 | M-H 15 | 16-19 | sample 20 | ...
    assert steps.tolist() == [0, 5, 10, 15, 20, 25, 30, 35, 40]
    s_all, is_sample = got["acceptance/is_sample"]
    assert s_all[is_sample == 1].tolist() == [0, 20, 40]
    plain, plain_metrics = _trajectory_runner(cls, 1.0, None)
    plain.run()
    ref = RC.streams_of(plain_metrics)
    lr = dict(zip(*got["lr"]))
    lr_ref = dict(zip(*ref["lr"]))
    ref_samples = ref["acceptance/is_sample"][0][ref["acceptance/is_sample"][1] == 1]
    assert ref_samples.tolist() == [0, 17, 34]
    for a, b in zip([0, 20, 40], ref_samples.tolist()):
        assert lr[a] == lr_ref[b], (a, b, lr[a], lr_ref[b])      # bit-exact host doubles
    # and the scheduler has advanced exactly once per minibatch + once at construction
    assert runner.scheduler.last_epoch == plain.scheduler.last_epoch == 2 * 2 * 8


@pytest.mark.gpu
@pytest.mark.parametrize("T", [1.0, 0.1])
def test_hmc_trajectories_and_temperature_on_gpu(T):
    """the extension end to end on the HIP path: L-step trajectories, tempered acceptance; the kinetic
    temperature logged right after each momentum refresh estimates T"""
    runner, metrics = _trajectory_runner(inference_reject.HMCRunnerReject, T, 3, device="cuda:0")
    runner.run()
    _check_trajectory_streams(metrics, runner, 3)
    assert runner.optimizer.param_groups[0]["temperature"] == T
    with pytest.raises(AssertionError):                    # without tempered=True the reference's assertion stands
        r2, _ = _trajectory_runner(inference_reject.HMCRunnerReject, 0.5, 3, device="cuda:0")
        r2.tempered = False
        r2.run()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["VerletSGLDReject_googleresnet", "VerletSGLDReject_convnet_laplace", "HMCReject"])
def test_chains_interleaved_on_streams_equal_chains_run_alone(name):
    """multichain.run_on_streams: FOUR chains (chain_id 0 .. 3), each on its own HIP stream of the one GPU (streams chosen
    so that no two share a hardware queue: multichain.concurrent_streams), advanced step by
    step from one process -- every chain's metric streams and samples are bit-identical to the same chain's ``run()``.
    (As in the reference, the INITIAL full-data pass iterates the shuffling loader before the per-cycle seed exists, i.e.
    it draws its order from torch's global generator -- and BatchNorm makes the potential depend on the batches'
    composition -- so each chain re-seeds the global generator where it begins, in both modes.)"""
    from bnn_priors_amd import multichain
    cfg = RC.CASES[name]
    dev = "cuda:0"

    def make(chain):
        train, test, (x, y) = RC.make_data(dev, cfg=cfg)
        model = RC.make_net(models, x, y, device=dev, cfg=cfg)
        metrics = MemoryMetrics()
        runner = _runner_class(name)(
            model=model, dataloader=train, dataloader_test=test, learning_rate=cfg["lr"],
            temperature=cfg["temperature"], momentum=cfg["momentum"], reject_samples=cfg["reject_samples"],
            metrics_saver=metrics, model_saver=None, seed=RC.SEED, chain_id=chain, cycle_seed=RC.CYCLE_SEED,
            **RC.RUN_KW)
        begin = runner.begin
        runner.begin = lambda: (torch.manual_seed(RC.SEED + chain), begin())[1]
        return runner, metrics

    chains = (0, 1, 2, 3)
    alone = []
    for chain in chains:
        runner, metrics = make(chain)
        runner.run()
        alone.append((RC.streams_of(metrics), {k: v.clone() for k, v in runner.get_samples().items()}))
    pairs = [make(chain) for chain in chains]
    streams = multichain.concurrent_streams(len(chains), dev)
    assert len(streams) >= 3, "this process has fewer than three hardware queues for its streams"
    multichain.run_on_streams([r for r, _ in pairs])
    for chain, (runner, metrics) in enumerate(pairs):
        s1, p1 = RC.streams_of(metrics), runner.get_samples()
        s0, p0 = alone[chain]
        assert sorted(s0) == sorted(s1)
        for k in s0:
            if k == "timestamps":
                continue
            assert np.array_equal(s0[k][0], s1[k][0]) and np.array_equal(s0[k][1], s1[k][1]), (chain, k)
        for k in p0:
            assert torch.equal(p0[k], p1[k]), (chain, k)
    assert not np.array_equal(alone[0][0]["potential"][1], alone[1][0]["potential"][1])     # the chains do differ


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["VerletSGLDReject_googleresnet", "VerletSGLDReject_convnet_laplace", "VerletSGLD"])
def test_lazy_batches_in_the_captured_step_change_nothing(name, monkeypatch):
    """minibatches gathered inside the captured step's one staging launch (inference.LazyBatch, the default) against
    minibatches gathered by the batch source and copied in: same metric streams, same samples, same bits"""
    outs = []
    for lazy in (True, False):
        monkeypatch.setattr(inference, "LAZY_BATCHES", lazy)
        cfg = RC.CASES[name]
        dev = "cuda:0"
        train, test, (x, y) = RC.make_data(dev, cfg=cfg)
        model = RC.make_net(models, x, y, device=dev, cfg=cfg)
        metrics = MemoryMetrics()
        torch.manual_seed(RC.SEED)
        runner = _runner_class(name)(
            model=model, dataloader=train, dataloader_test=test, learning_rate=cfg["lr"],
            temperature=cfg["temperature"], momentum=cfg["momentum"], reject_samples=cfg["reject_samples"],
            metrics_saver=metrics, model_saver=None, seed=RC.SEED, chain_id=0, **RC.run_kw(cfg),
            **({"cycle_seed": RC.CYCLE_SEED} if "Reject" in name else {}))
        runner._fused = False          # (the dense classifier: the generic captured-autograd path)
        runner.run()
        assert runner._graphed not in (None, False)
        outs.append((RC.streams_of(metrics), {k: v.clone() for k, v in runner.get_samples().items()}))
    (s0, p0), (s1, p1) = outs
    assert sorted(s0) == sorted(s1)
    for k in s0:
        if k == "timestamps":
            continue
        assert np.array_equal(s0[k][0], s1[k][0]) and np.array_equal(s0[k][1], s1[k][1]), k
    for k in p0:
        assert torch.equal(p0[k], p1[k]), k
