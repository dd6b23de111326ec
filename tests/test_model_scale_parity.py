"""Model-scale parity of the BASELINE configurations that no reference golden can reach directly.

* BASELINE configs[4]: ``HMCRunnerReject(trajectory_length=50, tempered=True)`` on googleresnet with the student-t
  weight prior at T in {1, 0.1, 0.01} (extends mcmc/hmc.py:25-79, inference_reject.py:115-157,182-189);
* BASELINE configs[3] at its real batch size: ``VerletSGLDRunnerReject`` on googleresnet, batch 128.

Both run the product runner (captured-graph steps, graph-captured exact passes) with ``oracle.samplers`` in lock-step
(tests/shadow.py): every transition against the oracle's on the same inputs, the trajectory's energy bookkeeping
accumulated independently by the oracle, M-H step indices and accept / reject flags identical.
"""
import numpy as np
import pytest
import torch

import runner_cases as RC
from shadow import check_mh_points, shadowed
from bnn_priors_amd import inference_reject, models
from bnn_priors_amd.storage import MemoryMetrics

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _run(base, kind, cfg, run_kw, **extra):
    train, test, (x, y) = RC.make_data(DEV, cfg=cfg)
    model = RC.make_net(models, x, y, device=DEV, cfg=cfg)
    metrics = MemoryMetrics()
    torch.manual_seed(RC.SEED)
    runner = shadowed(base, kind)(
        model=model, dataloader=train, dataloader_test=test, learning_rate=cfg["lr"], temperature=cfg["temperature"],
        momentum=cfg["momentum"], reject_samples=True, metrics_saver=metrics, model_saver=None, seed=RC.SEED,
        chain_id=0, cycle_seed=RC.CYCLE_SEED, **run_kw, **extra)
    runner.run()
    return runner, RC.streams_of(metrics)


# 256 examples, batch 32: 8 minibatches per epoch; 2 cycles x (6 warm-up + 1 sampling) epochs = 112 leapfrog steps.
# L = 50: inside a cycle the trajectories have 50 steps (ends inside epoch 6, no sample), then 6 steps up to the
# sampling epoch's end (sample stored, count restarts) -- M-H rows at step indices 51, 58 | 109, 116.
HMC_KW = dict(epochs_per_cycle=7, warmup_epochs=6, sample_epochs=1, skip=1, metrics_skip=10, cycles=2,
              precond_update=1, sampling_decay="cosine")


@pytest.mark.parametrize("T", [1.0, 0.1, 0.01])
def test_config5_hmc_L50_student_t_googleresnet_against_the_oracle(T):
    cfg = dict(RC.CASES["HMCReject_googleresnet_studentt"], n=256, batch=32, temperature=T,
               lr=RC.CASES["HMCReject_googleresnet_studentt"]["lr"] * T)
    runner, got = _run(inference_reject.HMCRunnerReject, "hmc", cfg, HMC_KW, trajectory_length=50,
                       tempered=(T != 1.0))
    log = runner.shadow.log
    assert runner._graphed not in (None, False), "ordinary steps must have gone through the captured graph"
    assert log["steps"] == 112 and log["refresh"] == 1 + 4
    steps, rej = got["acceptance/rejected"]
    assert steps.tolist() == [0, 51, 58, 109, 116]                     # M-H step indices (quirk 6: each consumes one)
    s_all, is_sample = got["acceptance/is_sample"]
    assert s_all[is_sample == 1].tolist() == [0, 58, 116]
    check_mh_points(log, 4)
    assert rej[1:].tolist() == [int(r["rejected"]) for r in log["mh"]]
    # the weight prior of the final Linear really is Student-t, and every other prior Gaussian (exp_utils.py:186-190)
    from bnn_priors_amd import _hip
    kinds = [int(k) for k in runner.optimizer.engine.seg_host["prior_kind"]]
    assert kinds.count(_hip.PRIOR_STUDENT_T) == 1 and set(kinds) <= {_hip.PRIOR_NONE, _hip.PRIOR_NORMAL,
                                                                      _hip.PRIOR_STUDENT_T}
    # kinetic temperature right after a momentum refresh estimates T (d = 272,474: within a per cent)
    t_steps, t_all = got["est_temperature/all"]
    assert abs(t_all[t_steps == 0][0] / T - 1) < 0.02, (t_all[:3], T)
    print("config5 T =", T, [{k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()} for r in log["mh"]],
          log["max_err"])


def test_config3_googleresnet_batch128_trajectory_against_the_oracle():
    "VerletSGLDReject, batch 128 (the north-star shape): 2 cycles x 2 epochs x 3 minibatches, M-H at both samples"
    cfg = dict(RC.CASES["VerletSGLDReject_googleresnet"], n=384, batch=128)
    runner, got = _run(inference_reject.VerletSGLDRunnerReject, "verlet", cfg, RC.RUN_KW)
    log = runner.shadow.log
    assert runner._graphed not in (None, False)
    assert log["steps"] == 12
    steps, rej = got["acceptance/rejected"]
    assert steps.tolist() == [0, 7, 14]
    check_mh_points(log, 2)
    assert rej[1:].tolist() == [int(r["rejected"]) for r in log["mh"]]
    print("config3 batch 128", log["mh"], log["max_err"])
