"""North-star acceptance criterion: "accept-rate / kinetic-temperature diagnostics matching the
reference within 1 %".  The oracle's samplers are bit-identical to the imported reference
(tests/test_oracle_goldens.py), so the reference side is the oracle runner on the CPU; the other
side is the HIP product.  Same synthetic data, same Philox key, a run long enough for
sampling-phase averages (trajectories decorrelate through fp differences, statistics must not)."""
import numpy as np
import pytest
import torch

import runner_cases as RC
from bnn_priors_amd import inference_reject, models
from bnn_priors_amd.storage import MemoryMetrics

pytestmark = pytest.mark.gpu

KW = dict(epochs_per_cycle=6, warmup_epochs=3, sample_epochs=3, skip=1, metrics_skip=4, cycles=6,
          precond_update=1, sampling_decay="cosine")


def _run(device, use_hip):
    from test_runners import _with_oracle_sampler
    train, test, (x, y) = RC.make_data(device)
    model = RC.make_net(models, x, y, device=device)
    metrics = MemoryMetrics()
    torch.manual_seed(RC.SEED)
    cls = inference_reject.VerletSGLDRunnerReject
    if not use_hip:
        cls = _with_oracle_sampler(cls)
    runner = cls(model=model, dataloader=train, dataloader_test=test, learning_rate=0.002,
                 temperature=1.0, momentum=0.98, reject_samples=True, metrics_saver=metrics,
                 model_saver=None, cycle_seed=RC.CYCLE_SEED,
                 **({"seed": RC.SEED, "chain_id": 0} if use_hip else {}), **KW)
    runner.run()
    return RC.streams_of(metrics)


def test_temperature_and_acceptance_diagnostics_within_one_percent():
    ref = _run("cpu", use_hip=False)
    hip = _run("cuda:0", use_hip=True)
    # identical bookkeeping streams
    for k in ("acceptance/is_sample", "lr", "temperature"):
        assert np.array_equal(ref[k][0], hip[k][0]) and np.array_equal(ref[k][1], hip[k][1]), k
    n = len(ref["est_temperature/all"][1])
    late = slice(n // 3, None)    # after the first two cycles
    for k in ("est_temperature/all", "est_config_temp/all"):
        a, b = ref[k][1][late].mean(), hip[k][1][late].mean()
        assert abs(a - b) <= 0.01 * abs(a), (k, a, b)
    # acceptance: mean min(1, exp(-dE/T)) over the M-H points and the reject decisions themselves
    is_s = ref["acceptance/is_sample"][1] > 0
    steps_s = ref["acceptance/is_sample"][0][is_s]

    def at(stream, steps):
        s, v = stream
        return v[np.isin(s, steps)]
    de_r, de_h = at(ref["delta_energy"], steps_s)[1:], at(hip["delta_energy"], steps_s)[1:]
    acc_r = np.minimum(1.0, np.exp(-de_r)).mean()
    acc_h = np.minimum(1.0, np.exp(-de_h)).mean()
    assert abs(acc_r - acc_h) <= 0.01 * max(acc_r, 1e-12) + 1e-3, (acc_r, acc_h)
    rej_r, rej_h = ref["acceptance/rejected"][1], hip["acceptance/rejected"][1]
    assert abs(rej_r.mean() - rej_h.mean()) <= 0.01 + 1.0 / len(rej_r), (rej_r, rej_h)
