"""Pins ``oracle/runner.py::cpu_leapfrog`` -- the loop ``bench.py`` times as ``cpu_baseline`` -- to the
reference's own loop body (bnn_priors/inference_reject.py:86-113, inference.py:215-223):

* ``test_cpu_leapfrog_matches_reference_fixture`` (runs everywhere): K steps of ``cpu_leapfrog`` on the oracle's own
  plain-torch network (oracle/nets.py; a reference-shaped model only supplies the initial values) reproduce the parameter vector / potentials / delta_energy that the IMPORTED REFERENCE
  produced for the same inputs and torch seed (tests/golden/cpu_loop.npz, make_cpu_loop_golden.py);
* ``test_cpu_leapfrog_matches_imported_reference`` (needs /root/reference): the same comparison live,
  statement for statement, bit for bit.
"""
import os

import numpy as np
import pytest
import torch

K_STEPS = 25
LOOP_KW = dict(num_data=60000.0, lr=0.01, momentum=0.994, temperature=1.0, steps_per_cycle=469 * 50,
               metrics_skip=10, seed=4321)


def loop_inputs(factory):
    "16 synthetic MNIST-shaped minibatches and a constructor of the case's network from ``factory.get_model``"
    g = torch.Generator().manual_seed(99)
    batches = [(torch.rand(128, 784, generator=g), torch.randint(0, 10, (128,), generator=g)) for _ in range(16)]

    def make():
        torch.manual_seed(0)
        kw = dict(width=50, depth=3, weight_prior="gaussian", weight_loc=0., weight_scale=2 ** .5,
                  bias_prior="gaussian", bias_loc=0., bias_scale=1., batchnorm=True, weight_prior_params={},
                  bias_prior_params={})
        net = factory.get_model(batches[0][0], batches[0][1], "classificationdensenet", **kw)
        torch.manual_seed(1)
        factory.he_initialize(net)
        return net
    return batches, make


def _flat(model):
    return torch.cat([p.detach().reshape(-1) for p in model.parameters()]).double().numpy().copy()


def reference_loop(model, batches, k, *, num_data, lr, momentum, temperature, steps_per_cycle, metrics_skip, seed):
    """the reference's objects driven through its own statements (needs the reference importable)"""
    import bnn_priors.inference_reject as ref_rej
    from bnn_priors.utils import get_cosine_schedule

    class _Set(torch.utils.data.Dataset):
        def __len__(self):
            return int(num_data)
    loader = torch.utils.data.DataLoader(_Set(), batch_size=128)
    runner = ref_rej.VerletSGLDRunnerReject(
        model=model, dataloader=loader, dataloader_test=loader, epochs_per_cycle=50, warmup_epochs=45,
        sample_epochs=5, learning_rate=lr, skip=1, metrics_skip=metrics_skip, temperature=temperature,
        momentum=momentum, sampling_decay="cosine", cycles=1, precond_update=1, metrics_saver=None,
        model_saver=None, reject_samples=True)
    torch.manual_seed(seed)
    opt = runner.optimizer = runner._make_optimizer(list(model.parameters()))
    sched = torch.optim.lr_scheduler.LambdaLR(opt, get_cosine_schedule(steps_per_cycle))
    x, y = batches[0]
    _, _, potential0, _ = runner._model_potential_and_grad(x, y)
    opt.sample_momentum()
    opt.initial_step(calc_metrics=True, save_state=True)
    u0 = potential0.item()
    potentials, des = [], []
    for step in range(1, k + 1):
        x, y = batches[step % len(batches)]
        store = (step % metrics_skip) == 0
        loss, log_prior, potential, acc = runner._model_potential_and_grad(x, y)   # inference_reject.py:91
        opt.step(calc_metrics=store)                                                # :94
        potentials.append(potential.item())
        if store:
            des.append(opt.delta_energy(u0, potential))                             # :98
        sched.step()                                                                # :112
    return _flat(model), potentials, des


def oracle_loop(model, batches, k, *, num_data, lr, momentum, temperature, steps_per_cycle, metrics_skip, seed):
    """the same K steps through oracle/runner.py ON oracle/nets.py's restatement of the network (the thing bench.py
    times): ``model`` only supplies the initial parameter values"""
    from oracle import nets as N
    from oracle import runner as R
    from oracle.runner import get_cosine_schedule
    from oracle.samplers import RefVerletSGLD
    model = N.load_parameters(N.densenet(), model)
    torch.manual_seed(seed)
    opt = RefVerletSGLD(list(model.parameters()), lr=lr, num_data=num_data, momentum=momentum,
                        temperature=temperature)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, get_cosine_schedule(steps_per_cycle))
    x, y = batches[0]
    opt.zero_grad()
    potential0 = model.split_potential_and_acc(x, y, num_data)[2]
    potential0.backward()
    for p in opt.param_groups[0]["params"]:
        p.grad.clamp_(min=-1e6, max=1e6)
    opt.sample_momentum()
    opt.initial_step(calc_metrics=True, save_state=True)
    u0 = potential0.item()
    potentials, des = [], []
    for step in range(1, k + 1):
        x, y = batches[step % len(batches)]
        out = R.cpu_leapfrog(model, opt, sched, x, y, step, num_data, metrics_skip=metrics_skip,
                             initial_potential=u0)
        potentials.append(out["potential"].item())
        if out["delta_energy"] is not None:
            des.append(out["delta_energy"])
    return _flat(model), potentials, des


def test_cpu_leapfrog_matches_reference_fixture(golden_dir):
    from bnn_priors_amd import models
    old = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        batches, make = loop_inputs(models)
        theta, potentials, des = oracle_loop(make(), batches, K_STEPS, **LOOP_KW)
    finally:
        torch.set_num_threads(old)
    z = np.load(os.path.join(golden_dir, "cpu_loop.npz"))
    # same torch ops in the same order under the same seed: differences can only come from the BLAS build
    np.testing.assert_allclose(theta, z["theta"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(potentials, z["potentials"], rtol=2e-6)
    np.testing.assert_allclose(des, z["delta_energy"], rtol=1e-4, atol=1e-3)
    assert len(des) == K_STEPS // LOOP_KW["metrics_skip"]


@pytest.mark.reference
def test_cpu_leapfrog_matches_imported_reference():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import ref_stubs
    ref_stubs.install()
    import bnn_priors.exp_utils as ref_exp
    from bnn_priors_amd import models
    old = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        batches, make_ref = loop_inputs(ref_exp)
        _, make_own = loop_inputs(models)
        t_ref, p_ref, d_ref = reference_loop(make_ref(), batches, K_STEPS, **LOOP_KW)
        t_own, p_own, d_own = oracle_loop(make_own(), batches, K_STEPS, **LOOP_KW)
    finally:
        torch.set_num_threads(old)
    assert np.array_equal(t_ref, t_own)                     # bit for bit: identical op sequence
    assert p_ref == p_own
    np.testing.assert_allclose(d_own, d_ref, rtol=1e-12)


@pytest.mark.parametrize("name,xshape,prior", [("classificationdensenet", (784,), "gaussian"),
                                               ("classificationconvnet", (784,), "laplace"),
                                               ("googleresnet", (3, 32, 32), "student-t")])
def test_oracle_nets_agree_with_the_reference_shaped_models(name, xshape, prior):
    """oracle/nets.py against the models whose goldens are pinned to the reference (same state, CPU): identical
    parameter lists, and loss / log-prior / potential / accuracy of one batch equal to fp32 rounding"""
    from bnn_priors_amd import models
    from oracle import nets as N
    g = torch.Generator().manual_seed(3)
    x, y = torch.randn((16,) + xshape, generator=g), torch.randint(0, 10, (16,), generator=g)
    torch.manual_seed(0)
    ref = models.get_model(x[:2], torch.tensor([0, 9]), name, width=50, depth=3, weight_prior=prior,
                           weight_scale=2 ** .5, bias_prior="gaussian", bias_scale=1.)
    models.he_initialize(ref)
    own = N.load_parameters(N.BUILDERS[name](weight_prior=prior), ref)
    # (the reference-shaped models keep nn.DataParallel's "module." level in their names; the oracle's do not)
    assert [n for n, _ in own.named_parameters()] == [n.replace("net.module.", "net.") for n, _ in ref.named_parameters()]
    a = ref.split_potential_and_acc(x, y, 50000.)
    b = own.split_potential_and_acc(x, y, 50000.)
    for u, v, what in zip(a[:4], b[:4], ("loss", "log_prior", "potential", "acc")):
        torch.testing.assert_close(u, v, rtol=1e-5, atol=1e-6, msg=what)
