"""The reference's own six hot-path tests, restated against the HIP samplers
(testing/test_sgld.py:13-80, testing/test_verlet_sgld.py:58-211,
testing/test_hmc.py:17-135): same targets, hyper-parameters, call sequences and
thresholds; written for this code base, running on the MI355X."""
import math

import numpy as np
import pytest
import scipy.stats
import torch

from helpers import default_dtype

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _imports():
    from bnn_priors_amd import mcmc, models
    return mcmc, models


def _dot(a, b):
    return (a.reshape(-1).double() @ b.reshape(-1).double()).item()


def _snapshot(opt, with_grad=False):
    out = []
    for p, st in opt.state.items():
        row = [p.detach().clone()]
        if with_grad:
            row.append(p.grad.detach().clone())
        row.append(st['momentum_buffer'].detach().clone())
        out.append(row)
    return list(zip(*out))


def _allclose_all(xs, ys):
    return [torch.allclose(a, b) for a, b in zip(xs, ys)]


def test_sgd_equivalence():
    "SGLD(T=0, momentum=.9, num_data=1) == torch.optim.SGD(momentum=.9)  (test_sgld.py:61-80)"
    mcmc, models = _imports()
    model = models.GaussianModel(N=1, D=5, mean=0.5, std=0.25).to(DEV)
    lr, momentum = 1.25, 0.9
    sgld = mcmc.SGLD(model.parameters(), lr=lr, num_data=1, momentum=momentum, temperature=0.)
    sgld.sample_momentum()
    sgd = torch.optim.SGD(model.parameters(), lr=lr, momentum=momentum)
    initial = {k: v.detach().clone() for k, v in model.state_dict().items()}
    for _ in range(4):
        sgld.step(model.potential_avg_closure)
    after_sgld = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.load_state_dict(initial)
    for _ in range(4):
        sgd.step(model.potential_avg_closure)
    for k, v in model.state_dict().items():
        assert torch.allclose(after_sgld[k], v), k


def test_hmc_reversible():
    "leapfrog forward, negate momenta, leapfrog back (test_hmc.py:17-65), float64"
    mcmc, models = _imports()
    with default_dtype(torch.float64):
        torch.manual_seed(3)
        N = 10
        x = torch.randn(N, 1, device=DEV)
        y = x.sin()
        model = models.DenseNet(1, 1, 10, noise_std=0.1).to(DEV)

        def loss():
            model.zero_grad()
            v = model.potential_avg(x, y, eff_num_data=1.)
            v.backward()
            return v
        hmc = mcmc.HMC(model.parameters(), lr=0.01, num_data=N, raise_on_nan=True,
                       raise_on_no_grad=True)
        for _, st in hmc.state.items():
            st['preconditioner'] = torch.rand(()).item() + 0.2
        hmc.sample_momentum()
        p0, m0 = _snapshot(hmc)
        hmc.initial_step(loss)
        p1, m_half = _snapshot(hmc)
        hmc.step(loss)
        p2, m_3half = _snapshot(hmc)
        hmc.final_step(loss)
        p2_alt, m2 = _snapshot(hmc)
        assert not any(_allclose_all(p0, p1)) and not any(_allclose_all(p1, p2))
        assert all(_allclose_all(p2, p2_alt))
        assert not any(_allclose_all(m0, m_half)) and not any(_allclose_all(m_half, m_3half))
        assert not any(_allclose_all(m_3half, m2))
        for _, st in hmc.state.items():
            st['momentum_buffer'].neg_()
        hmc.initial_step(loss)
        p1_alt, m_3half_neg = _snapshot(hmc)
        assert all(_allclose_all(p1, p1_alt))
        assert all(_allclose_all(m_3half, [-m for m in m_3half_neg]))
        hmc.step(loss)
        p0_alt, m_half_neg = _snapshot(hmc)
        assert all(_allclose_all(p0, p0_alt))
        assert all(_allclose_all(m_half, [-m for m in m_half_neg]))
        hmc.final_step(loss)
        p0_alt2, m0_neg = _snapshot(hmc)
        assert all(_allclose_all(p0, p0_alt2))
        assert all(_allclose_all(m0, [-m for m in m0_neg]))


def test_verlet_accept_prob_closed_form(n_samples=10):
    """delta_energy == sum_t C (g1.g1 - g0.g0) + sum_steps -1/2 (th1-th0).(g1+g0) + (U1-U0),
    C = lr M^2 / 8   (test_verlet_sgld.py:148-211)"""
    mcmc, models = _imports()
    torch.manual_seed(145)
    model = models.NealFunnelT().to(DEV)
    sgld = mcmc.VerletSGLD(model.parameters(), lr=1 / 32, num_data=1, momentum=127 / 128,
                           temperature=3 / 4)
    time_step_sq = sgld.param_groups[0]['lr']
    model.sample_all_priors()
    preconds = []
    for p in model.parameters():
        st = sgld.state[p]
        st['preconditioner'] = (torch.rand(()).item() + 0.2) / math.sqrt(4)
        preconds.append(st['preconditioner'])
    sgld.sample_momentum()
    states = []
    U0 = model.potential_avg_closure().item()
    states.append(_snapshot(sgld, with_grad=True))
    sgld.initial_step()
    for s in range(1, n_samples):
        model.potential_avg_closure()
        states.append(_snapshot(sgld, with_grad=True))
        sgld.step()
        if s == n_samples - 1:
            U1 = model.potential_avg_closure().item()
            sgld.final_step()
            states.append(_snapshot(sgld, with_grad=True))
    ref = 0.
    _, g0s, _ = states[0]
    _, g1s, _ = states[-1]
    for g0, g1, M in zip(g0s, g1s, preconds):
        ref += (time_step_sq * M ** 2 / 8) * (_dot(g1, g1) - _dot(g0, g0))
    point = 0.
    group = sgld.param_groups[0]
    for g0, g1, p in zip(g0s, g1s, group['params']):
        p.grad = g0
        point -= sgld._point_energy(group, p, sgld.state[p])
        p.grad = g1
        point += sgld._point_energy(group, p, sgld.state[p])
    assert np.allclose(ref, point)
    for i in range(1, len(states)):
        th0, g0s, _ = states[i - 1]
        th1, g1s, _ = states[i]
        for a0, a1, g0, g1 in zip(th0, th1, g0s, g1s):
            ref += -.5 * _dot(a1 - a0, g1 + g0)
    ref += U1 - U0
    got = sgld.delta_energy(U0, U1)
    assert np.allclose(ref, got), f"{ref} != {got}"


def _distribution_checks(opt, n_vars, n_dim, mean, std_eff, temp_scale, check_kinetic=True):
    """The reference's probabilistic criteria (Anderson-Darling normality at 15 %, KS p >= 0.3 for
    the parameters and for the per-tensor temperatures) as pass flags -- a CORRECT sampler passes
    each only 70-85 % of the time -- plus moment checks at 4 standard errors, which are both
    sharper (2.5 % on variance / temperatures) and essentially never fail for a correct sampler."""
    params = np.empty(n_vars * n_dim)
    kin, cfg = np.empty(n_vars), np.empty(n_vars)
    for i, (p, st) in enumerate(opt.state.items()):
        params[i * n_dim:(i + 1) * n_dim] = p.detach().cpu().numpy()
        kin[i], cfg[i] = st['est_temperature'], st['est_config_temp']
    res = scipy.stats.anderson(params, dist='norm')
    assert res.significance_level[0] == 15
    ok = {"normal": res.statistic < res.critical_values[0]}
    ok["variance"] = scipy.stats.ks_1samp(
        params, lambda x: scipy.stats.norm.cdf(x, loc=mean, scale=std_eff), method='exact').pvalue >= 0.3
    chi2 = lambda x: scipy.stats.chi2.cdf(x, df=n_dim, loc=0., scale=temp_scale / n_dim)  # noqa: E731
    ok["config_temp"] = scipy.stats.ks_1samp(cfg, chi2, method='exact').pvalue >= 0.3
    if check_kinetic:
        ok["kinetic_temp"] = scipy.stats.ks_1samp(kin, chi2, method='exact').pvalue >= 0.3
    n = params.size
    assert abs(params.mean() - mean) < 4 * std_eff / math.sqrt(n), ("mean", params.mean())
    assert abs(params.var() / std_eff ** 2 - 1) < 4 * math.sqrt(2 / n), ("variance", params.var())
    t_err = 4 * temp_scale * math.sqrt(2 / n)
    assert abs(cfg.mean() - temp_scale) < t_err, ("config temperature", cfg.mean())
    if check_kinetic:
        assert abs(kin.mean() - temp_scale) < t_err, ("kinetic temperature", kin.mean())
    return ok


def _run_verlet_preservation(seed, n_vars=50, n_dim=1000, n_samples=200, mh_freq=4):
    mcmc, models = _imports()
    with default_dtype(torch.float64):
        torch.manual_seed(seed)
        mean, std, T = 1., 2., 3 / 4
        model = models.GaussianModel(N=n_vars, D=n_dim, mean=mean, std=std).to(DEV)
        opt = mcmc.VerletSGLD(model.parameters(), lr=1 / 32, num_data=1, momentum=0.9, temperature=T,
                              seed=1000 + seed)
        model.sample_all_priors()
        with torch.no_grad():
            for p in model.parameters():
                p.sub_(mean).mul_(T ** .5).add_(mean)
        for _, st in opt.state.items():
            st['preconditioner'] = (torch.rand(()).item() + 0.2) / math.sqrt(4)
        opt.sample_momentum()
        acc_sum, acc_n, prev = 0., 0, None
        for step in range(n_samples + 1):
            if step % mh_freq == 0:
                if step != 0:
                    loss = opt.final_step(model.potential_avg_closure).item()
                    de = opt.delta_energy(prev, loss)
                    rejected, _ = opt.maybe_reject(de)
                    if rejected:
                        with torch.no_grad():
                            assert np.allclose(prev, model.potential_avg(None, None, 1.).item())
                    acc_n += 1
                    acc_sum += min(1., math.exp(-de))
                    if step == n_samples:
                        break
                prev = opt.initial_step(model.potential_avg_closure, save_state=True).item()
            else:
                opt.step(model.potential_avg_closure)
        return acc_sum / acc_n, _distribution_checks(opt, n_vars, n_dim, mean, std * T ** .5, T)


def test_verlet_distribution_preservation():
    """test_verlet_sgld.py:58-146.  The four probabilistic assertions pass jointly
    ~1/3 of the time for a CORRECT sampler (reference's own note, :215-219), and the
    noise stream here is Philox, not mt19937, so the reference's hand-picked seed does
    not transfer: require the acceptance bar and 4-sigma moment checks on every seed, and every
    individual probabilistic assertion to hold on at least one of 6 (fixed) seeds."""
    passes = {}
    for seed in range(6):
        acc, ok = _run_verlet_preservation(seed)
        assert acc > 0.6, acc  # "Was 0.73 at commit 56988f7"
        for k, v in ok.items():
            passes[k] = passes.get(k, 0) + int(v)
    # expected pass rates 0.85 / 0.7 / 0.7 / 0.7 per seed; P(0 of 6 | p = 0.7) = 0.07 %
    assert all(v >= 1 for v in passes.values()), passes


def test_hmc_distribution_preservation(n_vars=50, n_dim=1000, n_samples=100, resample=4):
    "test_hmc.py:68-135 (float32, as in the reference)"
    mcmc, models = _imports()
    passes, accs = {}, []
    for seed in range(6):
        torch.manual_seed(122 + seed)
        mean, std = 1., 2.
        model = models.GaussianModel(N=n_vars, D=n_dim, mean=mean, std=std).to(DEV)
        opt = mcmc.HMC(model.parameters(), lr=1 / 32, num_data=1, seed=2000 + seed)
        model.sample_all_priors()
        for _, st in opt.state.items():
            st['preconditioner'] = (torch.rand(()).item() + 0.2) / math.sqrt(std)
        acc_sum, acc_n, prev = 0., 0, None
        for step in range(n_samples + 1):
            if step % resample == 0:
                if step != 0:
                    loss = opt.final_step(model.potential_avg_closure).item()
                    de = opt.delta_energy(prev, loss)
                    rejected, _ = opt.maybe_reject(de)
                    if rejected:
                        with torch.no_grad():
                            assert np.allclose(prev, model.potential_avg(None, None, 1.).item())
                    acc_n += 1
                    acc_sum += min(1., math.exp(-de))
                    if step == n_samples:
                        break
                opt.sample_momentum()
                prev = opt.initial_step(model.potential_avg_closure, save_state=True).item()
            else:
                opt.step(model.potential_avg_closure)
        accs.append(acc_sum / acc_n)
        for k, v in _distribution_checks(opt, n_vars, n_dim, mean, std, 1.0).items():
            passes[k] = passes.get(k, 0) + int(v)
    assert min(accs) > 0.6, accs  # "Was 0.65 at commit 56988f7"
    assert all(v >= 1 for v in passes.values()), passes


def test_sgld_distribution_preservation(n_vars=50, n_dim=1000, n_samples=200):
    "test_sgld.py:13-59 (kinetic check is commented out in the reference too)"
    mcmc, models = _imports()
    passes = {}
    for seed in range(6):
        torch.manual_seed(123 + seed)
        mean, std, T = 1., 2., 3 / 4
        model = models.GaussianModel(N=n_vars, D=n_dim, mean=mean, std=std).to(DEV)
        opt = mcmc.SGLD(model.parameters(), lr=1 / 512, num_data=1, momentum=0.9, temperature=T,
                        seed=3000 + seed)
        model.sample_all_priors()
        with torch.no_grad():
            for p in model.parameters():
                p.sub_(mean).mul_(T ** .5).add_(mean)
        for _, st in opt.state.items():
            st['preconditioner'] = (torch.rand(()).item() + 0.2) / math.sqrt(std)
        opt.sample_momentum()
        for _ in range(n_samples):
            opt.step(model.potential_avg_closure)
        for k, v in _distribution_checks(opt, n_vars, n_dim, mean, std * T ** .5, T,
                                         check_kinetic=False).items():
            passes[k] = passes.get(k, 0) + int(v)
    assert all(v >= 1 for v in passes.values()), passes


def test_errors_match_reference():
    mcmc, models = _imports()
    p = torch.nn.Parameter(torch.zeros(8, device=DEV))
    opt = mcmc.VerletSGLD([p], lr=0.1, num_data=1, momentum=0.9)
    with pytest.raises(RuntimeError, match="No gradient"):
        opt.step()
    p.grad = torch.ones_like(p)
    with pytest.raises(RuntimeError, match="sample_momentum"):
        opt.step()
    with pytest.raises(AssertionError):
        mcmc.SGLD([p], lr=-1., num_data=1)
    hmc = mcmc.HMC([p], lr=0.1, num_data=1)
    hmc.sample_momentum()
    hmc.param_groups[0]['temperature'] = 0.5
    with pytest.raises(AssertionError):
        hmc.step()
    hmc.param_groups[0]['temperature'] = 1.0
    p.grad = torch.full_like(p, float('nan'))
    with pytest.raises(ValueError, match="not finite"):
        hmc.step()
    assert mcmc.SGLD([p], lr=0.1, num_data=1).delta_energy(0., 1.) == math.inf
    sg = mcmc.SGLD([p], lr=0.1, num_data=1, momentum=0.5)
    with pytest.raises(AssertionError):
        sg.step(save_state=True)


def test_state_views_are_built_once_and_rebound_tensors_are_adopted():
    """``state[p]['momentum_buffer']`` etc. are views of the engine's arenas, built once per arena (engine.views): the
    same objects after every transition; a tensor a user rebinds the key to (the reference's samplers own plain tensors:
    mcmc/sgld.py:57-69) is copied into the arena at the next transition and the view is restored -- the same parameters,
    bit for bit, as writing the values into the view in place."""
    mcmc, models = _imports()

    def run(rebind):
        torch.manual_seed(11)
        model = models.GaussianModel(N=1, D=7, mean=0.5, std=0.25).to(DEV)
        opt = mcmc.VerletSGLD(model.parameters(), lr=0.01, num_data=1, momentum=0.9, temperature=1., seed=3)
        opt.sample_momentum()
        opt.initial_step(model.potential_avg_closure, save_state=True)
        p = next(iter(model.parameters()))
        keys = ("momentum_buffer", "prev_parameter", "prev_grad", "prev_momentum_buffer")
        views = {k: opt.state[p][k] for k in keys}
        opt.step(model.potential_avg_closure)
        opt.final_step(model.potential_avg_closure)
        opt.sample_momentum()
        opt.initial_step(model.potential_avg_closure, save_state=True)
        for k, v in views.items():
            assert opt.state[p][k] is v, k
        if rebind:
            opt.state[p]["momentum_buffer"] = torch.full_like(p, 0.25)
        else:
            opt.state[p]["momentum_buffer"].fill_(0.25)
        opt.step(model.potential_avg_closure, calc_metrics=False)
        assert opt.state[p]["momentum_buffer"] is views["momentum_buffer"]
        return p.detach().clone(), opt.state[p]["momentum_buffer"].clone()

    (ta, ma), (tb, mb) = run(True), run(False)
    assert torch.equal(ta, tb) and torch.equal(ma, mb)
