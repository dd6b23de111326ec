"""Fused prior kernel (rows P1-P2): value and gradient of the element-wise priors against
torch.distributions through autograd -- the formulation the reference uses
(prior/base.py:57-58, prior/loc_scale.py:34-35,66-67,74-77; its own pins:
testing/test_priors.py:127-137)."""
import numpy as np
import pytest
import torch

from bnn_priors_amd import prior as P


def test_closed_forms_cpu():
    "SURVEY.md Appendix A closed forms == torch.distributions (float64, CPU)"
    torch.manual_seed(0)
    th = torch.randn(1000, dtype=torch.float64) * 3
    for cls, dist, kw in ((P.Normal, torch.distributions.Normal, {}),
                          (P.Laplace, torch.distributions.Laplace, {}),
                          (P.StudentT, torch.distributions.StudentT, {"df": 3}),
                          (P.Cauchy, torch.distributions.Cauchy, {})):
        loc, scale = 0.3, 1.7
        t = th.clone().requires_grad_(True)
        args = ((kw["df"], loc, scale) if kw else (loc, scale))
        lp = dist(*[torch.tensor(a, dtype=torch.float64) for a in args]).log_prob(t).sum()
        lp.backward()
        d = th - loc
        if cls is P.Normal:
            g = -d / scale ** 2
        elif cls is P.Laplace:
            g = -torch.sign(d) / scale
        elif cls is P.Cauchy:
            g = -2 * d / (scale ** 2 + d ** 2)
        else:
            g = -(3 + 1) * d / (3 * scale ** 2 + d ** 2)
        assert torch.allclose(t.grad, g, rtol=1e-12, atol=1e-14)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_fused_prior_kernel_matches_autograd(dtype):
    from bnn_priors_amd import mcmc
    dev = "cuda:0"
    old = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        torch.manual_seed(1)
        priors = [P.Normal((5000,), 0.1, 0.7), P.Laplace((4097,), -0.2, 1.3),
                  P.StudentT((33, 5), 0.0, 0.4, df=3), P.Cauchy((1025,), 0.3, 0.9),
                  P.Normal((7,), 0., torch.linspace(0.5, 1.5, 7))]     # last one: not fusable
        model = torch.nn.ModuleList(priors).to(dev)
        free = torch.nn.Parameter(torch.randn(10, device=dev))          # parameter without a prior
        params = [pr.p for pr in model] + [free]
        N = 123.0
        opt = mcmc.VerletSGLD(params, lr=0.01, num_data=N, momentum=0.9)
        leftover = opt.fuse_priors(model)
        assert leftover == [model[4]]
        g0 = [torch.randn_like(p) for p in params]
        # reference formulation
        for p, g in zip(params, g0):
            p.grad = g.clone()
        lp = sum(pr.log_prob() for pr in model[:4])
        (lp / -N).backward()
        want = [p.grad.clone() for p in params]
        # fused kernel
        for p, g in zip(params, g0):
            p.grad = g.clone()
        opt.add_prior_gradient(calc_log_prior=True)
        tol = dict(rtol=2e-6, atol=1e-7) if dtype == torch.float32 else dict(rtol=1e-13, atol=1e-15)
        for a, b in zip(params, want):
            torch.testing.assert_close(a.grad, b, **tol)
        got_lp = opt.fused_log_prior().item()
        assert got_lp == pytest.approx(lp.item(), rel=3e-6 if dtype == torch.float32 else 1e-12)
        per_seg = opt.engine.fetch_state()[:, -1]
        for i, pr in enumerate(model[:4]):
            assert per_seg[i] == pytest.approx(pr.log_prob().item(), rel=3e-6 if dtype == torch.float32 else 1e-12)
        assert per_seg[4] == 0.0 and per_seg[5] == 0.0
    finally:
        torch.set_default_dtype(old)


# ------------------------------------------------------------------ fixtures from the imported reference
def _prior_cases(golden_dir):
    import os
    z = np.load(os.path.join(golden_dir, "priors.npz"))
    keys = sorted({k.rsplit("|", 1)[0] for k in z.files})
    return z, keys


def _build(name, extra, shape, loc, scale, dtype):
    kw = {}
    for item in filter(None, extra.split("_")):
        for field in ("hyperscale", "rate", "beta", "df"):
            if item.startswith(field):
                kw[field] = float(item[len(field):])
    old = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        torch.manual_seed(0)
        return P.get_prior(name)(shape, loc, scale, **kw)
    finally:
        torch.set_default_dtype(old)


def _load_case(z, key):
    name, extra, shape, loc, scale, dt = key.split("|")
    dtype = torch.float32 if dt == "float32" else torch.float64
    pr = _build(name, extra, tuple(int(s) for s in shape.split("x")), float(loc), float(scale), dtype)
    with torch.no_grad():
        pr.p.copy_(torch.from_numpy(z[key + "|theta"]).to(dtype))
        hyper = pr.scale_link() or (getattr(pr, "beta", None) if isinstance(getattr(pr, "beta", None), P.Prior) else None)
        if hyper is not None:
            hyper.p.fill_(float(z[key + "|hyper_p"]))
    return pr, hyper, dtype


def test_prior_modules_match_reference_fixtures(golden_dir):
    """log_prob and its gradient of every prior family of the hot path against values captured from the
    imported reference (tests/golden/make_prior_goldens.py; prior/loc_scale.py:34-96,
    hierarchical.py:17-104, transformed.py:55-87): pins the autograd formulation of the product's modules"""
    z, keys = _prior_cases(golden_dir)
    assert len(keys) == 96
    for key in keys:
        pr, hyper, dtype = _load_case(z, key)
        tol = dict(rel=2e-5, abs=2e-5) if dtype == torch.float32 else dict(rel=1e-11, abs=1e-11)
        lps = [m.log_prob() for _, m in P.named_priors(pr)]
        assert float(lps[0]) == pytest.approx(float(z[key + "|log_prob"]), **tol), key
        total = sum(lps)
        want = torch.from_numpy(z[key + "|grad_theta"]).to(dtype)
        if isinstance(total, torch.Tensor) and total.requires_grad:
            total.backward()
            got = pr.p.grad if pr.p.grad is not None else torch.zeros_like(pr.p)
        else:
            got = torch.zeros_like(pr.p)
        torch.testing.assert_close(got, want, rtol=tol["rel"] * 5, atol=tol["abs"])
        if hyper is not None:
            assert float(hyper()) == pytest.approx(float(z[key + "|hyper_value"]), **tol), key
            assert float(lps[1]) == pytest.approx(float(z[key + "|hyper_log_prob"]), **tol), key
            g = 0.0 if hyper.p.grad is None else float(hyper.p.grad)
            assert g == pytest.approx(float(z[key + "|grad_hyper"]), rel=tol["rel"] * 20, abs=tol["abs"] * 20), key


@pytest.mark.gpu
def test_fused_prior_hook_matches_reference_fixtures(golden_dir):
    """the HIP prior hook (sgmcmc_prior_grad: closed-form gradient of -log p / N into g, log-density in fp64,
    chain-rule term of hierarchical scales) against the reference's values for every fusable case"""
    from bnn_priors_amd import mcmc
    z, keys = _prior_cases(golden_dir)
    dev, N, seen = "cuda:0", 77.0, set()
    for key in keys:
        pr, hyper, dtype = _load_case(z, key)
        name = key.split("|")[0]
        pr = pr.to(dev)
        hyper = pr.scale_link()
        params = [pr.p] + ([hyper.p] if hyper is not None else [])
        opt = mcmc.VerletSGLD(params, lr=0.01, num_data=N, momentum=0.9)
        leftover = opt.fuse_priors(pr)
        if name == "gennorm_uniform":
            assert leftover                     # the shape parameter's hyper-prior stays in autograd
            continue
        assert not leftover, key
        seen.add(name)
        g0 = torch.randn(pr.p.shape, generator=torch.Generator().manual_seed(5)).to(dtype).to(dev)
        pr.p.grad = g0.clone()
        opt.add_prior_gradient(calc_log_prior=True)
        tol = dict(rtol=1e-4, atol=2e-6) if dtype == torch.float32 else dict(rtol=1e-10, atol=1e-12)
        want = g0 - torch.from_numpy(z[key + "|grad_theta"]).to(dtype).to(dev) / N
        torch.testing.assert_close(pr.p.grad, want, **tol)
        lp_want = float(z[key + "|log_prob"]) if name != "improper" else 0.0
        if hyper is not None:
            lp_want += float(z[key + "|hyper_log_prob"])
            g_h = -float(z[key + "|grad_hyper"]) / N
            assert float(hyper.p.grad) == pytest.approx(g_h, rel=tol["rtol"] * 5, abs=tol["atol"] * 5), key
        assert opt.fused_log_prior().item() == pytest.approx(lp_want, rel=1e-5 if dtype == torch.float32 else 1e-11,
                                                             abs=1e-4 if dtype == torch.float32 else 1e-10), key
    assert {"gaussian", "laplace", "student-t", "cauchy", "gennorm", "gaussian_gamma", "laplace_gamma",
            "student-t_gamma", "gaussian_uniform", "laplace_uniform", "student-t_uniform", "horseshoe"} <= seen


# ------------------------------------------------------------------ priors built by name and left to autograd (round 5)
def _by_name_cases(golden_dir):
    import os
    z = np.load(os.path.join(golden_dir, "priors_by_name.npz"))
    return z, sorted({k.rsplit("|", 1)[0] for k in z.files})


def _build_by_name(key, z):
    import json
    name, a, b, extra, shape, dt = key.split("|")
    dtype = torch.float32 if dt == "float32" else torch.float64
    kw = {k: json.loads(v) for k, v in (item.split("=", 1) for item in filter(None, extra.split(";")))}
    old = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        torch.manual_seed(0)
        pr = P.get_prior(name)(tuple(int(s) for s in shape.split("x")), float(a), float(b), **kw)
    finally:
        torch.set_default_dtype(old)
    with torch.no_grad():
        pr.p.copy_(torch.from_numpy(z[key + "|theta"]).to(dtype))
        if key + "|weights" in z.files:
            pr.mixture_weights.copy_(torch.from_numpy(z[key + "|weights"]).to(dtype))
    return name, pr, dtype


def test_priors_by_name_match_reference_fixtures(golden_dir):
    """``get_prior("lognormal" | "uniform" | "mixture" | "scale_mixture")`` builds the reference's prior (its table:
    prior/mixture.py:17-50) instead of raising: the model-level log-prior, its gradient w.r.t. the tensor and the
    mixture logits, the transformed value and the state_dict's keys against values captured from the imported reference
    (tests/golden/make_prior_by_name_goldens.py).  None of them is handed to the HIP hook (fused_spec() is None)."""
    import json
    z, keys = _by_name_cases(golden_dir)
    assert len(keys) == 40 and {k.split("|")[0] for k in keys} == {"lognormal", "uniform", "mixture", "scale_mixture"}
    for key in keys:
        name, pr, dtype = _build_by_name(key, z)
        tol = dict(rel=2e-5, abs=2e-5) if dtype == torch.float32 else dict(rel=1e-11, abs=1e-11)
        assert pr.fused_spec() is None, key
        assert sorted(pr.state_dict().keys()) == json.loads(str(z[key + "|state_keys"])), key
        assert len(list(pr.parameters())) == int(z[key + "|n_parameters"]), key
        total = sum(m.log_prob() for _, m in P.named_priors(pr))
        assert float(total) == pytest.approx(float(z[key + "|log_prior"]), **tol), key
        if isinstance(total, torch.Tensor) and total.requires_grad:
            total.backward()
        got = pr.p.grad if pr.p.grad is not None else torch.zeros_like(pr.p)
        torch.testing.assert_close(got.double(), torch.from_numpy(z[key + "|grad_theta"]), rtol=tol["rel"] * 5, atol=tol["abs"])
        torch.testing.assert_close(pr().detach().double(), torch.from_numpy(z[key + "|value"]), rtol=tol["rel"], atol=tol["abs"])
        if key + "|grad_weights" in z.files:
            torch.testing.assert_close(pr.mixture_weights.grad.double(), torch.from_numpy(z[key + "|grad_weights"]),
                                       rtol=tol["rel"] * 5, atol=tol["abs"])


def test_a_mixtures_components_are_views_not_priors_of_their_own():
    torch.manual_seed(3)
    m = P.get_prior("mixture")((6, 2), 0.0, 0.5, components="g_l_s")
    assert [n for n, _ in m.named_parameters()] == ["p", "mixture_weights"]          # one tensor, one logit vector
    assert all(c.p is m.p and c.log_prob() == 0. and c.fused_spec() is None and c.is_component for c in m.components)
    with pytest.raises(KeyError):
        P.get_prior("no-such-prior")           # an unknown name stays an error, with the table in the message


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["lognormal", "uniform", "mixture", "scale_mixture"])
def test_by_name_priors_run_through_the_sampler_as_leftover(name):
    """a dense classifier whose WEIGHT prior is one of the by-name families: the optimizer leaves exactly those priors
    to autograd (the bias priors stay in the HIP hook), announces it once, and the gradient of the average potential
    equals the reference formulation's (models/base.py:72-77 through autograd) -- then one sampler transition runs"""
    import copy
    import warnings
    from bnn_priors_amd import mcmc, models, potential
    dev, N = "cuda:0", 512.0
    torch.manual_seed(0)
    x, y = torch.rand(64, 784), torch.randint(0, 10, (64,))
    net = models.get_model(x, y, "classificationdensenet", width=16, depth=3, weight_prior=name, weight_loc=0.,
                           weight_scale=2 ** .5 if name != "uniform" else 1.0, bias_prior="gaussian", bias_scale=1.).to(dev)
    x, y = x.to(dev), y.to(dev)
    ref = copy.deepcopy(net)
    opt = mcmc.VerletSGLD(net.parameters(), lr=1e-4, num_data=N, momentum=0.9, temperature=1.0, seed=3)
    potential._noticed.clear()
    with warnings.catch_warnings(record=True) as seen:
        warnings.simplefilter("always")
        pot = potential.Potential(net, opt, N)
        potential.Potential(net, opt, N)
    assert sum("differentiated by autograd" in str(w.message) for w in seen) == 1      # one notice per family set
    want_left = {id(pr) for n_, pr in P.named_priors(net) if "weight_prior" in n_ and not getattr(pr, "is_component", False)}
    assert {id(pr) for pr in pot.leftover} == want_left and pot.fast
    loss, log_prior, pot_value, acc = pot.minibatch(x, y, True)
    _, lp_ref, potential_ref, _, _ = ref.split_potential_and_acc(x, y, N)
    potential_ref.backward()
    parts = dict(fused=float(opt.fused_log_prior()), leftover=float(pot._leftover_log_prior()), ref=float(lp_ref),
                 ref_parts={n_: float(pr.log_prob()) for n_, pr in P.named_priors(ref) if not getattr(pr, "is_component", False)},
                 own_parts={n_: float(pr.log_prob()) for n_, pr in P.named_priors(net) if not getattr(pr, "is_component", False)})
    assert float(log_prior) == pytest.approx(float(lp_ref), rel=2e-5, abs=1e-3), parts
    assert float(pot_value) == pytest.approx(float(potential_ref), rel=2e-5, abs=1e-5), parts
    for (n_, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        want = q.grad if q.grad is not None else torch.zeros_like(q)
        torch.testing.assert_close(p.grad, want, rtol=2e-4, atol=2e-6, msg=lambda m: f"{name} {n_}: {m}")
    opt.sample_momentum()
    opt.initial_step(save_state=False)
    assert all(torch.isfinite(p).all() for p in net.parameters())


# ------------------------------------------------------------------ the last 13 names of the reference's table (round 6)
REFERENCE_TABLE = ["gaussian", "convcorrnormal", "convcorrnormal_fitted_ls", "convcorrnormal_empirical", "convcorrnormal_gamma",
                   "datadrivencorrnormal", "datadrivencorrdoublegamma", "fixedcov_normal", "fixedcov_gennorm", "lognormal",
                   "laplace", "cauchy", "student-t", "uniform", "improper", "gaussian_gamma", "gaussian_uniform", "horseshoe",
                   "laplace_gamma", "laplace_uniform", "student-t_gamma", "student-t_uniform", "gennorm", "gennorm_uniform",
                   "gaussian_empirical", "laplace_empirical", "student-t_empirical", "gennorm_empirical", "scale_mixture",
                   "mixture", "scale_mixture_empirical"]       # bnn_priors/prior/mixture.py:17-50, in its order


def test_get_prior_knows_every_name_of_the_reference_table():
    from bnn_priors_amd.prior import loc_scale
    assert len(REFERENCE_TABLE) == 31 and sorted(loc_scale._table()) == sorted(REFERENCE_TABLE)
    for name in REFERENCE_TABLE:
        assert issubclass(P.get_prior(name), P.Prior), name


def _remaining_cases(golden_dir):
    import os
    z = np.load(os.path.join(golden_dir, "priors_remaining.npz"))
    return z, sorted({k.rsplit("|", 1)[0] for k in z.files})


def _covariance(n):
    "tests/golden/make_prior_remaining_goldens.py: covariance()"
    i = np.arange(n)
    return 0.3 * np.exp(-np.abs(i[:, None] - i[None, :]) / 2.0) + 0.05 * np.eye(n)


def _build_remaining(key, z):
    "the constructor call of tests/golden/make_prior_remaining_goldens.py: build(), then the fixture's parameter values"
    import json
    name, a, b, extra, shape, dt = key.split("|")
    dtype = torch.float32 if dt == "float32" else torch.float64
    kw = {k: json.loads(v) for k, v in (item.split("=", 1) for item in filter(None, extra.split(";")))}
    shape = tuple(int(s) for s in shape.split("x"))
    old = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        torch.manual_seed(0)
        loc, scale = float(a), (b if b == "cov" else float(b))
        n_pos = shape[-2] * shape[-1] if len(shape) >= 2 else 1
        if name.startswith("convcorrnormal"):
            loc = torch.zeros(n_pos) + loc
        if scale == "cov":
            scale = torch.from_numpy(_covariance(n_pos)).to(dtype)
            loc = torch.zeros(n_pos) + loc
        pr = P.get_prior(name)(shape, loc, scale, **kw)
    finally:
        torch.set_default_dtype(old)
    names = [k.split("|param:", 1)[1] for k in z.files if k.startswith(key + "|param:")]
    params = dict(pr.named_parameters())
    with torch.no_grad():
        for n_ in names:
            params[n_].copy_(torch.from_numpy(z[key + "|param:" + n_]).to(dtype).reshape(params[n_].shape))
    return name, pr, dtype, names


def test_remaining_priors_match_reference_fixtures(golden_dir):
    """the empirical-Bayes families, ``scale_mixture_empirical``, the correlated / fixed-covariance convolution priors and
    the data-driven marginals (reference: prior/empirical_bayes.py:14-58, mixture.py:154-178, loc_scale.py:38-63,
    hierarchical.py:32-39, conv_loc_scale.py:62-140, transformed.py:83-95): model-level log-prior, its gradient w.r.t.
    EVERY parameter (learnable scales, shape parameters, lengthscales, mixture logits), the value and the stored keys
    against the imported reference (tests/golden/make_prior_remaining_goldens.py)."""
    import json
    z, keys = _remaining_cases(golden_dir)
    assert len(keys) == 96 and {k.split("|")[0] for k in keys} == set(REFERENCE_TABLE) - {
        "gaussian", "lognormal", "laplace", "cauchy", "student-t", "uniform", "improper", "gaussian_gamma", "gaussian_uniform",
        "horseshoe", "laplace_gamma", "laplace_uniform", "student-t_gamma", "student-t_uniform", "gennorm", "gennorm_uniform",
        "scale_mixture", "mixture"}
    for key in keys:
        name, pr, dtype, names = _build_remaining(key, z)
        tol = dict(rel=3e-5, abs=3e-5) if dtype == torch.float32 else dict(rel=1e-10, abs=1e-10)
        want_keys = json.loads(str(z[key + "|state_keys"]))
        assert sorted(pr.state_dict().keys()) == want_keys, key
        assert len(list(pr.parameters())) == int(z[key + "|n_parameters"]), key
        total = sum(m.log_prob() for _, m in P.named_priors(pr))
        assert float(total) == pytest.approx(float(z[key + "|log_prior"]), **tol), key
        total.backward()
        params = dict(pr.named_parameters())
        for n_ in names:
            got = params[n_].grad if params[n_].grad is not None else torch.zeros_like(params[n_])
            want = torch.from_numpy(z[key + "|grad:" + n_]).reshape(got.shape)
            torch.testing.assert_close(got.double(), want, rtol=tol["rel"] * 10, atol=tol["abs"] * 10, msg=lambda m: f"{key} {n_}: {m}")
        torch.testing.assert_close(pr().detach().double(), torch.from_numpy(z[key + "|value"]), rtol=tol["rel"], atol=tol["abs"])
        # which route a sampler would take: Normal / Laplace with a learnable scale are the hook's (linked to a
        # PositiveImproper segment); a learnable shape parameter, a mixture, a multivariate density are autograd's
        fused = pr.fused_spec() is not None
        assert fused == (name in ("gaussian_empirical", "laplace_empirical", "datadrivencorrnormal")), key
        if name in ("gaussian_empirical", "laplace_empirical"):
            assert pr.scale_link().fused_spec()[0] == 9 and np.isnan(pr.fused_spec()[2])


@pytest.mark.gpu
def test_learnable_scales_go_through_the_hook_like_the_reference(golden_dir):
    """``gaussian_empirical`` / ``laplace_empirical`` on the HIP path: the weight segment reads its scale from the
    PositiveImproper hyper segment (SGMCMC_PRIOR_IMPROPER_SOFTPLUS), the hyper-parameter's gradient is the chain-rule term
    of the dls reduction -- both against the reference's autograd values"""
    from bnn_priors_amd import mcmc
    z, keys = _remaining_cases(golden_dir)
    dev, N, done = "cuda:0", 61.0, 0
    for key in keys:
        name = key.split("|")[0]
        if name not in ("gaussian_empirical", "laplace_empirical"):
            continue
        _, pr, dtype, _ = _build_remaining(key, z)
        pr = pr.to(dev)
        hyper = pr.scale_link()
        opt = mcmc.VerletSGLD([pr.p, hyper.p], lr=0.01, num_data=N, momentum=0.9)
        assert not opt.fuse_priors(pr), key
        g0 = torch.randn(pr.p.shape, generator=torch.Generator().manual_seed(5)).to(dtype).to(dev)
        pr.p.grad = g0.clone()
        opt.add_prior_gradient(calc_log_prior=True)
        tol = dict(rtol=1e-4, atol=2e-6) if dtype == torch.float32 else dict(rtol=1e-10, atol=1e-12)
        want = g0 - torch.from_numpy(z[key + "|grad:p"]).to(dtype).to(dev) / N
        torch.testing.assert_close(pr.p.grad, want, **tol)
        g_h = -float(z[key + "|grad:scale.p"]) / N
        assert float(hyper.p.grad) == pytest.approx(g_h, rel=tol["rtol"] * 5, abs=tol["atol"] * 5), key
        assert opt.fused_log_prior().item() == pytest.approx(float(z[key + "|log_prior"]), rel=1e-5 if dtype == torch.float32 else 1e-11,
                                                             abs=1e-4 if dtype == torch.float32 else 1e-10), key
        done += 1
    assert done == 16


@pytest.mark.gpu
@pytest.mark.parametrize("name,kw", [("gaussian_empirical", {}), ("student-t_empirical", {}), ("gennorm_empirical", {}),
                                     ("scale_mixture_empirical", {}), ("datadrivencorrdoublegamma", {"concentration": 0.8})])
def test_remaining_names_run_through_the_sampler(name, kw):
    """a dense classifier whose WEIGHT prior is one of round 6's names: the gradient of the average potential equals the
    reference formulation's (models/base.py:72-77 through autograd on a copy of the net) whichever route the prior takes
    -- the hook (gaussian_empirical: nothing left over) or autograd (the others) -- and one sampler transition runs"""
    import copy
    import warnings
    from bnn_priors_amd import mcmc, models, potential
    dev, N = "cuda:0", 512.0
    torch.manual_seed(0)
    x, y = torch.rand(64, 784), torch.randint(0, 10, (64,))
    net = models.get_model(x, y, "classificationdensenet", width=16, depth=3, weight_prior=name, weight_loc=0.,
                           weight_scale=2 ** .5, bias_prior="gaussian", bias_scale=1., weight_prior_params=kw).to(dev)
    x, y = x.to(dev), y.to(dev)
    ref = copy.deepcopy(net)
    opt = mcmc.VerletSGLD(net.parameters(), lr=1e-4, num_data=N, momentum=0.9, temperature=1.0, seed=3)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        pot = potential.Potential(net, opt, N)
    assert pot.fast and bool(pot.leftover) == (name != "gaussian_empirical")
    loss, log_prior, pot_value, acc = pot.minibatch(x, y, True)
    _, lp_ref, potential_ref, _, _ = ref.split_potential_and_acc(x, y, N)
    potential_ref.backward()
    assert float(log_prior) == pytest.approx(float(lp_ref), rel=2e-5, abs=1e-3)
    assert float(pot_value) == pytest.approx(float(potential_ref), rel=2e-5, abs=1e-5)
    for (n_, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        want = q.grad if q.grad is not None else torch.zeros_like(q)
        got = p.grad if p.grad is not None else torch.zeros_like(p)
        torch.testing.assert_close(got, want, rtol=2e-4, atol=2e-6, msg=lambda m: f"{name} {n_}: {m}")
    opt.sample_momentum()
    opt.initial_step(save_state=False)
    assert all(torch.isfinite(p).all() for p in net.parameters())
