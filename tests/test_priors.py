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
