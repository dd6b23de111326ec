import copy, sys, warnings
import torch
sys.path.insert(0, ".")
from bnn_priors_amd import mcmc, models, potential
from bnn_priors_amd import prior as P
dev, N = "cuda:0", 512.0
for name in ("mixture", "scale_mixture", "lognormal"):
    torch.manual_seed(0)
    x, y = torch.rand(64, 784), torch.randint(0, 10, (64,))
    net = models.get_model(x, y, "classificationdensenet", width=16, depth=3, weight_prior=name, weight_loc=0.,
                           weight_scale=2 ** .5, bias_prior="gaussian", bias_scale=1.).to(dev)
    x, y = x.to(dev), y.to(dev)
    print(name, "params:", [(n, tuple(p.shape)) for n, p in net.named_parameters()][:12])
    for n_, pr in P.named_priors(net):
        if hasattr(pr, "components"):
            print("  shared p:", [c.p is pr.p for c in pr.components], "p device", pr.p.device)
    ref = copy.deepcopy(net)
    opt = mcmc.VerletSGLD(net.parameters(), lr=1e-4, num_data=N, momentum=0.9, temperature=1.0, seed=3)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        pot = potential.Potential(net, opt, N)
    pot.minibatch(x, y, True)
    g_own = {n: p.grad.clone() for n, p in net.named_parameters()}
    # reference formulation on ref
    _, lp_ref, potential_ref, _, _ = ref.split_potential_and_acc(x, y, N)
    potential_ref.backward()
    g_ref = {n: (p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for n, p in ref.named_parameters()}
    # likelihood only / prior only on a third copy through plain autograd
    third = copy.deepcopy(ref); third.zero_grad()
    f = third.net(x); lik = torch.nn.functional.cross_entropy(f, y); lik.backward()
    g_lik = {n: (p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for n, p in third.named_parameters()}
    third.zero_grad(); (third.log_prior() / -N).backward()
    g_pri = {n: (p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for n, p in third.named_parameters()}
    for n in g_own:
        d = (g_own[n] - g_ref[n]).abs().max().item()
        d2 = (g_lik[n] + g_pri[n] - g_ref[n]).abs().max().item()
        d3 = (g_own[n] - g_lik[n]).abs().max().item()
        print(f"  {n:45s} own-ref {d:.3e}  (lik+pri)-ref {d2:.3e}  own-lik {d3:.3e}  |pri| {g_pri[n].abs().max().item():.3e} |lik| {g_lik[n].abs().max().item():.3e}")
