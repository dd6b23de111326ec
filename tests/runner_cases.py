"""Runner-level parity cases, shared by the golden
generator and the tests: a 1,024-example synthetic MNIST-shaped training set
(8 minibatches of 128 per epoch), 256 test examples, classificationdensenet,
2 cycles x 2 epochs (1 warm-up + 1 sampling), metrics every 10 steps."""
import numpy as np
import torch

SEED = 777
CYCLE_SEED = 4000
CASES = {
    "VerletSGLDReject": dict(momentum=0.994, temperature=1.0, lr=0.01, reject_samples=True),
    "VerletSGLDReject_noreject": dict(momentum=0.994, temperature=1.0, lr=0.01, reject_samples=False,
                                      runner="VerletSGLDReject"),
    "VerletSGLD": dict(momentum=0.9, temperature=1.0, lr=0.01, reject_samples=True),
    "SGLD": dict(momentum=0.9, temperature=1.0, lr=0.001, reject_samples=False),
    "HMCReject": dict(momentum=1.0, temperature=1.0, lr=0.0005, reject_samples=True),
    "OurHMC": dict(momentum=1.0, temperature=1.0, lr=0.0005, reject_samples=True),
    "SGLDReject": dict(momentum=0.9, temperature=1.0, lr=0.001, reject_samples=False),
}
# BASELINE configs[2]-[4]: the other model families / priors (small synthetic sets)
CASES.update({
    "VerletSGLDReject_convnet_laplace": dict(momentum=0.98, temperature=1.0, lr=0.002, reject_samples=True,
                                             runner="VerletSGLDReject", model="classificationconvnet",
                                             weight_prior="laplace", n=512),
    "VerletSGLDReject_googleresnet": dict(momentum=0.98, temperature=1.0, lr=0.002, reject_samples=True,
                                          runner="VerletSGLDReject", model="googleresnet",
                                          weight_prior="gaussian", n=128, batch=32, image=True),
    "HMCReject_googleresnet_studentt": dict(momentum=1.0, temperature=1.0, lr=0.0002, reject_samples=True,
                                            runner="HMCReject", model="googleresnet",
                                            weight_prior="student-t", n=128, batch=32, image=True),
})
STREAMS_EXACT = ("acceptance/is_sample", "acceptance/rejected", "lr", "temperature")
STREAMS_FLOAT = ("delta_energy", "total_energy", "est_temperature/all", "est_config_temp/all",
                 "potential", "log_prior", "loss", "acc", "test/loss", "test/acc")
RUN_KW = dict(epochs_per_cycle=2, warmup_epochs=1, sample_epochs=1, skip=1, metrics_skip=10,
              cycles=2, precond_update=1, sampling_decay="cosine")


def make_data(device="cpu", cfg=None):
    cfg = cfg or {}
    n, bs = cfg.get("n", 1024), cfg.get("batch", 128)
    nt = max(bs, n // 4)
    g = torch.Generator().manual_seed(5)
    shape = (3, 32, 32) if cfg.get("image") else (784,)
    x = torch.rand((n,) + shape, generator=g)
    y = torch.randint(0, 10, (n,), generator=g)
    xt = torch.rand((nt,) + shape, generator=g)
    yt = torch.randint(0, 10, (nt,), generator=g)
    mk = torch.utils.data.TensorDataset
    train = torch.utils.data.DataLoader(mk(x.to(device), y.to(device)), batch_size=bs,
                                        shuffle=True, drop_last=False)
    test = torch.utils.data.DataLoader(mk(xt.to(device), yt.to(device)), batch_size=bs,
                                       shuffle=False, drop_last=False)
    return train, test, (x, y)


def make_net(models_mod, x, y, device="cpu", exp_utils=None, cfg=None):
    "the case's network (default classificationdensenet), He-initialised, deterministic"
    cfg = cfg or {}
    torch.manual_seed(0)
    factory = exp_utils if exp_utils is not None else models_mod
    kw = dict(width=50, depth=3, weight_prior=cfg.get("weight_prior", "gaussian"), weight_loc=0.,
              weight_scale=2 ** .5, bias_prior="gaussian", bias_loc=0., bias_scale=1., batchnorm=True,
              weight_prior_params={}, bias_prior_params={})
    net = factory.get_model(x, y, cfg.get("model", "classificationdensenet"), **kw)
    torch.manual_seed(1)
    factory.he_initialize(net)
    return net.to(device)


def streams_of(metrics):
    "name -> (steps, values) for every logged key"
    out = {}
    for name in metrics.names():
        s, v = metrics.column(name)
        keep = ~np.isnan(v)
        out[name] = (s[keep], v[keep])
    return out
