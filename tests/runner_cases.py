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
# Runner options the reference exposes beyond the defaults of RUN_KW (round 4; ``run`` overrides RUN_KW per case):
# the descent phase at T = 0 and thinning (inference.py:56-66,141-142), the ``stairs`` / ``flat`` schedules (:96-108),
# ``precond_update`` None / k > 1 (:167; inference_reject.py:163 tests (epoch + 1) % k), ``data_mult`` (:72).
CASES.update({
    # 4 epochs per cycle = 1 descent (T = 0) + 1 warm-up + 2 sampling of which every 2nd is stored
    "VerletSGLDReject_descent_skip2": dict(momentum=0.994, temperature=1.0, lr=0.01, reject_samples=True,
                                           runner="VerletSGLDReject",
                                           run=dict(epochs_per_cycle=4, warmup_epochs=1, sample_epochs=2, skip=2)),
    "SGLD_descent_skip2": dict(momentum=0.9, temperature=1.0, lr=0.001, reject_samples=False, runner="SGLD",
                               run=dict(epochs_per_cycle=4, warmup_epochs=1, sample_epochs=2, skip=2)),
    # StepLR(150 * len(dataloader), 0.1): one minibatch per epoch, so the stair is reached at step 150 of 152
    "VerletSGLD_stairs": dict(momentum=0.9, temperature=1.0, lr=0.01, reject_samples=True, runner="VerletSGLD",
                              n=128, batch=128,
                              run=dict(cycles=1, epochs_per_cycle=152, warmup_epochs=150, sample_epochs=2,
                                       sampling_decay="stairs")),
    "VerletSGLDReject_stairs": dict(momentum=0.994, temperature=1.0, lr=0.01, reject_samples=True,
                                    runner="VerletSGLDReject", run=dict(sampling_decay=False)),
    "SGLD_flat_noprecond": dict(momentum=0.9, temperature=1.0, lr=0.001, reject_samples=False, runner="SGLD",
                                run=dict(sampling_decay="flat", precond_update=None)),
    "VerletSGLDReject_noprecond": dict(momentum=0.994, temperature=1.0, lr=0.01, reject_samples=True,
                                       runner="VerletSGLDReject", run=dict(precond_update=None)),
    # preconditioner refreshed every 2nd epoch: after epochs 1, 3 of a cycle in the reject runners, 0, 2 in the plain ones
    "VerletSGLDReject_precond2": dict(momentum=0.994, temperature=1.0, lr=0.01, reject_samples=True,
                                      runner="VerletSGLDReject",
                                      run=dict(epochs_per_cycle=4, warmup_epochs=2, sample_epochs=2, precond_update=2)),
    "VerletSGLD_precond2": dict(momentum=0.9, temperature=1.0, lr=0.01, reject_samples=True, runner="VerletSGLD",
                                run=dict(epochs_per_cycle=4, warmup_epochs=2, sample_epochs=2, precond_update=2)),
    "VerletSGLD_datamult2": dict(momentum=0.9, temperature=1.0, lr=0.01, reject_samples=True, runner="VerletSGLD",
                                 run=dict(data_mult=2.0)),
    "VerletSGLDReject_datamult2": dict(momentum=0.994, temperature=1.0, lr=0.01, reject_samples=True,
                                       runner="VerletSGLDReject", run=dict(data_mult=2.0)),
})
STREAMS_EXACT = ("acceptance/is_sample", "acceptance/rejected", "lr", "temperature")
STREAMS_FLOAT = ("delta_energy", "total_energy", "est_temperature/all", "est_config_temp/all",
                 "potential", "log_prior", "loss", "acc", "test/loss", "test/acc")
RUN_KW = dict(epochs_per_cycle=2, warmup_epochs=1, sample_epochs=1, skip=1, metrics_skip=10,
              cycles=2, precond_update=1, sampling_decay="cosine")


def run_kw(cfg):
    "RUN_KW with the case's ``run`` overrides"
    return {**RUN_KW, **(cfg or {}).get("run", {})}


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
