"""Fused dense-classifier kernel (csrc/mlp_hip.inc) against a plain PyTorch fp32
reference of the same op: gradients of mean cross-entropy of a 3-layer ReLU MLP."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _run_kernel(X, Y, idx, Ws, batch, temp=1.0):
    from bnn_priors_amd import _hip
    W1, b1, W2, b2, W3, b3 = Ws
    sizes = [t.numel() for t in Ws]
    offs, acc = [], 0
    for n in sizes:
        offs.append(acc)
        acc += -(-n // 4) * 4
    slices = -(-batch // _hip.MLP_ROWS)
    gpart = torch.full((slices, acc), float("nan"), device=DEV)
    loss_part = torch.zeros(slices, device=DEV)
    corr_part = torch.zeros(slices, device=DEV)
    A = _hip.MlpArgs(X=X.data_ptr(), Y=Y.data_ptr(), idx=idx.data_ptr() if idx is not None else None,
                     W1=W1.data_ptr(), b1=b1.data_ptr(), W2=W2.data_ptr(), b2=b2.data_ptr(),
                     W3=W3.data_ptr(), b3=b3.data_ptr(), gpart=gpart.data_ptr(),
                     loss_part=loss_part.data_ptr(), correct_part=corr_part.data_ptr(),
                     gpart_stride=acc, off_W1=offs[0], off_b1=offs[1], off_W2=offs[2], off_b2=offs[3],
                     off_W3=offs[4], off_b3=offs[5], batch=batch, in_features=W1.shape[1],
                     hidden1=W1.shape[0], hidden2=W2.shape[0], out_features=W3.shape[0],
                     inv_softmax_temp=1.0 / temp)
    _hip.check(_hip.lib().sgmcmc_mlp_fwdbwd(ctypes.byref(A),
                                            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
               "mlp_fwdbwd")
    torch.cuda.synchronize()
    grads = [gpart[:, o:o + n].sum(0).view(t.shape) for o, n, t in zip(offs, sizes, Ws)]
    return grads, loss_part.sum().item() / batch, corr_part.sum().item() / batch, gpart


@pytest.mark.parametrize("dims,batch,n_data,temp", [
    ((784, 50, 50, 10), 128, 1000, 1.0),      # the BASELINE classificationdensenet
    ((784, 50, 50, 10), 96, 1000, 1.0),       # last minibatch of an MNIST epoch (60000 % 128)
    ((20, 8, 8, 10), 32, 32, 1.0),            # the tiny golden net
    ((36, 64, 33, 16), 100, 300, 2.5),        # ragged slice, full-width tiles, softmax temperature
    ((4, 1, 1, 2), 5, 9, 1.0),
])
def test_mlp_fwdbwd_matches_torch(dims, batch, n_data, temp):
    IN, H1, H2, OUT = dims
    g = torch.Generator().manual_seed(3)
    X = torch.randn(n_data, IN, generator=g).to(DEV)
    Y = torch.randint(0, OUT, (n_data,), generator=g).to(DEV)
    idx = torch.randperm(n_data, generator=g)[:batch].to(DEV)
    mk = lambda *s: (torch.randn(*s, generator=g) * (1.5 / max(s[-1], 1) ** 0.5)).to(DEV).requires_grad_(True)  # noqa: E731
    Ws = [mk(H1, IN), mk(H1), mk(H2, H1), mk(H2), mk(OUT, H2), mk(OUT)]
    grads, loss, acc, gpart = _run_kernel(X, Y, idx, [w.detach() for w in Ws], batch, temp)

    x, y = X[idx], Y[idx]
    h1 = F.relu(F.linear(x, Ws[0], Ws[1]))
    h2 = F.relu(F.linear(h1, Ws[2], Ws[3]))
    f = F.linear(h2, Ws[4], Ws[5]) / temp
    ref_loss = F.cross_entropy(f, y)
    ref_loss.backward()
    assert loss == pytest.approx(ref_loss.item(), rel=2e-6, abs=1e-6)
    assert acc == pytest.approx((f.argmax(1) == y).float().mean().item(), abs=1e-7)
    for got, w in zip(grads, Ws):
        scale = w.grad.abs().max().item() + 1e-12
        # same fp32 products, different summation order than rocBLAS: a few ulp of the largest term
        assert (got - w.grad).abs().max().item() <= 2e-6 * scale + 1e-9, (got - w.grad).abs().max().item() / scale
    # no partial outside the tensors' extents was touched except the 4-alignment pads (left NaN)
    assert torch.isnan(gpart).sum().item() == gpart.shape[0] * sum((-t.numel()) % 4 for t in Ws)


def test_mlp_fwdbwd_identity_index_and_determinism():
    g = torch.Generator().manual_seed(4)
    X = torch.randn(128, 784, generator=g).to(DEV)
    Y = torch.randint(0, 10, (128,), generator=g).to(DEV)
    mk = lambda *s: (torch.randn(*s, generator=g) * 0.05).to(DEV)  # noqa: E731
    Ws = [mk(50, 784), mk(50), mk(50, 50), mk(50), mk(10, 50), mk(10)]
    a = _run_kernel(X, Y, None, Ws, 128)
    b = _run_kernel(X, Y, torch.arange(128, device=DEV), Ws, 128)
    for u, v in zip(a[0], b[0]):
        assert torch.equal(u, v)
    assert a[1] == b[1]


def test_fused_exact_gradient_pass_matches_autograd():
    """full-data gradient through the fused kernel (mega-batches, fp64 accumulation) vs the
    reference formulation through autograd (inference_reject.py:18-33)"""
    import sys
    import runner_cases as RC
    from bnn_priors_amd import inference_reject, models
    from bnn_priors_amd.storage import MemoryMetrics
    cfg = RC.CASES["VerletSGLDReject"]
    g = torch.Generator().manual_seed(9)
    n = 5000                                  # 2 full mega-batches + a ragged one
    x = torch.rand(n, 784, generator=g).to(DEV)
    y = torch.randint(0, 10, (n,), generator=g).to(DEV)
    train = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(x, y), batch_size=128, shuffle=True)
    test = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(x[:0], y[:0]), batch_size=128)
    model = RC.make_net(models, x.cpu(), y.cpu(), device=DEV)
    runner = inference_reject.VerletSGLDRunnerReject(
        model=model, dataloader=train, dataloader_test=test, learning_rate=cfg["lr"],
        temperature=1.0, momentum=0.9, metrics_saver=MemoryMetrics(), seed=1, **RC.RUN_KW)
    runner.optimizer = runner._make_optimizer(runner._params)
    fused = runner._fused_dense()
    assert fused is not None
    loss_f, lp_f, pot_f = fused.exact()
    got = [p.grad.clone() for p in runner._params]
    loss_a, lp_a, pot_a = runner._potential().exact(runner._batches())
    want = [p.grad.clone() for p in runner._params]
    assert loss_f.item() == pytest.approx(float(loss_a), rel=2e-6)
    assert lp_f.item() == pytest.approx(float(lp_a), rel=1e-9)
    assert pot_f.item() == pytest.approx(float(pot_a), rel=2e-6)
    for a, b in zip(got, want):
        scale = b.abs().max().item()
        assert (a - b).abs().max().item() <= 3e-6 * scale + 1e-10


def _chain(seed, chain_id, n_data=2000):
    "a runner of the BASELINE dense classifier on its own synthetic device-resident set, ready to leapfrog"
    import runner_cases as RC
    from bnn_priors_amd import inference_reject, models
    from bnn_priors_amd.storage import MemoryMetrics
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(n_data, 784, generator=g).to(DEV)
    y = torch.randint(0, 10, (n_data,), generator=g).to(DEV)
    train = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(x, y), batch_size=128, shuffle=True)
    test = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(x[:0], y[:0]), batch_size=128)
    torch.manual_seed(seed)
    model = models.get_model(x.cpu()[:2], torch.tensor([0, 9]), "classificationdensenet", width=50, depth=3,
                             weight_prior="gaussian", weight_scale=2 ** .5, bias_prior="gaussian", bias_scale=1.)
    models.he_initialize(model)
    model = model.to(DEV)
    runner = inference_reject.VerletSGLDRunnerReject(
        model=model, dataloader=train, dataloader_test=test, learning_rate=0.01, temperature=1.0, momentum=0.994,
        metrics_saver=MemoryMetrics(), seed=99, chain_id=chain_id, epochs_per_cycle=2, warmup_epochs=1,
        sample_epochs=1, skip=1, metrics_skip=10, cycles=1, precond_update=1, sampling_decay="cosine")
    runner.begin()
    return runner


def _snapshot(runner):
    opt = runner.optimizer
    return ([p.detach().clone() for p in runner._params],
            [opt.state[p]["momentum_buffer"].clone() for p in runner._params],
            [opt.state[p]["square_avg"].clone() for p in runner._params])


def test_chains_stepped_together_are_bit_identical_to_chains_stepped_alone():
    """sgmcmc_dense_step_multi (grid dimension y = chain): 3 chains with their own weights, data, Philox streams --
    after 25 lock-step leapfrog steps every chain's theta / m / v and its metric row equal, bit for bit, the same
    chain stepped alone through the single-chain direct path"""
    from bnn_priors_amd.fused_dense import MultiChainDense
    K, steps = 3, 25
    g = np.random.default_rng(5)
    idx = [[g.choice(2000, 128, replace=False).astype(np.int64) for _ in range(K)] for _ in range(steps)]

    alone, rows_alone = [], []
    for c in range(K):
        r = _chain(10 + c, c)
        f = r._fused_dense()
        assert f is not None and f.direct and f.split
        row = None
        for t in range(steps):
            out = f.replay(idx[t][c], metrics=(t % 10 == 9))
            row = out or row
            r.scheduler.step()
        r.optimizer.engine.flush()
        alone.append(_snapshot(r))
        rows_alone.append(row)

    runners = [_chain(10 + c, c) for c in range(K)]
    multi = MultiChainDense([r._fused_dense() for r in runners])
    rows = None
    for t in range(steps):
        out = multi.step(idx[t], metrics=(t % 10 == 9))
        rows = out or rows
        for r in runners:
            r.scheduler.step()
    for r in runners:
        r.optimizer.engine.flush()
    for c, r in enumerate(runners):
        for a, b in zip(_snapshot(r), alone[c]):
            for u, v in zip(a, b):
                assert torch.equal(u, v), f"chain {c}"
        for k in ("loss", "acc", "log_prior", "energy"):
            assert rows[c][k] == rows_alone[c][k], (c, k)
    # different chains did different things
    assert not torch.equal(_snapshot(runners[0])[0][0], _snapshot(runners[1])[0][0])
    # and a single-chain "batch" works too
    one = _chain(10, 0)
    m1 = MultiChainDense([one._fused_dense()])
    for t in range(steps):
        m1.step([idx[t][0]], metrics=(t % 10 == 9))
        one.scheduler.step()
    one.optimizer.engine.flush()
    for a, b in zip(_snapshot(one), alone[0]):
        for u, v in zip(a, b):
            assert torch.equal(u, v)
