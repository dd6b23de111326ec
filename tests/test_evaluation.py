"""Posterior-predictive evaluation (reference bnn_priors/exp_utils.py:250-340): the grouped
multi-sample path -- all stored samples stacked, one vmapped forward per test batch, fp64 log-mean-exp on the device
-- against the sample-by-sample statement of the reference's loop, for the three BASELINE networks."""
import numpy as np
import pytest
import torch

from bnn_priors_amd import evaluation as ev
from bnn_priors_amd import models


def _setup(name, n=96, E=5, device="cpu"):
    torch.manual_seed(0)
    x = torch.rand(n, 3, 32, 32) if name == "googleresnet" else torch.rand(n, 784)
    y = torch.arange(n) % 10
    net = models.get_model(x, y, name, width=50, depth=3, weight_prior="gaussian", weight_loc=0., weight_scale=2 ** .5,
                           bias_prior="gaussian", bias_loc=0., bias_scale=1., batchnorm=True, weight_prior_params={},
                           bias_prior_params={})
    models.he_initialize(net)
    net = net.to(device).eval()
    g = torch.Generator().manual_seed(1)
    samples = {}
    for k, v in net.state_dict().items():
        if v.is_floating_point():
            noise = 0.05 * torch.randn((E,) + tuple(v.shape), generator=g).to(device)
            samples[k] = (v.unsqueeze(0) + noise * v.abs().mean()).clone()
            if k.endswith("running_var"):
                samples[k] = samples[k].abs() + 0.5
        else:
            samples[k] = v.unsqueeze(0).repeat((E,) + (1,) * v.dim())
    loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(x.to(device), y.to(device)), batch_size=32)
    return net, loader, samples, y


@pytest.mark.parametrize("name", ["classificationdensenet", "classificationconvnet", "googleresnet"])
def test_grouped_tables_equal_the_per_sample_loop_cpu(name):
    net, loader, samples, y = _setup(name)
    lps_b, acc_b = ev._predictive_tables_batched(net, loader, samples, y, 5, 10)
    old = ev.BATCHED
    ev.BATCHED = False
    try:
        lps, acc, _, kind = ev.predictive_tables(net, loader, samples)
    finally:
        ev.BATCHED = old
    assert kind == "cat"
    torch.testing.assert_close(lps_b, lps, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(acc_b, acc, rtol=1e-4, atol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["classificationdensenet", "classificationconvnet", "googleresnet"])
def test_grouped_evaluation_on_gpu_matches_the_loop_and_is_taken(name, monkeypatch):
    net, loader, samples, y = _setup(name, device="cuda:0")
    used = []
    real = ev._predictive_tables_batched
    monkeypatch.setattr(ev, "_predictive_tables_batched", lambda *a: (used.append(1), real(*a))[1])
    want = ev.evaluate_model(net, loader, samples)          # the default: this repo's kernels, sample by sample
    assert not used and not ev.BATCHED                      # (the library's batched layers are not entered by default)
    monkeypatch.setattr(ev, "BATCHED", True)
    got = ev.evaluate_model(net, loader, samples)
    assert used                                             # the grouped path ran when asked for (and did not fall back)
    for k in want:
        assert got[k] == pytest.approx(want[k], rel=2e-4, abs=2e-4), k
    from bnn_priors_amd import conv
    assert conv.ENABLED                                     # the layer switches are restored


def test_row_groups_concatenate_consecutive_batches_in_order():
    x = torch.arange(100, dtype=torch.float32).view(100, 1)
    y = torch.arange(100)
    loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(x, y), batch_size=16)
    groups = list(ev._row_groups(loader, "cpu", 40))                  # an in-order TensorDataset loader: sliced, not iterated
    assert [len(gx) for gx, _ in groups] == [32, 32, 32, 4]           # whole batches per group; the ragged batch on its own
    assert torch.equal(torch.cat([gx for gx, _ in groups]), x) and torch.equal(torch.cat([gy for _, gy in groups]), y)
    shuffled = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(x, y), batch_size=16, sampler=list(range(100)))
    groups = list(ev._row_groups(shuffled, "cpu", 40))                # any other loader is iterated: 16 + 16 | 16 + 16 | 16 + 16 + 4
    assert [len(gx) for gx, _ in groups] == [32, 32, 36]
    assert torch.equal(torch.cat([gx for gx, _ in groups]), x) and torch.equal(torch.cat([gy for _, gy in groups]), y)
    assert [len(gx) for gx, _ in ev._row_groups(loader, "cpu", 0)] == [16] * 6 + [4]      # 0: the loader's own batches
    assert [len(gx) for gx, _ in ev._row_groups(loader, "cpu", 8)] == [16] * 6 + [4]      # never splits a batch


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["classificationconvnet", "googleresnet"])
def test_evaluating_several_batches_per_forward_gives_the_same_rows(name, monkeypatch):
    """E = 1 (the per-epoch evaluation): the loader's batches concatenated up to EVAL_ROWS rows per forward -- the
    per-image kernels give every row the same bits as batch-by-batch evaluation"""
    net, loader, samples, y = _setup(name, n=200, E=1, device="cuda:0")
    tables = {}
    for rows in (0, 96, 1024):
        monkeypatch.setattr(ev, "EVAL_ROWS", rows)
        lps, acc, labels, kind = ev.predictive_tables(net, loader, samples)
        tables[rows] = (lps.clone(), acc.clone())
    for rows in (96, 1024):
        assert torch.equal(tables[rows][0], tables[0][0]) and torch.equal(tables[rows][1], tables[0][1]), rows


@pytest.mark.gpu
def test_a_model_can_be_copied_and_pickled_after_evaluation_and_new_buffers_get_a_new_capture():
    """the captured evaluation forwards live outside the module (a hipGraph is not picklable): deepcopy / torch.save of
    an evaluated model work, and a re-registered BatchNorm buffer is not replayed from the old capture's address"""
    import copy
    import io
    net, loader, samples, y = _setup("googleresnet", n=64, E=1, device="cuda:0")
    lps0, acc0, _, _ = ev.predictive_tables(net, loader, samples)
    assert ev._eval_graphs.get(net)                                  # a capture exists ...
    assert "_eval_graphs" not in net.__dict__                        # ... and not inside the module
    twin = copy.deepcopy(net)
    buf = io.BytesIO()
    torch.save(net, buf)
    lps1, _, _, _ = ev.predictive_tables(twin, loader, samples)
    assert torch.equal(lps0, lps1)
    # a fresh running_var tensor (another storage): the key changes, the result follows the new statistics
    bn = next(m for m in net.modules() if isinstance(m, torch.nn.BatchNorm2d))
    n_before = len(ev._eval_graphs[net])
    bn.running_var = bn.running_var.clone() * 4.0
    one = dict(samples)
    first = next(k for k in net.state_dict() if k.endswith("running_var"))     # (module order = state_dict order)
    one[first] = samples[first] * 4.0
    lps2, _, _, _ = ev.predictive_tables(net, loader, one)
    assert len(ev._eval_graphs[net]) == n_before + 1
    assert not torch.equal(lps0, lps2)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [256, 5])
def test_eval_batchnorm_in_the_convolution_epilogue_changes_no_bit(n, monkeypatch):
    """Evaluation mode: conv3x3 -> BatchNorm (running statistics) -> (+ shortcut) -> ReLU is one launch
    (sgmcmc_conv3x3_bn_eval: the BatchNorm applied to the accumulator tile); the logits of googleresnet carry the bits of
    the separate launches (conv3x3 + bn_eval), and agree with PyTorch's own layers in float64."""
    from bnn_priors_amd import conv
    torch.manual_seed(3)
    x = torch.randn(n, 3, 32, 32).cuda()
    net = models.get_model(x.cpu()[:2], torch.tensor([0, 9]), "googleresnet", width=50, depth=3, weight_prior="gaussian",
                           weight_scale=2 ** .5, bias_prior="gaussian", bias_scale=1.).cuda()
    models.he_initialize(net)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                m.running_mean.normal_(0, 0.2)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.2)
    net.eval()
    calls = {"n": 0}
    real = conv.conv3x3_bn_eval
    monkeypatch.setattr(conv, "conv3x3_bn_eval", lambda *a, **k: (calls.__setitem__("n", calls["n"] + 1), real(*a, **k))[1])
    with torch.no_grad():
        fused = net.net(x)
        assert calls["n"] == 16            # every 3x3 of the trunk but the stem and the two strided ones
        monkeypatch.setattr(conv, "CONV_BN_EVAL", False)
        plain = net.net(x)
        assert calls["n"] == 16
    assert torch.equal(fused, plain)
    ref = net.double()
    monkeypatch.setattr(conv, "ENABLED", False)
    with torch.no_grad():
        want = ref.net(x.double())
    torch.testing.assert_close(fused.double(), want, rtol=1e-4, atol=1e-4)


def test_resident_copy_of_a_host_test_set_follows_the_loader_object_and_its_data():
    """the device copy of a host-resident test set is cached per loader OBJECT (weakly) and invalidated by
    in-place edits of the host tensors -- never by id() / data_ptr() alone, which a later loader can reuse."""
    import gc
    import torch
    from bnn_priors_amd import evaluation as ev

    def groups(loader):
        return [(x.clone(), y.clone()) for x, y in ev._row_groups(loader, "cpu", 1024)]

    class _Fake(str):          # a device that is not the tensors': forces the copy branch on a CPU-only box
        pass
    x, y = torch.arange(20.).view(10, 2), torch.arange(10)
    loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(x, y), batch_size=4)
    # (on one device nothing is copied and nothing cached)
    assert torch.equal(groups(loader)[0][0], x) and len(ev._resident_cache) == 0
    calls = []
    real_to = torch.Tensor.to

    def counting_to(self, *a, **k):
        calls.append(1)
        return real_to(self, "cpu").clone()
    dev = torch.device("meta")          # stands in for "another device": _resident_on only compares and calls .to()
    torch.Tensor.to = counting_to
    try:
        a1 = ev._resident_on(loader, (x[:10], y[:10]), dev)
        a2 = ev._resident_on(loader, (x[:10], y[:10]), dev)
        assert a1[0] is a2[0] and len(calls) == 2                    # second call: the cached copy
        x.mul_(2)                                                    # in-place edit of the host tensor
        a3 = ev._resident_on(loader, (x[:10], y[:10]), dev)
        assert a3[0] is not a1[0] and torch.equal(a3[0], x) and len(calls) == 4
        loader2 = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(x.clone(), y.clone()), batch_size=4)
        b = ev._resident_on(loader2, (loader2.dataset.tensors[0][:10], loader2.dataset.tensors[1][:10]), dev)
        assert b[0] is not a3[0] and len(ev._resident_cache) == 2
        del loader2, b
        gc.collect()
        assert len(ev._resident_cache) == 1                          # the entry went with its loader
    finally:
        torch.Tensor.to = real_to
        ev._resident_cache.clear()


# ---- an independent pin (round 6): the tables against a float64 plain-torch evaluation of the ORACLE's nets -----------
def _oracle_tables(name, samples, x, y, E):
    """lps [E, N], acc [E, N, C] in float64 from oracle/nets.py's restatement of the network (torch.nn.functional layers,
    evaluation-mode BatchNorm) loaded with the same state dicts -- statement by statement what exp_utils.py:250-283 does:
    load sample e, preds = model(x), lps[e] = preds.log_prob(y), acc[e] = preds.logits (normalised)."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import nets as oracle_nets
    net = oracle_nets.BUILDERS[name](weight_prior="gaussian").double().eval()
    lps, acc = [], []
    for e in range(E):
        state = {k.replace("net.module.", "net."): v[e].detach().cpu().to(torch.float64 if v.is_floating_point() else v.dtype)
                 for k, v in samples.items()}
        net.load_state_dict(state)
        with torch.no_grad():
            preds = net(x.double())
        lps.append(preds.log_prob(y))
        acc.append(preds.logits)
    return torch.stack(lps), torch.stack(acc)


def _check_against_oracle(name, device):
    net, loader, samples, y = _setup(name, n=96, E=3, device=device)
    x = torch.cat([bx for bx, _ in loader]).cpu()
    lps, acc, labels, kind = ev.predictive_tables(net, loader, samples)
    want_lps, want_acc = _oracle_tables(name, samples, x, y, 3)
    assert kind == "cat" and torch.equal(labels.cpu(), y)
    torch.testing.assert_close(lps.cpu(), want_lps, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(acc.cpu(), want_acc, rtol=1e-5, atol=1e-5)
    # ... and the ensemble numbers the runner logs, from the oracle's tables by the reference's formula (exp_utils.py:300-321)
    got = ev.evaluate_model(net, loader, samples)
    lp_ens = (want_lps.logsumexp(0) - np.log(3)).mean().item()
    ens = want_acc.logsumexp(0) - np.log(3)
    assert got["lp_ensemble"] == pytest.approx(lp_ens, rel=1e-5, abs=1e-5)
    assert got["lp_last"] == pytest.approx(want_lps[-1].mean().item(), rel=1e-5, abs=1e-5)
    assert got["acc_ensemble"] == pytest.approx(ens.argmax(-1).eq(y).double().mean().item(), abs=1e-9)
    assert got["acc_last"] == pytest.approx(want_acc[-1].argmax(-1).eq(y).double().mean().item(), abs=1e-9)


@pytest.mark.parametrize("name", ["classificationdensenet", "classificationconvnet", "googleresnet"])
def test_tables_match_a_float64_evaluation_of_the_oracle_nets_cpu(name):
    _check_against_oracle(name, "cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["classificationdensenet", "classificationconvnet", "googleresnet"])
def test_tables_match_a_float64_evaluation_of_the_oracle_nets_gpu(name):
    "the HIP evaluation path (captured forwards, conv + BatchNorm-eval epilogue kernels) against the same oracle tables"
    _check_against_oracle(name, "cuda:0")
