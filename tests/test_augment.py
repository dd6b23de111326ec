"""On-device random crop + flip (csrc/augment_hip.inc) against its numpy restatement (oracle/augment.py):
bit for bit; plus the properties that stand in for the reference's torchvision stream (uniform offsets,
fair flips), and the runners' batch source using it."""
import numpy as np
import pytest
import torch

from oracle import augment as ref
from bnn_priors_amd import augment


def test_oracle_properties():
    rows = np.arange(20000)
    dx, dy, fl = ref.decisions(rows, seed=99, stream=3, draw=5, pad=4, flip=True)
    assert dx.min() == -4 and dx.max() == 4 and dy.min() == -4 and dy.max() == 4
    counts = np.bincount((dx + 4) * 9 + (dy + 4), minlength=81)
    assert counts.min() > 0.75 * 20000 / 81 and counts.max() < 1.25 * 20000 / 81      # uniform over the 81 offsets
    assert abs(fl.mean() - 0.5) < 0.02
    assert not np.array_equal(dx, ref.decisions(rows, 99, 3, 6, 4, True)[0])           # a new pass, new crops
    assert not np.array_equal(dx, ref.decisions(rows, 99, 4, 5, 4, True)[0])           # another chain
    assert np.array_equal(dx, ref.decisions(rows, 99, 3, 5, 4, True)[0])               # reproducible
    assert not ref.decisions(rows, 99, 3, 5, 4, False)[2].any()
    data = np.arange(2 * 3 * 4 * 6, dtype=np.float32).reshape(2, 3, 4, 6)
    assert np.array_equal(ref.gather(data, [1, 0, 1], 1, 0, 0, pad=0, flip=False), data[[1, 0, 1]])
    # a flip without a shift is a mirror image; zeros enter where the crop leaves the image
    rows = np.arange(64) % 2
    out = ref.gather(data, rows, 7, 0, 0, pad=0, flip=True)
    _, _, fl = ref.decisions(rows, 7, 0, 0, 0, True)
    for b in range(64):
        assert np.array_equal(out[b], data[rows[b]][:, :, ::-1] if fl[b] else data[rows[b]])
    out = ref.gather(data + 1, np.zeros(200, dtype=int), 7, 0, 0, pad=3, flip=False)
    dx, dy, _ = ref.decisions(np.zeros(200, dtype=int), 7, 0, 0, 3, False)
    assert np.array_equal((out == 0).any(axis=(1, 2, 3)), (dx != 0) | (dy != 0))


def _semantics_fixture():
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "augment_semantics.json")) as f:
        return json.load(f)


def test_oracle_reproduces_the_hand_computed_crop_and_flip_fixture():
    """tests/golden/augment_semantics.json: RandomCrop(padding) then RandomHorizontalFlip (cifar.py:158-159) written out
    by hand on a 4 x 4 image -- offset sign convention, zero fill, crop BEFORE flip, flip orientation.  The numpy
    restatement must give those arrays for the recorded (row, draw); the kernel is held to the same arrays below."""
    fx = _semantics_fixture()
    data = np.asarray(fx["images"], dtype=np.float32)[:, None]          # [2, 1, 4, 4]
    for case in fx["cases"]:
        dx, dy, fl = ref.decisions(np.array([case["row"]]), fx["seed"], fx["stream"], case["draw"], fx["pad"], True)
        assert (int(dx[0]), int(dy[0]), bool(fl[0])) == (case["dx"], case["dy"], case["flipped"]), case["name"]
        got = ref.gather(data, [case["row"]], fx["seed"], fx["stream"], case["draw"], fx["pad"], True)
        assert np.array_equal(got[0, 0], np.asarray(case["out"], dtype=np.float32)), case["name"]
    # the offsets cover exactly {-pad .. pad}: nine values for the reference's padding = 4
    dx, dy, _ = ref.decisions(np.arange(5000), fx["seed"], fx["stream"], 0, 4, True)
    assert sorted(set(dx.tolist())) == list(range(-4, 5)) and sorted(set(dy.tolist())) == list(range(-4, 5))


@pytest.mark.gpu
def test_kernel_reproduces_the_hand_computed_crop_and_flip_fixture():
    fx = _semantics_fixture()
    data = torch.tensor(fx["images"], dtype=torch.float32)[:, None].cuda()
    aug = augment.RandomCropFlip(pad=fx["pad"], flip=True, seed=fx["seed"], stream=fx["stream"])
    for case in fx["cases"]:
        got = aug.gather(data, torch.tensor([case["row"]]).cuda(), case["draw"]).cpu().numpy()
        assert np.array_equal(got[0, 0], np.asarray(case["out"], dtype=np.float32)), case["name"]


@pytest.mark.gpu
@pytest.mark.parametrize("shape,pad,flip", [((500, 3, 32, 32), 4, True), ((64, 1, 28, 28), 2, False),
                                            ((9, 5, 6, 10), 0, True), ((9, 2, 7, 5), 3, True)])
def test_kernel_matches_oracle_bit_for_bit(shape, pad, flip):
    g = torch.Generator().manual_seed(sum(shape))
    data = torch.randn(shape, generator=g)
    idx = torch.randint(0, shape[0], (128,), generator=g)
    aug = augment.RandomCropFlip(pad=pad, flip=flip, seed=0x1234_5678_9ABC, stream=5)
    for draw in (0, 1, 2 ** 33 + 7):
        got = aug.gather(data.cuda(), idx.cuda(), draw).cpu().numpy()
        want = ref.gather(data.numpy(), idx.numpy(), aug.seed, aug.stream, draw, pad, flip)
        assert np.array_equal(got, want)
    fill = [0.5 - c for c in range(shape[1])]             # the reference's border on normalised data: -mean/std
    filled = augment.RandomCropFlip(pad=pad, flip=flip, seed=7, stream=1, fill=fill)
    assert np.array_equal(filled.gather(data.cuda(), idx.cuda(), 3).cpu().numpy(),
                          ref.gather(data.numpy(), idx.numpy(), 7, 1, 3, pad, flip, fill=fill))
    assert aug.gather(data.cuda(), idx[:0].cuda(), 0).shape == (0,) + shape[1:]
    with pytest.raises(ValueError):
        aug.gather(data, idx, 0)                      # CPU tensor: no fallback


@pytest.mark.gpu
def test_batch_source_augments_every_traversal_anew():
    from bnn_priors_amd.inference import _BatchSource
    g = torch.Generator().manual_seed(0)
    x, y = torch.randn(40, 3, 8, 8, generator=g).cuda(), torch.randint(0, 10, (40,), generator=g).cuda()
    ds = augment.AugmentedTensorDataset(x, y, augment.RandomCropFlip(pad=2, flip=True, seed=11, stream=1))
    dl = torch.utils.data.DataLoader(ds, batch_size=16, shuffle=False)
    src = _BatchSource(dl, torch.device("cuda:0"))
    assert src.fast and src.augment is ds.augment
    passes = []
    for _ in range(2):
        xs, ys = zip(*list(src))
        assert [len(b) for b in xs] == [16, 16, 8] and torch.equal(torch.cat(ys), y)
        passes.append(torch.cat(xs))
    assert ds.draw == 2 and not torch.equal(passes[0], passes[1])
    want = ref.gather(x.cpu().numpy(), np.arange(40), 11, 1, 2, 2, True)
    assert np.array_equal(passes[1].cpu().numpy(), want)
    # item access (a plain DataLoader) uses the current draw
    xi, yi = ds[7]
    assert np.array_equal(xi.cpu().numpy(), want[7]) and yi == y[7]
    # shuffled: same samples, same augmentation per ROW whatever the batch order
    dl2 = torch.utils.data.DataLoader(ds, batch_size=16, shuffle=True, generator=torch.Generator().manual_seed(3))
    src2 = _BatchSource(dl2, torch.device("cuda:0"))
    xs, ys = zip(*list(src2))
    # (the DataLoader draws its base seed from the same generator before the sampler's permutation)
    gen = torch.Generator().manual_seed(3)
    torch.empty((), dtype=torch.int64).random_(generator=gen)
    perm = torch.randperm(40, generator=gen)
    want3 = ref.gather(x.cpu().numpy(), perm.numpy(), 11, 1, 3, 2, True)
    assert np.array_equal(torch.cat(xs).cpu().numpy(), want3)


@pytest.mark.gpu
def test_reject_runner_runs_on_an_augmented_set():
    from bnn_priors_amd import inference_reject, models
    from bnn_priors_amd.storage import MemoryMetrics
    g = torch.Generator().manual_seed(1)
    x = torch.randn(64, 3, 32, 32, generator=g).cuda()
    y = torch.randint(0, 10, (64,), generator=g).cuda()
    ds = augment.AugmentedTensorDataset(x, y, augment.RandomCropFlip(seed=5))
    train = torch.utils.data.DataLoader(ds, batch_size=16, shuffle=True)
    test = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(x[:16], y[:16]), batch_size=16)
    torch.manual_seed(0)
    model = models.get_model(x, y, "googleresnet", width=50, depth=3, weight_prior="gaussian", weight_loc=0.,
                             weight_scale=2 ** .5, bias_prior="gaussian", bias_loc=0., bias_scale=1.,
                             batchnorm=True, weight_prior_params={}, bias_prior_params={}).cuda()
    metrics = MemoryMetrics()
    runner = inference_reject.VerletSGLDRunnerReject(
        model=model, dataloader=train, dataloader_test=test, epochs_per_cycle=2, warmup_epochs=1, sample_epochs=1,
        learning_rate=1e-3, skip=1, metrics_skip=2, temperature=1.0, momentum=0.98, sampling_decay="cosine", cycles=1,
        precond_update=1, metrics_saver=metrics, model_saver=None, reject_samples=True, seed=3, chain_id=0, cycle_seed=9)
    runner.run()
    assert ds.draw >= 4                      # begin() + 2 epochs + the exact pass, each a new traversal
    _, loss = metrics.column("loss")
    assert np.isfinite(loss[~np.isnan(loss)]).all()
    assert runner.get_samples()["net.module.0.weight_prior.p"].shape[0] == 1


@pytest.mark.gpu
def test_kernel_output_obeys_the_reference_pipelines_contract():
    """The reference augments with torchvision's RandomCrop(32, padding=4) + RandomHorizontalFlip()
    (bnn_priors/data/CIFAR/cifar.py:136-172): every output is a 32x32 window of the ZERO-padded (4 pixels
    per side) image at an offset uniform over {0..8}^2, mirrored with probability 1/2, independently per
    image and per traversal.  torchvision's random stream cannot be replayed, so the contract is checked on
    the HIP kernel's OUTPUT itself -- by exhaustive search over the 81 x 2 candidate windows built with
    plain torch ops, not through the numpy restatement of the kernel."""
    g = torch.Generator().manual_seed(12)
    n = 4000
    data = torch.randn(n, 3, 32, 32, generator=g)
    padded = torch.nn.functional.pad(data, (4, 4, 4, 4))                     # zero padding, as RandomCrop's default fill
    aug = augment.RandomCropFlip(pad=4, flip=True, seed=2024, stream=0)
    idx = torch.arange(n)
    seen = []
    for draw in (0, 1):
        out = aug.gather(data.cuda(), idx.cuda(), draw).cpu()
        found = torch.full((n, 3), -1, dtype=torch.int64)
        for oy in range(9):
            for ox in range(9):
                win = padded[:, :, oy:oy + 32, ox:ox + 32]
                for fl, cand in ((0, win), (1, win.flip(-1))):
                    hit = (cand == out).flatten(1).all(1) & (found[:, 0] < 0)
                    found[hit] = torch.tensor([ox, oy, fl])
        assert (found[:, 0] >= 0).all()                      # every output IS such a window
        ox, oy, fl = found[:, 0].numpy(), found[:, 1].numpy(), found[:, 2].numpy()
        counts = np.bincount(ox * 9 + oy, minlength=81)
        expect = n / 81
        assert counts.min() > 0.55 * expect and counts.max() < 1.5 * expect
        chi2 = ((counts - expect) ** 2 / expect).sum()       # 80 degrees of freedom: mean 80, sd 12.6
        assert chi2 < 80 + 5 * 12.65
        assert abs(fl.mean() - 0.5) < 4 * 0.5 / n ** .5
        assert abs(np.corrcoef(ox, oy)[0, 1]) < 0.06 and abs(np.corrcoef(ox, fl)[0, 1]) < 0.06
        seen.append(found)
    same = (seen[0] == seen[1]).all(1).float().mean().item()
    assert same < 3.0 / 162                                   # a new traversal draws new windows
    other = augment.RandomCropFlip(pad=4, flip=True, seed=2024, stream=1)       # another chain
    out_b = other.gather(data.cuda(), idx.cuda(), 0).cpu()
    out_a = aug.gather(data.cuda(), idx.cuda(), 0).cpu()
    assert (out_a == out_b).flatten(1).all(1).float().mean().item() < 3.0 / 162


@pytest.mark.gpu
def test_stage_batch_copies_up_to_three_buffers_in_one_launch_bit_exactly():
    """sgmcmc_stage_batch: minibatch + pinned argument block into a captured step's static inputs; pure data
    movement, so bit-exact -- including a size that is not a multiple of the 16-byte vector and of the
    4096-byte block, and a pinned-host source."""
    import ctypes
    from bnn_priors_amd import _hip
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(3)
    x = torch.randn((128, 3, 32, 32), generator=g, device=dev)
    y = torch.randint(0, 10, (128,), generator=g, device=dev)
    args = torch.randint(0, 255, (204,), dtype=torch.uint8).pin_memory()
    for n in (3, 2, 1):
        xd, yd, ad = torch.zeros_like(x), torch.zeros_like(y), torch.zeros(204, dtype=torch.uint8, device=dev)
        srcs, dsts = [x, y, args][:n], [xd, yd, ad][:n]
        src = (ctypes.c_void_p * n)(*[t.data_ptr() for t in srcs])
        dst = (ctypes.c_void_p * n)(*[t.data_ptr() for t in dsts])
        nb = (ctypes.c_int64 * n)(*[t.numel() * t.element_size() for t in srcs])
        _hip.check(_hip.lib().sgmcmc_stage_batch(src, dst, nb, n, None, None, torch.cuda.current_stream(dev).cuda_stream), "stage")
        torch.cuda.synchronize(dev)
        for s, d in zip(srcs, dsts):
            assert torch.equal(s.to(dev), d)
    # rejected: a size that is not a multiple of 4, a misaligned pointer
    src = (ctypes.c_void_p * 1)(x.data_ptr() + 4)
    dst = (ctypes.c_void_p * 1)(xd.data_ptr())
    nb = (ctypes.c_int64 * 1)(64)
    assert _hip.lib().sgmcmc_stage_batch(src, dst, nb, 1, None, None, 0) != 0
    src = (ctypes.c_void_p * 1)(x.data_ptr())
    nb = (ctypes.c_int64 * 1)(63)
    assert _hip.lib().sgmcmc_stage_batch(src, dst, nb, 1, None, None, 0) != 0


@pytest.mark.gpu
@pytest.mark.parametrize("augmented", [True, False])
def test_lazy_batches_staged_in_one_launch_are_the_gathered_tensors_bit_for_bit(augmented):
    """LazyBatch.stage (gather [+ crop / flip] + labels + riding copies in ONE launch: csrc/augment_hip.inc,
    gather_stage_kernel) leaves exactly what the batch source's ordinary iteration yields -- same rows, same draw --
    for image sets with augmentation and for plain row gathers (flattened MNIST-shaped rows), ragged last batch
    included, and consumes the loader's RNG identically."""
    from bnn_priors_amd.inference import LazyBatch, _BatchSource
    dev = "cuda:0"
    g = torch.Generator().manual_seed(3)
    n = 300
    x = (torch.randn(n, 3, 32, 32, generator=g) if augmented else torch.rand(n, 784, generator=g)).to(dev)
    y = torch.randint(0, 10, (n,), generator=g).to(dev)

    def source():
        if augmented:
            ds = augment.AugmentedTensorDataset(x, y, augment.RandomCropFlip(pad=4, flip=True, seed=11, stream=2,
                                                                             fill=[0.1, -0.2, 0.3]))
        else:
            ds = torch.utils.data.TensorDataset(x, y)
        return _BatchSource(torch.utils.data.DataLoader(ds, batch_size=128, shuffle=True), torch.device(dev))

    torch.manual_seed(5)
    plain = [(a.clone(), b.clone()) for a, b in source()]
    state_after = torch.get_rng_state()
    torch.manual_seed(5)
    lazies = list(source().lazy_batches())
    assert torch.equal(torch.get_rng_state(), state_after)                  # same RNG consumption
    assert [len(b) for b, _ in lazies] == [128, 128, 44] and all(t is None for _, t in lazies)
    extra_src = torch.arange(64, dtype=torch.float32, device=dev)
    for (lb, _), (px, py) in zip(lazies, plain):
        assert isinstance(lb, LazyBatch) and lb.shapes == (tuple(px.shape), tuple(py.shape))
        mx, my = lb.materialize()
        assert torch.equal(mx, px) and torch.equal(my, py)
        xd, yd = torch.full_like(px, -7.0), torch.full_like(py, -7)
        extra_dst = torch.zeros_like(extra_src)
        assert lb.stageable(xd, yd)
        lb.stage(xd, yd, [(extra_src.data_ptr(), extra_dst.data_ptr(), extra_src.numel() * 4)], None, None,
                 torch.cuda.current_stream().cuda_stream)
        assert torch.equal(xd, px) and torch.equal(yd, py) and torch.equal(extra_dst, extra_src)
        lb.stage(xd.zero_(), yd.zero_(), [], None, None, torch.cuda.current_stream().cuda_stream)      # no riders
        assert torch.equal(xd, px) and torch.equal(yd, py)
