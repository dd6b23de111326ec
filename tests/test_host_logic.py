"""Host-side logic of round 4 that needs no GPU: the exact pass's group-size choice, the batch source's lazy minibatches
(same rows, same RNG consumption as its ordinary iteration), in-place filling of a consumer's buffers."""
import torch

from bnn_priors_amd import graphed
from bnn_priors_amd.inference import LazyBatch, _BatchSource


def test_group_size_leaves_the_fewest_leftovers_and_deals_groups_evenly():
    assert graphed.pick_group(390, 128, 3, "auto") == 5            # 78 groups = 26 per lane, nothing left over
    assert graphed.pick_group(468, 128, 3, "auto") == 4            # 117 groups = 39 per lane
    assert graphed.pick_group(50, 128, 3, "auto") == 5             # 10 groups, nothing left over
    assert graphed.pick_group(4, 32, 3, "auto") == 4               # one group
    assert graphed.pick_group(390, 256, 3, "auto") == 4            # 1,024 rows per launch at most (the fused head)
    assert graphed.pick_group(390, 128, 3, 16) == 8 and graphed.pick_group(390, 128, 3, 1) == 1
    assert graphed.pick_group(None, 128, 3, "auto") == 1             # unknown pass length: never grouped


def _source(shuffle, n=300):
    g = torch.Generator().manual_seed(3)
    x, y = torch.rand(n, 7, generator=g), torch.randint(0, 10, (n,), generator=g)
    ds = torch.utils.data.TensorDataset(x, y)
    return _BatchSource(torch.utils.data.DataLoader(ds, batch_size=128, shuffle=shuffle), torch.device("cpu")), x, y


def test_lazy_batches_are_the_ordinary_minibatches_and_consume_the_same_random_numbers():
    for shuffle in (True, False):
        torch.manual_seed(5)
        src, x, y = _source(shuffle)
        plain = [(a.clone(), b.clone()) for a, b in src]
        after = torch.get_rng_state()
        torch.manual_seed(5)
        src, _, _ = _source(shuffle)
        lazies = list(src.lazy_batches())
        assert torch.equal(torch.get_rng_state(), after)
        assert [len(b) for b, _ in lazies] == [128, 128, 44]
        for (lb, none), (px, py) in zip(lazies, plain):
            assert isinstance(lb, LazyBatch) and none is None and lb.shapes == (tuple(px.shape), tuple(py.shape))
            mx, my = lb.materialize()
            assert torch.equal(mx, px) and torch.equal(my, py)
        assert src.n_full_batches() == 2 and src.example()[0].shape == (128, 7)


def test_filling_writes_minibatches_into_the_consumers_buffers():
    torch.manual_seed(9)
    src, x, y = _source(True)
    plain = [(a.clone(), b.clone()) for a, b in src]
    torch.manual_seed(9)
    src, _, _ = _source(True)
    bx, by = torch.zeros(128, 7), torch.zeros(128, dtype=torch.int64)
    with src.filling(lambda rows: (bx, by) if rows == 128 else None):
        got = []
        for gx, gy in src:
            if len(gx) == 128:
                assert gx.data_ptr() == bx.data_ptr() and gy.data_ptr() == by.data_ptr()      # produced in place
            got.append((gx.clone(), gy.clone()))
    assert src._provider is None
    for (gx, gy), (px, py) in zip(got, plain):
        assert torch.equal(gx, px) and torch.equal(gy, py)


def test_a_group_provider_changes_nothing_where_a_group_cannot_be_gathered_at_once():
    """``filling(provider, group=(G, group_provider))``: on a source that cannot gather on the device (the CPU here) the
    group provider is asked, declined by ``_stage_into``, and every minibatch takes the ordinary route -- same rows."""
    torch.manual_seed(9)
    src, x, y = _source(True, n=700)
    plain = [(a.clone(), b.clone()) for a, b in src]
    torch.manual_seed(9)
    src, _, _ = _source(True, n=700)
    gx, gy = torch.zeros(256, 7), torch.zeros(256, dtype=torch.int64)
    asked = []
    with src.filling(lambda rows: None, group=(2, lambda rows: (asked.append(rows), (gx, gy))[1])):
        got = [(a.clone(), b.clone()) for a, b in src]
    assert src._provider is None and src._group is None
    assert asked and all(r == 256 for r in asked)
    assert len(got) == len(plain) == 6
    for (a, b), (px, py) in zip(got, plain):
        assert torch.equal(a, px) and torch.equal(b, py)


def test_gradient_tags_carry_the_masked_flag_and_die_with_the_tensor_version():
    "bnlink: (partial, n, masked) while the gradient tensor is untouched; an in-place change or a disabled link: None"
    from bnn_priors_amd import bnlink
    t = torch.zeros(4)
    part = torch.zeros(2, 3, 2, dtype=torch.float64)
    assert bnlink.sums_of(t) is None
    bnlink.tag_gradient(t, part, 3, masked=True)
    got = bnlink.sums_of(t)
    assert got[0] is part and got[1] == 3 and got[2] is True
    bnlink.tag_gradient(t, part, 3)
    assert bnlink.sums_of(t)[2] is False
    t.add_(1.0)
    assert bnlink.sums_of(t) is None
    out = torch.ones(2, 3)
    bnlink.tag_output(out, torch.zeros(2, 3), torch.zeros(2, 3))
    assert bnlink.source_of(out)[0] is not None and bnlink.source_of(torch.ones(2, 3)) == (None, None)
