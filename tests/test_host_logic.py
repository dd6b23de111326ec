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
    assert graphed.pick_group(None, 128, 3, "auto") == 4


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
