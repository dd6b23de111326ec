"""Multi-chain path: independent chains, one exchange at the end.
The exchange -- posterior-predictive ensemble over all chains' samples via two small
all-reduces -- is exercised here with world_size 2 on the gloo backend (CPU); on the GPUs the
same code runs over RCCL.  The reference computes the ensemble from the gathered tables
(exp_utils.py:300-321); the test checks both are the same numbers."""
import math
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _tables(rank, E, N, C):
    g = torch.Generator().manual_seed(100 + rank)
    logits = torch.randn(E, N, C, generator=g, dtype=torch.float64) * 3
    acc = logits - logits.logsumexp(-1, keepdim=True)
    y = torch.randint(0, C, (N,), generator=torch.Generator().manual_seed(7))
    lps = acc.gather(-1, y.view(1, N, 1).expand(E, N, 1)).squeeze(-1)
    return lps, acc, y


def _worker(rank, world, port, sizes, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from bnn_priors_amd.evaluation import ensemble_across_chains
        lps, acc, y = _tables(rank, sizes[rank], 50, 10)
        lp, ens = ensemble_across_chains(lps, acc)
        if rank == 0:
            torch.save((lp, ens), out)
    finally:
        dist.destroy_process_group()


def test_ensemble_across_chains_gloo_world2(tmp_path):
    sizes = [3, 5]        # chains may hold different numbers of samples
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, _free_port(), sizes, out), nprocs=2, join=True)
    lp, ens = torch.load(out)
    parts = [_tables(r, sizes[r], 50, 10) for r in range(2)]
    lps = torch.cat([p[0] for p in parts])
    acc = torch.cat([p[1] for p in parts])
    ref_lp = lps.logsumexp(0) - math.log(lps.shape[0])       # exp_utils.py:300-305
    ref_ens = acc.logsumexp(0) - math.log(acc.shape[0])      # exp_utils.py:309
    torch.testing.assert_close(lp, ref_lp, rtol=1e-12, atol=1e-12)
    torch.testing.assert_close(ens, ref_ens, rtol=1e-12, atol=1e-12)


def test_ensemble_single_process_matches_evaluate_model():
    from bnn_priors_amd.evaluation import ensemble_across_chains
    lps, acc, y = _tables(0, 4, 30, 10)
    lp, ens = ensemble_across_chains(lps, acc)
    torch.testing.assert_close(lp, lps.logsumexp(0) - math.log(4))
    torch.testing.assert_close(ens, acc.logsumexp(0) - math.log(4))


def test_chain_streams_are_disjoint():
    "chain_id selects a Philox stream: same seed, different chains -> unrelated noise"
    from oracle import noise
    a = noise.normals(1234, 0, 5, 0, 0, 4096)
    b = noise.normals(1234, 1, 5, 0, 0, 4096)
    assert abs(float((a * b).mean())) < 0.06 and not (a == b).any()


def _gather_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from bnn_priors_amd.evaluation import gather_samples
        E = 2 + rank
        samples = {"a.p": torch.full((E, 3, 2), float(rank)), "steps": torch.arange(E) + 10 * rank}
        got = gather_samples(samples)
        if rank == 0:
            torch.save(got, out)
        else:
            assert got is None
    finally:
        dist.destroy_process_group()


def test_gather_samples_gloo_world2(tmp_path):
    out = str(tmp_path / "g.pt")
    mp.spawn(_gather_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    assert got["a.p"].shape == (5, 3, 2)
    assert got["a.p"][:2].eq(0).all() and got["a.p"][2:].eq(1).all()
    assert got["steps"].tolist() == [0, 1, 10, 11, 12]


# ---- several chains on ONE GPU: streams that own a hardware queue (bnn_priors_amd/multichain.py) ----------------------
@pytest.mark.gpu
def test_concurrent_streams_really_run_side_by_side():
    """multichain.concurrent_streams: the streams it hands out are distinct, none of them is an excluded one, the set is
    measured once per device -- and K chains of spin kernels on them take one chain's time, not K (two streams on one
    hardware queue, what round 4's chains 3 and 4 had, take twice as long)."""
    from bnn_priors_amd import multichain
    dev = torch.device("cuda", 0)
    picked = multichain.concurrent_streams(3, dev)
    assert 2 <= len(picked) <= 3 and len({s.cuda_stream for s in picked}) == len(picked)
    again = multichain.concurrent_streams(3, dev)
    assert [s.cuda_stream for s in again] == [s.cuda_stream for s in picked]              # memoised
    rest = multichain.concurrent_streams(3, dev, exclude=[picked[0]])
    assert picked[0].cuda_stream not in {s.cuda_stream for s in rest}
    alone = min(multichain._spin_time(picked[:1], dev, 400_000, 3) for _ in range(3))
    together = min(multichain._spin_time(picked, dev, 400_000, 3) for _ in range(3))
    assert together < 1.5 * alone, (alone, together)
    assert len(multichain.spread(picked, 5)) == 5


@pytest.mark.gpu
def test_lanes_are_distinct_streams_and_never_a_chains_main_stream(monkeypatch):
    """multichain.lanes (round 6): the exact pass's helper streams are DISTINCT and are none of the streams reserved for a
    chain's own launches; when the measured pool has fewer left, fresh streams make up the number instead of one stream
    being handed out twice.  SGMCMC_STREAM_PROBE=0 hands out fresh streams without measuring."""
    from bnn_priors_amd import multichain
    dev = torch.device("cuda", 0)
    mains = multichain.concurrent_streams(2, dev)
    monkeypatch.setattr(multichain, "_reserved", {})
    multichain.reserve(mains, dev)
    got = multichain.lanes(3, dev, exclude=[torch.cuda.current_stream(dev)])
    handles = [s.cuda_stream for s in got]
    assert len(set(handles)) == 3 and not set(handles) & {s.cuda_stream for s in mains}
    assert torch.cuda.current_stream(dev).cuda_stream not in handles
    many = multichain.lanes(12, dev)            # more than any pool holds: still all distinct
    assert len({s.cuda_stream for s in many}) == 12
    monkeypatch.setenv("SGMCMC_STREAM_PROBE", "0")
    fresh = multichain.concurrent_streams(3, dev)
    assert len({s.cuda_stream for s in fresh}) == 3
