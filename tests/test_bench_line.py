"""bench.py's stdout contract: the ONE line the driver parses stays below 4 KB and carries the fields the record
needs (round 4's 26 KB line was not parsed); the headline roofline row is the device symbol with the largest time per
step; committed per-kernel measurements taken on other kernel sources are marked stale, not mixed in."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

FULL = os.path.join(ROOT, "profiles", "r04_bench_googleresnet_driver_args.json")


def _full_record():
    "round 4's full one-GPU record (every table this script can emit), as the stub of a run"
    with open(FULL) as f:
        out = json.loads(f.read().strip().splitlines()[-1])
    out.update(ranks_seen=1, backend=None, devices=[{"rank": 0, "local_rank": 0, "index": 0, "device": "AMD Instinct MI355X"}])
    return out


def test_compact_line_is_small_and_complete():
    out = _full_record()
    top, other = bench.headline_rooflines(out["roofline_kernels"])
    out["roofline"] = top
    out["roofline_mfma" if other["bound"] == "mfma" else "roofline_hbm"] = other
    out["roofline_step"] = bench.step_roofline(out["roofline_kernels"], out["ms_per_step"])
    out["cpu_baseline"].update(cpu_model="AMD EPYC 9575F 64-Core Processor", block_steps_per_s=[20.1, 21.2, 20.7],
                               threads_calibration_steps_per_s={str(2 ** k): 10.0 + k for k in range(8)})
    text = bench.compact_line(out, os.path.join(ROOT, "bench_detail.json"))
    assert len(text) < 4096 and "\n" not in text
    line = json.loads(text)
    # round 6: the headline fraction is the IN-STEP one, the isolated-launch figure beside it; the whole step against the
    # MFMA peak; the CPU leg says what it ran on and how steady it was
    assert line["roofline"]["frac_source"] == "in_step" and line["roofline"]["frac"] <= line["roofline"]["frac_isolated"] + 0.05
    assert line["roofline"]["achieved"] == pytest.approx(line["roofline"]["frac"] * line["roofline"]["peak"], rel=2e-3)
    for k in ("flops_per_step", "achieved_tflops", "peak", "frac", "launches_per_step", "gpu_busy_us"):
        assert k in line["roofline_step"], k
    assert 30e9 < line["roofline_step"]["flops_per_step"] < 33e9 and 0 < line["roofline_step"]["frac"] < 1
    for k in ("cpu_model", "block_steps_per_s", "threads_calibration_steps_per_s"):
        assert k in line["cpu_baseline"], k
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "roofline_sampler", "roofline_flat_arena",
              "cpu_baseline", "samples_per_sec", "speedup_vs_cpu", "detail", "ranks_seen"):
        assert k in line, k
    assert set(line["config"]) == {"workload", "params", "chains", "step_path"}
    for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_kernel_us", "launches_per_step"):
        assert k in line["roofline"], k
    for k in ("value", "unit", "cores", "host_cpus", "kind", "sample"):
        assert k in line["cpu_baseline"], k
    assert line["detail"] == "bench_detail.json"
    assert line["value"] == out["value"] and line["roofline"]["frac"] == top["frac"]


def test_compact_line_carries_chains_per_gpu_as_steps_and_as_samples():
    out = _full_record()
    out["chains_per_gpu"] = {str(k): {"aggregate_steps_per_s": 1000.0 * k ** .5, "per_chain_steps_per_s": 1.0, "us_per_lockstep": 1.0,
                                      "host_issue_us_per_lockstep": 1.0, "block_us_per_lockstep": [1.0, 1.0, 1.0],
                                      "aggregate_samples_per_s": 2.0 + k / 10} for k in (1, 2, 4)}
    out["chains_per_gpu"].update(distinct_hw_queues=6, method="x" * 100)
    line = json.loads(bench.compact_line(out, None))
    assert line["chains_per_gpu"] == {"1": 1000.0, "2": 1000.0 * 2 ** .5, "4": 2000.0}
    assert line["chains_per_gpu_samples_per_s"] == {"1": 2.1, "2": 2.2, "4": 2.4}
    assert len(json.dumps(line)) < 4096


def test_spread_cycles_over_fewer_streams():
    from bnn_priors_amd import multichain
    assert multichain.spread(["a", "b", "c"], 5) == ["a", "b", "c", "a", "b"] and multichain.spread(["a"], 2) == ["a", "a"]


def test_compact_line_with_eight_ranks_and_every_optional_table():
    out = _full_record()
    out["n_gpus"] = out["ranks_seen"] = 8
    out["devices"] = [{"rank": r, "local_rank": r, "index": r, "device": "AMD Instinct MI355X"} for r in range(8)]
    out["exchange"] = dict(backend="nccl (RCCL)", chains=8, samples_per_chain=8, n_test=10000, classes=10, ensemble_ms=1.0,
                           ensemble_max_abs_err=0.0, ensemble_matches_single_process=True, gather_ms=2.0,
                           gathered_bytes_rank0=1, gather_order_checked=True, collectives="x" * 200)
    text = bench.compact_line(out, None)
    assert len(text) < 4096
    assert json.loads(text)["detail"] is None


def test_headline_is_the_symbol_with_the_largest_time_per_step():
    rows = _full_record()["roofline_kernels"]
    top, other = bench.headline_rooflines(rows)
    per_symbol = {}
    for r in rows:
        if r["launches_per_step"]:
            per_symbol[bench._symbol(r)] = per_symbol.get(bench._symbol(r), 0) + r["in_step_us"] * r["launches_per_step"]
    assert top["us_per_step"] == pytest.approx(max(per_symbol.values()), abs=0.06)
    assert other is not None and other["bound"] != top["bound"]
    # pooled over shapes: work / time, both summed over the launches of a step
    assert 0 < top["frac"] <= 1 and top["frac_in_step"] == top["frac"] and top["frac_source"] == "in_step"
    assert 0 < top["frac_isolated"] <= 1


def test_committed_measurements_of_other_sources_are_marked_stale(tmp_path, monkeypatch):
    from bnn_priors_amd import _hip
    rows = [dict(kernel="conv::conv3x3_kernel<16,32,8,stats>", shape=dict(n=128), unit="TFLOP/s", peak=157.3,
                 algorithmic_flops_per_launch=6e8, avg_kernel_us=10.0, launches_per_step=6, bound="mfma", traffic=None)]
    fake_root = tmp_path
    (fake_root / "profiles").mkdir()
    body = {"kernels": {rows[0]["kernel"]: {"in_step_us": 10.0, "launches_per_step": 6.0}}, "source": "x"}
    monkeypatch.setattr(bench.os.path, "abspath", lambda p: str(fake_root / "bench.py") if p == bench.__file__ else os.path.abspath(p))
    (fake_root / "profiles" / "in_step_us.json").write_text(json.dumps(dict(body, source_sha="0" * 16)))
    got = bench.attach_in_step([dict(rows[0])])[0]
    assert got.get("stale") is True and "frac_in_step" not in got and "in_step_us" not in got
    (fake_root / "profiles" / "in_step_us.json").write_text(json.dumps(dict(body, source_sha=_hip.library_sha())))
    got = bench.attach_in_step([dict(rows[0])])[0]
    assert "stale" not in got and got["frac_in_step"] == pytest.approx(6e8 / 10e-6 / 1e12 / 157.3, rel=1e-3)
    top, _ = bench.headline_rooflines([dict(rows[0], stale=True)])
    assert top["stale"] is True and "frac_in_step" not in top and top["frac_source"] == "isolated"
    assert top["frac"] == top["frac_isolated"]


def test_gpus_flag_launches_ranks_unless_already_a_rank(monkeypatch):
    import subprocess
    import types
    calls = []
    monkeypatch.setattr(subprocess, "call", lambda cmd: calls.append(cmd) or 0)
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 2)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3"])
    assert bench.self_launch(types.SimpleNamespace(gpus=1, backend="nccl")) is None and not calls
    assert bench.self_launch(types.SimpleNamespace(gpus=2, backend="nccl")) == 0
    cmd = calls[0]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=2" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "2", "--steps", "3"]
    with pytest.raises(SystemExit):          # RCCL: one device per rank, or fail loudly
        bench.self_launch(types.SimpleNamespace(gpus=4, backend="nccl"))
    assert bench.self_launch(types.SimpleNamespace(gpus=4, backend="gloo")) == 0     # (ranks may share a GPU over gloo)
    monkeypatch.setenv("WORLD_SIZE", "2")
    assert bench.self_launch(types.SimpleNamespace(gpus=2, backend="nccl")) is None  # already a rank: run the bench


def test_eight_ranks_get_disjoint_cpu_shares_and_a_per_rank_summary():
    """one-command 8-GPU readiness (round 6): `python bench.py --gpus 8` deals the host's CPUs into 8 disjoint contiguous
    shares (256 CPUs -> 32 each), refuses the RCCL path loudly with fewer than 8 devices (test above), and its line
    carries every rank's own rate next to rank 0's solo rate."""
    shares = bench.cpu_shares(range(256), 8)
    assert [len(s) for s in shares] == [32] * 8 and sorted(c for s in shares for c in s) == list(range(256))
    assert all(s == list(range(s[0], s[0] + 32)) for s in shares)
    assert [len(s) for s in bench.cpu_shares(range(10), 4)] == [2, 2, 2, 2]          # (the remainder stays unused)
    assert bench.cpu_shares([3], 2) == [[3], [3]]                                    # fewer CPUs than ranks: shared, never empty
    got = bench.per_rank_summary([1200.0, 1180.0, 1210.0, 1190.0, 1205.0, 1195.0, 1185.0, 1215.0], solo=1240.0)
    assert got["min"] == 1180.0 and got["max"] == 1215.0 and got["ranks"] == 8 and got["solo_rank0"] == 1240.0
    assert got["median"] == 1197.5 and got["efficiency_vs_solo"] == round(1197.5 / 1240.0, 4)
    assert bench.per_rank_summary([5.0], None)["efficiency_vs_solo"] is None
    out = _full_record()
    out["n_gpus"] = out["ranks_seen"] = 8
    out["per_rank_steps_per_s"] = got
    assert json.loads(bench.compact_line(out, None))["per_rank_steps_per_s"] == got


def _run_bench(*flags, timeout=600):
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True,
                       timeout=timeout, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines          # stdout carries exactly one line
    assert len(lines[0]) < 4096
    return json.loads(lines[0])


@pytest.mark.gpu
def test_gpus_2_over_gloo_starts_two_ranks_by_itself(tmp_path):
    """`python bench.py --gpus 2` with no launcher around it: two ranks, one chain each, the exchange checked against the
    single-process formula (over gloo the two ranks share this box's one GPU: plumbing, never a scaling number)."""
    detail = tmp_path / "detail.json"
    line = _run_bench("--gpus", "2", "--backend", "gloo", "--steps", "5", "--warmup", "2", "--samples", "0",
                      "--cpu-budget", "0", "--other-workloads", "0", "--sweep-log2", "0", "--no-kernel-timing",
                      "--stream-chains", "", "--detail", str(detail))
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["backend"] == "gloo"
    assert [d["rank"] for d in line["devices"]] == [0, 1] and [d["local_rank"] for d in line["devices"]] == [0, 1]
    assert line["exchange"]["ensemble_matches_single_process"] and line["exchange"]["gather_order_checked"]
    assert line["config"]["chains"] == 2 and line["value"] > 0
    pr = line["per_rank_steps_per_s"]          # every rank's own rate, and rank 0's when it ran alone right before
    assert pr["ranks"] == 2 and 0 < pr["min"] <= pr["median"] <= pr["max"] and pr["solo_rank0"] > 0
    assert 0 < pr["efficiency_vs_solo"] < 1.5
    full = json.loads(detail.read_text())
    assert full["value"] == line["value"] and "timing" in full and full["exchange"]["chains"] == 2


@pytest.mark.gpu
def test_one_gpu_line_carries_roofline_and_names_its_detail_file(tmp_path):
    detail = tmp_path / "detail.json"
    line = _run_bench("--steps", "5", "--warmup", "2", "--samples", "0", "--cpu-budget", "0", "--other-workloads", "0",
                      "--sweep-log2", "20", "--stream-chains", "", "--detail", str(detail))
    assert line["n_gpus"] == 1 and line["ranks_seen"] == 1
    assert line["roofline"]["bound"] in ("hbm", "mfma") and 0 < line["roofline"]["frac"] <= 1
    assert ("roofline_mfma" in line) != ("roofline_hbm" in line)
    assert 0 < line["roofline_sampler"]["frac"] <= 1 and 0 < line["roofline_flat_arena"]["frac"] <= 1
    full = json.loads(detail.read_text())
    assert len(full["roofline_kernels"]) >= 18 and full["source_sha"]
    assert "other_workloads" not in line


@pytest.mark.gpu
def test_a_sub_run_of_other_workloads_is_read_from_its_side_file():
    import types
    got = bench.other_workloads(types.SimpleNamespace(metrics_skip=10), only="densenet")
    (name, row), = got.items()
    assert "error" not in row, row
    assert row["value"] > 0 and row["timed_steps"] > 0 and "dense" in row["step_path"]
