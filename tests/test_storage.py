"""The HDF5 sample / metrics store against the reference's own store test
(testing/test_exp_utils.py:27-80) and against ``h5dump``'s view of the file layout."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest
import torch

from bnn_priors_amd import _h5, storage

pytestmark = pytest.mark.skipif(not _h5.available(), reason="libhdf5 (>= 1.10) not found on this machine")
H5DUMP = shutil.which("h5dump") or ("/opt/conda/bin/h5dump" if os.path.exists("/opt/conda/bin/h5dump") else None)


INT_FILL = -2 ** 63          # what the store writes where an int64 column has no value (exp_utils.py:505-513)
# the scenario of the reference's store test (testing/test_exp_utils.py:27-80) as a TABLE: column -> (stride, dtype).
# A column receives a value at step -1 and at every multiple of its stride; 101 rows (steps -1 .. 99); chunk 13;
# the writer flushes at multiples of 31 while a SWMR reader looks at the file.
COLUMNS = {"re_step": (1, np.int64), "step5": (5, np.int64), "step11": (11, np.float64), "step23": (23, np.int64)}
STEPS = np.arange(-1, 100)


def _expected_column(stride, dtype):
    "the dense column the file must hold: value step // stride where written, the dtype's fill everywhere else"
    written = (STEPS == -1) | (STEPS % stride == 0)
    fill = np.nan if dtype is np.float64 else INT_FILL
    return np.where(written, STEPS // stride, fill).astype(dtype), written


def test_metric_columns_equal_a_dense_model_of_the_reference_scenario(tmp_path):
    """HDF5Metrics against a numpy model of what the reference's store must contain for its own test scenario:
    every column compared WHOLE (values, fills, dtype), the row count a concurrent SWMR reader sees after each
    flush, the chunking / checksum filter of the datasets."""
    fname = tmp_path / "metrics_test.h5"
    seen_by_reader = {}
    with storage.HDF5Metrics(fname, "w", chunk_size=13) as metrics:
        for step in STEPS.tolist():
            for key, (stride, dtype) in COLUMNS.items():
                if step == -1 or step % stride == 0:
                    metrics.add_scalar(key, dtype(step // stride).item(), step)
            if step % 31 == 0:
                metrics.flush()
                with _h5.File(fname, "r", swmr=True) as reader:        # the writer still holds the file
                    seen_by_reader[step] = {k: len(reader[k]) for k in reader.keys()}
    # a reader that opens mid-run sees every row flushed so far, in every column
    assert sorted(seen_by_reader) == [0, 31, 62, 93]
    for step, lengths in seen_by_reader.items():
        assert set(lengths) == set(COLUMNS) | {"steps", "timestamps"}
        assert set(lengths.values()) == {step + 2}, (step, lengths)

    with _h5.File(fname, "r") as f:
        assert sorted(f.keys()) == sorted(set(COLUMNS) | {"steps", "timestamps"})
        assert np.array_equal(f["steps"][:], STEPS) and f["steps"].dtype == np.int64
        ts = f["timestamps"][:]
        assert ts.shape == (101,) and ts.dtype == np.float64 and not np.isnan(ts).any()
        for key, (stride, dtype) in COLUMNS.items():
            want, written = _expected_column(stride, dtype)
            got = f[key][:]
            assert got.dtype == dtype and got.shape == want.shape, key
            assert np.array_equal(got[written], want[written]), key
            if dtype is np.float64:
                assert np.isnan(got[~written]).all(), key
            else:
                assert (got[~written] == INT_FILL).all(), key
            assert f[key].creation_properties() == ((13,), [3]), key       # chunk 13; filter 3 = Fletcher-32


def test_metrics_errors_and_nested_names(tmp_path):
    with storage.HDF5Metrics(tmp_path / "m.h5", "w", chunk_size=4) as m:
        m.add_scalar("est_temperature/all", 1.5, 0)
        m.add_scalar("acceptance/rejected", 0, 0)
        m.add_scalar("est_temperature/all", 2.5, 3)
        with pytest.raises(ValueError, match="step went backwards"):
            m.add_scalar("est_temperature/all", 2.5, 2)
        m.flush()
        m.add_scalar("late_key", 1.0, 4)
        with pytest.raises(KeyError, match="SWMR"):
            m.flush()
        del m._cache["late_key"]
    with _h5.File(tmp_path / "m.h5") as f:
        assert sorted(f.dataset_names()) == ["acceptance/rejected", "est_temperature/all", "steps", "timestamps"]
        assert np.array_equal(f["steps"][:], [0, 3, 4])
        assert np.array_equal(f["est_temperature/all"][:2], [1.5, 2.5]) and np.isnan(f["est_temperature/all"][2])
        assert np.array_equal(f["acceptance/rejected"][:], [0, -2 ** 63, -2 ** 63])


def _net():
    torch.manual_seed(3)
    return torch.nn.Sequential(torch.nn.Linear(3, 4), torch.nn.BatchNorm1d(4), torch.nn.Linear(4, 2))


def test_model_saver_round_trip_and_layout(tmp_path):
    path = tmp_path / "samples.h5"
    net = _net()
    want = []
    with storage.HDF5ModelSaver(path, "w") as saver:
        for i in range(4):
            with torch.no_grad():
                for p in net.parameters():
                    p.add_(0.25 * (i + 1))
                net[1].num_batches_tracked += 1
            saver.add_state_dict(net.state_dict(), 10 * i + 7)
            want.append({k: v.clone() for k, v in net.state_dict().items()})
        mid = saver.load_samples()                      # readable while the writer is open
        assert mid["steps"].tolist() == [7, 17, 27, 37]
    for out in (storage.load_samples(path), mid):
        assert set(out) == set(want[0]) | {"steps", "timestamps"}
        for k in want[0]:
            assert out[k].dtype == want[0][k].dtype
            assert torch.equal(out[k], torch.stack([w[k] for w in want])), k
    sub = storage.load_samples(path, idx=slice(1, 3), keep_steps=False)
    assert "steps" not in sub and "timestamps" not in sub
    assert torch.equal(sub["0.weight"], torch.stack([w["0.weight"] for w in want[1:3]]))
    assert storage.load_samples(path, idx=-1)["steps"].item() == 37
    with _h5.File(path) as f:
        assert f["0.weight"].creation_properties() == ((1, 4, 3), [3])
        assert f["1.num_batches_tracked"].shape == (4,) and f["1.num_batches_tracked"].dtype == np.int64
    with pytest.raises(TypeError, match="NaN"):
        with storage.HDF5ModelSaver(tmp_path / "bad.h5", "w") as saver:
            saver.add_state_dict({"flag": torch.zeros(3, dtype=torch.bool)}, 0)


@pytest.mark.skipif(H5DUMP is None, reason="h5dump not installed")
def test_file_layout_as_h5dump_sees_it(tmp_path):
    "an independent reader (the HDF5 tools) agrees on type, extent, chunking, checksum and fill value"
    path = tmp_path / "samples.h5"
    with storage.HDF5ModelSaver(path, "w") as saver:
        for i in range(3):
            saver.add_state_dict(_net().state_dict(), i)
    out = subprocess.run([H5DUMP, "-H", "-p", "-d", "/0.weight", str(path)], capture_output=True, text=True,
                         check=True).stdout
    squeezed = " ".join(out.split())
    for piece in ("DATATYPE H5T_IEEE_F32LE", "SIMPLE { ( 3, 4, 3 ) / ( H5S_UNLIMITED, 4, 3 ) }",
                  "CHUNKED ( 1, 4, 3 )", "CHECKSUM FLETCHER32", "VALUE nan"):
        assert piece in squeezed, (piece, out)
    data = subprocess.run([H5DUMP, "-d", "/steps", str(path)], capture_output=True, text=True, check=True).stdout
    assert "(0): 0, 1, 2" in data


def test_load_samples_falls_back_to_torch_files(tmp_path):
    path = tmp_path / "samples.pt"
    torch.save({"w": torch.arange(6.).reshape(3, 2), "steps": torch.tensor([1, 2, 3])}, path)
    out = storage.load_samples(path, idx=slice(0, 2))
    assert torch.equal(out["w"], torch.arange(4.).reshape(2, 2))


def test_reject_samples_rewinds_rejected_rows(tmp_path):
    mpath = tmp_path / "metrics.h5"
    with storage.HDF5Metrics(mpath, "w", chunk_size=8) as m:
        for step, (is_sample, rejected) in enumerate([(1, 0), (0, None), (1, 1), (1, 0), (0, None), (1, 1)]):
            m.add_scalar("acceptance/is_sample", is_sample, step)
            if rejected is not None:
                m.add_scalar("acceptance/rejected", rejected, step)
    samples = {"w": torch.tensor([[0.], [2.], [3.], [5.]]), "steps": torch.tensor([0, 2, 3, 5])}
    with _h5.File(mpath) as f:
        storage.reject_samples_(samples, f)
    assert samples["w"].flatten().tolist() == [0., 0., 3., 3.]
    with storage.HDF5Metrics(tmp_path / "plain.h5", "w", chunk_size=8) as m:
        m.add_scalar("loss", 1.0, 0)
    with _h5.File(tmp_path / "plain.h5") as f:
        assert storage.reject_samples_(samples, f) is samples


@pytest.mark.gpu
def test_runner_through_hdf5_sinks_matches_memory_sinks(tmp_path):
    "the same run logged to HDF5 files and to the in-memory sinks: identical streams and samples"
    from bnn_priors_amd import inference_reject, models
    import runner_cases as RC

    def run(metrics, saver):
        cfg = RC.CASES["VerletSGLDReject"]
        train, test, (x, y) = RC.make_data("cuda:0")
        model = RC.make_net(models, x, y, device="cuda:0")
        torch.manual_seed(RC.SEED)
        runner = inference_reject.VerletSGLDRunnerReject(
            model=model, dataloader=train, dataloader_test=test, learning_rate=cfg["lr"],
            temperature=cfg["temperature"], momentum=cfg["momentum"], reject_samples=True,
            metrics_saver=metrics, model_saver=saver, seed=RC.SEED, chain_id=0,
            cycle_seed=RC.CYCLE_SEED, **RC.RUN_KW)
        runner.run()
        return runner.get_samples()

    mem = storage.MemoryMetrics()
    want = run(mem, None)
    with storage.HDF5Metrics(tmp_path / "metrics.h5", "w", chunk_size=4) as hm, \
            storage.HDF5ModelSaver(tmp_path / "samples.h5", "w") as hs:
        got = run(hm, hs)
    assert set(got) == set(want)
    for k in want:
        assert torch.equal(got[k].to(want[k].device), want[k]), k
    with _h5.File(tmp_path / "metrics.h5") as f:
        names = f.dataset_names()
        assert set(mem.names()) <= set(names)
        steps, _ = mem.column("loss")
        assert np.array_equal(f["steps"][:], steps)
        for k in mem.names():
            col = f[k][:]
            _, v = mem.column(k)
            if col.dtype == np.int64:
                v = np.where(np.isnan(v), -2.0 ** 63, v).astype(np.int64)
            assert np.array_equal(col, v, equal_nan=col.dtype != np.int64), k
        assert f["acceptance/rejected"].dtype == np.int64
    s = storage.load_samples(tmp_path / "samples.h5")
    assert s["steps"].tolist() == [17, 34]


@pytest.mark.gpu
def test_a_device_state_dict_reaches_the_host_in_one_copy_per_dtype_with_the_same_values():
    "storage.state_dict_to_host: concatenated on the device, copied once, split into views -- key order and values kept"
    import torch
    from bnn_priors_amd import models
    from bnn_priors_amd.storage import state_dict_to_host
    x = torch.randn(2, 3, 32, 32)
    net = models.get_model(x, torch.tensor([0, 9]), "googleresnet", width=50, depth=3, weight_prior="gaussian",
                           weight_scale=2 ** .5, bias_prior="gaussian", bias_scale=1.).cuda()
    sd = net.state_dict()
    sd["strided"] = torch.randn(6, 4, device="cuda").t()           # (a non-contiguous entry)
    host = state_dict_to_host(sd)
    assert list(host) == list(sd)
    for k, v in sd.items():
        assert not host[k].is_cuda and host[k].dtype == v.dtype and host[k].shape == v.shape
        assert torch.equal(host[k], v.cpu()), k


@pytest.mark.skipif(H5DUMP is None, reason="h5dump / h5ls not installed (conda's hdf5 package has them in the build container)")
def test_store_read_back_by_the_hdf5_tools():
    """files written through _h5.py read back by a reader that is not _h5.py (h5ls / h5dump): every dataset's name, type,
    extent, chunking, fletcher32 filter, fill value and VALUES -- tests/golden/h5_independent_read.py, whose output is
    committed as tests/golden/h5_independent_read.txt"""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import h5_independent_read as H
    text = H.main()
    assert text.count("values equal") == 10 and "agree with what was stored" in text
    committed = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "h5_independent_read.txt")).read()
    assert [ln for ln in committed.splitlines() if "values equal" in ln] == [ln for ln in text.splitlines() if "values equal" in ln]
