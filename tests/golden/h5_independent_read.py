"""One independent read of the sample / metrics store (reference: bnn_priors/exp_utils.py:409-551,
testing/test_exp_utils.py:27-80).

Writes a sample file with ``storage.HDF5ModelSaver`` and a metrics file with ``storage.HDF5Metrics`` (both through this
package's ctypes binding ``_h5.py``), then reads them back with a reader that is NOT ``_h5.py`` -- the HDF5 group's own
command-line tools ``h5ls`` / ``h5dump`` (conda's hdf5 package; h5py is not installed in this image) -- and compares, per
dataset: name, shape and maximum shape, element type, chunk shape, the fletcher32 filter, the fill value, and every
VALUE (``h5dump -b LE``: the dataset's raw little-endian bytes) against the tensors that were stored.

    python tests/golden/h5_independent_read.py [--write-report]     ->  tests/golden/h5_independent_read.txt

tests/test_storage.py::test_store_read_back_by_the_hdf5_tools runs the same comparison wherever the tools exist.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def tool(name):
    return shutil.which(name) or (f"/opt/conda/bin/{name}" if os.path.exists(f"/opt/conda/bin/{name}") else None)


H5TYPE = {np.dtype("float32"): "H5T_IEEE_F32LE", np.dtype("float64"): "H5T_IEEE_F64LE", np.dtype("int64"): "H5T_STD_I64LE"}


def make_state(i):
    "a small net's state_dict with the reference's kinds of entries: float tensors, 0-d buffers, an int64 counter"
    g = torch.Generator().manual_seed(100 + i)
    return {"net.module.0.weight_prior.p": torch.randn(4, 3, generator=g),
            "net.module.0.weight_prior.loc": torch.tensor(0.0), "net.module.0.weight_prior.scale": torch.tensor(0.5),
            "net.module.0.bias_prior.p": torch.randn(4, generator=g),
            "net.module.1.running_mean": torch.randn(4, generator=g).double(),
            "net.module.1.num_batches_tracked": torch.tensor(7 + i)}


def write_files(root):
    from bnn_priors_amd import storage
    spath, mpath = os.path.join(root, "samples.h5"), os.path.join(root, "metrics.h5")
    states, steps = [make_state(i) for i in range(4)], [7, 17, 27, 37]
    with storage.HDF5ModelSaver(spath, "w") as saver:
        for st, step in zip(states, steps):
            saver.add_state_dict(st, step)
    rows = []
    with storage.HDF5Metrics(mpath, "w", chunk_size=4) as m:          # (several chunks: 10 rows)
        for step in range(0, 50, 5):
            m.add_scalar("loss", 0.5 + step, step)
            if step % 10 == 0:
                m.add_scalar("acceptance/is_sample", step // 10 % 2, step, dtype=np.int64)
            rows.append(step)
    want_samples = {k: np.stack([np.asarray(s[k].numpy()) for s in states]) for k in states[0]}
    want_samples["steps"] = np.array(steps, dtype=np.int64)
    want_metrics = {"steps": np.array(rows, dtype=np.int64), "loss": np.array([0.5 + s for s in rows]),
                    "acceptance/is_sample": np.array([(s // 10 % 2) if s % 10 == 0 else -2 ** 63 for s in rows], dtype=np.int64)}
    return spath, want_samples, mpath, want_metrics


def read_with_tools(path):
    "{dataset name: dict(header facts, values)} as h5ls / h5dump see the file"
    h5ls, h5dump = tool("h5ls"), tool("h5dump")
    listing = subprocess.run([h5ls, "-r", path], capture_output=True, text=True, check=True).stdout
    names = [ln.split()[0].replace("\\ ", " ") for ln in listing.splitlines() if " Dataset " in ln]
    out = {}
    for name in names:
        head = " ".join(subprocess.run([h5dump, "-H", "-p", "-d", name, path], capture_output=True, text=True,
                                       check=True).stdout.split())
        m = re.search(r"DATASPACE SIMPLE \{ \( ([^)]*) \) / \( ([^)]*) \) \}", head)
        info = dict(type=re.search(r"DATATYPE (\S+)", head).group(1),
                    shape=tuple(int(v) for v in m.group(1).split(",")),
                    maxshape=tuple(None if "UNLIMITED" in v else int(v) for v in m.group(2).split(",")),
                    chunks=tuple(int(v) for v in re.search(r"CHUNKED \( ([^)]*) \)", head).group(1).split(",")),
                    fletcher32="CHECKSUM FLETCHER32" in head,
                    fill=re.search(r"FILLVALUE \{.*? VALUE (\S+)", head).group(1))
        with tempfile.NamedTemporaryFile(suffix=".bin") as raw:
            subprocess.run([h5dump, "-d", name, "-b", "LE", "-o", raw.name, path], capture_output=True, check=True)
            info["bytes"] = open(raw.name, "rb").read()
        out[name.lstrip("/")] = info
    return out


def compare(path, want, report, chunk_rows=None):
    got = read_with_tools(path)
    extra = set(got) - set(want) - {"timestamps"}
    assert not extra and set(want) <= set(got), (extra, set(want) - set(got))
    for name, arr in want.items():
        g = got[name]
        assert g["type"] == H5TYPE[arr.dtype], (name, g["type"])
        assert g["shape"] == arr.shape and g["maxshape"] == (None,) + arr.shape[1:], (name, g["shape"], g["maxshape"])
        assert g["chunks"] == ((chunk_rows or 1),) + arr.shape[1:], (name, g["chunks"])
        assert g["fletcher32"], name
        fill_ok = (g["fill"].lower().lstrip("-") == "nan") if arr.dtype.kind == "f" else (int(g["fill"]) == -2 ** 63)
        assert fill_ok, (name, g["fill"])
        values = np.frombuffer(g["bytes"], dtype=arr.dtype.newbyteorder("<")).reshape(arr.shape)
        assert np.array_equal(values, arr), name
        report.append(f"  {name:40s} {g['type']:15s} shape {str(g['shape']):12s} max {str(g['maxshape']):16s} "
                      f"chunks {str(g['chunks']):10s} fletcher32 {g['fletcher32']} fill {g['fill']:>22s}  "
                      f"values equal ({arr.size} elements)")
    ts = got.get("timestamps")
    assert ts is not None and ts["type"] == "H5T_IEEE_F64LE" and ts["shape"] == (len(want["steps"]),)
    report.append(f"  {'timestamps':40s} {ts['type']:15s} shape {str(ts['shape']):12s} (wall-clock values: not compared)")


def main(write_report=False):
    report = ["Independent read of files written through bnn_priors_amd/_h5.py (storage.HDF5ModelSaver, storage.HDF5Metrics)",
              "reader: " + subprocess.run([tool("h5dump"), "--version"], capture_output=True, text=True).stdout.strip()
              + " (h5ls -r for the names; h5dump -H -p for type / extent / chunking / filters / fill value; h5dump -b LE for the values)", ""]
    with tempfile.TemporaryDirectory() as root:
        spath, want_s, mpath, want_m = write_files(root)
        report.append("samples.h5 (HDF5ModelSaver: one row per stored sample, chunk = one sample):")
        compare(spath, want_s, report)
        report.append("metrics.h5 (HDF5Metrics, chunk_size = 4: rows keyed by step, NaN / INT64_MIN where a key was not logged):")
        compare(mpath, want_m, report, chunk_rows=4)
    report.append("")
    report.append("every dataset: names, element types, extents (first axis unlimited), chunk shapes, the fletcher32 filter, "
                  "fill values and all values agree with what was stored")
    text = "\n".join(report) + "\n"
    if write_report:
        with open(os.path.join(HERE, "h5_independent_read.txt"), "w") as f:
            f.write(text)
    return text


if __name__ == "__main__":
    if tool("h5dump") is None or tool("h5ls") is None:
        raise SystemExit("h5dump / h5ls not found (conda's hdf5 package provides them in the build container)")
    print(main("--write-report" in sys.argv), end="")
