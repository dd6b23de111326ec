"""Per-leapfrog-step kernel breakdown from a rocprofv3 kernel trace CSV:
    step_summary.py <kernel_trace.csv> [--steps 30] [--marker step_kernel]
A step = the dispatches between two consecutive launches of the sampler's update kernel (the marker); the
last --steps complete steps of the trace are averaged.  Prints launches per step, GPU-busy time per step, the
step period (marker to marker) and, per kernel, launches per step, mean duration and time per step.

--json <file>: the same means keyed by bench.py's `roofline_kernels` row names (profiles/in_step_us.json: what bench.py
attaches as `in_step_us` / `frac_in_step`).  A BatchNorm kernel's name does not carry its shape: it is the shape of
the convolution next to it in the stream -- the producer right before an apply launch, the consumer right after a
backward launch (a BatchNorm's dx feeds the gradient launch of the convolution that produced its input)."""
import argparse
import csv
import json
import re
from collections import Counter

ap = argparse.ArgumentParser()
ap.add_argument("csv")
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--marker", default="step_kernel")
ap.add_argument("--top", type=int, default=40)
ap.add_argument("--json", default=None)
ap.add_argument("--source", default=None, help="what to record as the trace's origin in the JSON")
a = ap.parse_args()
rows = list(csv.DictReader(open(a.csv)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if a.marker in r["Kernel_Name"]]
# keep markers that delimit steps of the steady state: the last `steps` + 1 whose spacing is regular
gaps = [marks[i + 1] - marks[i] for i in range(len(marks) - 1)]
common = Counter(gaps).most_common(1)[0][0]
good = [i for i in range(len(gaps)) if gaps[i] == common]
good = good[-a.steps:]
n = len(good)
cnt, dur = Counter(), Counter()
busy = period = 0
for i in good:
    seg = rows[marks[i]:marks[i + 1]]
    period += int(rows[marks[i + 1]]["Start_Timestamp"]) - int(rows[marks[i]]["Start_Timestamp"])
    for r in seg:
        k = r["Kernel_Name"][:100]
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        cnt[k] += 1
        dur[k] += d
        busy += d
print(f"{n} steps averaged; {common} launches per step; GPU busy {busy / n / 1e3:.1f} us per step; "
      f"step period {period / n / 1e3:.1f} us")
print(f"{'per step':>9} {'mean us':>9} {'us/step':>9} {'share':>6}  kernel")
for k, v in dur.most_common(a.top):
    print(f"{cnt[k] / n:9.2f} {v / cnt[k] / 1e3:9.2f} {v / n / 1e3:9.1f} {100 * v / busy:5.1f}%  {k}")


def conv_shape(name):
    "(channels, side) of the activation a convolution kernel PRODUCES (forward) / whose gradient it CONSUMES (backward)"
    m = re.search(r"conv::conv3x3(?:_bwd)?_kernel<(\d+), (\d+)", name)
    if m:
        return int(m.group(1)), int(m.group(2))
    m = re.search(r"convdown::(?:fwd|bwd)_kernel<(\d+), (\d+)", name)
    if m:
        return 2 * int(m.group(1)), int(m.group(2)) // 2
    if "convstem::" in name:
        return 16, 32
    return None


def row_name(rows_of_step, i):
    "bench.py's roofline row for dispatch i of a step, or None"
    name = rows_of_step[i]["Kernel_Name"]
    m = re.search(r"conv::conv3x3_kernel<(\d+), (\d+), (\d+), false, true>", name)
    if m:
        return "conv::conv3x3_kernel<%s,%s,%s,stats>" % m.groups()
    m = re.search(r"conv::conv3x3_bwd_kernel<(\d+), (\d+), (\d+), (true|false), (true|false)(?:, (?:true|false))?>", name)
    if m:       # (a third flag since round 4: the many-minibatch instantiation; the step runs `false`)
        c, hw, r, epi, sums = m.groups()
        tag = ",".join(t for t, on in (("ADD", epi), ("SUMS", sums)) if on == "true")
        return "conv::conv3x3_bwd_kernel<%s,%s,%s%s>" % (c, hw, r, "," + tag if tag else "")
    m = re.search(r"bn::(apply_kernel|bwd_dx_kernel)<(true|false), (true|false)(?:, (true|false))?>", name)
    if m:
        kind, relu, res, rs = m.groups()
        step = -1 if kind == "apply_kernel" else 1           # producer before an apply, consumer after a backward
        j, shape = i + step, None
        while 0 <= j < len(rows_of_step) and shape is None:
            shape = conv_shape(rows_of_step[j]["Kernel_Name"])
            j += step
        if shape is None:
            return None
        flags = ",".join(t for t, on in (("relu", relu), ("residual", res), ("rsums", rs)) if on == "true")
        return "bn::%s<%s> %d@%d^2" % (kind, flags, shape[0], shape[1])
    return None


if a.json:
    acc = {}
    for i in good:
        seg = rows[marks[i]:marks[i + 1]]
        for k in range(len(seg)):
            key = row_name(seg, k)
            if key is None:
                continue
            d = int(seg[k]["End_Timestamp"]) - int(seg[k]["Start_Timestamp"])
            e = acc.setdefault(key, [0, 0])
            e[0] += 1
            e[1] += d
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from bnn_priors_amd import _hip
    out = {"source": a.source or a.csv, "source_sha": _hip.library_sha(), "steps_averaged": n, "launches_per_step": common,
           "gpu_busy_us_per_step": round(busy / n / 1e3, 1),
           "collected_by": "rocprofv3 --kernel-trace over bench.py (tools/prof_workload.sh), tools/step_summary.py --json",
           "kernels": {k: {"in_step_us": round(v[1] / v[0] / 1e3, 3), "launches_per_step": round(v[0] / n, 3)}
                       for k, v in sorted(acc.items())}}
    with open(a.json, "w") as f:
        json.dump(out, f, indent=1)
