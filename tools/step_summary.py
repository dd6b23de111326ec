"""Per-leapfrog-step kernel breakdown from a rocprofv3 kernel trace CSV:
    step_summary.py <kernel_trace.csv> [--steps 30] [--marker step_kernel]
A step = the dispatches between two consecutive launches of the sampler's update kernel (the marker); the
last --steps complete steps of the trace are averaged.  Prints launches per step, GPU-busy time per step, the
step period (marker to marker) and, per kernel, launches per step, mean duration and time per step."""
import argparse
import csv
from collections import Counter

ap = argparse.ArgumentParser()
ap.add_argument("csv")
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--marker", default="step_kernel")
ap.add_argument("--top", type=int, default=40)
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
