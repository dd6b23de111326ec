"""Summarise a rocprofv3 kernel trace CSV: per-kernel GPU time in the steady state.
    trace_summary.py <kernel_trace.csv> [top-N] [last-ms]
The steady state is the second half of the dispatches, or -- when MIOpen's find mode or other
set-up work dominates the trace -- the last ``last-ms`` milliseconds (the timed steps run last)."""
import csv
import sys
from collections import Counter

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 14
if len(sys.argv) > 3:
    cut = int(rows[-1]["End_Timestamp"]) - int(float(sys.argv[3]) * 1e6)
    sub = [r for r in rows if int(r["Start_Timestamp"]) >= cut]
else:
    sub = rows[len(rows) // 2:]
t0, t1 = int(sub[0]["Start_Timestamp"]), int(sub[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in sub)
print(f"steady-state window {(t1 - t0) / 1e6:.2f} ms, GPU busy {busy / 1e6:.2f} ms, {len(sub)} dispatches")
c, d = Counter(), Counter()
for r in sub:
    k = r["Kernel_Name"][:90]
    c[k] += 1
    d[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, v in d.most_common(top):
    print(f"{v / 1e3:10.1f} us total {100 * v / busy:5.1f}% {c[k]:6d} x {v / c[k] / 1e3:8.2f} us  {k}")
