"""Summarise a rocprofv3 kernel trace CSV: steady-state (second half) per-kernel GPU time."""
import csv
import sys
from collections import Counter

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sub = rows[len(rows) // 2:]
t0, t1 = int(sub[0]["Start_Timestamp"]), int(sub[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in sub)
print(f"steady-state window {(t1 - t0) / 1e6:.2f} ms, GPU busy {busy / 1e6:.2f} ms, {len(sub)} dispatches")
c, d = Counter(), Counter()
for r in sub:
    k = r["Kernel_Name"][:90]
    c[k] += 1
    d[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, v in d.most_common(int(sys.argv[2]) if len(sys.argv) > 2 else 14):
    print(f"{v / 1e3:10.1f} us total {c[k]:6d} x {v / c[k] / 1e3:8.2f} us  {k}")
