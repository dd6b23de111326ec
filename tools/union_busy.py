"""GPU occupancy of a multi-stream run from a rocprofv3 kernel trace CSV: union of the kernels' [start, end) intervals
(time with AT LEAST one kernel resident), the sum of their durations, and the overlap histogram -- over the last
`--frac` of the trace (the steady state of the largest chain count).
    union_busy.py <kernel_trace.csv> [--frac 0.3]"""
import argparse
import csv

ap = argparse.ArgumentParser()
ap.add_argument("csv")
ap.add_argument("--frac", type=float, default=0.3)
a = ap.parse_args()
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in csv.DictReader(open(a.csv))]
rows.sort()
t_end = max(r[1] for r in rows)
t_beg = min(r[0] for r in rows)
lo = t_end - a.frac * (t_end - t_beg)
rows = [r for r in rows if r[0] >= lo]
span = rows[-1][1] - rows[0][0]
ev = []
for s, e, _, _ in rows:
    ev.append((s, 1))
    ev.append((e, -1))
ev.sort()
depth, last, hist = 0, ev[0][0], {}
for t, d in ev:
    hist[depth] = hist.get(depth, 0) + (t - last)
    depth += d
    last = t
total = sum(e - s for s, e, _, _ in rows)
queues = sorted({q for _, _, _, q in rows})
print(f"{len(rows)} dispatches on {len(queues)} queue(s) over {span / 1e3:.0f} us; sum of durations {total / 1e3:.0f} us "
      f"({total / span:.2f} x span); at least one kernel resident {100 * (1 - hist.get(0, 0) / span):.1f}% of the span")
for k in sorted(hist):
    print(f"  {k} kernel(s) resident: {100 * hist[k] / span:5.1f}%")
names = {}
for s, e, n, _ in rows:
    k = n[:60]
    c = names.setdefault(k, [0, 0])
    c[0] += 1
    c[1] += e - s
for k, (c, d) in sorted(names.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"  {c:6d} x {d / c / 1e3:8.2f} us  {k}")
