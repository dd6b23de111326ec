"""side-by-side table of steady-state summaries (tools/step_summary.py output): python tools/ab_table.py a.txt b.txt ..."""
import os, re, sys
cols, heads = [], []
for f in sys.argv[1:]:
    d = {}
    lines = open(f).read().splitlines()
    heads.append(lines[0])
    for l in lines[2:]:
        m = re.match(r"\s*([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+[\d.]+%\s+(.*)", l)
        if m:
            d[m.group(4)[:86]] = (float(m.group(1)), float(m.group(2)), float(m.group(3)))
    cols.append((os.path.basename(f)[:-4], d))
for (n, _), h in zip(cols, heads):
    print(f"{n:>14}: {h}")
keys = sorted(set().union(*[d for _, d in cols]), key=lambda k: -max(d.get(k, (0, 0, 0))[2] for _, d in cols))
print(f"{'n/step':>6} " + " ".join(f"{n[:9]:>9}" for n, _ in cols) + "  kernel (mean us per launch)")
for k in keys:
    n = max(d.get(k, (0, 0, 0))[0] for _, d in cols)
    print(f"{n:6.0f} " + " ".join(f"{d[k][1]:9.2f}" if k in d else f"{'-':>9}" for _, d in cols) + "  " + re.sub(r"\(.*", "", k.replace("void ", ""))[:70])
