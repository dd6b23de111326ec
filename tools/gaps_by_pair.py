import csv, sys, collections
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
rows=rows[len(rows)//2:]
g=collections.defaultdict(list)
for a,b in zip(rows[:-1],rows[1:]):
    gap=(int(b["Start_Timestamp"])-int(a["End_Timestamp"]))/1e3
    g[(a["Kernel_Name"][:50],b["Kernel_Name"][:50])].append(gap)
for k,v in sorted(g.items(), key=lambda kv:-len(kv[1]))[:8]:
    v.sort(); print(f"{len(v):6d} median {v[len(v)//2]:6.2f}  {k[0]} -> {k[1]}")
allg=sorted(x for v in g.values() for x in v)
n=len(allg)
print("all gaps: n", n, "p50 %.2f p90 %.2f p95 %.2f p99 %.2f max %.2f" % tuple(allg[int(n*q)] for q in (0.5,0.9,0.95,0.99,0.9999)), " >3us:", sum(1 for x in allg if x>3), " >7us:", sum(1 for x in allg if x>7))
