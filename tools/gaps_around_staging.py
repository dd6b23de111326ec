import csv, sys, collections
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
rows=rows[len(rows)//2:]
gaps=collections.defaultdict(list)
for a,b in zip(rows[:-1],rows[1:]):
    g=(int(b["Start_Timestamp"])-int(a["End_Timestamp"]))/1e3
    na,nb=a["Kernel_Name"][:40],b["Kernel_Name"][:40]
    if "gather_stage" in na or "gather_stage" in nb:
        gaps[(na,nb)].append(g)
    else:
        gaps[("other","other")].append(g)
for k,v in gaps.items():
    v.sort()
    print(f"{len(v):6d} median {v[len(v)//2]:7.2f} mean {sum(v)/len(v):7.2f} us  {k[0]} -> {k[1]}")
