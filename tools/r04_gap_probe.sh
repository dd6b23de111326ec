cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/pg -o p -- python /root/repo/tools/exact_pass_probe.py --workload convnet --product-source 1 --passes 8 > /tmp/pg.log 2>&1
tail -1 /tmp/pg.log | cut -c1-200
python - <<'PY'
import csv
rows=list(csv.DictReader(open('/tmp/pg/p_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last 60% of the trace: find gaps > 5 ms between end of any kernel and start of next
t_end=0; gaps=[]
for i,r in enumerate(rows):
    s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    if t_end and s-t_end>3e6: gaps.append((i,(s-t_end)/1e6,rows[i-1]['Kernel_Name'][:60],r['Kernel_Name'][:60],rows[i-1]['Stream_Id'],r['Stream_Id'], rows[i-1]['Queue_Id'], r['Queue_Id']))
    t_end=max(t_end,e)
print(len(rows),"dispatches; gaps > 3 ms:")
for g in gaps[-25:]: print(g)
PY
