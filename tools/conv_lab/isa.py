"""condensed instruction trace of one kernel of an assembly listing: python isa.py file.s substring [first..last]"""
import re, sys
s = open(sys.argv[1]).read()
sym = [x for x in re.findall(r'^(_Z\S+):', s, re.M) if sys.argv[2] in x][0]
i = s.index('\n' + sym + ':'); j = s.index('s_endpgm', i)
lines = [l.strip() for l in s[i:j].splitlines() if l.startswith('\t') and l.strip() and not l.strip().startswith(('.', ';'))]
keys = ('global_load', 'global_store', 's_waitcnt', 's_barrier', 'ds_write', 'ds_read', 's_cbranch', 's_memrealtime', 'v_mfma', 'v_accvgpr')
print(sym, len(lines), 'instructions')
last = None; run = 0
for k, l in enumerate(lines):
    if not l.startswith(keys) and not re.match(r'\.?LBB', l): continue
    op = l.split()[0]
    tag = op if op not in ('s_waitcnt', 's_cbranch_execz', 's_cbranch_scc1', 's_cbranch_scc0', 's_cbranch_vccnz', 's_cbranch_vccz', 's_cbranch_execnz') else l[:60]
    if tag == last: run += 1; continue
    if last is not None: print(f"{start:5d} {last}" + (f"  x{run}" if run > 1 else ""))
    last, run, start = tag, 1, k
print(f"{start:5d} {last}" + (f"  x{run}" if run > 1 else ""))
