"""Kernels of an assembly listing whose global loads are waited for one by one: for every kernel, the number of
`global_load* ... s_waitcnt vmcnt(0)` pairs with fewer than GAP instructions and no other global load in between, and how
many of them sit inside a loop (a backward branch follows) -- the `for (i = tid; ...) lds[i] = src[i]` shape that compiles
to load -> wait -> use per iteration (two-phase staging).  python tools/isa_chains.py file.s [substring]"""
import re, sys
s = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else ""
GAP = 14
for m in re.finditer(r'^(_Z\S+):\s*(?:;.*)?$', s, re.M):
    sym = m.group(1)
    if want not in sym:
        continue
    i = m.end(); j = s.find('s_endpgm', i)
    if j < 0:
        continue
    lines = [l.strip() for l in s[i:j].splitlines() if l.strip() and not l.strip().startswith((';', '.'))or re.match(r'\s*\.LBB', l)]
    labels = {}
    for k, l in enumerate(lines):
        mm = re.match(r'(\.LBB\S+):', l)
        if mm: labels[mm.group(1)] = k
    pairs = []
    last_load = None
    for k, l in enumerate(lines):
        if l.startswith('global_load') or l.startswith('buffer_load'):
            last_load = k
        elif l.startswith('s_waitcnt') and 'vmcnt(0)' in l and last_load is not None and k - last_load < GAP:
            pairs.append(k); last_load = None
    in_loop = 0
    for k in pairs:
        for kk in range(k, min(k + 40, len(lines))):
            mm = re.match(r's_cbranch_\w+\s+(\.LBB\S+)', lines[kk])
            if mm and labels.get(mm.group(1), 10 ** 9) < k:
                in_loop += 1; break
    if pairs:
        print(f"{len(pairs):3d} load->wait pairs, {in_loop:2d} in loops, {len(lines):5d} instr  {sym[:110]}")
