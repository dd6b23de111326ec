"""Per-kernel HBM-side traffic from two counter-only rocprofv3 passes (tools/pmc_hbm.sh):
    pmc_hbm_summary.py <FETCH_SIZE counter_collection.csv> <WRITE_SIZE counter_collection.csv> [--tail 0.5]

FETCH_SIZE / WRITE_SIZE are rocprofiler's derived counters in KiB, built from the L2's memory-side request
counters (TCC_EA0_RDREQ / _WRREQ): bytes that left the XCDs' L2s towards Infinity Cache / HBM.  On gfx950
FETCH_SIZE tallies the 128-byte requests of wide coalesced reads (16 B per lane) at 64 bytes, i.e. reports HALF
of their bytes (MI355X_MICROARCH.md "HBM"): the `read x2` column applies that correction and is the one to
compare with a byte count for kernels whose reads are such accesses (all of the ones named below); WRITE_SIZE is
used as reported.  Averages are per launch over the last --tail fraction of the run's dispatches.

For the kernels whose algorithmic byte count is known the ratio traffic / algorithmic
is printed: ~1 means every byte crosses the L2 boundary once, > 1 re-reads, < 1 hits in L2 across launches
(working sets below the 32 MiB of L2 / 256 MiB of Infinity Cache never reach HBM at all).
"""
import argparse
import collections
import csv
import re


def load(path, counter):
    csv.field_size_limit(1 << 30)
    rows = {}
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                rows[int(r["Dispatch_Id"])] = (r["Kernel_Name"], float(r["Counter_Value"]),
                                               int(r.get("Grid_Size", 0) or 0))
    return rows


BN_SHAPE = None      # (n, channels, side) of the one stage tools/bn_pmc.py ran (--bn-shape): the names do not carry it


def algorithmic_bytes(name, grid):
    "bytes one launch has to move at least once (reads, writes), or None"
    m = re.search(r"bn::(apply_kernel|bwd_dx_kernel)<(true|false), (true|false)", name)
    if m and BN_SHAPE:
        n, c, hw = BN_SHAPE
        act = 4 * n * c * hw * hw
        if m.group(1) == "apply_kernel":      # reads x (+ residual), writes y
            return act * (2 if m.group(3) == "true" else 1), act
        # reads dout, out (ReLU mask), x; writes dx -- the ReLU-less instantiation (the step's launches since round 4: the
        # gradient arrives masked, bnlink.PREMASK) does not read `out`
        return (3 if m.group(2) == "true" else 2) * act, act
    m = re.search(r"step_kernel(_stream|_indirect)?<float, \d+(, (true|false), (\d))?", name)
    if m:                                  # 28 B per element: 4 reads + 3 writes of 4 B
        items = 4 if (m.group(1) == "_stream" or m.group(4) is None) else int(m.group(4))
        elems = grid // 256 * 1024 * items
        return 16 * elems, 12 * elems
    m = re.search(r"conv3x3_kernel<(\d+), (\d+), (\d+), (true|false), (true|false)>", name)
    if m:
        c, hw = int(m.group(1)), int(m.group(2))
        n = grid // 256 // ((hw // 8) * (c // 16))
        act = 4 * n * c * hw * hw
        return act + 36 * c * c, act
    m = re.search(r"conv3x3_bwd_kernel<(\d+), (\d+), (\d+), (true|false), (true|false)(?:, (?:true|false))?>", name)
    if m:
        c, hw = int(m.group(1)), int(m.group(2))
        # grid = dgrad blocks (n * bands * ct) + wrw blocks (n * bands / 2 * ct)
        n = grid // 256 * 2 // (3 * (hw // 8) * (c // 16))
        act = 4 * n * c * hw * hw
        slabs = n * (hw // 8) // 2
        extra = (act if m.group(4) == "true" else 0) + (2 * act if m.group(5) == "true" else 0)   # epilogue operands
        # reads x, dy, w (+ the shortcut's gradient -- stored masked by its producer since round 4: its activation is not
        # read --, + the next BatchNorm's y / out); writes dx + partial slabs
        return 2 * act + 36 * c * c + extra, act + 36 * c * c * slabs
    m = re.search(r"fused_bwd_kernel<(\d+), (\d+), (\d+), (true|false)>", name)
    if m:
        c, hw = int(m.group(1)), int(m.group(2))
        n = grid // 256 * 2 // (3 * (hw // 8) * (c // 16))
        act = 4 * n * c * hw * hw
        slabs = n * (hw // 8) // 2
        extra = 2 * act if m.group(4) == "true" else 0              # the shortcut's dout / out in the epilogue
        return 4 * act + 36 * c * c + extra, act + 36 * c * c * slabs   # reads x, dout, out, y, w; writes dx + slabs
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_csv")
    ap.add_argument("write_csv")
    ap.add_argument("--tail", type=float, default=0.5)
    ap.add_argument("--top", type=int, default=30)
    ap.add_argument("--bn-shape", default=None, help="n,channels,side of the BatchNorm stage of tools/bn_pmc.py")
    a = ap.parse_args()
    global BN_SHAPE
    if a.bn_shape:
        BN_SHAPE = tuple(int(v) for v in a.bn_shape.split(","))
    fetch, write = load(a.fetch_csv, "FETCH_SIZE"), load(a.write_csv, "WRITE_SIZE")
    print(f"dispatches with FETCH_SIZE: {len(fetch)}, with WRITE_SIZE: {len(write)}; tail {a.tail:g} summarised")
    agg = collections.OrderedDict()
    for rows, col in ((fetch, 0), (write, 1)):
        ids = sorted(rows)
        for d in ids[int(len(ids) * (1 - a.tail)):]:
            name, kib, grid = rows[d]
            e = agg.setdefault((name, grid), [[0, 0.0], [0, 0.0]])
            e[col][0] += 1
            e[col][1] += kib * 1024
    print(f"{'n':>5} {'read MB':>10} {'read x2 MB':>11} {'write MB':>10} {'algo R/W MB':>15} {'traffic/algo':>13}  kernel (grid)")
    order = sorted(agg.items(), key=lambda kv: -(2 * kv[1][0][1] + kv[1][1][1]))[:a.top]
    for (name, grid), ((nf, bf), (nw, bw)) in order:
        rd = bf / max(nf, 1)
        wr = bw / max(nw, 1)
        alg = algorithmic_bytes(name, grid)
        if alg:
            ratio = f"{(2 * rd + wr) / (alg[0] + alg[1]):.3f} (R {2 * rd / alg[0]:.3f}, W {wr / alg[1]:.3f})"
            algs = f"{alg[0] / 1e6:.2f}/{alg[1] / 1e6:.2f}"
        else:
            ratio, algs = "", ""
        print(f"{max(nf, nw):5d} {rd / 1e6:10.3f} {2 * rd / 1e6:11.3f} {wr / 1e6:10.3f} {algs:>15} {ratio:>13}  "
              f"{name[:84]} ({grid})")


if __name__ == "__main__":
    main()
