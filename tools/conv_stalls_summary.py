"""Join the counter passes of tools/conv_stalls.sh per kernel and print where the wave-cycles go.

Units (MI355X_MICROARCH.md, per-instruction constants): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count QUAD-cycles
summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs; SQ_BUSY_CYCLES cycles summed over shader
engines' SQs.  A v_mfma_f32_16x16x4_f32 occupies its SIMD's matrix pipe for 32 cycles, so

    mfma_busy_expected = 32 * N_mfma            N_mfma = flop / 2048   (16 x 16 x 4 x 2 flop per wave-instruction)
    pipe utilisation   = mfma_busy / (1024 SIMDs x kernel duration x clock)

The clock is not exported: GRBM_GUI_ACTIVE (cycles the GPU was active during the dispatch; under counter collection
dispatches are serialised, so it includes the launch's own ramp) stands in for duration x clock.  `algorithmic` is
flop / duration / 157.3 TF from the pass's own kernel trace, for reconciliation with bench.py's roofline rows."""
import collections
import csv
import re
import sys

csv.field_size_limit(1 << 30)
passes, trace = sys.argv[1:4], sys.argv[4] if len(sys.argv) > 4 else None
val = collections.defaultdict(lambda: collections.defaultdict(list))     # kernel -> counter -> per-dispatch values
for p in passes:
    per = collections.defaultdict(dict)
    name = {}
    for row in csv.DictReader(open(p, newline="")):
        d = int(row["Dispatch_Id"])
        per[d][row["Counter_Name"]] = per[d].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
        name[d] = row["Kernel_Name"]
    for d, cs in per.items():
        for c, v in cs.items():
            if c == "GRBM_GUI_ACTIVE" and p != passes[0]:
                continue
            val[name[d]][c].append(v)
dur = collections.defaultdict(list)
if trace:
    for row in csv.DictReader(open(trace, newline="")):
        dur[row["Kernel_Name"]].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))


def mean(xs, skip=1):
    xs = xs[skip:] if len(xs) > skip else xs          # the first dispatch of a kernel carries its cold start
    return sum(xs) / max(len(xs), 1)


def flops(kname):
    m = re.search(r"conv3x3(_bwd)?_kernel<(\d+), (\d+)", kname)
    if not m:
        return None
    c, hw = int(m.group(2)), int(m.group(3))
    return (2 if m.group(1) else 1) * 2.0 * 128 * hw * hw * c * c * 9


print(__doc__.split("\n\n")[1])
print()
hdr = ("kernel", "us", "algor.", "pipe", "busy/exp", "parked", "issue-st", "of it LDS", "active", "LDS act", "VALU act",
       "VMEM act", "LDS ins/MFMA", "bank conf")
print("%-52s %6s %6s %6s %8s %7s %8s %9s %7s %8s %8s %8s %12s %9s" % hdr)
for k in sorted(val, key=lambda k: -mean(val[k].get("SQ_WAVE_CYCLES", [0]))):
    f = flops(k)
    if f is None:
        continue
    v = {c: mean(x) for c, x in val[k].items()}
    wc = v.get("SQ_WAVE_CYCLES", 0.0) or 1.0
    us = mean(dur.get(k, [0])) / 1e3
    n_mfma = f / 2048.0
    busy = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    gui = v.get("GRBM_GUI_ACTIVE", 0.0) or 1.0
    short = re.sub(r"void conv::|\(.*", "", k)
    print("%-52s %6.2f %6.3f %6.3f %8.3f %6.1f%% %7.1f%% %8.1f%% %6.1f%% %7.1f%% %7.1f%% %7.1f%% %12.2f %8.1f%%" % (
        short[:52], us, f / (us * 1e-6) / 157.3e12 if us else 0, busy / (1024.0 * gui), busy / (32.0 * n_mfma),
        100 * v.get("SQ_WAIT_ANY", 0) / wc, 100 * v.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * v.get("SQ_WAIT_INST_LDS", 0) / wc,
        100 * v.get("SQ_ACTIVE_INST_ANY", 0) / wc, 100 * v.get("SQ_ACTIVE_INST_LDS", 0) / wc,
        100 * v.get("SQ_ACTIVE_INST_VALU", 0) / wc, 100 * v.get("SQ_ACTIVE_INST_VMEM", 0) / wc,
        v.get("SQ_INSTS_LDS", 0) / n_mfma, 100 * v.get("SQ_LDS_BANK_CONFLICT", 0) / max(v.get("SQ_LDS_IDX_ACTIVE", 0), 1.0)))
print()
print("columns: us = mean dispatch duration in the first pass's kernel trace (serialised by the profiler); algor. = flop / us / "
      "157.3 TF; pipe = SQ_VALU_MFMA_BUSY_CYCLES / (1024 x GRBM_GUI_ACTIVE); busy/exp = SQ_VALU_MFMA_BUSY_CYCLES / (32 x N_mfma) "
      "(1.0 = the counter counts what the arithmetic says); parked = SQ_WAIT_ANY (s_waitcnt / barrier), issue-st = "
      "SQ_WAIT_INST_ANY, active = SQ_ACTIVE_INST_ANY, each as a share of SQ_WAVE_CYCLES; LDS ins/MFMA = SQ_INSTS_LDS per "
      "wave-MFMA; bank conf = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE")
print("raw means per dispatch:")
for k in sorted(val):
    if flops(k) is None:
        continue
    print(" ", re.sub(r"void conv::|\(.*", "", k)[:60], {c: round(mean(x)) for c, x in sorted(val[k].items())})
