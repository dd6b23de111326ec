"""Join the counter passes of tools/conv_stalls.sh per kernel and print where the wave-cycles go.

Units (MI355X_MICROARCH.md, per-instruction constants): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count QUAD-cycles
summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs; SQ_BUSY_CYCLES cycles summed over shader
engines' SQs.  A v_mfma_f32_16x16x4_f32 occupies its SIMD's matrix pipe for 32 cycles, so

    mfma_busy_expected = 32 * N_mfma            N_mfma = flop / 2048   (16 x 16 x 4 x 2 flop per wave-instruction)
    pipe utilisation   = mfma_busy / (1024 SIMDs x kernel duration x clock)

The clock is not exported (GRBM_GUI_ACTIVE as collected here is summed over the XCDs and covers more than the
dispatch), so `pipe` uses the dispatch duration of the pass's own kernel trace at the nominal 2.4 GHz; it equals
`algor.` = flop / duration / 157.3 TF by construction when busy/exp = 1 -- which is the reconciliation: the counter
counts exactly 32 cycles per v_mfma_f32_16x16x4_f32, and the fraction of peak is the algorithmic one (the round-3
`MfmaUtil` file, normalised by rocprofiler's gfx94x formula, was below it and is retired).
`wave us` = SQ_WAVE_CYCLES x 4 / SQ_WAVES at 2.4 GHz: the mean lifetime of a wave; `mfma us` = the time the two waves
that share a SIMD need for their MFMAs at the pipe's rate (2 x N_mfma_per_wave x 32 cycles): what is left of `wave us`
is spent with no MFMA issued on that SIMD unless the co-resident workgroup is out of phase."""
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
hdr = ("kernel", "us", "algor.", "pipe", "busy/exp", "wave us", "mfma us", "parked", "issue-st", "of it LDS", "active", "LDS act", "VALU act",
       "VMEM act", "LDS ins/MFMA", "bank conf")
print("%-52s %6s %6s %6s %8s %7s %7s %7s %8s %9s %7s %8s %8s %8s %12s %9s" % hdr)
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
    waves = v.get("SQ_WAVES", 0.0) or 1.0
    wave_us = wc * 4.0 / waves / 2400.0
    resident = 2.0 if "bwd" not in short and True else 2.0
    mfma_us = resident * (n_mfma / waves) * 32.0 / 2400.0
    print("%-52s %6.2f %6.3f %6.3f %8.3f %7.2f %7.2f %6.1f%% %7.1f%% %8.1f%% %6.1f%% %7.1f%% %7.1f%% %7.1f%% %12.2f %8.1f%%" % (
        short[:52], us, f / (us * 1e-6) / 157.3e12 if us else 0, busy / (1024.0 * us * 2400.0) if us else 0, busy / (32.0 * n_mfma),
        wave_us, mfma_us,
        100 * v.get("SQ_WAIT_ANY", 0) / wc, 100 * v.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * v.get("SQ_WAIT_INST_LDS", 0) / wc,
        100 * v.get("SQ_ACTIVE_INST_ANY", 0) / wc, 100 * v.get("SQ_ACTIVE_INST_LDS", 0) / wc,
        100 * v.get("SQ_ACTIVE_INST_VALU", 0) / wc, 100 * v.get("SQ_ACTIVE_INST_VMEM", 0) / wc,
        v.get("SQ_INSTS_LDS", 0) / n_mfma, 100 * v.get("SQ_LDS_BANK_CONFLICT", 0) / max(v.get("SQ_LDS_IDX_ACTIVE", 0), 1.0)))
print()
print("columns: us = mean dispatch duration in the first pass's kernel trace (serialised by the profiler); algor. = flop / us / "
      "157.3 TF; pipe = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x us x 2.4 GHz); busy/exp = SQ_VALU_MFMA_BUSY_CYCLES / (32 x N_mfma) "
      "(1.0 = the counter counts what the arithmetic says); parked = SQ_WAIT_ANY (s_waitcnt / barrier), issue-st = "
      "SQ_WAIT_INST_ANY, active = SQ_ACTIVE_INST_ANY, each as a share of SQ_WAVE_CYCLES; LDS ins/MFMA = SQ_INSTS_LDS per "
      "wave-MFMA; bank conf = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE")
print("raw means per dispatch:")
for k in sorted(val):
    if flops(k) is None:
        continue
    print(" ", re.sub(r"void conv::|\(.*", "", k)[:60], {c: round(mean(x)) for c, x in sorted(val[k].items())})
