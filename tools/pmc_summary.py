"""Per-kernel MFMA utilisation from a `rocprofv3 --pmc MfmaUtil SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
--kernel-trace --output-format csv` pass (tools/pmc_mfma.sh).  Only the steady-state tail of the run
(the last ``--tail`` fraction of the dispatches) is summarised.

MfmaUtil is rocprofiler-sdk's own derived counter for gfx950 (counter_defs.yaml):
    100 * sum(SQ_VALU_MFMA_BUSY_CYCLES) / (max(GRBM_GUI_ACTIVE) * SIMD_NUM)
i.e. the share of SIMD-cycles (1024 SIMDs) in which the matrix pipe was busy while the kernel ran.
Per kernel name we report the GUI_ACTIVE-weighted mean, i.e. sum(busy) / (sum(active) * 1024)."""
import argparse
import collections
import csv
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--tail", type=float, default=0.5)
    ap.add_argument("--top", type=int, default=25)
    a = ap.parse_args()
    csv.field_size_limit(1 << 30)
    per_dispatch = collections.defaultdict(dict)
    names = {}
    with open(a.csv, newline="") as f:
        for row in csv.DictReader(f):
            d = int(row["Dispatch_Id"])
            per_dispatch[d][row["Counter_Name"]] = float(row["Counter_Value"])
            names[d] = row["Kernel_Name"]
    ids = sorted(per_dispatch)
    ids = ids[int(len(ids) * (1 - a.tail)):]
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
    for d in ids:
        c = per_dispatch[d]
        busy, act = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("GRBM_GUI_ACTIVE", 0.0)
        e = agg[names[d]]
        e[0] += 1; e[1] += busy; e[2] += act; e[3] += c.get("MfmaUtil", 0.0) * act
    tot_busy = sum(e[1] for e in agg.values())
    tot_act = sum(e[2] for e in agg.values())
    # GRBM_GUI_ACTIVE as exported is either max-reduced (per-XCD clock) or summed over the 8 XCDs;
    # calibrate with the tool's own MfmaUtil so that the per-kernel figure uses ITS normalisation
    tot_util_w = sum(e[3] for e in agg.values())
    print(f"dispatches summarised: {len(ids)} (tail {a.tail:g} of the run); kernels: {len(agg)}")
    print(f"all kernels: MfmaUtil (GUI_ACTIVE-weighted) = {tot_util_w / max(tot_act, 1):.2f} %   "
          f"raw busy/(active*1024) = {100 * tot_busy / max(tot_act * 1024, 1):.2f} %")
    print(f"{'n':>6} {'share of GPU-active':>20} {'MfmaUtil %':>11}  kernel")
    for k, e in sorted(agg.items(), key=lambda kv: -kv[1][2])[:a.top]:
        print(f"{e[0]:6d} {100 * e[2] / max(tot_act, 1):19.1f}% {e[3] / max(e[2], 1):11.2f}  {k[:110]}")


if __name__ == "__main__":
    sys.exit(main())
