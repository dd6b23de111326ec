"""profiles/<round>_conv_kernels_pmc_hbm.txt (tools/pmc_hbm_summary.py's table of a `rocprofv3 --pmc FETCH_SIZE
WRITE_SIZE` pass over tools/conv_pmc.py) -> profiles/pmc_traffic.json: HBM bytes per launch for the kernels bench.py
lists in `roofline_kernels`, keyed by bench.py's row names.  bench.py cannot collect counters itself (they need the
rocprofv3 wrapper and their own pass); it copies these per-launch figures into `roofline.traffic` and names this file
as the source.  `python tools/pmc_to_json.py profiles/r04_conv_kernels_pmc_hbm.txt profiles/r04_bn_16x32_pmc_hbm.txt ... > profiles/pmc_traffic.json`"""
import json
import os
import re
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bnn_priors_amd import _hip  # noqa: E402

rows = {}
for path in sys.argv[1:]:
    stage = re.search(r"bn_(\d+)x(\d+)", path)           # a BatchNorm pass of one stage (tools/r04_pmc.sh names it)
    for line in open(path):
        m = re.match(r"\s*(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+(.*?)void ((?:conv|bn)::\w+)<([^>]*)>", line)
        if not m:
            continue
        n, read_mb, read2_mb, write_mb, _, name, args = m.groups()
        a = [v.strip() for v in args.split(",")]
        if name == "conv::conv3x3_kernel" and a[3:] == ["false", "true"]:
            key = f"conv::conv3x3_kernel<{a[0]},{a[1]},{a[2]},stats>"
        elif name == "conv::conv3x3_bwd_kernel":
            tag = ",".join(t for t, on in (("ADD", a[3]), ("SUMS", a[4])) if on == "true")
            key = f"conv::conv3x3_bwd_kernel<{a[0]},{a[1]},{a[2]}" + ("," + tag if tag else "") + ">"
        elif name == "conv::fused_bwd_kernel":
            key = f"conv::fused_bwd_kernel<{a[0]},{a[1]},{a[2]}>"
        elif name in ("bn::apply_kernel", "bn::bwd_dx_kernel") and stage:
            flags = ",".join(t for t, on in zip(("relu", "residual", "rsums"), a) if on == "true")
            key = f"{name}<{flags}> {stage.group(1)}@{stage.group(2)}^2"
        else:
            continue
        # (gfx950: FETCH_SIZE reports half of the bytes of wide coalesced reads -- MI355X_MICROARCH.md, HBM section --
        # so the doubled column; WRITE_SIZE as reported: it matches the algorithmic write bytes of these kernels)
        rows[key] = {"read_bytes": round(float(read2_mb) * 1e6), "write_bytes": round(float(write_mb) * 1e6),
                     "launches_averaged": int(n), "from": path}
        ratio = re.search(r"([\d.]+)/([\d.]+)\s+([\d.]+) \(R", line)      # algorithmic read / write MB, traffic / algorithmic
        if ratio:
            rows[key]["algorithmic_bytes"] = round((float(ratio.group(1)) + float(ratio.group(2))) * 1e6)
            rows[key]["traffic_over_algorithmic"] = float(ratio.group(3))
print(json.dumps({"source": ", ".join(sys.argv[1:]), "source_sha": _hip.library_sha(),
                  "collected_by": "tools/pmc_hbm.sh (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate counter-only passes "
                                  "with --kernel-trace) over tools/conv_pmc.py and tools/bn_pmc.py at n = 128",
                  "kernels": rows}, indent=1))
