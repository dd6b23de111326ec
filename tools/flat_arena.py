"""The fused sampler kernel on one flat fp32 segment (bandwidth-bound regime).
    python tools/flat_arena.py [--log2 26] [--iters 20] [--sweep]
Prints GB/s of algorithmic traffic (28 B/element) from HIP events around the kernel."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--log2", type=int, default=26)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--sweep", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda", 0)
for lg in (range(18, 30, 2) if a.sweep else [a.log2]):
    print(json.dumps(bench.flat_arena_point(lg, dev, iters=a.iters)))
