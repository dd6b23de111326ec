import cProfile, pstats, sys, io
sys.argv = ["bench.py", "--steps", "4000", "--warmup", "300", "--cpu-budget", "0", "--sweep-log2", "0", "--no-kernel-timing"]
sys.path.insert(0, "/root/repo")
import bench
pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
