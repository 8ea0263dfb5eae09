"""cProfile of bench.py's eagerly issued incremental-aggregation steps (host side)."""
import cProfile, pstats, sys, os, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = ["bench.py", "--no-cpu-baseline", "--sweep", "off", "--epoch-pass", "off", "--steps", "100",
            "--graphs", "off", "--mode", os.environ.get("MODE", "vr")]
os.environ["BENCH_PRIME_SECONDS"] = "0.3"
import bench
pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(40)
print(s.getvalue(), file=sys.stderr)
