"""cProfile of bench.py's step loop (host side)."""
import cProfile, pstats, sys, os, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = ["bench.py", "--no-cpu-baseline", "--sweep", "off", "--epoch-pass", "off", "--steps", "60", "--graphs", os.environ.get("GRAPHS", "off")]
os.environ["BENCH_PRIME_SECONDS"] = "0.5"
import bench
pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45)
print(s.getvalue(), file=sys.stderr)
