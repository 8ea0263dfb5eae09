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
st = pstats.Stats(pr, stream=s)
st.sort_stats("cumtime").print_stats("eager_step|VR_|aggregate.py|nn.py|ops.py|optim.py|base.py|run_backward|backward|addmm|mm|bmm|empty", 60)
print(s.getvalue(), file=sys.stderr)
