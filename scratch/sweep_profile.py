"""Host-side profile of the refresh sweep (mini_inference_vr over all partitions): cProfile of
bench.py with the sweep on, reduced to what runs below _sweep_layers."""
import cProfile, pstats, sys, os, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = ["bench.py", "--no-cpu-baseline", "--sweep", "on", "--epoch-pass", "off", "--steps", "5",
            "--warmup", "2"]
os.environ["BENCH_PRIME_SECONDS"] = "0.2"
import bench
pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
s = io.StringIO()
st = pstats.Stats(pr, stream=s)
st.sort_stats("cumulative").print_stats("_sweep|run_layer|_aggregate|_push_rows|_update|spmm|copy_row_chunks|apply|forward_after|_folded|_embed", 40)
print(s.getvalue(), file=sys.stderr)
