"""cProfile of the device batch builder's host side (30 new products-sized batches)."""
import cProfile, pstats, os, sys
sys.argv = [sys.argv[0]]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(ROOT, "scratch", "build_trace.py")).read()
head, loop = src.split("for p in range(3):")
exec(head)
for p in range(3):
    build(p)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for p in range(10, 40):
    build(p)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
