"""cProfile of train.run on one of scratch/train_config.py's configurations (argv[1])."""
import cProfile, pstats, os, sys, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(ROOT, "scratch", "train_config.py")).read()
head = src.split("t0 = time.perf_counter()")[0]
exec(head)
conf["model"]["params"][c["dataset"]]["epochs"] = 6
pr = cProfile.Profile()
pr.enable()
res = train.run(conf)
pr.disable()
for h in res["history"]:
    print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in h.items()}), flush=True)
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(60)
