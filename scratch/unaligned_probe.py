"""Which aggregation calls of a training-driver run get operands whose rows are not 16-byte
aligned (they fall to the whole-row kernel)?  argv[1]: a configuration of train_config.py."""
import os, sys, traceback, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(ROOT, "scratch", "train_config.py")).read()
head = src.split("t0 = time.perf_counter()")[0]
exec(head)
from torch_geometric_autoscale import ops
seen = collections.Counter()


def wrap(name):
    inner = getattr(ops, name)

    def f(*a, **k):
        import torch
        bad = []
        for i, t in list(enumerate(a)) + list(k.items()):
            if isinstance(t, torch.Tensor) and t.dim() == 2 and t.dtype == torch.float32:
                if t.stride(0) % 4 or t.data_ptr() % 16:
                    bad.append((i, tuple(t.shape), t.stride()))
        if bad:
            st = traceback.extract_stack(limit=9)[:-1]
            key = (name, str(bad), " < ".join(f"{os.path.basename(s.filename)}:{s.lineno}" for s in st[-7:]))
            seen[key] += 1
        return inner(*a, **k)
    setattr(ops, name, f)


for n in ("spmm_csr", "spmm_csr_incagg"):
    wrap(n)
conf["model"]["params"][c["dataset"]]["epochs"] = 2
train.run(conf)
for k, v in seen.most_common(12):
    print(v, k)
