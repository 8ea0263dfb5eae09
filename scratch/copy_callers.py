"""Who calls aten::copy_ on big tensors during training-driver steps?  Wraps Tensor.copy_,
.contiguous, .clone and torch.cat and counts Python call sites by bytes moved.  argv[1]: a
configuration of train_config.py."""
import os, sys, traceback, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(ROOT, "scratch", "train_config.py")).read()
head = src.split("t0 = time.perf_counter()")[0]
exec(head)
import torch
seen = collections.Counter()
count = collections.Counter()


def site():
    st = traceback.extract_stack(limit=8)[:-2]
    return " < ".join(f"{os.path.basename(s.filename)}:{s.lineno}" for s in st[-4:])


def wrap(owner, name, size_of):
    inner = getattr(owner, name)

    def f(*a, **k):
        out = inner(*a, **k)
        try:
            n = size_of(out, a)
            if n >= (1 << 22):
                key = (name, site())
                seen[key] += n
                count[key] += 1
        except Exception:
            pass
        return out
    setattr(owner, name, f)


wrap(torch.Tensor, "copy_", lambda out, a: out.numel() * out.element_size())
wrap(torch.Tensor, "contiguous", lambda out, a: 0 if out.data_ptr() == a[0].data_ptr() else out.numel() * out.element_size())
wrap(torch.Tensor, "clone", lambda out, a: out.numel() * out.element_size())
wrap(torch, "cat", lambda out, a: out.numel() * out.element_size())
conf["model"]["params"][c["dataset"]]["epochs"] = 3
train.run(conf)
for k, v in seen.most_common(14):
    print(f"{v / 1e6:9.1f} MB in {count[k]:4d} calls  {k[0]:10s} {k[1]}")
