"""Generates tests/golden/relabel_golden.npz by RUNNING THE REFERENCE's own relabel
operators (csrc/relabel.cpp + csrc/cpu/relabel_cpu.cpp, compiled in place by
oracle/build_ref.py).  Run it in a process of its own, in the authoring container
(where /root/reference exists):

    python tests/golden/make_relabel_fixtures.py

The fixture holds only data: seeded synthetic inputs and the outputs the reference
produced for them.  Cases cover both operators, bipartite True/False, with and
without edge values (float32 / float64 / int64 values), contiguous partition-like
index ranges, unions of ranges, random subsets, a node without edges, an empty
index set and duplicate indices.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import build_ref  # noqa: E402


def random_graph(rng, n, avg_deg, isolated=()):
    """Symmetric, de-duplicated, sorted CSR without self loops."""
    m = n * avg_deg // 2
    a = rng.integers(0, n, size=m)
    b = rng.integers(0, n, size=m)
    keep = a != b
    a, b = a[keep], b[keep]
    for v in isolated:
        k = (a != v) & (b != v)
        a, b = a[k], b[k]
    r = np.concatenate([a, b])
    c = np.concatenate([b, a])
    key = np.unique(r * n + c)
    r, c = key // n, key % n
    rowptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(r, minlength=n), out=rowptr[1:])
    return rowptr, c.astype(np.int64)


def ring_with_chord():
    n = 8
    adj = {i: {(i - 1) % n, (i + 1) % n} for i in range(n)}
    adj[1].add(5)
    adj[5].add(1)
    rowptr = [0]
    col = []
    for i in range(n):
        col += sorted(adj[i])
        rowptr.append(len(col))
    return np.array(rowptr, dtype=np.int64), np.array(col, dtype=np.int64)


def main():
    one_hop, within = build_ref.load()
    rng = np.random.default_rng(20261111)
    cases = {}
    graphs = {"kat": ring_with_chord()}
    graphs["g50"] = random_graph(rng, 50, 4, isolated=(7,))
    graphs["g300"] = random_graph(rng, 300, 8)
    graphs["g800"] = random_graph(rng, 800, 12, isolated=(0, 799))

    def idx_sets(name, n):
        if name == "kat":
            return {"survey": np.array([0, 1, 5, 6, 7]), "all": np.arange(n)}
        out = {
            "range": np.arange(n // 4, n // 2),
            "two_ranges": np.concatenate([np.arange(0, n // 8), np.arange(n // 2, n // 2 + n // 8)]),
            "random": rng.permutation(n)[: n // 5],
            "empty": np.zeros(0, dtype=np.int64),
            "with_isolated": np.array([7 if n == 50 else 0, 1, 2, 3]),
            "dups": np.array([3, 4, 3, 5, 4]),
        }
        return out

    k = 0
    for gname, (rowptr, col) in graphs.items():
        n = rowptr.size - 1
        nnz = col.size
        values = {
            "none": None,
            "f32": rng.standard_normal(nnz).astype(np.float32),
            "f64": np.arange(nnz, dtype=np.float64),
            "i64": rng.integers(-5, 5, size=nnz).astype(np.int64),
        }
        for iname, idx in idx_sets(gname, n).items():
            for vname, val in values.items():
                if gname != "kat" and iname not in ("range", "random") and vname in ("f64", "i64"):
                    continue
                for bip in (True, False):
                    for opname, op in (("one_hop", one_hop), ("within", within)):
                        tv = None if val is None else torch.from_numpy(val.copy())
                        o_rp, o_col, o_val, o_nid = op(
                            torch.from_numpy(rowptr.copy()), torch.from_numpy(col.copy()), tv,
                            torch.from_numpy(idx.astype(np.int64).copy()), bip)
                        tag = f"c{k:03d}"
                        k += 1
                        cases[f"{tag}/meta"] = np.array(
                            [gname, iname, vname, str(int(bip)), opname])
                        cases[f"{tag}/idx"] = idx.astype(np.int64)
                        cases[f"{tag}/out_rowptr"] = o_rp.numpy()
                        cases[f"{tag}/out_col"] = o_col.numpy()
                        cases[f"{tag}/out_n_id"] = o_nid.numpy()
                        if o_val is not None:
                            cases[f"{tag}/out_value"] = o_val.numpy()
        cases[f"graph/{gname}/rowptr"] = rowptr
        cases[f"graph/{gname}/col"] = col
        for vname, val in values.items():
            if val is not None:
                cases[f"graph/{gname}/value_{vname}"] = val
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "relabel_golden.npz")
    np.savez_compressed(out, **cases)
    print(f"wrote {out}: {k} cases, {os.path.getsize(out) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
