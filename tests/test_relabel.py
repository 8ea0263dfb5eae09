"""The product's batch builder (host C++ behind the C ABI, bound as
torch.ops.torch_geometric_autoscale.relabel_*) -- bit-exact against the reference-made
fixtures, against the oracle on larger random inputs, and, when the reference build is
present (authoring container), against the reference itself run in a subprocess."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import cpu_path as cp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ns():
    import torch_geometric_autoscale  # noqa: F401
    return torch.ops.torch_geometric_autoscale


def test_product_relabel_matches_reference_fixtures(ns, golden_relabel):
    g = golden_relabel
    tags = sorted({k.split("/")[0] for k in g.files if k.startswith("c")})
    for tag in tags:
        gname, iname, vname, bip, op = g[f"{tag}/meta"]
        rowptr = torch.from_numpy(g[f"graph/{gname}/rowptr"])
        col = torch.from_numpy(g[f"graph/{gname}/col"])
        val = None if vname == "none" else torch.from_numpy(g[f"graph/{gname}/value_{vname}"])
        fn = ns.relabel_one_hop if op == "one_hop" else ns.relabel_one_hop_within_batch
        rp, c, v, nid = fn(rowptr, col, val, torch.from_numpy(g[f"{tag}/idx"]), bip == "1")
        assert rp.dtype == c.dtype == nid.dtype == torch.int64
        assert np.array_equal(rp.numpy(), g[f"{tag}/out_rowptr"]), tag
        assert np.array_equal(c.numpy(), g[f"{tag}/out_col"]), tag
        assert np.array_equal(nid.numpy(), g[f"{tag}/out_n_id"]), tag
        if val is None:
            assert v is None
        else:
            assert v.dtype == val.dtype and np.array_equal(v.numpy(), g[f"{tag}/out_value"]), tag


def _big_graph(seed, n=60000, deg=16):
    import torch_geometric_autoscale as tga
    adj, ptr = tga.synthetic_graph(n, n * deg, 12, seed=seed)
    rowptr, col, _ = adj.csr()
    val = torch.arange(col.numel(), dtype=torch.float32)
    return rowptr, col, val, ptr


@pytest.mark.parametrize("bipartite", [True, False])
def test_product_relabel_matches_oracle_on_partition_batches(ns, bipartite):
    rowptr, col, val, ptr = _big_graph(1)
    for parts in ([0], [3, 7], [11, 2, 5]):
        idx = torch.cat([torch.arange(int(ptr[p]), int(ptr[p + 1])) for p in parts])
        for fn, ofn in ((ns.relabel_one_hop, cp.relabel_one_hop),
                        (ns.relabel_one_hop_within_batch, cp.relabel_one_hop_within_batch)):
            got = fn(rowptr, col, val, idx, bipartite)
            want = ofn(rowptr.numpy(), col.numpy(), val.numpy(), idx.numpy(), bipartite)
            for a, b in zip(got, want):
                assert np.array_equal(a.numpy(), b)
    # repeated calls reuse the per-thread map: results must not depend on call history
    idx = torch.arange(int(ptr[4]), int(ptr[5]))
    a = ns.relabel_one_hop(rowptr, col, None, idx, True)
    ns.relabel_one_hop(rowptr, col, None, torch.arange(100), True)
    b = ns.relabel_one_hop(rowptr, col, None, idx, True)
    assert all(torch.equal(x, y) for x, y in zip((a[0], a[1], a[3]), (b[0], b[1], b[3])))


def test_relabel_error_behaviour(ns):
    rowptr, col = torch.tensor([0, 1, 2]), torch.tensor([1, 0])
    with pytest.raises(RuntimeError, match="one-dimensional"):
        ns.relabel_one_hop(rowptr, col, torch.zeros(2, 1), torch.tensor([0]), True)
    with pytest.raises(RuntimeError):
        ns.relabel_one_hop(rowptr, col, None, torch.tensor([2]), True)      # id outside the graph
    with pytest.raises(RuntimeError):
        ns.relabel_one_hop(rowptr, torch.tensor([5, 0]), None, torch.tensor([0]), True)
    rp, c, v, nid = ns.relabel_one_hop(rowptr, col, None, torch.zeros(0, dtype=torch.int64), True)
    assert rp.tolist() == [0] and c.numel() == 0 and nid.numel() == 0 and v is None


@pytest.mark.skipif(not os.path.isfile(os.path.join(ROOT, "oracle", "_ref", "ref_relabel.so")),
                    reason="reference build not present (only in the authoring container)")
def test_product_relabel_matches_live_reference(ns, tmp_path):
    """Fresh seeds, not in the committed fixtures: the reference's own operators run in a
    subprocess (both libraries register the same operator names, so never in this process)."""
    rowptr, col, val, ptr = _big_graph(7, n=20000)
    idx = torch.cat([torch.arange(int(ptr[2]), int(ptr[3])), torch.arange(int(ptr[9]), int(ptr[10]))])
    inp, outp = tmp_path / "in.npz", tmp_path / "out.npz"
    np.savez(inp, rowptr=rowptr.numpy(), col=col.numpy(), val=val.numpy(), idx=idx.numpy())
    code = f"""
import sys, numpy as np, torch
sys.path.insert(0, {ROOT!r})
from oracle import build_ref
one_hop, within = build_ref.load()
d = np.load({str(inp)!r})
t = lambda k: torch.from_numpy(d[k])
out = {{}}
for name, fn in (("one_hop", one_hop), ("within", within)):
    for bip in (True, False):
        rp, c, v, nid = fn(t("rowptr"), t("col"), t("val"), t("idx"), bip)
        for k, a in zip(("rp", "c", "v", "nid"), (rp, c, v, nid)):
            out[f"{{name}}_{{int(bip)}}_{{k}}"] = a.numpy()
np.savez({str(outp)!r}, **out)
"""
    subprocess.check_call([sys.executable, "-c", code])
    ref = np.load(outp)
    for name, fn in (("one_hop", ns.relabel_one_hop), ("within", ns.relabel_one_hop_within_batch)):
        for bip in (True, False):
            got = fn(rowptr, col, val, idx, bip)
            for k, a in zip(("rp", "c", "v", "nid"), got):
                assert np.array_equal(a.numpy(), ref[f"{name}_{int(bip)}_{k}"]), (name, bip, k)
