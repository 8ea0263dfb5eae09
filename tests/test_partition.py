"""Partitioner stand-in (native BFS blocks + refinement) and the partition file formats."""
from collections import deque

import pytest
import torch

import torch_geometric_autoscale as tga
import importlib
M = importlib.import_module("torch_geometric_autoscale.metis")  # the package exports a function of that name
from torch_geometric_autoscale import ops


def _bfs_python(rowptr, col, n, num_parts, seeds):
    """Plain-Python statement of the block-growing rule the native routine implements."""
    target = -(-n // num_parts)
    cluster = [-1] * n
    cursor, part, filled = 0, 0, 0
    queue = deque()
    for _ in range(n):
        if not queue:
            while cluster[seeds[cursor]] != -1:
                cursor += 1
            queue.append(seeds[cursor])
            cluster[seeds[cursor]] = -2
        v = queue.popleft()
        cluster[v] = part
        filled += 1
        if filled == target and part < num_parts - 1:
            for w in queue:
                cluster[w] = -1
            queue.clear()
            part, filled = part + 1, 0
            continue
        for e in range(rowptr[v], rowptr[v + 1]):
            w = col[e]
            if cluster[w] == -1:
                cluster[w] = -2
                queue.append(w)
    return cluster


def _graph(n=3000, parts=7, seed=0):
    adj, _ = tga.synthetic_graph(n, n * 8, parts, cross_fraction=0.2, seed=seed)
    return adj


def test_native_bfs_matches_python_rule():
    adj = _graph()
    rowptr, col, _ = adj.csr()
    n = adj.size(0)
    for parts, seed in ((1, 0), (7, 0), (16, 3), (3000, 1)):
        seeds = torch.randperm(n, generator=torch.Generator().manual_seed(seed))
        got = ops.partition_bfs(rowptr, col, parts, seeds)
        want = _bfs_python(rowptr.tolist(), col.tolist(), n, parts, seeds.tolist())
        assert got.tolist() == want
        sizes = torch.bincount(got, minlength=parts)
        assert int(sizes.max()) <= -(-n // parts)


def test_refinement_lowers_the_cut_and_keeps_balance():
    adj = _graph(6000, 10, seed=2)
    rowptr, col, _ = adj.csr()
    n, parts = adj.size(0), 10
    base = M.bfs_clusters(adj, parts, seed=0)
    limit = int(-(-n // parts) * 1.03) + 1
    refined, moves = ops.partition_refine(rowptr, col, parts, base, limit, sweeps=6)
    assert moves > 0
    assert M.edge_cut(adj, refined) < M.edge_cut(adj, base)
    assert int(torch.bincount(refined, minlength=parts).max()) <= limit
    assert int(torch.bincount(refined, minlength=parts).min()) >= 1
    # the input vector is left alone; zero sweeps change nothing
    same, moves0 = ops.partition_refine(rowptr, col, parts, base, limit, sweeps=0)
    assert moves0 == 0 and torch.equal(same, base)
    with pytest.raises(RuntimeError):
        ops.partition_refine(rowptr, col, parts, torch.full((n,), parts), limit)


def test_metis_contract_and_permute():
    adj = tga.gcn_norm(_graph(2000, 5, seed=4).set_diag())
    n = adj.size(0)
    perm, ptr = tga.metis(adj, 6)
    assert ptr[0] == 0 and ptr[-1] == n and ptr.numel() == 7
    assert torch.equal(torch.sort(perm).values, torch.arange(n))
    data = tga.Data(adj_t=adj, num_nodes=n, x=torch.arange(n, dtype=torch.float32).view(-1, 1))
    out = tga.permute(data, perm)
    assert torch.equal(out.x[:, 0].long(), perm)
    # an externally supplied cluster vector is used as is
    cluster = torch.arange(n) % 4
    perm2, ptr2 = tga.metis(adj, 4, cluster=cluster)
    assert torch.equal(cluster[perm2], torch.sort(cluster).values)
    assert ptr2.tolist() == [0, 500, 1000, 1500, 2000]


def test_metis_graph_file_known_answer(tmp_path):
    # path 0-1-2 plus edge 0-2 and a self loop on 1 (dropped by the format)
    row = torch.tensor([0, 0, 1, 1, 1, 2, 2])
    col = torch.tensor([1, 2, 0, 1, 2, 0, 1])
    adj = tga.SparseTensor(row=row, col=col, sparse_sizes=(3, 3))
    p = tmp_path / "g.graph"
    M.write_metis_graph(adj, str(p))
    assert p.read_text() == "3 3\n2 3\n1 3\n1 2\n"
    part = tmp_path / "g.graph.part.2"
    part.write_text("0\n1\n0\n")
    cluster = M.read_metis_partition(str(part), 3)
    assert cluster.tolist() == [0, 1, 0]
    perm, ptr = tga.metis(adj, 2, cluster=cluster)
    assert perm.tolist() == [0, 2, 1] and ptr.tolist() == [0, 2, 3]
    with pytest.raises(ValueError):
        M.read_metis_partition(str(part), 4)


def test_partition_file_round_trip(tmp_path):
    adj = _graph(1500, 4, seed=6)
    perm, ptr = tga.metis(adj, 5)
    path = str(tmp_path / "parts.npz")
    M.save_partition(path, perm, ptr)
    perm2, ptr2 = M.load_partition(path, adj.size(0))
    assert torch.equal(perm, perm2) and torch.equal(ptr, ptr2)
    with pytest.raises(ValueError):
        M.load_partition(path, adj.size(0) + 1)
    bad = perm.clone()
    bad[0] = bad[1]
    M.save_partition(path, bad, ptr)
    with pytest.raises(ValueError):
        M.load_partition(path)
