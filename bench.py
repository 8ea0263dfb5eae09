#!/usr/bin/env python3
"""Headline benchmark: edges aggregated per second of GCNII push/pull training steps on an
ogbn-products-sized partitioned graph (BASELINE.json configs[2]: GCN2, 150 parts, 8 layers,
hidden 256), one process per GPU.

A step = one partition batch through the hot path: 8 x (HIP CSR aggregation over the
batch's [B, M] one-hop adjacency -- its M - B out-of-batch source rows read in place from
the layer's History table -- and push of the B fresh rows into the next layer's table), the
dense GCNII updates, the loss, the backward pass (8 transposed aggregations) and the Adam
step.  Inputs (graph, features,
History tables) are resident in HBM before the timed region.  `value` counts every edge
that an aggregation launch processes: L * nnz(adj_t) forward; backward nnz(adj_t) for
layer 0 plus, for the other layers, the edges whose source row is in the batch (the
pulled out-of-batch rows are constants, no gradient is aggregated for them).

Prints ONE JSON line (rank 0).  Data is synthetic: a seeded power-law graph with the
public ogbn-products statistics planted in 150 communities (no dataset download).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "incagg-gnn_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E nominal
MODEL_NAMES = {"gcn2": "GCN2", "gcn": "GCN", "appnp": "APPNP", "sage": "GraphSAGE"}

# BASELINE.json configs[1..4] as concrete synthetic workloads (sizes: the public statistics of
# the named datasets; layer counts / partitions per batch from conf/model/*.yaml where
# BASELINE.json leaves them open).
PRESETS = {
    "products-gcn2": dict(dataset="products", parts=150, batch_parts=1, model="gcn2", layers=8,
                          hidden=256, tables="hbm", mode="gas"),
    "arxiv-gcn": dict(dataset="arxiv", parts=80, batch_parts=40, model="gcn", layers=3, hidden=256,
                      tables="hbm", mode="gas"),
    "reddit-appnp": dict(dataset="reddit", parts=200, batch_parts=100, model="appnp", layers=5,
                         hidden=256, tables="hbm", mode="gas"),
    "products-sage-host": dict(dataset="products", parts=150, batch_parts=1, model="sage", layers=3,
                               hidden=256, tables="host", mode="gas"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="products-gcn2", choices=sorted(PRESETS),
                    help="BASELINE.json configuration (the headline one by default); the flags "
                         "below override single values of it")
    ap.add_argument("--dataset", default=None)
    ap.add_argument("--parts", type=int, default=None)
    ap.add_argument("--batch-parts", type=int, default=None, help="partitions per batch")
    ap.add_argument("--model", default=None, choices=["gcn2", "gcn", "appnp", "sage"])
    ap.add_argument("--layers", type=int, default=None)
    ap.add_argument("--hidden", type=int, default=None)
    ap.add_argument("--tables", default=None, choices=["hbm", "host"],
                    help="History tables in HBM, or in pinned host memory behind the staging pools")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink nodes and edges together")
    ap.add_argument("--cross", type=float, default=0.15, help="fraction of inter-community edges")
    ap.add_argument("--mode", default=None, choices=["gas", "vr"],
                    help="push/pull training step (headline) or incremental-aggregation step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sweep", choices=["auto", "on", "off"], default="auto",
                    help="also time one full refresh sweep (mini_inference_vr) over all "
                         "partitions; auto = only on a single GPU")
    ap.add_argument("--force-dist", action="store_true",
                    help="take the partition-parallel code path even with one rank")
    ap.add_argument("--batch-norm", action="store_true",
                    help="GCN / GraphSAGE: batch norm after every layer but the last, as "
                         "conf/model/gcn.yaml has it for ogbn-arxiv (the presets leave it out)")
    ap.add_argument("--halo-ahead", choices=["on", "off"], default="on",
                    help="partition-parallel push/pull steps: exchange the out-of-batch rows of the "
                         "next step's batch while the current step computes (off: at the top of "
                         "the step that reads them)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--fused-incagg", action="store_true",
                    help="vr mode: A (x - M_in) + M_ag as one launch gathering both operands "
                         "(default: x - M_in as a pass of its own, then the plain aggregation)")
    ap.add_argument("--epoch-pass", choices=["auto", "on", "off"], default="auto",
                    help="after the timed steps, time one training pass over ALL of this rank's "
                         "partition batches (auto: single-GPU runs)")
    ap.add_argument("--launch-selftest", action="store_true",
                    help="exercise only the rank launcher: every rank joins a gloo group on the "
                         "CPU, the ranks all-reduce their numbers and rank 0 prints a JSON line "
                         "(no GPU needed)")
    ap.add_argument("--graphs", choices=["auto", "on", "off"], default="auto",
                    help="replay every step from a HIP graph recorded once per batch "
                         "(auto: on for a single GPU)")
    args = ap.parse_args()
    for k, v in PRESETS[args.config].items():
        if getattr(args, k) is None:
            setattr(args, k, v)
    return args


class KernelClock:
    """HIP-event timing of every aggregation launch inside the timed region, on the stream
    the kernels are launched on (torch's current stream)."""

    def __init__(self):
        self.records = []   # (start, end, algorithmic_bytes, gather_bytes, edges)
        self.enabled = False
        self.counting = False   # set-up pass: add up the edges every launch processes (reads
        self.edges = 0          # the row pointers back, i.e. waits for the device)
        # distinct source rows an edge array references (col data_ptr -> count), filled in the
        # counting pass: the call's n_cols is not that number when out-of-batch rows are read in
        # place from a whole History table
        self.sources = {}

    def wrap(self, ops):
        inner = ops.spmm_csr
        clock = self

        def timed(rowptr32, col32, val, x, n_cols, *a, **k):
            if clock.counting and rowptr32.numel() > 1:
                clock.edges += int(rowptr32[-1] - rowptr32[0])
                if col32.data_ptr() not in clock.sources:     # distinct source rows referenced
                    clock.sources[col32.data_ptr()] = int(torch.unique(col32).numel())
            if not clock.enabled:
                return inner(rowptr32, col32, val, x, n_cols, *a, **k)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            y = inner(rowptr32, col32, val, x, n_cols, *a, **k)
            e.record()
            n_rows, nnz, f = rowptr32.numel() - 1, col32.numel(), x.size(1)
            # compulsory traffic: indices + weights, row pointers, every distinct source row
            # once, every destination row once (source rows touched <= n_cols)
            n_src = clock.sources.get(col32.data_ptr(), min(n_cols, nnz))
            # (+ one read of the gate rows when the relu backward is fused into the epilogue)
            n_out = n_rows * (2 if k.get("gate") is not None else 1)
            algo = nnz * (4 + (4 if val is not None else 0)) + (n_rows + 1) * 4 \
                + n_src * f * 4 + n_out * f * 4
            gather = nnz * (f * 4 + 8) + n_out * f * 4
            clock.records.append((s, e, algo, gather, nnz))
            return y

        ops.spmm_csr = timed
        inner_inc = ops.spmm_csr_incagg

        def timed_inc(rowptr32, col32, val, x, m_in, m_ag, *a, **k):
            if clock.counting and rowptr32.numel() > 1:
                clock.edges += int(rowptr32[-1] - rowptr32[0])
            if not clock.enabled:
                return inner_inc(rowptr32, col32, val, x, m_in, m_ag, *a, **k)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            y = inner_inc(rowptr32, col32, val, x, m_in, m_ag, *a, **k)
            e.record()
            n_rows, nnz, f = rowptr32.numel() - 1, col32.numel(), x.size(1)
            # the plain aggregation's compulsory bytes plus one read each of the M_in and M_ag rows
            algo = nnz * (4 + (4 if val is not None else 0)) + (n_rows + 1) * 4 + 4 * n_rows * f * 4
            gather = nnz * (2 * f * 4 + 8) + 2 * n_rows * f * 4
            clock.records.append((s, e, algo, gather, nnz))
            return y

        ops.spmm_csr_incagg = timed_inc

    @staticmethod
    def empty_interval_ms(pairs=256, spin=0):
        """What a start/end event pair with NOTHING between them reads on this stream, behind
        ``spin`` cycles of spin work so that the host is ahead of the device as in the timed
        pass: the events' own share of every interval (each is a packet of its own that the
        command processor has to retire; 4-5 us per pair on MI355X)."""
        if spin:
            torch.cuda._sleep(spin)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
              for _ in range(pairs)]
        for s, e in ev:
            s.record()
            e.record()
        torch.cuda.synchronize()
        ms = sorted(s.elapsed_time(e) for s, e in ev)
        return ms[len(ms) // 2]

    def summary(self, empty_ms=0.0):
        """``empty_ms``: the reading of an empty event pair, taken off every interval."""
        if not self.records:
            return None
        raw = [s.elapsed_time(e) for s, e, *_ in self.records]
        ms = [max(t - empty_ms, 1e-6) for t in raw]
        algo = sum(r[2] for r in self.records)
        gather = sum(r[3] for r in self.records)
        tot_s = sum(ms) / 1e3
        return {"launches": len(ms), "avg_ms": sum(ms) / len(ms), "algo_GBps": algo / tot_s / 1e9,
                "gather_GBps": gather / tot_s / 1e9, "avg_interval_ms": sum(raw) / len(raw),
                "algo_bytes_per_launch": algo / len(ms)}


def pmc_traffic(args):
    """Fabric bytes per aggregation launch from the committed rocprofv3 PMC passes
    (profiles/pmc_traffic_current.json: separate FETCH_SIZE and WRITE_SIZE runs of this very
    command reduced by profiles/pmc_traffic.py, FETCH_SIZE doubled as gfx950 requires).
    Counters cannot be read from inside the process, so the figure is only reported for the
    workload it was measured on and only while the kernel source is the one it was measured
    with (SHA-256 recorded next to it); otherwise null."""
    import hashlib
    default = (args.dataset, args.parts, args.batch_parts, args.model, args.layers, args.hidden,
               args.tables, args.scale, args.cross, args.mode) == \
        ("products", 150, 1, "gcn2", 8, 256, "hbm", 1.0, 0.15, "gas")
    path = os.path.join(ROOT, "profiles", "pmc_traffic_current.json")
    src = os.path.join(ROOT, "incagg-gnn_amd", "csrc", "aggregate.hip")
    if not default or not os.path.isfile(path) or not os.path.isfile(src):
        return None
    with open(path) as f:
        rec = json.load(f)
    with open(src, "rb") as f:
        if rec.get("kernel_source_sha256") != hashlib.sha256(f.read()).hexdigest():
            return None
    return rec["traffic_bytes_per_launch"]


def _transposed(rowptr, col, val, m):
    b = rowptr.size - 1
    order = np.argsort(col, kind="stable")
    t_col = np.repeat(np.arange(b), np.diff(rowptr))[order].astype(np.int64)
    t_rowptr = np.zeros(m + 1, dtype=np.int64)
    np.cumsum(np.bincount(col, minlength=m), out=t_rowptr[1:])
    return t_rowptr, t_col, val[order]


def cpu_baseline_port(batches, layers, width, seconds):
    """The C / OpenMP restatement of the CSR aggregation (oracle/aggregate_oracle.c), forward
    + transposed, on the same batches, all host cores the box grants."""
    from oracle import cpu_path as cp
    cores = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
    cp.set_threads(cores)
    rng = np.random.default_rng(1)
    edges, t_total, used, prepared = 0, 0.0, 0, []
    for rowptr, col, val, m in batches:
        b = rowptr.size - 1
        x = rng.standard_normal((m, width)).astype(np.float32)
        dy = rng.standard_normal((b, width)).astype(np.float32)
        prepared.append((rowptr, col, val, x) + _transposed(rowptr, col, val, m) + (dy,))
    cp.spmm(*prepared[0][:4])  # warm
    while t_total < seconds:
        rowptr, col, val, x, t_rowptr, t_col, t_val, dy = prepared[used % len(prepared)]
        t0 = time.perf_counter()
        for _ in range(layers):
            cp.spmm(rowptr, col, val, x)
            cp.spmm(t_rowptr, t_col, t_val, dy)
        t_total += time.perf_counter() - t0
        edges += 2 * layers * col.size
        used += 1
    return {"value": edges / t_total, "unit": "edges/s", "cores": cores, "kind": "port",
            "sample": f"forward + transposed fp32 CSR aggregation of {used} partition-batch steps x "
                      f"{layers} layers at width {width} (dense updates, push/pull and optimizer "
                      f"not included), C/OpenMP oracle, {t_total:.1f} s"}


def cpu_baseline(batches, layers, width, seconds, table_rows):
    """The reference's CPU data path, restated with the operations the reference itself names
    (its own SpMM lives in torch-sparse, which is not installed): per layer
    `torch.sparse_csr_tensor(rowptr, col, val) @ x` forward and on the transposed matrix
    backward (MKL, all granted host cores), the History pull `emb.index_select(0, n_id[B:])`
    (history.py:38) and the chunk push `emb[o:o + c] = x[:B]` (history.py:63-65)."""
    cores = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(1)
    emb = torch.zeros(table_rows, width)
    prepared = []
    for rowptr, col, val, m, n_id, offset in batches:
        b = rowptr.size - 1
        t_rowptr, t_col, t_val = _transposed(rowptr, col, val, m)
        a = torch.sparse_csr_tensor(torch.from_numpy(rowptr), torch.from_numpy(col),
                                    torch.from_numpy(val), size=(b, m))
        at = torch.sparse_csr_tensor(torch.from_numpy(t_rowptr), torch.from_numpy(t_col),
                                     torch.from_numpy(t_val), size=(m, b))
        prepared.append((a, at, torch.randn(m, width, generator=g), torch.randn(b, width, generator=g),
                         n_id, int(offset), b, col.size))
    _ = prepared[0][0] @ prepared[0][2]     # warm
    edges, t_total, used = 0, 0.0, 0
    while t_total < seconds:
        a, at, x, dy, n_id, o, b, nnz = prepared[used % len(prepared)]
        t0 = time.perf_counter()
        for _ in range(layers):
            y = a @ x                                   # aggregation
            emb[o:o + b] = y                            # push of the fresh rows
            _halo = emb.index_select(0, n_id[b:])       # pull of the out-of-batch rows
            _dx = at @ dy                               # backward aggregation
        t_total += time.perf_counter() - t0
        edges += 2 * layers * nnz
        used += 1
    return {"value": edges / t_total, "unit": "edges/s", "cores": cores, "kind": "port",
            "sample": f"{used} partition-batch steps x {layers} layers at width {width}: "
                      "torch.sparse_csr_tensor @ x forward and transposed, History index_select "
                      "pull and slice-assign push (history.py:38,63-65), dense updates and "
                      f"optimizer not included; torch CPU ops on {cores} cores, {t_total:.1f} s"}


def launch_ranks(n, argv, check_devices=True):
    """`--gpus N` without a launcher around this script: start N fresh copies of it, one per GPU,
    with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set as torch.distributed.run
    would, wait for all of them and return the worst exit code.  The children inherit stdout, so
    rank 0's JSON line is this command's JSON line.  The parent never touches the GPU (counting
    devices does not initialise HIP) and a rank that fails takes the others down with it."""
    import socket
    import subprocess
    if check_devices:
        have = torch.cuda.device_count()
        if have < n:
            print(f"bench.py: --gpus {n} needs {n} visible GPUs, this machine shows {have}",
                  file=sys.stderr, flush=True)
            return 2
    port = os.environ.get("MASTER_PORT")
    if port is None:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = str(s.getsockname()[1])
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv),
                                      env=env))
    worst = 0
    pending = set(range(n))
    while pending:
        for r in sorted(pending):
            try:
                rc = procs[r].wait(timeout=0.2)
            except subprocess.TimeoutExpired:
                continue
            pending.discard(r)
            if rc != 0:
                worst = worst or rc
                print(f"bench.py: rank {r} exited with code {rc}; stopping the other ranks",
                      file=sys.stderr, flush=True)
                for q in pending:           # (exact PIDs of our own children)
                    procs[q].terminate()
    return worst


def launch_selftest(args):
    """The launcher's contract on a machine without GPUs: every rank sees its environment, the
    ranks meet over gloo, rank 0 prints one JSON line."""
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1:
        dist.init_process_group("gloo")
    t = torch.tensor([float(rank + 1)])
    if world > 1:
        dist.all_reduce(t)
    if os.environ.get("BENCH_SELFTEST_FAIL_RANK") == str(rank):
        sys.exit(7)
    if rank == 0:
        print(json.dumps({"selftest": True, "n_gpus": world, "rank_sum": float(t),
                          "local_rank": int(os.environ.get("LOCAL_RANK", "0")),
                          "gpus_flag": args.gpus}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:], check_devices=not args.launch_selftest))
    if args.launch_selftest:
        return launch_selftest(args)
    # Only the JSON line may reach stdout: communication libraries print banners there.
    # Everything else this process (and its native libraries) writes goes to stderr.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")
    assert torch.cuda.is_available(), "bench.py measures the HIP path: it needs a GPU"
    if torch.cuda.device_count() <= local_rank:
        sys.exit(f"bench.py: rank {rank} wants GPU {local_rank}, this machine shows "
                 f"{torch.cuda.device_count()}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    sharded = world > 1 or args.force_dist
    if sharded:
        if "MASTER_ADDR" not in os.environ:      # --force-dist without a launcher
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        __import__('torch_geometric_autoscale').distributed.enable_collective_records()
        dist.init_process_group("nccl", device_id=dev)

    import torch_geometric_autoscale as tga
    from torch_geometric_autoscale import ops, _lib
    from torch_geometric_autoscale.models import APPNP, GCN, GCN2, GraphSAGE
    from torch_geometric_autoscale.loader import partition_batch
    from torch_geometric_autoscale import distributed as D

    clock = KernelClock()
    clock.wrap(ops)

    t_setup = time.perf_counter()
    data, f_in, n_cls = tga.get_data(None, args.dataset, num_parts=args.parts, scale=args.scale,
                                     cross_fraction=args.cross, seed=0, features=False, device=dev,
                                     keep_on_device=True)
    n = data.num_nodes
    # self loops + symmetric normalisation on the GPU (main.py:147-151), then to the host,
    # where batches are built
    data.adj_t = tga.gcn_norm(data.adj_t.set_diag()).to("cpu")
    torch.cuda.empty_cache()
    ptr = data.ptr
    g = torch.Generator(device=dev).manual_seed(1)
    x_all = torch.randn(n, f_in, device=dev, generator=g)   # features live in HBM
    y_all = torch.randint(0, n_cls, (n,), device=dev, generator=g)

    # Rank r owns a contiguous block of partitions, i.e. a contiguous block of History rows;
    # every rank runs the same number of steps, each on one of its own partitions.
    ctx = D.PartitionContext(ptr, exercise_collectives=args.force_dist) if sharded else None
    parts_lo, parts_hi = rank * args.parts // world, (rank + 1) * args.parts // world
    my_parts = list(range(parts_lo, parts_hi))
    groups = [my_parts[i:i + args.batch_parts] for i in range(0, len(my_parts), args.batch_parts)]
    if sharded:
        # every rank takes part in every collective (plan building, halo exchanges, gradient
        # all-reduces) the same number of times: all ranks use as many batches as the rank with
        # the fewest (the shares differ by at most one partition)
        shares = [(r + 1) * args.parts // world - r * args.parts // world for r in range(world)]
        groups = groups[:min(-(-sh // args.batch_parts) for sh in shares)]
    epoch_pass = args.epoch_pass == "on" or (args.epoch_pass == "auto" and not sharded)
    need = len(groups) if epoch_pass else min(len(groups), args.steps + args.warmup)
    host_tables = args.tables == "host"
    loader_data = tga.Data(adj_t=data.adj_t, num_nodes=n)
    batches = []
    for grp in groups[:need]:
        if sharded:     # collective: builds the halo exchange plan of the batch
            sub = D.sharded_partition_batch(loader_data, ptr, grp, ctx,
                                            in_batch_only=(args.mode == "vr"))
        else:
            sub = partition_batch(loader_data, ptr, grp, in_batch_only=(args.mode == "vr"))
        adj = sub.data.adj_t.to(dev)
        nid_dev = sub.n_id.to(dev)
        batches.append({
            "adj": adj, "host": sub.data.adj_t, "b": sub.batch_size, "n_id": sub.n_id,
            "nid_dev": nid_dev, "offset": sub.offset, "count": sub.count,
            "x": x_all[nid_dev], "y": y_all[nid_dev[:sub.batch_size]], "edges": 0,
        })

    table_rows = ctx.local_rows if sharded else n      # row-sharded tables, never replicated
    halo_pipe = None
    table_dev = None if host_tables else dev
    pool = {}
    if host_tables:     # staging slots as main.py:177 sizes them: 2 x the largest batch (rows + halo)
        from torch_geometric_autoscale import ops as _o
        g_rowptr, g_col, _ = data.adj_t.csr()
        largest = 0
        for grp in groups:
            idx = torch.cat([torch.arange(int(ptr[p_]), int(ptr[p_ + 1])) for p_ in grp])
            largest = max(largest, int(_o.relabel_one_hop(g_rowptr, g_col, None, idx, True)[3].numel()))
        pool = dict(pool_size=2, buffer_size=2 * largest)
    if args.model == "gcn2":
        model = GCN2(table_rows, f_in, args.hidden, n_cls, args.layers, alpha=0.1, theta=0.5,
                     shared_weights=False, dropout=0.0, device=table_dev, **pool)
    elif args.model == "gcn":
        model = GCN(table_rows, f_in, args.hidden, n_cls, args.layers, dropout=0.0,
                    drop_input=False, batch_norm=args.batch_norm, residual=False, device=table_dev,
                    **pool)
    elif args.model == "sage":
        model = GraphSAGE(table_rows, f_in, args.hidden, n_cls, args.layers, dropout=0.0,
                          drop_input=False, batch_norm=args.batch_norm, residual=False,
                          device=table_dev, **pool)
    else:
        model = APPNP(table_rows, f_in, args.hidden, n_cls, args.layers, alpha=0.1, dropout=0.0,
                      device=table_dev, **pool)
    model = model.to(dev)
    if sharded:
        model.set_partition_context(ctx)
        if not host_tables:
            # out-of-batch rows are read in place as on a single GPU: own rows where they are,
            # rows of peers received into the tail of the table shard
            # (and, for the push/pull step, the rows of the next step's batch travel while the
            # current step computes: HaloPipeline gives consecutive batches separate tail regions)
            if args.mode != "vr":
                halo_pipe = D.HaloPipeline(model, [bt["adj"] for bt in batches],
                                           ahead=args.halo_ahead == "on")
            else:
                D.reserve_halo_tails(model, [bt["host"] for bt in batches])
    model.fused_incagg = args.fused_incagg
    g_cpu = torch.Generator().manual_seed(2)
    with torch.no_grad():
        for p_ in model.parameters():
            if p_.dim() > 1:
                p_.copy_((torch.rand(p_.shape, generator=g_cpu) * 2 - 1) *
                         (6.0 / (p_.size(0) + p_.size(1))) ** 0.5)
        for tables in (model.histories, model.histories_ag):   # non-trivial table contents
            for h in tables:
                if h.emb.is_cuda:
                    h.emb.normal_(0, 0.1, generator=g)
                else:
                    h.emb.normal_(0, 0.1)
    # (staging pools run on side streams with a host worker thread: nothing to record)
    use_graphs = (args.graphs == "on" or (args.graphs == "auto" and not sharded)) and not host_tables
    # Partition-parallel steps are not recorded whole (this torch build cannot record the
    # asynchronous collectives): the collectives -- halo exchanges at the top, the gradient
    # all-reduce at the end -- are issued eagerly around ONE recorded graph of everything between
    # them (forward with the foreign rows already in the table tails, loss, backward).
    split_graphs = sharded and args.graphs != "off" and not host_tables
    if split_graphs:
        use_graphs = False
    # Partition-parallel split steps: the trainable parameters live in ONE flat tensor (every
    # parameter of the model is a view of it) and the optimizer is given that one tensor with
    # the concatenated gradient -- the buffer the all-reduce works on.  Adam is element-wise:
    # the update is the same as over the parameter list.  Single-GPU steps keep the parameter
    # list (measured: no faster with the flat form there).
    # capturable: the step counter lives on the device.
    train_params = [p_ for p_ in model.parameters() if p_.requires_grad]
    # (every parameter starts 256-byte aligned inside the flat tensor: GEMM operands)
    sizes = [-(-p_.numel() // 64) * 64 for p_ in train_params]
    flat_w = torch.nn.Parameter(torch.zeros(sum(sizes), device=dev))
    off = 0
    for p_, n_ in zip(train_params, sizes):
        flat_w.data[off:off + p_.numel()] = p_.detach().reshape(-1)
        p_.data = flat_w.data[off:off + p_.numel()].view_as(p_)
        off += n_
    no_grad_yet = [torch.zeros(n_, device=dev) for n_ in sizes]   # (never written)

    def padded(g, n_):      # a gradient, zero-extended to its parameter's slot
        g = g.reshape(-1)
        return g if g.numel() == n_ else torch.nn.functional.pad(g, (0, n_ - g.numel()))
    # torch.optim.Adam's update and torch.nn.CrossEntropyLoss's value (main.py:60,160-165), each
    # in fewer, fuller launches (optim.Adam: step counts on the device, so recorded steps count on)
    from torch_geometric_autoscale.optim import Adam
    from torch_geometric_autoscale.nn import CrossEntropyLoss
    opt_flat = Adam([flat_w], lr=1e-3)
    opt = Adam(train_params, lr=1e-3)
    crit = CrossEntropyLoss()
    one = torch.ones((), device=dev)
    model.train()
    setup_s = time.perf_counter() - t_setup

    def clear_grads():
        for p_ in train_params:
            p_.grad = None

    def gathered_grads():
        return torch.cat([z if p_.grad is None else padded(p_.grad, z.numel())
                          for p_, z in zip(train_params, no_grad_yet)])

    flat_grads = None       # partition-parallel split steps: {batch: its recorded gradient}

    def step(i):
        k = i % len(batches)
        if flat_grads is not None:
            return split_step(k, replay=True)
        if graphs is not None and k in graphs:
            graphs.replay(k)
            return batches[k]["edges"]
        return eager_step(k)

    def split_body(k):      # the collective-free part of a partition-parallel step
        bt = batches[k]
        if args.mode == "vr":
            out = model.VR_call(bt["x"], bt["adj"], bt["b"], bt["nid_dev"], bt["offset"],
                                bt["count"])["out"]
        else:
            out = model(bt["x"], bt["adj"], bt["b"], bt["nid_dev"], bt["offset"], bt["count"],
                        halo_ready=True)["out"]
        clear_grads()
        crit(out, bt["y"]).backward(gradient=one)
        return gathered_grads()

    def split_step(k, replay):
        bt = batches[k]
        if halo_pipe is not None:
            halo_pipe.ready(k)      # this batch's foreign rows are in; the next batch's set off
        if replay:
            graphs.replay(k)
            grad = flat_grads[k]    # (recorded: the same memory in every replay)
        else:
            grad = split_body(k)
        D.allreduce_flat_gradients(grad, ctx)
        flat_w.grad = grad
        opt_flat.step()
        return bt["edges"]

    def eager_step(k):
        bt = batches[k]
        if args.mode == "vr":
            out = model.VR_call(bt["x"], bt["adj"], bt["b"], bt["nid_dev"], bt["offset"],
                                bt["count"])["out"]
        else:
            out = model(bt["x"], bt["adj"], bt["b"], bt["nid_dev"], bt["offset"], bt["count"])["out"]
        opt.zero_grad(set_to_none=True)
        loss = crit(out, bt["y"])
        loss.backward(gradient=one)     # (a resident 1.0: no fill launch per step)
        if sharded:
            D.allreduce_gradients(model, ctx)
        opt.step()
        return bt["edges"]

    # The refresh sweep (timed after the steps): its evaluation batches, resident in HBM, and on
    # a single GPU its recording -- the first call issues the sweep eagerly (the tables then
    # hold real layer inputs and aggregates) and records it, BEFORE any step is recorded.
    evals = sweep_graph = None
    if args.sweep == "on" or (args.sweep == "auto" and not sharded):
        feat = tga.Data(adj_t=data.adj_t, num_nodes=n)
        if sharded:
            evals = D.sharded_eval_batches(feat, ptr, ctx)
        else:
            evals = [partition_batch(feat, ptr, grp) for grp in groups]
        for sub in evals:                       # features of each batch resident in HBM
            if sub.batch_size:
                sub.data.x = x_all[sub.n_id.to(dev)]
                sub.data.adj_t.to(dev)
        if not sharded and not host_tables and args.graphs != "off":
            from torch_geometric_autoscale.graphs import SweepGraph
            sweep_graph = SweepGraph(model, evals, with_aggregates=True)
            sweep_graph()
            model.train()
            torch.cuda.synchronize()

    # Edges per step, counted once per batch in an instrumented eager step: every aggregation
    # launch adds the edges its rows address (forward launches the batch adjacency; backward
    # launches the transposed adjacency, restricted to in-batch source rows where the
    # out-of-batch rows are constants).  Also creates every lazily built per-batch structure.
    graphs = None
    for k, bt in enumerate(batches):
        clock.counting, clock.edges = True, 0
        eager_step(k)
        clock.counting = False
        bt["edges"] = clock.edges
    torch.cuda.synchronize()
    if split_graphs and args.mode != "vr" and not model.supports_halo_ready():
        split_graphs = False    # (e.g. a model that transforms rows before aggregating them)
    if split_graphs:
        from torch_geometric_autoscale.graphs import StepGraphs
        flat_grads = {}
        graphs = StepGraphs(dev)
        for k in range(len(batches)):
            graphs.warm(lambda: split_step(k, replay=False), times=1)
        if halo_pipe is not None:
            halo_pipe.drain()
        D.quiesce_collectives()
        recorded = True
        try:
            for k in range(len(batches)):
                graphs.capture(k, lambda: flat_grads.__setitem__(k, split_body(k)),
                               capture_error_mode="thread_local")
            torch.cuda.synchronize()
        except Exception as exc:     # fall back to eagerly issued steps, on every rank alike
            print(f"rank {rank}: recording the step failed ({exc!r}); steps are issued eagerly",
                  file=sys.stderr, flush=True)
            recorded = False
        agreed = torch.tensor([int(recorded)], device=dev)
        dist.all_reduce(agreed, op=dist.ReduceOp.MIN)
        if not int(agreed):
            graphs = flat_grads = None
            clear_grads()
    if use_graphs:
        # every batch's step is recorded once (untimed) and replayed from then on
        from torch_geometric_autoscale.graphs import StepGraphs
        graphs = StepGraphs(dev)
        for k in range(len(batches)):
            graphs.warm(lambda: eager_step(k), times=1)
        torch.cuda.synchronize()
        for k in range(len(batches)):
            graphs.capture(k, lambda: eager_step(k))
            if os.environ.get("BENCH_TRACE") and k % 10 == 9:
                torch.cuda.synchronize()
                print(f"captured {k + 1} step graphs, "
                      f"{torch.cuda.memory_allocated() / 2**30:.1f} GiB allocated", file=sys.stderr,
                      flush=True)
        torch.cuda.synchronize()

    # Untimed priming before the W warmup steps: every prepared batch once (batches differ in
    # shape: GEMM code objects and allocator blocks for all shapes exist afterwards), and at
    # least ~1.5 s of sustained work.  On a freshly started machine the first process sees a
    # one-off stall of ~90 ms roughly 0.7 s into sustained GPU load (measured on three fresh
    # boxes; absent from every later process), which would otherwise land inside the K steps.
    t_prime = time.perf_counter()
    prime_seconds = float(os.environ.get("BENCH_PRIME_SECONDS", "1.5"))
    while True:
        for i in range(len(batches)):
            step(i)
        torch.cuda.synchronize()
        done = torch.tensor([time.perf_counter() - t_prime >= prime_seconds], device=dev,
                            dtype=torch.int32)
        if sharded:                      # every rank runs the same number of passes
            dist.broadcast(done, 0)
        if int(done):
            break
    for i in range(args.warmup):
        step(i)
    if sharded:
        dist.barrier()
    torch.cuda.synchronize()
    clock.enabled = True
    t0 = time.perf_counter()
    edges = 0
    host_marks = [t0]
    for i in range(args.steps):
        edges += step(args.warmup + i)
        host_marks.append(time.perf_counter())
    torch.cuda.synchronize()
    if sharded:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    clock.enabled = False
    kernel_timing = "HIP events around every aggregation launch of the timed steps"
    empty_ms = 0.0
    if graphs is not None:
        # The timed steps were graph replays, into which no timing events can be placed: the
        # per-launch durations come from an eager pass over the same K steps, right away.
        # The events must bracket device time only.  An eager step takes the host longer to
        # issue than the GPU to run, so every step is enqueued behind spin work that lasts
        # 1.5x the host's issue time of a step: the GPU never catches up with the host, and every
        # start event is reached the moment the preceding kernel ends (otherwise the intervals
        # would include the host's time between recording the event and launching the kernel).
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        torch.cuda._sleep(20_000_000)
        e.record()
        torch.cuda.synchronize()
        spin_per_ms = 20_000_000 / s.elapsed_time(e)
        unrecorded = eager_step if flat_grads is None else (lambda k: split_step(k, replay=False))
        clock.enabled = True        # (issue time measured with the events in place)
        t_issue = time.perf_counter()
        unrecorded(args.warmup % len(batches))
        t_issue = (time.perf_counter() - t_issue) * 1e3
        torch.cuda.synchronize()
        clock.records.clear()
        spin = int(spin_per_ms * 1.5 * t_issue)
        for i in range(args.steps):
            torch.cuda._sleep(spin)
            unrecorded((args.warmup + i) % len(batches))
        torch.cuda.synchronize()
        clock.enabled = False
        empty_ms = clock.empty_interval_ms(spin=int(spin_per_ms * 8))
        kernel_timing = ("HIP events around every aggregation launch in an eager pass over the "
                         "same K steps run right after the timed region, each step enqueued "
                         f"behind {1.5 * t_issue:.1f} ms of spin work (1.5x the host's issue time of "
                         "an eager step) so that the intervals hold device time only (the timed "
                         "steps are HIP graph replays, which cannot carry timing events); "
                         f"avg_launch_ms = the mean interval (avg_interval_ms) less the "
                         f"{empty_ms * 1e3:.1f} us that an event pair with nothing between it "
                         "reads on the same stream (median of 256 pairs)")
    # ---- epoch time, part (i): one training pass over every partition batch of this rank
    # (outside the timed region; the same step() the K timed steps ran)
    epoch_s = None
    if epoch_pass and len(batches) == len(groups):
        torch.cuda.synchronize()
        if sharded:
            dist.barrier()
        t_ep = time.perf_counter()
        for i in range(len(batches)):
            step(i)
        torch.cuda.synchronize()
        if sharded:
            dist.barrier()
        epoch_s = time.perf_counter() - t_ep
    if os.environ.get("BENCH_TRACE"):   # host-side enqueue time of every step
        print("enqueue ms/step:", " ".join(f"{(b - a) * 1e3:.2f}" for a, b in
                                             zip(host_marks, host_marks[1:])), file=sys.stderr)
    _lib.check_device_errors()

    tot = torch.tensor([elapsed, float(edges)], dtype=torch.float64, device=dev)
    if world > 1:
        t_max = tot[:1].clone()
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
        e_sum = tot[1:].clone()
        dist.all_reduce(e_sum, op=dist.ReduceOp.SUM)
        elapsed, edges = float(t_max), float(e_sum)

    # ---- epoch time, part (ii): one layer-wise refresh sweep over every partition
    # (outside the timed region above; it is reported next to the headline value)
    sweep_s = sweep_edges = None
    sweep_launch = None
    if halo_pipe is not None:
        halo_pipe.drain()
    if evals is not None:
        torch.cuda.synchronize()
        if sharded:
            dist.barrier()
        t0 = time.perf_counter()
        if sweep_graph is not None:
            sweep_graph()
        else:
            model.eval()
            model.mini_inference_vr(evals)
        torch.cuda.synchronize()
        if sharded:
            dist.barrier()
        sweep_s = time.perf_counter() - t0
        sweep_edges = args.layers * sum(sub.data.adj_t.nnz() for sub in evals)
        sweep_launch = ("one HIP graph replay" if sweep_graph is not None and sweep_graph.recorded
                        else "eager")
        _lib.check_device_errors()

    ks = clock.summary(empty_ms)
    if rank == 0:
        bt = batches[0]
        line = {
            "metric": f"edges-aggregated/sec, ogbn-{args.dataset} {MODEL_NAMES[args.model]} training "
                      "step (forward + backward)",
            "value": edges / elapsed, "unit": "edges/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{MODEL_NAMES[args.model]} {args.layers}-layer hidden {args.hidden} on "
                            f"an ogbn-{args.dataset}-sized synthetic graph (N={n}, "
                            f"nnz={data.adj_t.nnz()}), {args.parts} partitions, {args.batch_parts} "
                            f"per step, {args.mode} training step, "
                            + ("batch norm after every hidden layer, " if args.batch_norm else "")
                            + "History tables "
                            + ("in pinned host memory behind the staging pools" if host_tables
                               else "resident in HBM"),
                "preset": args.config,
                "batch_rows": bt["b"], "batch_rows_with_halo": int(bt["n_id"].numel()),
                "batch_edges": bt["adj"].nnz(),
                "parallelism": f"partition-parallel x{world}" + (
                    ", row-sharded History tables, one all-to-all of halo rows per layer"
                    + (" issued one step ahead (the rows a step reads left their owners at the "
                       "top of the step before)" if halo_pipe is not None and halo_pipe.ahead
                       else "") + ", flat gradient all-reduce" if sharded else ""),
                "setup_seconds": round(setup_s, 1),
                "step_launch": ("collectives issued eagerly around one HIP graph replay of "
                                "forward + loss + backward per step" if flat_grads is not None
                                else "one HIP graph replay per step (recorded once per batch, "
                                "untimed)" if graphs is not None else "eager"),
            },
            "epoch": {
                "train_epoch_seconds": epoch_s if epoch_s is not None
                else elapsed / args.steps * len(groups),
                "train_epoch_measured": epoch_s is not None,
                "train_epoch_batches": len(groups),
                "refresh_sweep_seconds": sweep_s,
                "refresh_sweep_edges_per_s": None if not sweep_s else sweep_edges / sweep_s,
                "refresh_sweep_launch": sweep_launch,
                "note": "training epoch = one measured pass over all of this rank's partition "
                        "batches (when train_epoch_measured; else batches x step time); the sweep is "
                        "one mini_inference_vr over this rank's partitions (forward only, both "
                        "History kinds refreshed)",
            },
        }
        if ks:
            line["roofline"] = {
                "bound": "hbm", "achieved": ks["algo_GBps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": ks["algo_GBps"] / HBM_PEAK_GBPS, "traffic": pmc_traffic(args),
                "kernel": ("spmm_narrow_kernel" if 32 < model.histories[-1].embedding_dim <= 48 else "spmm_plan_kernel")
                + ", forward and transposed launches (one launch per call)",
                "avg_launch_ms": ks["avg_ms"], "avg_interval_ms": ks["avg_interval_ms"],
                "empty_event_pair_ms": empty_ms, "launches": ks["launches"],
                "algorithmic_bytes_per_launch": ks["algo_bytes_per_launch"],
                "gather_bound_GBps": ks["gather_GBps"],
                "timing": kernel_timing,
            }
        if not args.no_cpu_baseline and world == 1:     # (reported on single-GPU runs only)
            host = []
            for b_ in batches[:4]:
                rp, c, v = b_["host"].csr()
                host.append((rp.numpy(), c.numpy(), v.numpy().astype(np.float32),
                             int(b_["host"].size(1)), b_["n_id"], b_["offset"][0]))
            agg_width = n_cls if args.model == "appnp" else args.hidden   # models/appnp.py:24
            line["cpu_baseline"] = cpu_baseline(host, args.layers, agg_width, args.cpu_seconds, n)
            line["cpu_baseline_c_port"] = cpu_baseline_port([h[:4] for h in host], args.layers,
                                                            agg_width, args.cpu_seconds / 2)
        print(json.dumps(line), file=json_out, flush=True)
    if sharded:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
