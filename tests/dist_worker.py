"""Worker of tests/test_distributed_cpu.py: one rank of a partition-parallel run on the CPU
(gloo).  The HIP entry points are swapped for the CPU oracle (tests/cpu_shim.py), so what
is exercised is the sharding, the halo plans, the all-to-all routing and the gradient
all-reduce.  Launched with torch.distributed.run; writes its results to argv[1].

With DIST_WORKER_DEVICE=cuda (tests/test_gpu_distributed.py) the same script is one rank of a
run on the GPUs: backend "nccl" (RCCL), rank r on GPU LOCAL_RANK, the real HIP entry points,
tables in HBM."""
import contextlib
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT, os.path.join(ROOT, "incagg-gnn_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import model_cases as mc  # noqa: E402
from cpu_shim import cpu_kernels  # noqa: E402


PIPELINE_PARTS = 7      # partitions of the pipeline case
PIPELINE_MODEL = "gcn2"


def pipeline_steps(world):
    """Steps of the pipeline case: every rank's cycle wraps around at least twice."""
    return 2 * (PIPELINE_PARTS // world + 1) + 1


def pipeline_tables(model, n):
    """Whole tables of the pipeline case to start from: random rows, far from what the model
    would push, so that every step's pushes change what later steps read."""
    g = torch.Generator().manual_seed(11)
    return [torch.randn(n, h.embedding_dim, generator=g) for h in model.histories]


def pipeline_main(out_dir, dev, on_gpu):
    """Push/pull forward steps over this rank's partitions, one per step, with the exchange of
    out-of-batch rows running one step ahead (distributed.HaloPipeline); the outputs of every
    step and the tables at the end go to the result file."""
    from torch_geometric_autoscale import distributed as D
    data, ptr = mc.make_graph(parts=PIPELINE_PARTS)
    ctx = D.PartitionContext(ptr, exercise_collectives=on_gpu and dist.get_world_size() == 1)
    result = {"row_lo": ctx.row_lo, "row_hi": ctx.row_hi, "parts": ctx.local_parts}
    with (contextlib.nullcontext() if on_gpu else cpu_kernels()):
        model = mc.make_model(PIPELINE_MODEL, data, data.x.size(1), 5,
                              emb_device=dev if on_gpu else "cpu", num_nodes=ctx.local_rows)
        if on_gpu:
            model = model.to(dev)
        model.set_partition_context(ctx)
        model.eval()
        # (a batch's exchange plan is built, and later used, together with the peers' batches of
        # the same round: every rank brings as many batches as the rank with the smallest share)
        mine = ctx.local_parts[:-ctx.max_over_ranks(-len(ctx.local_parts))]
        result["parts"] = mine
        subs = [D.sharded_partition_batch(data, ptr, [p], ctx) for p in mine]
        pipe = D.HaloPipeline(model, [s.data.adj_t for s in subs])
        result["ahead"] = pipe.ahead
        result["regions"] = [s.data.adj_t._halo_plan.region for s in subs]
        for t, full in zip(model.histories, pipeline_tables(model, data.num_nodes)):
            t.emb.copy_(full[ctx.row_lo:ctx.row_hi])
        for i in range(pipeline_steps(ctx.world)):
            sub = subs[i % len(subs)]
            b = sub.batch_size
            pipe.ready(i % len(subs))
            x = sub.data.x.to(dev) if on_gpu else sub.data.x
            adj = sub.data.adj_t.to(dev) if on_gpu else sub.data.adj_t
            with torch.no_grad():
                ret = model(x, adj, b, sub.n_id, sub.offset, sub.count, halo_ready=True)
            result[f"step{i}"] = ret["out"].detach().clone()
        pipe.drain()
        # two more steps on batch 0, back to back: the first after the drain (nothing in flight),
        # the second with the look-ahead for batch 1 in flight, which is then dropped -- both
        # exchange at the top of the step itself
        for j in range(2):
            sub = subs[0]
            pipe.ready(0)
            x = sub.data.x.to(dev) if on_gpu else sub.data.x
            adj = sub.data.adj_t.to(dev) if on_gpu else sub.data.adj_t
            with torch.no_grad():
                ret = model(x, adj, sub.batch_size, sub.n_id, sub.offset, sub.count, halo_ready=True)
            result[f"extra{j}"] = ret["out"].detach().clone()
        pipe.drain()
        for i in range(1, model.num_layers):
            result[f"hist{i}"] = model.histories[i].emb.clone()
    if on_gpu:
        from torch_geometric_autoscale import _lib
        torch.cuda.synchronize()
        _lib.check_device_errors()
        result = {k: v.cpu() if torch.is_tensor(v) else v for k, v in result.items()}
    torch.save(result, os.path.join(out_dir, f"rank{ctx.rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def main():
    out_dir = sys.argv[1]
    names = sys.argv[2].split(",")
    in_place = len(sys.argv) > 3 and sys.argv[3] == "in_place"
    on_gpu = os.environ.get("DIST_WORKER_DEVICE", "cpu") == "cuda"
    if on_gpu:
        dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
        torch.cuda.set_device(dev)
        from torch_geometric_autoscale import distributed as D
        D.enable_collective_records()
        dist.init_process_group("nccl", device_id=dev)
    else:
        dev = torch.device("cpu")
        dist.init_process_group("gloo")
    from torch_geometric_autoscale import distributed as D
    if len(sys.argv) > 3 and sys.argv[3] == "pipeline":
        return pipeline_main(out_dir, dev, on_gpu)
    data, ptr = mc.make_graph()
    ctx = D.PartitionContext(ptr, exercise_collectives=on_gpu and dist.get_world_size() == 1)
    result = {"row_lo": ctx.row_lo, "row_hi": ctx.row_hi}

    def to_dev(t):
        return t.to(dev) if on_gpu else t

    with (contextlib.nullcontext() if on_gpu else cpu_kernels()):
        for name in names:
            model = mc.make_model(name, data, data.x.size(1), 5, emb_device=dev if on_gpu else "cpu",
                                  num_nodes=ctx.local_rows)
            if on_gpu:
                model = model.to(dev)
            model.set_partition_context(ctx)
            model.eval()
            batches = D.sharded_eval_batches(data, ptr, ctx, batch_size=1)
            calls = {"in_place": 0, "buffer": 0}
            if in_place:
                # room for the foreign rows behind every table's own rows: out-of-batch rows
                # are then read in place (own ones where they are, foreign ones in the tail)
                train_sub = D.sharded_partition_batch(data, ptr, [ctx.part_lo], ctx)
                need = D.reserve_halo_tails(model, batches + [train_sub])
                assert all(h.tail_rows == need for h in model.histories)
                originals = {}
                for attr, key in (("start_exchange_in_place", "in_place"), ("start_exchange", "buffer")):
                    originals[attr] = getattr(D.HaloPlan, attr)

                    def counted(self, *a, _f=originals[attr], _k=key, **k):
                        calls[_k] += 1
                        return _f(self, *a, **k)
                    setattr(D.HaloPlan, attr, counted)
            out = model.mini_inference_vr(batches)
            result[f"{name}/out"] = out.clone()
            for i in range(model.num_layers):
                result[f"{name}/hist{i}"] = model.histories[i].emb.clone()
                result[f"{name}/hist_ag{i}"] = model.histories_ag[i].emb.clone()
            # push/pull forward + backward on this rank's first partition, then the all-reduce
            sub = D.sharded_partition_batch(data, ptr, [ctx.part_lo], ctx)
            b = sub.batch_size
            x_dev, adj_dev, y_dev = to_dev(sub.data.x), to_dev(sub.data.adj_t), to_dev(sub.data.y[:b])
            ret = model(x_dev, adj_dev, b, sub.n_id, sub.offset, sub.count)
            result[f"{name}/gas_out"] = ret["out"].detach().clone()
            loss = torch.nn.functional.cross_entropy(ret["out"], y_dev)
            model.zero_grad()
            loss.backward()
            D.allreduce_gradients(model, ctx)
            result[f"{name}/grad"] = torch.cat([p.grad.reshape(-1) for p in model.parameters()
                                                if p.grad is not None])
            if in_place and name == "gcn2":     # (a model that reads every layer's rows in place)
                # the same step split at its collectives: exchanges first (foreign rows land in
                # the table tails), then a forward that contains no collective, gradients into
                # one flat buffer, one all-reduce of that buffer
                flat = D.flatten_gradients(model)
                for pending in model.start_halo_exchanges(adj_dev):
                    pending.wait()
                ret2 = model(x_dev, adj_dev, b, sub.n_id, sub.offset, sub.count, halo_ready=True)
                assert torch.allclose(ret2["out"], ret["out"], rtol=1e-5, atol=1e-6)
                model.zero_grad(set_to_none=False)
                torch.nn.functional.cross_entropy(ret2["out"], y_dev).backward()
                D.allreduce_flat_gradients(flat, ctx)
                # the flat buffer holds the same all-reduced mean gradient (zeros where a
                # parameter got no gradient, which the list of the first run leaves out)
                n1, n2 = float(flat.norm()), float(result[f"{name}/grad"].norm())
                assert abs(n1 - n2) <= 1e-4 * (1.0 + n2), (n1, n2)
                assert all(p.grad.data_ptr() >= flat.data_ptr() for p in model.parameters()
                           if p.requires_grad)
                for p in model.parameters():
                    p.grad = None
            # incremental-aggregation step: no halo traffic at all
            sub = D.sharded_partition_batch(data, ptr, [ctx.part_lo], ctx, in_batch_only=True)
            ret = model.VR_call(to_dev(sub.data.x), to_dev(sub.data.adj_t), sub.batch_size, sub.n_id,
                                sub.offset, sub.count)
            result[f"{name}/vr_out"] = ret["out"].detach().clone()
            if in_place:
                result[f"{name}/calls"] = dict(calls)
                for attr, fn in originals.items():
                    setattr(D.HaloPlan, attr, fn)
    if on_gpu:
        from torch_geometric_autoscale import _lib
        torch.cuda.synchronize()
        _lib.check_device_errors()
        result = {k: v.cpu() if torch.is_tensor(v) else v for k, v in result.items()}
    torch.save(result, os.path.join(out_dir, f"rank{ctx.rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
