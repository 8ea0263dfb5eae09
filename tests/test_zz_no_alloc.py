"""The plan builder that allocates nothing (incagg_spmm_plan_device_static), called through the
C ABI, against incagg_spmm_plan_device on the same row pointers: integer work, bit-exact.  Two
calls in a row share the library's fixed work array; the second input is smaller."""

import numpy as np
import pytest
import torch

from helpers import random_csr

pytestmark = pytest.mark.gpu


def _static_plan(L, rp_dev, hub_threshold, segment_edges, nnz):
    """Outputs are allocated here, before the call; the call itself only enqueues."""
    dev = rp_dev.device
    n_rows = rp_dev.numel() - 1
    cap = int(L.incagg_spmm_plan_items_bound(n_rows, nnz, hub_threshold, segment_edges))
    assert cap >= 0
    order = torch.full((n_rows,), -7, dtype=torch.int32, device=dev)
    items = torch.full((max(cap, 1), 2), -7, dtype=torch.int32, device=dev)
    multi = torch.full((max(min(n_rows, cap), 1),), -7, dtype=torch.int32, device=dev)
    counts = torch.full((6,), -7, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    torch.cuda.synchronize()
    before = torch.cuda.memory_stats(dev)["allocation.all.allocated"]
    free_before = torch.cuda.mem_get_info(dev)[0]
    rc = L.incagg_spmm_plan_device_static(rp_dev.data_ptr(), n_rows, hub_threshold, segment_edges,
                                          order.data_ptr(), items.data_ptr(), cap, multi.data_ptr(),
                                          multi.numel(), counts.data_ptr(), stream)
    assert rc == 0, L.incagg_last_error()
    torch.cuda.synchronize()
    # nothing was allocated: neither through torch nor from the driver
    assert torch.cuda.memory_stats(dev)["allocation.all.allocated"] == before
    assert torch.cuda.mem_get_info(dev)[0] == free_before
    return order, items, multi, counts.cpu().tolist()


def test_static_plan_builder_matches_the_workspace_builder():
    from torch_geometric_autoscale import _lib, ops
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    assert int(L.incagg_spmm_plan_device_static_max_rows()) == 262144
    rng = np.random.default_rng(31)
    big, _, _ = random_csr(rng, 200000, 50000, 6, hubs={0: 129, 5: 128, 17: 9000, 18: 4096, 19: 4097,
                                                        199999: 30000}, empty_rows=(1, 2, 77777),
                           weighted=False)
    small, _, _ = random_csr(rng, 3001, 9000, 14, hubs={3: 5000, 2999: 300}, weighted=False)
    # (the largest input the fixed array takes: 262 144 rows; then smaller ones)
    most = np.arange(262145, dtype=np.int64) * 3
    for rowptr, kw in ((most, (0, 0)), (big, (0, 0)), (small, (24, 128)), (big[:1000], (0, 0))):
        rp = torch.from_numpy(rowptr.astype(np.int32)).to(dev)
        nnz = int(rowptr[-1] - rowptr[0])
        want = ops.SpmmPlan.build(rp, nnz, hub_threshold=kw[0], segment_edges=kw[1])
        order, items, multi, counts = _static_plan(L, rp, kw[0], kw[1], nnz)
        assert counts == [want.n_hub, want.n_items, want.n_multi, want.hub_threshold,
                          want.segment_edges, 0]
        assert torch.equal(order, want.order)
        assert torch.equal(items[:want.n_items], want.items[:want.n_items])
        assert torch.equal(multi[:want.n_multi], want.multi_rows[:want.n_multi])
        # and the plan it built drives the aggregation to the same result
        if rowptr is small:
            col = torch.from_numpy(rng.integers(0, 9000, size=nnz).astype(np.int32)).to(dev)
            x = torch.randn(9000, 64, device=dev)
            plan = ops.SpmmPlan(order, items[:max(counts[1], 1)], multi[:max(counts[2], 1)],
                                counts[0], counts[1], counts[2], counts[3], counts[4])
            got = ops.spmm_csr(rp, col, None, x, 9000, plan=plan)
            ref = ops.spmm_csr(rp, col, None, x, 9000, plan=want)
            # (rows cut into several segments are combined with fp32 atomics: same plan, same
            # sums, rounding order free)
            assert torch.allclose(got, ref, rtol=1e-5, atol=1e-4)
    _lib.check_device_errors()


def test_static_plan_builder_rejects_what_does_not_fit():
    from torch_geometric_autoscale import _lib
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    n = 262145
    rp = torch.zeros(n + 1, dtype=torch.int32, device=dev)
    out = torch.zeros(n, dtype=torch.int32, device=dev)
    counts = torch.zeros(6, dtype=torch.int32, device=dev)
    rc = L.incagg_spmm_plan_device_static(rp.data_ptr(), n, 0, 0, out.data_ptr(), None, 0, None, 0,
                                          counts.data_ptr(), None)
    assert rc == -1 and "exceed" in _lib.last_error()
    # a capacity that is too small is reported in the device-side counts, nothing is overrun
    rowptr = np.arange(0, 5 * 300 + 1, 300, dtype=np.int32)          # 5 rows of 300 edges
    rp = torch.from_numpy(rowptr).to(dev)
    order = torch.zeros(5, dtype=torch.int32, device=dev)
    items = torch.full((2, 2), -7, dtype=torch.int32, device=dev)
    multi = torch.full((1,), -7, dtype=torch.int32, device=dev)
    rc = L.incagg_spmm_plan_device_static(rp.data_ptr(), 5, 128, 4096, order.data_ptr(),
                                          items.data_ptr(), 2, multi.data_ptr(), 1,
                                          counts.data_ptr(), None)
    assert rc == 0
    torch.cuda.synchronize()
    assert counts.cpu().tolist()[5] == 2 and int(multi[0]) == -7
