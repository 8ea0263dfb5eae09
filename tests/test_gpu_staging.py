"""GPU parity of the host-pinned <-> HBM staging operators against the numpy restatement of
csrc/cuda/async_cuda.cu:62-112,139-163, plus their precondition errors and the pool's
queue discipline under real streams."""
import numpy as np
import pytest
import torch

from oracle import cpu_path as cp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda", 0)


@pytest.fixture(scope="module")
def tga_ops():
    import torch_geometric_autoscale  # noqa: F401  (registers torch.ops)
    return torch.ops.torch_geometric_autoscale


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64, torch.int64])
@pytest.mark.parametrize("use_chunks", [True, False])
def test_read_write_async_roundtrip(tga_ops, dev, dtype, use_chunks):
    rng = np.random.default_rng(0)
    n, w, cap = 3000, 48, 2000
    src = torch.from_numpy(rng.integers(-1000, 1000, size=(n, w))).to(dtype).pin_memory()
    offset = torch.tensor([100, 2500, 7]) if use_chunks else None
    count = torch.tensor([300, 400, 0]) if use_chunks else None
    index = torch.from_numpy(rng.permutation(n)[:555].astype(np.int64))
    dst = torch.zeros(cap, w, dtype=dtype, device=dev)
    buf = torch.empty(cap, w, dtype=dtype).pin_memory()
    stream = torch.cuda.Stream(dev)
    with torch.cuda.stream(stream):
        tga_ops.read_async(src, offset, count, index, dst, buf)
    tga_ops.synchronize()
    stream.synchronize()
    want = np.zeros((cap, w), dtype=src.numpy().dtype)
    cp.read_async(src.numpy(), None if offset is None else offset.numpy(),
                  None if count is None else count.numpy(), index.numpy(), want)
    assert np.array_equal(dst.cpu().numpy(), want)

    # write the pulled chunk rows back somewhere else
    if use_chunks:
        back = torch.zeros(n, w, dtype=dtype).pin_memory()
        w_off, w_cnt = torch.tensor([0, 1000, 2000]), torch.tensor([300, 400, 0])
        with torch.cuda.stream(stream):
            tga_ops.write_async(dst, w_off, w_cnt, back)
        stream.synchronize()
        ref = np.zeros((n, w), dtype=src.numpy().dtype)
        cp.write_async(want, w_off.numpy(), w_cnt.numpy(), ref)
        assert np.array_equal(back.numpy(), ref)


def test_async_preconditions(tga_ops, dev):
    src = torch.zeros(10, 4).pin_memory()
    dst = torch.zeros(10, 4, device=dev)
    buf = torch.zeros(10, 4).pin_memory()
    idx = torch.tensor([1, 2])
    stream = torch.cuda.Stream(dev)
    with pytest.raises(RuntimeError, match="non-default"):
        tga_ops.read_async(src, None, None, idx, dst, buf)
    with torch.cuda.stream(stream):
        with pytest.raises(RuntimeError, match="pinned"):
            tga_ops.read_async(src, None, None, idx, dst, torch.zeros(10, 4))
        with pytest.raises(RuntimeError, match="GPU tensor"):
            tga_ops.read_async(src, None, None, idx, torch.zeros(10, 4), buf)
        with pytest.raises(RuntimeError, match="too small"):
            tga_ops.read_async(src, None, None, torch.arange(10).repeat(2), dst, buf)
        with pytest.raises(RuntimeError, match="Invalid index"):
            tga_ops.read_async(src, torch.tensor([8]), torch.tensor([3]), idx[:0], dst, buf)
        with pytest.raises(RuntimeError, match="Invalid index"):
            tga_ops.write_async(dst, torch.tensor([8]), torch.tensor([3]), src)
        with pytest.raises(RuntimeError, match="pinned"):
            tga_ops.write_async(dst, torch.tensor([0]), torch.tensor([3]), torch.zeros(10, 4))
        # a bad gathered row is reported by the matching synchronize(), not swallowed
        tga_ops.read_async(src, None, None, torch.tensor([3, 10]), dst, buf)
        with pytest.raises(RuntimeError, match="Invalid index"):
            tga_ops.synchronize()
    tga_ops.synchronize()   # nothing queued: no-op


@pytest.mark.parametrize("zero_copy", [True, False])
def test_pool_queue_discipline_on_device(dev, zero_copy):
    """More pulls queued than slots: buffers come back in FIFO order with the right rows,
    deferred pulls start when a slot is freed (pool.py:64-99)."""
    from torch_geometric_autoscale import AsyncIOPool
    rng = np.random.default_rng(1)
    n, w = 2000, 32
    table = torch.from_numpy(rng.standard_normal((n, w)).astype(np.float32)).pin_memory()
    pool = AsyncIOPool(pool_size=2, buffer_size=600, embedding_dim=w, zero_copy=zero_copy).to(dev)
    reqs = []
    for k in range(5):
        off = torch.tensor([100 * k, 1000 + 50 * k])
        cnt = torch.tensor([40 + k, 10])
        idx = torch.from_numpy(rng.permutation(n)[:200 + k].astype(np.int64))
        reqs.append((off, cnt, idx))
        pool.async_pull(table, off, cnt, idx)
    assert len(pool.pending) == 5
    for k, (off, cnt, idx) in enumerate(reqs):
        total = int(cnt.sum()) + idx.numel()
        got = pool.synchronize_pull()[:total].clone()
        want = np.zeros((total, w), dtype=np.float32)
        cp.read_async(table.numpy(), off.numpy(), cnt.numpy(), idx.numpy(), want)
        assert np.array_equal(got.cpu().numpy(), want), f"pull {k}"
        pool.free_pull()
    assert pool.pending == [] and pool.last_pull_slot == -1

    # pushes: slots recycle, the pinned table ends up with every chunk
    dst = torch.zeros(n, w).pin_memory()
    ref = np.zeros((n, w), dtype=np.float32)
    for k in range(5):
        rows = torch.from_numpy(rng.standard_normal((30, w)).astype(np.float32)).to(dev)
        off, cnt = torch.tensor([60 * k, 1500 + 20 * k]), torch.tensor([20, 10])
        pool.async_push(rows, off, cnt, dst)
        cp.write_async(rows.cpu().numpy(), off.numpy(), cnt.numpy(), ref)
    pool.synchronize_push()
    assert np.array_equal(dst.numpy(), ref)
