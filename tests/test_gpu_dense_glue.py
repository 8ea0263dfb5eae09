"""Dense-side helpers of the training step (column sums, mean softmax cross entropy, Adam) against
torch's own operators -- the ones the reference calls (main.py:60-90,160-165).  fp32 results
within rounding of the fp64 / torch values; tolerances written at every comparison."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda", 0)


@pytest.fixture(scope="module")
def ops():
    from torch_geometric_autoscale import ops
    return ops


@pytest.mark.parametrize("rows,width,pitch", [(128064, 256, 288), (16326, 47, 48), (16326, 47, 47),
                                              (5, 3, 3), (1, 256, 256), (100003, 100, 100),
                                              (70000, 300, 304), (4097, 1, 1)])
def test_column_sums_match_fp64(ops, dev, rows, width, pitch):
    g = torch.Generator(device=dev).manual_seed(rows + width)
    base = torch.randn(rows, pitch, device=dev, generator=g)
    x = base[:, :width]
    got = ops.column_sums(x)
    want = x.double().sum(0)
    # sums of `rows` standard normals are ~ sqrt(rows) large: every add rounds at that size
    tol = 1e-4 + 3e-6 * np.sqrt(rows)
    assert got.shape == (width,)
    assert float((got.double() - want).abs().max()) <= tol
    # fixed summation order: the same bits every time
    assert torch.equal(got, ops.column_sums(x))


@pytest.mark.parametrize("rows,classes,pitch", [(16326, 47, 48), (16326, 47, 47), (1, 41, 41),
                                                (5, 172, 176), (333, 1000, 1000), (4099, 2, 2)])
def test_softmax_cross_entropy_matches_torch(ops, dev, rows, classes, pitch):
    from torch_geometric_autoscale import _lib
    g = torch.Generator(device=dev).manual_seed(rows * 7 + classes)
    base = (torch.randn(rows, pitch, device=dev, generator=g) * 3).requires_grad_()
    y = torch.randint(0, classes, (rows,), device=dev, generator=g)
    logits = base[:, :classes]
    loss = ops.softmax_cross_entropy(logits, y)
    (loss * 1.7).backward()
    got_grad = base.grad[:, :classes].clone()
    ref_in = logits.detach().double().requires_grad_()
    ref = torch.nn.functional.cross_entropy(ref_in, y)
    (ref * 1.7).backward()
    assert abs(float(loss.detach()) - float(ref.detach())) <= 1e-5 * max(1.0, abs(float(ref.detach())))
    assert float((got_grad.double() - ref_in.grad).abs().max()) <= 1e-6 / rows + 1e-5 * float(
        ref_in.grad.abs().max())
    assert float(base.grad[:, classes:].abs().sum()) == 0.0       # padding columns untouched
    _lib.check_device_errors()


def test_softmax_cross_entropy_reports_bad_labels(ops, dev):
    from torch_geometric_autoscale import _lib
    logits = torch.randn(6, 5, device=dev, requires_grad=True)
    y = torch.tensor([0, 4, 5, 1, -1, 2], device=dev)
    loss = ops.softmax_cross_entropy(logits, y)
    loss.backward()
    with pytest.raises(RuntimeError):
        _lib.check_device_errors()
    good = [0, 1, 3, 5]
    ref = torch.nn.functional.cross_entropy(logits.detach()[good], y[good], reduction="sum") / 6
    assert abs(float(loss.detach()) - float(ref)) <= 1e-6
    assert float(logits.grad[[2, 4]].abs().sum()) == 0.0
    with pytest.raises(ValueError):
        ops.softmax_cross_entropy(logits, y[:3])
    with pytest.raises(ValueError):
        ops.softmax_cross_entropy(logits, y.int())
    _lib.check_device_errors()


def _twin_params(dev, shapes, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    ours = [torch.randn(s, device=dev, generator=g).requires_grad_() for s in shapes]
    theirs = [p.detach().clone().requires_grad_() for p in ours]
    return ours, theirs


def _set_grads(params_a, params_b, gen, skip=()):
    for i, (a, b) in enumerate(zip(params_a, params_b)):
        if i in skip:
            a.grad = b.grad = None
            continue
        gr = torch.randn(a.shape, device=a.device, generator=gen)
        a.grad, b.grad = gr.clone(), gr.clone()


def test_adam_matches_torch(dev):
    from torch_geometric_autoscale.optim import Adam
    shapes = [(256, 256), (47,), (256, 100), (1,), (3, 5), (12032,)] + [(17, 9)] * 25
    ours, theirs = _twin_params(dev, shapes, 0)
    groups = lambda ps: [dict(params=ps[:3], weight_decay=5e-4), dict(params=ps[3:], weight_decay=0.0)]
    opt = Adam(groups(ours), lr=1e-2)
    ref = torch.optim.Adam(groups(theirs), lr=1e-2)
    gen = torch.Generator(device=dev).manual_seed(1)
    for it in range(6):
        _set_grads(ours, theirs, gen, skip=(4,) if it % 2 else ())
        opt.step()
        ref.step()
    for a, b in zip(ours, theirs):
        # same formulas in fp32; torch evaluates the bias corrections in double on the host
        a, b = a.detach(), b.detach()
        assert float((a - b).abs().max()) <= 2e-6 + 1e-5 * float(b.abs().max())
    assert [float(opt.state[p]["step"]) for p in ours[:6]] == [6.0, 6.0, 6.0, 6.0, 3.0, 6.0]
    sd = opt.state_dict()
    opt2 = Adam(groups([p.detach().clone().requires_grad_() for p in ours]), lr=1e-2)
    opt2.load_state_dict(sd)
    assert float(opt2.state[opt2.param_groups[1]["params"][1]]["step"]) == 3.0
    assert torch.equal(opt2.state[opt2.param_groups[0]["params"][0]]["exp_avg"],
                       opt.state[ours[0]]["exp_avg"])
    opt.zero_grad()
    assert all(p.grad is None for p in ours)
    with pytest.raises(TypeError):
        Adam([torch.zeros(3, requires_grad=True)])


def test_adam_counts_steps_inside_a_replayed_graph(dev):
    """The step count lives on the device: replays of ONE recorded step advance it."""
    from torch_geometric_autoscale.optim import Adam
    ours, theirs = _twin_params(dev, [(300, 7), (64,)], 3)
    opt, ref = Adam(ours, lr=3e-3), torch.optim.Adam(theirs, lr=3e-3)
    grads = [torch.randn(p.shape, device=dev) for p in ours]
    for p, q, g in zip(ours, theirs, grads):
        p.grad, q.grad = g, g.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        opt.step()                      # warm-up on the capture stream
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            opt.step()
    torch.cuda.current_stream().wait_stream(side)
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    steps = int(float(opt.state[ours[0]]["step"]))
    assert steps == 4                   # 1 eager + 3 replays (recording itself runs nothing)
    for _ in range(steps):
        ref.step()
    for a, b in zip(ours, theirs):
        assert float((a - b).abs().max()) <= 2e-6 + 1e-5 * float(b.abs().max())


@pytest.mark.parametrize("rows,width,pitch", [(16326, 256, 288), (1000, 47, 48), (7, 5, 5)])
def test_relu_and_gate_add_rows(ops, dev, rows, width, pitch):
    g = torch.Generator(device=dev).manual_seed(rows)
    base = torch.randn(rows, pitch, device=dev, generator=g)
    x = base[:, :width]
    want = x.clamp(min=0)
    keep = base[:, width:].clone()
    assert torch.equal(ops.relu_rows_(x), want)
    assert torch.equal(base[:, width:], keep)              # padding untouched
    dst = torch.randn(rows, pitch, device=dev, generator=g)[:, :width]
    src = torch.randn(rows, width, device=dev, generator=g)
    gate = torch.randn(rows, pitch, device=dev, generator=g)[:, :width]
    want = dst + torch.where(gate > 0, src, torch.zeros_like(src))
    assert torch.equal(ops.gate_add_rows_(dst, src, gate), want)
    with pytest.raises(ValueError):
        ops.gate_add_rows_(dst, src[:-1], gate)


@pytest.mark.parametrize("n_layers,c", [(8, 256), (3, 20), (1, 7)])
def test_gcn2_weight_fold_matches_the_formulas(ops, dev, n_layers, c):
    """Folded GCNII weights of all layers and their gradient against plain torch."""
    import math
    from torch_geometric_autoscale.nn import _FoldedWeights
    g = torch.Generator(device=dev).manual_seed(c)
    alpha = 0.1
    betas = tuple(math.log(0.5 / (l + 1) + 1) for l in range(n_layers))
    ws = torch.randn(2 * n_layers, c, c, device=dev, generator=g, requires_grad=True)
    w1f, w2cat = _FoldedWeights.apply(ws, alpha, betas)
    eye = torch.eye(c, device=dev)
    ref = ws.detach().clone().requires_grad_()
    r1 = torch.stack([(1 - alpha) * (b * ref[l] + (1 - b) * eye) for l, b in enumerate(betas)])
    r2 = torch.cat([alpha * (b * ref[n_layers + l] + (1 - b) * eye) for l, b in enumerate(betas)], 1)
    assert torch.allclose(w1f, r1, rtol=1e-6, atol=1e-7)
    assert torch.allclose(w2cat, r2, rtol=1e-6, atol=1e-7)
    d1 = torch.randn(w1f.shape, device=dev, generator=g)
    d2 = torch.randn(w2cat.shape, device=dev, generator=g)
    torch.autograd.backward([w1f, w2cat], [d1, d2])
    torch.autograd.backward([r1, r2], [d1, d2])
    assert torch.allclose(ws.grad, ref.grad, rtol=1e-6, atol=1e-7)
    with pytest.raises(ValueError):
        ops.gcn2_fold_weights(ws.detach(), alpha, betas[:-1] if n_layers > 1 else betas + (0.5,))


@pytest.mark.parametrize("rows,width", [(116480, 41), (16326, 256), (3, 5), (1, 1)])
@pytest.mark.parametrize("n", [1, 2, 5, 16])
def test_rows_sum_matches_torch(ops, dev, rows, width, n):
    """scale * (s_0 + ... + s_{n-1}) over operands of different row pitches (contiguous,
    16-byte-aligned padded, unaligned padded); the result has the aggregation's pitch and its
    padding is not relied on.  fp32 sums in the order of the operands: exact against the same
    chain in torch."""
    g = torch.Generator(device=dev).manual_seed(rows + 7 * n)
    pitches = [width, (width + 15) // 16 * 16 + 32, width + 3]
    srcs = [torch.randn(rows, pitches[k % 3], device=dev, generator=g)[:, :width] for k in range(n)]
    got = ops.rows_sum(srcs, 0.1)
    want = srcs[0].clone()
    for s in srcs[1:]:
        want += s
    want *= 0.1
    assert got.shape == (rows, width)
    assert torch.equal(got, want)
    assert got.stride(0) == ops.row_pitch(width) or rows == 1
    with pytest.raises(ValueError):
        ops.rows_sum([])
    with pytest.raises(ValueError):
        ops.rows_sum(srcs + [srcs[0][:, :max(width - 1, 0)]])


@pytest.mark.parametrize("rows,k,n", [(16326, 256, 256), (1000, 64, 128), (130, 32, 100), (64, 256, 4),
                                      (5, 96, 256), (4097, 128, 36)])
@pytest.mark.parametrize("trans_w", [False, True])
def test_rows_gemm_matches_float64(rows, k, n, trans_w):
    """The hand-written fp32 matrix-core product against a float64 product of the same fp32
    operands: plain, accumulating into a column block of a wider matrix, and with the fused
    History push (with and without the activation) into a range of table rows -- everything
    outside the addressed rows / columns untouched."""
    from torch_geometric_autoscale import ops, _lib
    dev = torch.device("cuda", 0)
    assert ops.rows_gemm_supported(k, n)
    g = torch.Generator(device="cpu").manual_seed(rows + k + n)
    xbuf = torch.randn(rows, k + 8, generator=g).to(dev)
    x = xbuf[:, :k]
    w = torch.randn((n, k) if trans_w else (k, n), generator=g).to(dev) * 0.1
    ref = x.double() @ (w.double().t() if trans_w else w.double())
    scale = float(ref.abs().max())
    tol = 2e-6 * scale * (k ** 0.5)
    y = ops.rows_gemm(x, w, trans_w)
    assert y.shape == (rows, n) and float((y.double() - ref).abs().max()) <= tol
    assert torch.allclose(y, x @ (w.t() if trans_w else w), rtol=1e-4, atol=1e-4 * scale)
    # accumulate into a block of a wider matrix + push max(0, .) into table rows [7, 7 + rows)
    wide = torch.randn(rows, 3 * n + 4, generator=g).to(dev)
    before = wide.clone()
    blk = wide[:, n:2 * n]
    table = torch.full((rows + 20, n + 12), 3.0, device=dev)
    out = ops.rows_gemm(x, w, trans_w, out=blk, accumulate=True, table=table, table_row0=7,
                        relu_push=True)
    assert out.data_ptr() == blk.data_ptr()
    want = before[:, n:2 * n].double() + ref
    assert float((blk.double() - want).abs().max()) <= tol + 1e-6 * float(want.abs().max())
    assert torch.equal(wide[:, :n], before[:, :n]) and torch.equal(wide[:, 2 * n:], before[:, 2 * n:])
    assert torch.equal(table[7:7 + rows, :n], torch.relu(blk))
    assert bool((table[:7] == 3.0).all()) and bool((table[7 + rows:] == 3.0).all())
    assert bool((table[:, n:] == 3.0).all())
    table.fill_(3.0)
    ops.rows_gemm(x, w, trans_w, out=blk, table=table, table_row0=0)
    assert torch.equal(table[:rows, :n], blk)
    # table only: the accumulated-into matrix is read, not written
    blk.copy_(before[:, n:2 * n])
    table.fill_(3.0)
    ops.rows_gemm(x, w, trans_w, out=blk, accumulate=True, table=table, table_row0=3,
                  relu_push=True, table_only=True)
    assert torch.equal(blk, before[:, n:2 * n])
    assert float((table[3:3 + rows, :n].double() - torch.relu(want)).abs().max()) <= \
        tol + 1e-6 * float(want.abs().max())
    # a bias per output column: alone, and on top of the accumulated-into matrix
    bias = torch.randn(n, generator=g).to(dev)
    yb = ops.rows_gemm(x, w, trans_w, bias=bias)
    assert float((yb.double() - (ref + bias.double())).abs().max()) <= tol + 1e-6
    blk.copy_(before[:, n:2 * n])
    ops.rows_gemm(x, w, trans_w, out=blk, accumulate=True, bias=bias)
    assert float((blk.double() - (want + bias.double())).abs().max()) <= \
        tol + 2e-6 * float(want.abs().max())
    with pytest.raises(ValueError):
        ops.rows_gemm(x, w, trans_w, bias=bias[:-1].contiguous())
    with pytest.raises(Exception):
        ops.rows_gemm(x, w, trans_w, out=blk, table=table, table_row0=21)
    with pytest.raises(Exception):
        ops.rows_gemm(xbuf[:, :k - 1], w[:, :k - 1] if trans_w else w[:k - 1], trans_w)
    _lib.check_device_errors()


@pytest.mark.parametrize("rows,width,pitch", [(16326, 256, 260), (5000, 40, 40), (2, 4, 4),
                                              (70001, 128, 132), (333, 1028, 1028)])
def test_batch_norm_rows_matches_float64(ops, rows, width, pitch):
    """Training-mode batch norm over rows (incagg_batch_norm_forward_f32 / _backward_f32) against
    torch's batch_norm on float64 copies: output, saved statistics, running statistics, and the
    three gradients; columns with a mean far from zero (the shifted moments must not cancel)."""
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu").manual_seed(rows + width)
    x64 = torch.randn(rows, width, generator=g, dtype=torch.float64)
    x64 = x64 * torch.linspace(0.1, 3.0, width, dtype=torch.float64) \
        + torch.linspace(-50.0, 200.0, width, dtype=torch.float64)
    w64 = torch.rand(width, generator=g, dtype=torch.float64) + 0.5
    b64 = torch.randn(width, generator=g, dtype=torch.float64)
    dy64 = torch.randn(rows, width, generator=g, dtype=torch.float64)
    rm64 = torch.randn(width, generator=g, dtype=torch.float64)
    rv64 = torch.rand(width, generator=g, dtype=torch.float64) + 0.5
    momentum, eps = 0.1, 1e-5
    # (the float64 side works on the values the kernels get: rounded to float32)
    x64, w64, b64, dy64, rm64, rv64 = (t.float().double() for t in (x64, w64, b64, dy64, rm64, rv64))

    xr = x64.clone().requires_grad_(True)
    wr, br = w64.clone().requires_grad_(True), b64.clone().requires_grad_(True)
    rm_ref, rv_ref = rm64.clone(), rv64.clone()
    want = torch.nn.functional.batch_norm(xr, rm_ref, rv_ref, wr, br, True, momentum, eps)
    want.backward(dy64)

    buf = torch.zeros(rows, pitch, device=dev)
    x = buf[:, :width]
    x.copy_(x64)
    dybuf = torch.zeros(rows, pitch, device=dev)
    dy = dybuf[:, :width]
    dy.copy_(dy64)
    w, b = w64.float().to(dev), b64.float().to(dev)
    rm, rv = rm64.float().to(dev), rv64.float().to(dev)
    assert ops.batch_norm_supported(x)
    y, mean, rstd = ops.batch_norm_forward(x, w, b, rm, rv, momentum, eps)
    scale = float(want.detach().abs().max())
    assert float((y.double().cpu() - want.detach()).abs().max()) <= 2e-5 * max(scale, 1.0)
    np.testing.assert_allclose(mean[:width].cpu().numpy(), x64.mean(0).numpy(), rtol=1e-6, atol=1e-5)
    np.testing.assert_allclose(rstd[:width].cpu().numpy(),
                               (1.0 / torch.sqrt(x64.var(0, unbiased=False) + eps)).numpy(),
                               rtol=2e-5)
    np.testing.assert_allclose(rm.cpu().numpy(), rm_ref.numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(rv.cpu().numpy(), rv_ref.numpy(), rtol=2e-5, atol=1e-6)
    dx, dw, db = ops.batch_norm_backward(dy, x, w, mean, rstd)
    tol = 2e-5 * max(float(xr.grad.abs().max()), 1e-3) + 1e-6
    assert float((dx.double().cpu() - xr.grad).abs().max()) <= tol + 2e-4 * float(xr.grad.abs().max())
    np.testing.assert_allclose(dw[:width].cpu().numpy(), wr.grad.numpy(), rtol=2e-4,
                               atol=2e-4 * float(wr.grad.abs().max()))
    np.testing.assert_allclose(db[:width].cpu().numpy(), br.grad.numpy(), rtol=2e-4,
                               atol=2e-4 * float(br.grad.abs().max()))
    # in place over the incoming gradient
    dx2, _, _ = ops.batch_norm_backward(dy, x, w, mean, rstd, dx=dy)
    assert dx2.data_ptr() == dy.data_ptr() and torch.equal(dx2, dx)
    # without affine parameters and without running statistics
    y0, m0, r0 = ops.batch_norm_forward(x, None, None, None, None, momentum, eps)
    want0 = torch.nn.functional.batch_norm(x64, None, None, None, None, True, momentum, eps)
    assert float((y0.double().cpu() - want0).abs().max()) <= 2e-5 * max(float(want0.abs().max()), 1.0)
    from torch_geometric_autoscale import _lib
    _lib.check_device_errors()


def test_batch_norm_rows_module_path_and_errors(ops):
    """nn.batch_norm_rows takes the kernel path in training mode (same result as the module, the
    module's buffers updated, gradients through autograd) and the module itself in evaluation
    mode; unsupported layouts are refused by the entry point."""
    from torch_geometric_autoscale import nn as tnn
    dev = torch.device("cuda")
    torch.manual_seed(3)
    bn = torch.nn.BatchNorm1d(64).to(dev)
    ref = torch.nn.BatchNorm1d(64).to(dev)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_()
    ref.load_state_dict(bn.state_dict())
    x = torch.randn(3000, 64, device=dev) * 2 + 1
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ya, yb = tnn.batch_norm_rows(bn, xa), ref(xb)
    assert float((ya - yb).detach().abs().max()) <= 1e-4
    go = torch.randn_like(ya)
    ya.backward(go)
    yb.backward(go)
    assert float((xa.grad - xb.grad).abs().max()) <= 1e-4 * float(xb.grad.abs().max()) + 1e-6
    assert float((bn.weight.grad - ref.weight.grad).abs().max()) <= 1e-3 * float(ref.weight.grad.abs().max())
    assert float((bn.bias.grad - ref.bias.grad).abs().max()) <= 1e-3 * float(ref.bias.grad.abs().max())
    assert int(bn.num_batches_tracked) == 1
    assert float((bn.running_mean - ref.running_mean).abs().max()) <= 1e-5
    assert float((bn.running_var - ref.running_var).abs().max()) <= 1e-4
    bn.eval()
    ref.eval()
    assert torch.equal(tnn.batch_norm_rows(bn, x), bn(x))
    with pytest.raises(ValueError):
        ops.batch_norm_forward(torch.randn(10, 6, device=dev), None, None, None, None, 0.1, 1e-5)
    with pytest.raises(ValueError):
        ops.batch_norm_forward(torch.randn(1, 8, device=dev), None, None, None, None, 0.1, 1e-5)


@pytest.mark.parametrize("rows,width,pitches", [(16326, 256, (260, 288, 2080)), (5000, 41, (48, 48, 48)),
                                                (333, 41, (41, 41, 41)), (1, 4, (4, 8, 4))])
def test_gate_rows_is_the_relu_backward(ops, rows, width, pitches):
    """incagg_gate_rows_f32 = torch's threshold_backward on matrices with three different row
    pitches (bit-exact: a select), also in place over the incoming gradient; and through
    nn.relu_into's autograd function."""
    from torch_geometric_autoscale import nn as tnn
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(rows)
    bufs = [torch.randn(rows, p, device=dev, generator=g) for p in pitches]
    dy, h, out = (b[:, :width] for b in bufs)
    want = torch.ops.aten.threshold_backward(dy, h, 0)
    got = ops.gate_rows(dy, h, out=out)
    assert got.data_ptr() == out.data_ptr() and torch.equal(got, want)
    assert torch.equal(ops.gate_rows(dy, h), want)
    dy2 = dy.clone()
    assert torch.equal(ops.gate_rows(dy2, h, out=dy2), want)
    if width % 4 == 0 or pitches[0] != width:
        x = ops.rows_empty(rows, width, dev).copy_(h).requires_grad_(True)
        y = tnn.relu_into(x * 1.0)
        y.backward(dy)
        assert torch.equal(x.grad, want)
