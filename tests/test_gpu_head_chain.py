"""The interpolation head's dense chain as one kernel (csrc/pps_head_chain_impl.h through train_ops.head_chain; source/poco_model.py:400-414 in
train()) against (i) the same chain as separate launches (head_input, rows_layer x 2, query_attn_pool: the nodes it replaces -- same rounding
points, so the stored tensors agree to the last bit except where the accumulation order inside an MFMA moves a value across a rounding boundary)
and (ii) a float64 torch twin of the reference's formula on the 16-bit operands.
It is what the training graph runs (train_graph.HEAD_CHAIN; PPS_HEAD_CHAIN=0 selects the separate launches)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _case(nq, k, n, seed, dt):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale)
    table = r(n, 256).to(dt)
    ids = torch.randint(0, n, (nq * k,), generator=g)
    pts = torch.rand(n, 3, generator=g) - 0.5
    query = torch.rand(nq, 3, generator=g) - 0.5
    wx = r(256, 3, scale=0.5)
    w2, w3, wq = r(256, 256, scale=1 / 16), r(256, 256, scale=1 / 16), r(64, 256, scale=1 / 8)
    b2, b3, bq = r(256, scale=0.1), r(256, scale=0.1), r(64, scale=0.1)
    dev = lambda t: t.to(DEV)
    return [dev(t) for t in (table, ids, pts, query, wx, w2, b2, w3, b3, wq, bq)]


def _separate(table, ids, pts, query, k, wx, w2, b2, w3, b3, wq, bq):
    from ppsurf_amd import train_ops
    h1 = train_ops.head_input(table, ids, pts, query, k, wx)
    y2 = train_ops.rows_layer(train_ops.Act(h1, None, True), w2, b2, None, True)
    y3 = train_ops.rows_layer(y2, w3, b3, None, True)
    return train_ops.query_attn_pool(y3.raw, wq, bq, k)


def _twin(table, ids, pts, query, k, wx, w2, b2, w3, b3, wq, bq, dt):
    rd = lambda t: t.float().to(dt).double()                          # every stored tensor is rounded to the storage type
    d = lambda t: t.double()
    rel = (d(query).repeat_interleave(k, dim=0) - d(pts)[ids])
    h1 = rd(d(table)[ids] + rel @ d(wx).t())
    y2 = rd(torch.relu(h1) @ rd(w2).t() + d(b2))
    y3 = rd(torch.relu(y2) @ rd(w3).t() + d(b3))
    qy = rd(torch.relu(y3) @ rd(wq).t() + d(bq))
    nq = query.shape[0]
    att = torch.softmax(qy.view(nq, k, 64), dim=1).mean(dim=2)
    return (att.unsqueeze(2) * torch.relu(y3).view(nq, k, 256)).sum(1)


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('nq,k,n', [(300, 64, 4000), (37, 64, 500), (41, 37, 700), (1, 64, 64), (2003, 64, 10000)])
def test_head_chain_equals_the_separate_launches_and_the_formula(nq, k, n, dt):
    from ppsurf_amd import train_ops
    args = _case(nq, k, n, 7 + nq, dt)
    table, ids, pts, query, wx, w2, b2, w3, b3, wq, bq = args
    leaves = [table.clone().requires_grad_(True), wx.clone().requires_grad_(True)] + [t.clone().requires_grad_(True) for t in (w2, b2, w3, b3, wq, bq)]
    leaves2 = [t.detach().clone().requires_grad_(True) for t in leaves]
    with torch.autocast('cuda', dtype=dt):
        fused = train_ops.head_chain(leaves[0], ids, pts, query, k, leaves[1], (leaves[2], leaves[3]), (leaves[4], leaves[5]), (leaves[6], leaves[7]))
        sep = _separate(leaves2[0], ids, pts, query, k, leaves2[1], *leaves2[2:])
    assert fused.dtype == dt and fused.shape == (nq, 256)
    ref = _twin(table, ids, pts, query, k, wx, w2, b2, w3, b3, wq, bq, dt)
    scale = float(ref.abs().max())
    eps = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11
    # (i) the nodes it replaces: values that cross a rounding boundary in one of the three stored layers move the pooled row by a few ulp
    assert float((fused.double() - sep.double()).abs().max()) <= 4 * eps * scale
    # (ii) the formula: storage rounding of three layers of 256-term sums + the pooled row itself
    assert float((fused.double() - ref).abs().max()) <= 6 * eps * scale
    gout = torch.randn(nq, 256, generator=torch.Generator().manual_seed(3)).to(DEV).to(dt)
    gf = torch.autograd.grad(fused, leaves, gout)
    gs = torch.autograd.grad(sep, leaves2, gout)
    for name, a, b in zip(('table', 'wx', 'w2', 'b2', 'w3', 'b3', 'wq', 'bq'), gf, gs):
        assert a.shape == b.shape and a.dtype == b.dtype
        tol = 0.04 * float(b.double().abs().max()) + 1e-6           # (16-bit gradients of a ReLU network whose stored activations differ in a few ulp)
        if name == 'bq':
            # a softmax does not see a constant added to its logits: the exact gradient of fc_query's bias is ZERO, what both paths return is the
            # rounding noise of a 19 200-term sum of 16-bit values that cancel -- compared against that noise level, not against itself
            tol = 2e-4
        d = a.double() - b.double()
        # a stored activation that sits within an ulp of zero takes the other side of its ReLU in one of the two paths: single entries move by the
        # whole contribution of a row, so entries are held to 3 x the bar and the tensor as a whole (l2) to the bar
        if name == 'bq':
            assert float(d.abs().max()) <= tol, name
        else:
            assert float(d.abs().max()) <= 3 * tol and float(d.norm()) <= 0.02 * float(b.double().norm()) + 1e-6, name


def _launch(args, nq, k, dt):
    from ppsurf_amd import _lib
    table, ids, pts, query, wx, w2, b2, w3, b3, wq, bq = args
    L = _lib.lib()
    rows = nq * k
    pad = (rows + 255) // 256 * 256
    h1, y2, y3 = (torch.zeros((pad, 256), device=DEV, dtype=dt) for _ in range(3))
    qy = torch.zeros((pad, 64), device=DEV, dtype=dt)
    ws = torch.empty((L.pps_head_chain_ws_bytes(),), device=DEV, dtype=torch.uint8)
    _lib.check(L.pps_head_chain_fwd(table.data_ptr(), ids.data_ptr(), pts.data_ptr(), query.data_ptr(), nq, k, 1 if dt == torch.bfloat16 else 2,
                                    wx.data_ptr(), w2.data_ptr(), b2.data_ptr(), w3.data_ptr(), b3.data_ptr(), wq.data_ptr(), bq.data_ptr(),
                                    h1.data_ptr(), y2.data_ptr(), y3.data_ptr(), qy.data_ptr(), ws.data_ptr(),
                                    torch.cuda.current_stream().cuda_stream), 'pps_head_chain_fwd')
    torch.cuda.synchronize()
    return h1[:rows], y2[:rows], y3[:rows], qy[:rows]


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('nq,n,reps', [(20000, 100000, 60), (50000, 250000, 20), (24000, 120000, 20), (12000, 60000, 20), (2003, 10000, 40), (777, 3000, 40)])
def test_head_chain_is_run_to_run_identical_and_h1_is_head_input_to_the_bit(dt, nq, n, reps):
    """Repeated launches at the fit batch's size (20 000 queries: 5000 row units; and at the per-rank batches of a data-parallel fit, B = 25 / 12 / 6
    shapes: 50 000 / 24 000 / 12 000 queries, and two small sizes with a partial last unit): every stored tensor equal to the first launch's, and
    h1 equal to the separate head_input kernel's.  (The gather phase once lost this -- one channel of a 16-row tile in ~1 unit of 10^4, from the
    compiler's paired form of the Wx products: the note in csrc/pps_head_chain_impl.h; that form differed in 165 of 300 launches at 20 000
    queries.  ADVICE r5: one size is not a guard.)"""
    from ppsurf_amd import train_ops
    k = 64
    args = _case(nq, k, n, 11, dt)
    first = [t.clone() for t in _launch(args, nq, k, dt)]
    with torch.no_grad():
        ref = train_ops.head_input(args[0], args[1], args[2], args[3], k, args[4])
    assert torch.equal(first[0], ref.view(nq * k, 256))
    for rep in range(reps):
        out = _launch(args, nq, k, dt)
        for name, a, b in zip(('h1', 'y2', 'y3', 'qy'), out, first):
            assert torch.equal(a, b), (name, rep)


def test_head_chain_is_what_the_training_graph_runs():
    """interp_attention (train_graph) with the chain kernel and with the separate launches: same output and parameter gradients within bf16."""
    from ppsurf_amd import modules, train_graph
    import contextlib
    import io
    torch.manual_seed(5)
    with contextlib.redirect_stdout(io.StringIO()):
        proj = modules.InterpAttentionKHeadsNet(256, 256, 64).to(DEV).train() if hasattr(modules, 'InterpAttentionKHeadsNet') else None
    if proj is None:
        pytest.skip('no standalone head module')
    b, n, q, k = 2, 3000, 500, 64
    lat = torch.randn(b, n, 256, device=DEV)
    pts = torch.rand(b, n, 3, device=DEV) - 0.5
    qry = torch.rand(b, q, 3, device=DEV) - 0.5
    ids = torch.randint(0, n, (b, q, k), device=DEV)
    outs = []
    for flag in (True, False):
        train_graph.HEAD_CHAIN = flag
        try:
            proj.zero_grad()
            with torch.autocast('cuda', dtype=torch.bfloat16):
                y = train_graph.interp_attention(proj, lat, pts, qry, ids)
            y.float().square().mean().backward()
            outs.append((y.detach().float(), {nme: p.grad.detach().clone() for nme, p in proj.named_parameters() if p.grad is not None}))
        finally:
            train_graph.HEAD_CHAIN = True
            train_graph.release_step_caches()
    (ya, ga), (yb, gb) = outs
    assert float((ya - yb).abs().max()) <= 2.0 ** -6 * float(yb.abs().max())
    assert ga.keys() == gb.keys() and len(ga) >= 10
    for nme in ga:
        if nme == 'fc_query.bias':                  # exactly zero in exact arithmetic (softmax is shift-invariant): both are cancellation noise
            assert float(ga[nme].abs().max()) <= 1e-2 * float(ga['fc_query.weight'].abs().max()) + 1e-6
            continue
        assert float((ga[nme] - gb[nme]).abs().max()) <= 0.03 * float(gb[nme].abs().max()) + 1e-7, nme


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16])
def test_first_use_self_check_passes_and_is_cached(dt):
    """train_ops.head_chain_trusted (what train_graph asks before it uses the chain kernel): the synthetic case agrees with the separate launches."""
    from ppsurf_amd import train_ops
    train_ops._head_chain_checked.pop(dt, None)
    assert train_ops.head_chain_trusted(dt) is True
    assert train_ops._head_chain_checked[dt] is True and train_ops.head_chain_trusted(dt) is True


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('nq,k', [(300, 64), (41, 37), (1, 64), (2003, 64), (7, 1)])
def test_rebuilt_attention_gradient_is_the_stored_one_without_its_rounding(nq, k, dt, monkeypatch):
    """d y3 of (fc_query, attention pooling): the pooling's share relu'(y3) a[q, j] dpooled[q, c] rebuilt inside fc_query's input-gradient kernel
    (pps_attn_pool_bwd_weights + pps_rows_layer_bwd_attn, the default) against the same share written to memory and read back (PPS_ATTN_GRAD=stored:
    pps_attn_pool_bwd + dx_add) and against float64 autograd of the formula on the same 16-bit operands.  The two differ by ONE 16-bit rounding of
    the share (the stored path rounds it before the add): d wq / d bq are EQUAL, d y3 agrees within that rounding and is no further from the
    formula.  k = 37: row / k is a multiply-high in the kernel; 1517 / 7 rows: partial 32-row tiles."""
    from ppsurf_amd import train_ops
    g = torch.Generator().manual_seed(100 + nq + k)
    y3 = (torch.randn(nq * k, 256, generator=g)).to(dt).to(DEV)
    wq = (torch.randn(64, 256, generator=g) / 8).to(DEV)
    bq = (torch.randn(64, generator=g) * 0.1).to(DEV)
    gout = torch.randn(nq, 256, generator=g).to(dt).to(DEV)
    res = {}
    for mode in ('rebuilt', 'stored'):
        monkeypatch.setenv('PPS_ATTN_GRAD', mode)
        leaves = [y3.clone().requires_grad_(True), wq.clone().requires_grad_(True), bq.clone().requires_grad_(True)]
        with torch.autocast('cuda', dtype=dt):
            pooled = train_ops.query_attn_pool(leaves[0], leaves[1], leaves[2], k)
        res[mode] = (pooled.detach(),) + torch.autograd.grad(pooled, leaves, gout)
    assert all(torch.equal(a, b) for a, b in zip(res['rebuilt'][:1] + res['rebuilt'][2:], res['stored'][:1] + res['stored'][2:]))
    # the formula in float64 on the stored operands (qy rounded to the storage type like the kernels' saved tensor)
    yd = y3.double().requires_grad_(True)
    qy = (torch.relu(yd) @ wq.to(dt).double().t() + bq.double())
    qy = qy + (qy.detach().to(dt).double() - qy.detach())                    # value rounded, gradient straight through
    att = torch.softmax(qy.view(nq, k, 64), dim=1).mean(dim=2)
    ref = torch.autograd.grad((att.unsqueeze(2) * torch.relu(yd).view(nq, k, 256)).sum(1), yd, gout.double())[0]
    a, b = res['rebuilt'][1].double(), res['stored'][1].double()
    eps = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11
    scale = float(ref.abs().max())
    assert float((a - b).abs().max()) <= 2 * eps * scale
    ea, eb = float((a - ref).norm()), float((b - ref).norm())
    assert ea <= 1.02 * eb + 1e-9, (ea, eb)
    assert float((a - ref).abs().max()) <= 0.03 * scale
