"""Fused dense layer of the training step (pps_rows_train.hip through train_ops.rows_layer) against the same layer written with torch
ops: forward output, BatchNorm affine + running statistics, and every gradient.  The torch twin uses the operands the kernels use
(inputs, weights and the row gradient G rounded to bf16, fp32 accumulation), so the tolerances are those of bf16 STORAGE of the
results (2^-8 relative) plus the accumulation order, not of a different algorithm."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


DT16 = [torch.bfloat16, torch.float16]


def _bf(t, dt=torch.bfloat16):
    return t.to(dt).float()


class _RoundGrad(torch.autograd.Function):
    """identity whose gradient is rounded to bf16: the row gradient G = gy + gS + 2 y gQ enters the kernels' products (and db) as a bf16 MFMA
    operand -- like the bf16 grad_input of the reference's batch-norm backward under autocast.  Adding the per-channel constant gS to
    bf16 values and rounding again is NOT unbiased within a binade, so an unrounded twin differs systematically in the sums over rows."""

    @staticmethod
    def forward(ctx, t, dt):
        ctx.dt = dt
        return t.clone()

    @staticmethod
    def backward(ctx, g):
        return g.float().to(ctx.dt).to(g.dtype), None


def _twin(x, aff, relu, w, b, gamma, beta, eps, dt=torch.bfloat16):
    """float64 twin on the bf16-rounded operands.  Returns y (unrounded), out_affine or None."""
    a = x.double()
    if aff is not None:
        a = a * aff[0].double() + aff[1].double()
        if relu:
            a = torch.relu(a)
    a = a.float().to(dt).double() if aff is not None else a                    # the MFMA operand is 16-bit
    y = a @ w.to(dt).double().t()
    if b is not None:
        y = y + b.double()
    y = _RoundGrad.apply(y, dt)
    if gamma is None:
        return y, None
    yr = y.float().to(dt).double()                                            # statistics are those of the stored y
    yr = y + (yr - y).detach()
    m = yr.shape[0]
    mean = yr.sum(0) / m
    var = (yr * yr).sum(0) / m - mean * mean
    sc = gamma.double() / torch.sqrt(var + eps)
    return y, torch.stack([sc, beta.double() - mean * sc])


CASES = [(5000, 64, 64, True, True, True), (3001, 64, 128, True, True, True), (4100, 128, 256, True, True, True),
         (2500, 256, 256, False, False, True), (1999, 256, 64, True, False, True), (70, 64, 64, False, True, False),
         (40000, 128, 128, True, True, True), (33, 256, 128, True, True, True)]


@pytest.mark.parametrize('dt', DT16)
@pytest.mark.parametrize('rows,cin,cout,affine,bn,bias', CASES)
def test_rows_layer_matches_the_torch_twin(rows, cin, cout, affine, bn, bias, dt):
    from ppsurf_amd import train_ops
    g = torch.Generator().manual_seed(rows + cin + cout)
    rnd = lambda *s: torch.randn(*s, generator=g)
    x = _bf(rnd(rows, cin) * 1.5 + 0.3, dt).to(DEV)
    w = (rnd(cout, cin) / cin ** 0.5).to(DEV)
    b = (rnd(cout) * 0.2).to(DEV) if bias else None
    aff = torch.stack([rnd(cin) * 0.3 + 1.0, rnd(cin) * 0.4]).to(DEV) if affine else None
    gamma = (rnd(cout) * 0.2 + 1.0).to(DEV) if bn else None
    beta = (rnd(cout) * 0.3).to(DEV) if bn else None
    gy = _bf(rnd(rows, cout), dt).to(DEV)
    ga = rnd(2, cout).to(DEV) * rows ** 0.5 if bn else None
    eps, mom = 1e-5, 0.1

    class Holder:
        pass
    hold = None
    if bn:
        hold = Holder()
        hold.weight, hold.bias = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
        hold.running_mean, hold.running_var = torch.zeros(cout, device=DEV), torch.ones(cout, device=DEV)
        hold.momentum, hold.eps = mom, eps

    # ---- ours
    xo = x.to(dt).requires_grad_(True)
    wo = w.clone().requires_grad_(True)
    bo = b.clone().requires_grad_(True) if bias else None
    ao = aff.clone().requires_grad_(True) if affine else None
    out = train_ops.rows_layer(train_ops.Act(xo, ao, affine), wo, bo, hold, relu=True)
    loss = (out.raw.float() * gy).sum()
    if bn:
        loss = loss + (out.affine * ga).sum()
    loss.backward()
    torch.cuda.synchronize()

    # ---- twin
    xt = x.clone().requires_grad_(True)
    wt = w.clone().requires_grad_(True)
    bt = b.clone().requires_grad_(True) if bias else None
    at = aff.clone().requires_grad_(True) if affine else None
    gt = gamma.clone().requires_grad_(True) if bn else None
    bet = beta.clone().requires_grad_(True) if bn else None
    # straight-through bf16 rounding of the weights (the twin differentiates wrt the fp32 master weights like the kernel does)
    wq = wt + (wt.to(dt).float() - wt).detach()
    y, oa = _twin(xt, at, affine, wq, bt, gt, bet, eps, dt)
    lt = (y * gy.double()).sum()
    if bn:
        lt = lt + (oa * ga.double()).sum()
    lt.backward()

    def close(ours, theirs, rel, name):
        theirs = theirs.float()
        scale = float(theirs.detach().abs().max()) + 1e-12
        err = float((ours.detach().float() - theirs.detach()).abs().max())
        assert err <= rel * scale, '{}: max abs err {:.3e} vs scale {:.3e}'.format(name, err, scale)

    close(out.raw, y.detach(), 6e-3, 'y')                               # bf16 storage of y: 2^-8
    if bn:
        close(out.affine, oa.detach(), 2e-4, 'out_affine')
        yr = y.detach().float().to(dt).double()
        mean = yr.mean(0)
        var_unb = yr.var(0, unbiased=True)
        close(hold.running_mean, (mom * mean), 2e-4, 'running_mean')
        close(hold.running_var, (1 - mom) + mom * var_unb, 2e-4, 'running_var')
        close(hold.weight.grad, gt.grad, 2e-2, 'dgamma')
        close(hold.bias.grad, bet.grad, 2e-2, 'dbeta')
    # gradients: G and act(x) enter the products rounded to bf16 (2^-8 per element, averaging down over the contraction); dx is stored in bf16
    close(xo.grad, xt.grad, 1.5e-2, 'dx')
    close(wo.grad, wt.grad, 1.5e-2, 'dw')
    if bias:
        close(bo.grad, bt.grad, 1.5e-2, 'db')
    if affine:
        close(ao.grad, at.grad, 1.5e-2, 'd_in_affine')


def test_rows_layer_is_deterministic():
    from ppsurf_amd import train_ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(20000, 128, generator=g).to(DEV).to(torch.bfloat16)
    w = (torch.randn(256, 128, generator=g) / 11).to(DEV)
    outs = []
    for _ in range(2):
        xo, wo = x.clone().requires_grad_(True), w.clone().requires_grad_(True)

        class H:
            weight, bias = torch.ones(256, device=DEV, requires_grad=True), torch.zeros(256, device=DEV, requires_grad=True)
            running_mean, running_var, momentum, eps = torch.zeros(256, device=DEV), torch.ones(256, device=DEV), 0.1, 1e-5
        out = train_ops.rows_layer(train_ops.Act(xo), wo, None, H, relu=True)
        (out.raw.float().square().sum() + out.affine.sum()).backward()
        outs.append((out.raw.detach().clone(), out.affine.detach().clone(), xo.grad.clone(), wo.grad.clone(), H.weight.grad.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize('dt', DT16)
def test_pointnet_with_fused_row_layers_is_as_close_to_fp32_as_the_separate_ops(dt):
    """train_graph.pointnet under bf16 autocast, fused row layers vs the separate ops (library GEMMs + fused BatchNorm op), both measured
    against the fp32 run of the same graph: the max-pool of the STN makes bf16 gradients differ by whole arg-max switches, so the two
    bf16 paths are not compared with each other but by their distance to fp32."""
    import contextlib
    import io
    from ppsurf_amd import modules, synthetic, train_graph
    with contextlib.redirect_stdout(io.StringIO()):
        net = modules.PPSurfNetwork(in_channels=3, latent_size=256, out_channels=2, k=64, num_pts_local=50, pointnet_latent_size=256)
    if dt == torch.bfloat16:
        net.load_state_dict(synthetic.network_state_dict('ppsurf', num_pts_local=50))
    # (fp16: the constructor's default initialisation -- the formula-filled weights drive the raw conv outputs past 65504, the separate-op
    # path then returns NaN and nothing can be compared)
    pn = net.point_net.to(DEV).train()
    state = {k: v.clone() for k, v in pn.state_dict().items()}
    g = torch.Generator().manual_seed(3)
    patches = (torch.randn(2000, 50, 3, generator=g) * 0.4).to(DEV)
    gout = torch.randn(2000, 256, generator=g).to(DEV)
    res = {}
    for mode in ('fp32', 'separate', 'fused'):
        pn.load_state_dict(state)
        pn.zero_grad(set_to_none=True)
        train_graph.FUSED_ROWS = mode == 'fused'
        try:
            with torch.autocast('cuda', dtype=dt, enabled=mode != 'fp32'):
                feat, trans2 = train_graph.pointnet(pn, patches)
            (feat.float() * gout).sum().backward()
        finally:
            train_graph.FUSED_ROWS = True
        res[mode] = (feat.detach().float(), trans2.detach().float(), {k: p.grad.detach().clone() for k, p in pn.named_parameters() if p.grad is not None},
                     {k: v.clone() for k, v in pn.state_dict().items() if 'running' in k})

    def dist(a, b):
        return float((a - b).abs().max())

    ref = res['fp32']
    for i, name in ((0, 'features'), (1, 'trans2')):
        scale = float(ref[i].abs().max())
        assert dist(res['fused'][i], ref[i]) <= 2.5 * dist(res['separate'][i], ref[i]) + 1e-2 * scale, name
    assert set(res['fused'][2]) == set(ref[2]) == set(res['separate'][2])
    gmax = max(float(v.abs().max()) for v in ref[2].values())
    worse = []
    for k, g32 in ref[2].items():
        scale = float(g32.abs().max())
        if scale < 1e-6 * gmax:
            continue                                                    # biases in front of a BatchNorm: the gradient is rounding noise everywhere
        # in the root-mean-square sense: ONE arg-max switch of the STN's max-pool moves single entries of the early layers' gradients by 10 % of
        # the largest entry in either 16-bit path (conv0a.weight in fp16: 971 fused, 319 separate of 6500), which says nothing about the path
        ef, es = float((res['fused'][2][k] - g32).norm()), float((res['separate'][2][k] - g32).norm())
        if ef > 2.5 * es + 2e-2 * float(g32.norm()):
            worse.append((k, ef, es, float(g32.norm())))
    assert not worse, worse
    for k, v32 in ref[3].items():
        assert dist(res['fused'][3][k], v32) <= 2.5 * dist(res['separate'][3][k], v32) + 2e-3 * float(v32.abs().max()) + 1e-5, k


@pytest.mark.parametrize('dt', DT16)
@pytest.mark.parametrize('q,k', [(37, 50), (5, 64), (300, 1), (2049, 20)])
def test_patch_attention_pooling_matches_torch(q, k, dt):
    from ppsurf_amd import train_ops
    g = torch.Generator().manual_seed(q * 100 + k)
    h = (torch.randn(q, k, 256, generator=g)).to(DEV).to(dt)
    v = (torch.randn(256, generator=g) * 0.2).to(DEV)
    gp = torch.randn(q, 256, generator=g).to(DEV)
    ho, vo = h.clone().requires_grad_(True), v.clone().requires_grad_(True)
    out = train_ops.patch_attn(ho, vo)
    (out * gp).sum().backward()
    ht, vt = h.double().requires_grad_(True), v.double().requires_grad_(True)
    a = torch.softmax(ht @ vt, dim=1)
    ref = (a.unsqueeze(-1) * ht).sum(1)
    (ref * gp.double()).sum().backward()
    assert torch.allclose(out.double(), ref, rtol=1e-4, atol=1e-5)
    assert float((ho.grad.double() - ht.grad).abs().max()) <= 6e-3 * float(ht.grad.abs().max())          # dh is stored in bf16
    assert torch.allclose(vo.grad.double(), vt.grad, rtol=2e-4, atol=2e-4 * float(vt.grad.abs().max()))


@pytest.mark.parametrize('dt', DT16)
def test_act_max_matches_the_materialised_maximum(dt):
    from ppsurf_amd import train_ops
    g = torch.Generator().manual_seed(11)
    raw = torch.randn(90 * 50, 256, generator=g).to(DEV).to(dt)
    aff = torch.stack([torch.randn(256, generator=g), torch.randn(256, generator=g)]).to(DEV)      # negative scales included
    go = torch.randn(90, 256, generator=g).to(DEV)
    ro, ao = raw.clone().requires_grad_(True), aff.clone().requires_grad_(True)
    out = train_ops.act_max(train_ops.Act(ro, ao, True), 90, 50)
    (out * go).sum().backward()
    rt, at = raw.float().requires_grad_(True), aff.clone().requires_grad_(True)
    ref = torch.relu(rt * at[0] + at[1]).view(90, 50, 256).max(dim=1)[0]
    (ref * go).sum().backward()
    assert torch.allclose(out, ref, rtol=1e-6, atol=1e-6)
    assert torch.allclose(ao.grad, at.grad, rtol=1e-4, atol=1e-4)
    # ties between equal bf16 values may pick another row: compare the gradient summed over the patch
    assert torch.allclose(ro.grad.float().view(90, 50, 256).sum(1), rt.grad.view(90, 50, 256).sum(1), rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize('dt', DT16)
@pytest.mark.parametrize('groups,p', [(90, 50), (333, 20), (7, 2), (40, 64), (20000, 50), (1000, 255)])
def test_layer_and_patch_maximum_as_one_node_equal_the_two_nodes(dt, groups, p):
    """train_ops.rows_layer_max (the gradient comes back per patch; pps_rows_layer_bwd_pooled rebuilds its rows on load) against
    act_max(rows_layer(...)) with the scattered [rows, 256] gradient tensor: the same kernels on the same values -- equal outputs and gradients."""
    from ppsurf_amd import train_ops
    g = torch.Generator().manual_seed(groups + p)
    rows = groups * p
    assert train_ops.rows_layer_max_supported(rows, 128, 256, groups, p) and not train_ops.rows_layer_max_supported(rows, 64, 256, groups, p)
    assert not train_ops.rows_layer_max_supported(groups, 128, 256, groups, 1) and not train_ops.rows_layer_max_supported(rows + 1, 128, 256, groups, p)
    x = (torch.randn(rows, 128, generator=g) * 0.7).to(DEV).to(dt)
    in_aff = torch.stack([torch.rand(128, generator=g) + 0.5, torch.randn(128, generator=g) * 0.3]).to(DEV)
    bn = torch.nn.BatchNorm1d(256).to(DEV).train()
    with torch.no_grad():
        bn.weight.copy_(torch.randn(256, generator=g))                     # negative scales: the minimum wins there
        bn.bias.copy_(torch.randn(256, generator=g) * 0.2)
    w = (torch.randn(256, 128, generator=g) * 0.1).to(DEV)
    b = (torch.randn(256, generator=g) * 0.1).to(DEV)
    go = torch.randn(groups, 256, generator=g).to(DEV)
    res = []
    for fused in (True, False):
        xs, affs, ws, bs = x.clone().requires_grad_(True), in_aff.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        bn.zero_grad(set_to_none=True)
        bn.running_mean.zero_()
        bn.running_var.fill_(1.0)
        act = train_ops.Act(xs, affs, True)
        if fused:
            out = train_ops.rows_layer_max(act, ws, bs, bn, True, groups, p)
        else:
            out = train_ops.act_max(train_ops.rows_layer(act, ws, bs, bn, True), groups, p)
        (out * go).sum().backward()
        res.append((out.detach(), xs.grad, affs.grad, ws.grad, bs.grad, bn.weight.grad.clone(), bn.bias.grad.clone(), bn.running_mean.clone(), bn.running_var.clone()))
    names = ('out', 'dx', 'd_in_affine', 'dw', 'db', 'dgamma', 'dbeta', 'running_mean', 'running_var')
    for name, a, c in zip(names, res[0], res[1]):
        assert torch.equal(a, c), (name, float((a.float() - c.float()).abs().max()))


@pytest.mark.parametrize('dt', DT16)
def test_interpolation_head_with_fused_row_layers_is_as_close_to_fp32_as_the_separate_ops(dt):
    import contextlib
    import io
    from ppsurf_amd import modules, synthetic, train_graph
    with contextlib.redirect_stdout(io.StringIO()):
        net = modules.PPSurfNetwork(in_channels=3, latent_size=256, out_channels=2, k=64, num_pts_local=50, pointnet_latent_size=256)
    net.load_state_dict(synthetic.network_state_dict('ppsurf', num_pts_local=50))
    proj = net.projection.to(DEV).train()
    g = torch.Generator().manual_seed(8)
    b, n, q, k = 2, 3000, 400, 64
    lat0 = (torch.randn(b, n, 256, generator=g) * 0.5).to(DEV)
    pts = torch.rand(b, n, 3, generator=g).to(DEV)
    query = torch.rand(b, q, 3, generator=g).to(DEV)
    ids = torch.randint(0, n, (b, q, k), generator=g).to(DEV)
    gout = torch.randn(b, q, 256, generator=g).to(DEV)
    res = {}
    for mode in ('fp32', 'separate', 'fused'):
        proj.zero_grad(set_to_none=True)
        lat = lat0.clone().requires_grad_(True)
        train_graph.FUSED_ROWS = mode == 'fused'
        try:
            with torch.autocast('cuda', dtype=dt, enabled=mode != 'fp32'):
                out = train_graph.interp_attention(proj, lat, pts, query, ids, last_layer=True)
            (out.float() * gout).sum().backward()
        finally:
            train_graph.FUSED_ROWS = True
            train_graph.release_step_caches()
        grads = {k_: p.grad.detach().clone() for k_, p in proj.named_parameters() if p.grad is not None}
        grads['latents'] = lat.grad.detach().clone()
        res[mode] = (out.detach().float(), grads)
    dist = lambda a, c: float((a - c).abs().max())
    ref = res['fp32']
    assert dist(res['fused'][0], ref[0]) <= 2.5 * dist(res['separate'][0], ref[0]) + 1e-2 * float(ref[0].abs().max())
    assert set(res['fused'][1]) == set(ref[1])
    worse = []
    for name, g32 in ref[1].items():
        scale = float(g32.abs().max())
        ef, es = dist(res['fused'][1][name], g32), dist(res['separate'][1][name], g32)
        if ef > 2.5 * es + 2e-2 * scale:
            worse.append((name, ef, es, scale))
    assert not worse, worse


@pytest.mark.parametrize('dt', DT16)
def test_rows3_layer_matches_torch(dt):
    from ppsurf_amd import train_ops
    g = torch.Generator().manual_seed(21)
    rows = 7013
    x = (torch.randn(rows, 3, generator=g) * 0.5).to(DEV)
    w, b = (torch.randn(64, 3, generator=g)).to(DEV), (torch.randn(64, generator=g) * 0.1).to(DEV)
    gy = _bf(torch.randn(rows, 64, generator=g), dt).to(DEV)
    ga = (torch.randn(2, 64, generator=g) * 30).to(DEV)

    class H:
        weight, bias = (torch.rand(64, generator=g) + 0.5).to(DEV).requires_grad_(True), torch.randn(64, generator=g).to(DEV).requires_grad_(True)
        running_mean, running_var, momentum, eps = torch.zeros(64, device=DEV), torch.ones(64, device=DEV), 0.1, 1e-5
    wo, bo = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    with torch.autocast('cuda', dtype=dt):
        out = train_ops.rows3_layer(x, wo, bo, H, True)
    assert out.raw.dtype == dt
    ((out.raw.float() * gy).sum() + (out.affine * ga).sum()).backward()
    wt, bt = w.double().requires_grad_(True), b.double().requires_grad_(True)
    gt, bet = H.weight.detach().double().requires_grad_(True), H.bias.detach().double().requires_grad_(True)
    y = x.double() @ wt.t() + bt
    yr = y + (y.float().to(dt).double() - y).detach()
    mean = yr.mean(0)
    var = (yr * yr).mean(0) - mean * mean
    sc = gt / torch.sqrt(var + 1e-5)
    aff = torch.stack([sc, bet - mean * sc])
    ((y * gy.double()).sum() + (aff * ga.double()).sum()).backward()
    rel = lambda a, c: float((a.detach().double() - c.detach()).abs().max()) / (float(c.detach().abs().max()) + 1e-12)
    assert rel(out.raw.float(), y.detach()) <= 6e-3
    assert rel(out.affine, aff.detach()) <= 2e-4
    assert rel(wo.grad, wt.grad) <= 2e-3 and rel(bo.grad, bt.grad) <= 2e-3
    assert rel(H.weight.grad, gt.grad) <= 2e-3 and rel(H.bias.grad, bet.grad) <= 2e-3
    assert rel(H.running_mean, 0.1 * mean.detach()) <= 2e-4


@pytest.mark.parametrize('dt', DT16)
@pytest.mark.parametrize('nq,p,affine', [(37, 50, True), (5, 64, False), (130, 17, True)])
def test_patch_transform_matches_torch(nq, p, affine, dt):
    from ppsurf_amd import train_ops
    g = torch.Generator().manual_seed(nq + p)
    x = _bf(torch.randn(nq * p, 64, generator=g), dt).to(DEV)
    t = _bf(torch.randn(nq, 64, 64, generator=g) * 0.3, dt).to(DEV)
    aff = torch.stack([torch.randn(64, generator=g) * 0.3 + 1, torch.randn(64, generator=g) * 0.3]).to(DEV) if affine else None
    go = _bf(torch.randn(nq * p, 64, generator=g), dt).to(DEV)
    xo, to_ = x.to(dt).requires_grad_(True), t.to(dt).requires_grad_(True)
    ao = aff.clone().requires_grad_(True) if affine else None
    out = train_ops.patch_transform(train_ops.Act(xo, ao, affine), to_, p)
    (out.float() * go).sum().backward()
    xt, tt = x.double().requires_grad_(True), t.double().requires_grad_(True)
    at = aff.double().requires_grad_(True) if affine else None
    a = torch.relu(xt * at[0] + at[1]) if affine else xt
    a = a + (a.float().to(dt).double() - a).detach()                                     # the operand is 16-bit
    tm = tt + torch.eye(64, device=DEV, dtype=torch.float64)
    tm = tm + (tm.float().to(dt).double() - tm).detach()
    ref = torch.bmm(a.view(nq, p, 64), tm.transpose(1, 2)).reshape(nq * p, 64)
    (ref * go.double()).sum().backward()
    rel = lambda u, v: float((u.detach().double() - v.detach()).abs().max()) / (float(v.detach().abs().max()) + 1e-12)
    assert rel(out.float(), ref.detach()) <= 6e-3
    assert rel(xo.grad.float(), xt.grad) <= 8e-3
    assert rel(to_.grad.float(), tt.grad) <= 8e-3
    if affine:
        assert rel(ao.grad, at.grad) <= 2e-3
