"""Fused dense layer of the training step (pps_rows_train.hip through train_ops.rows_layer) against the same layer written with torch
ops: forward output, BatchNorm affine + running statistics, and every gradient.  The torch twin uses the operands the kernels use
(inputs, weights and the row gradient G rounded to bf16, fp32 accumulation), so the tolerances are those of bf16 STORAGE of the
results (2^-8 relative) plus the accumulation order, not of a different algorithm."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _bf(t):
    return t.to(torch.bfloat16).float()


class _RoundGrad(torch.autograd.Function):
    """identity whose gradient is rounded to bf16: the row gradient G = gy + gS + 2 y gQ enters the kernels' products (and db) as a bf16 MFMA
    operand -- like the bf16 grad_input of the reference's batch-norm backward under autocast.  Adding the per-channel constant gS to
    bf16 values and rounding again is NOT unbiased within a binade, so an unrounded twin differs systematically in the sums over rows."""

    @staticmethod
    def forward(ctx, t):
        return t.clone()

    @staticmethod
    def backward(ctx, g):
        return g.float().to(torch.bfloat16).to(g.dtype)


def _twin(x, aff, relu, w, b, gamma, beta, eps):
    """float64 twin on the bf16-rounded operands.  Returns y (unrounded), out_affine or None."""
    a = x.double()
    if aff is not None:
        a = a * aff[0].double() + aff[1].double()
        if relu:
            a = torch.relu(a)
    a = a.float().to(torch.bfloat16).double() if aff is not None else a        # the MFMA operand is bf16
    y = a @ w.to(torch.bfloat16).double().t()
    if b is not None:
        y = y + b.double()
    y = _RoundGrad.apply(y)
    if gamma is None:
        return y, None
    yr = y.float().to(torch.bfloat16).double()                                # statistics are those of the stored y
    yr = y + (yr - y).detach()
    m = yr.shape[0]
    mean = yr.sum(0) / m
    var = (yr * yr).sum(0) / m - mean * mean
    sc = gamma.double() / torch.sqrt(var + eps)
    return y, torch.stack([sc, beta.double() - mean * sc])


CASES = [(5000, 64, 64, True, True, True), (3001, 64, 128, True, True, True), (4100, 128, 256, True, True, True),
         (2500, 256, 256, False, False, True), (1999, 256, 64, True, False, True), (70, 64, 64, False, True, False),
         (40000, 128, 128, True, True, True), (33, 256, 128, True, True, True)]


@pytest.mark.parametrize('rows,cin,cout,affine,bn,bias', CASES)
def test_rows_layer_matches_the_torch_twin(rows, cin, cout, affine, bn, bias):
    from ppsurf_amd import train_ops
    g = torch.Generator().manual_seed(rows + cin + cout)
    rnd = lambda *s: torch.randn(*s, generator=g)
    x = _bf(rnd(rows, cin) * 1.5 + 0.3).to(DEV)
    w = (rnd(cout, cin) / cin ** 0.5).to(DEV)
    b = (rnd(cout) * 0.2).to(DEV) if bias else None
    aff = torch.stack([rnd(cin) * 0.3 + 1.0, rnd(cin) * 0.4]).to(DEV) if affine else None
    gamma = (rnd(cout) * 0.2 + 1.0).to(DEV) if bn else None
    beta = (rnd(cout) * 0.3).to(DEV) if bn else None
    gy = _bf(rnd(rows, cout)).to(DEV)
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
    xo = x.to(torch.bfloat16).requires_grad_(True)
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
    wq = wt + (wt.to(torch.bfloat16).float() - wt).detach()
    y, oa = _twin(xt, at, affine, wq, bt, gt, bet, eps)
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
        yr = y.detach().float().to(torch.bfloat16).double()
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
