"""Dense-layer kernels of the training step (csrc/pps_gemm_train.hip: pps_gemm_nt_16, pps_gemm_tn_16, pps_transpose_cast_pieces) against plain fp32
matrix products of the SAME 16-bit operands -- the kernels accumulate in fp32, so they must agree to fp32 summation order (and to one rounding of
the result where it is stored in 16 bits).  Shapes: every dense layer of the fit step that is not a fused row layer (FKAConv encoder 1x1 convs and
(1,16) convs, source/base/nn.py:438-450,508-554,571,650; head table / fc_value / fc8, poco_model.py:405-417; STN fc layers nn.py:183-188; MLP
nn.py:376-417), plus ragged ones."""
import numpy as np
import pytest
import torch

from ppsurf_amd import train_ops, train_graph as tg

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
# (rows, K, N)
SHAPES = [(100000, 48, 64), (100000, 64, 32), (100000, 512, 32), (25000, 1024, 64), (6250, 2048, 128), (1560, 4096, 256), (390, 8192, 512),
          (390, 512, 1024), (390, 2048, 1024), (1560, 1536, 512), (100000, 192, 64), (100000, 64, 256), (20000, 64, 4096), (20000, 256, 8),
          (20000, 256, 256), (777, 40, 24), (1, 8, 8), (130, 72, 200)]


def _ops(m, k, n, dt, seed=0):
    g = torch.Generator(device=DEV).manual_seed(seed)
    x = torch.randn((m, k), device=DEV, generator=g).to(dt)
    w = (torch.randn((n, k), device=DEV, generator=g) / np.sqrt(k)).to(dt)
    b = torch.randn((n,), device=DEV, generator=g)
    return x, w, b


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('m,k,n', SHAPES)
def test_gemm_nt_forward_and_input_gradient(m, k, n, dt):
    x, w, b = _ops(m, k, n, dt)
    ref = x.float() @ w.float().t() + b
    y32 = train_ops.gemm_nt(x, w, b, out_f32=True)
    scale = float(ref.abs().max())
    assert float((y32 - ref).abs().max()) <= 2e-5 * scale * max(1.0, np.sqrt(k) / 8)         # fp32 accumulation, another summation order
    y = train_ops.gemm_nt(x, w, b)
    assert y.dtype == dt and torch.equal(y, y32.to(dt))                                          # the 16-bit result is the fp32 result rounded once
    if n % 8 == 0:
        g16 = y                                                                                    # any [m, n] 16-bit tensor serves as a gradient
        wt = w.t().contiguous()                                                                    # [k, n] image
        dx = train_ops.gemm_nt(g16, wt, None, out_f32=True)
        refdx = g16.float() @ w.float()
        assert float((dx - refdx).abs().max()) <= 2e-5 * float(refdx.abs().max()) * max(1.0, np.sqrt(n) / 8)


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('m,k,n', [s for s in SHAPES if s[2] % 8 == 0])
def test_gemm_tn_weight_gradient(m, k, n, dt):
    x, w, _ = _ops(m, k, n, dt, seed=1)
    g = (torch.randn((m, n), device=DEV) * 0.1).to(dt)
    dw = train_ops.gemm_tn(g, x)
    ref = (g.double().t() @ x.double())
    assert dw.dtype == torch.float32 and dw.shape == (n, k)
    err = float((dw.double() - ref).abs().max())
    assert err <= 1e-5 * float(ref.abs().max()) + 1e-6 * np.sqrt(m), err
    assert torch.equal(train_ops.gemm_tn(g, x), dw)                                               # slab partials summed in a fixed order: reproducible


def test_gemm_operand_pitches_and_transposed_images():
    """Row pitches larger than the row (column slices of a wider tensor) are read in place; pps_transpose_cast_pieces makes the [K, N] images of all
    matrix-shaped parameters of a module in one launch, equal to a plain cast + transpose."""
    big = torch.randn((3000, 320), device=DEV).to(torch.bfloat16)
    x = big[:, 64:64 + 128]                                       # pitch 320, offset 128 bytes
    w = torch.randn((40, 128), device=DEV).to(torch.bfloat16)
    y = train_ops.gemm_nt(x, w, None, out_f32=True)
    assert float((y - x.float() @ w.float().t()).abs().max()) < 1e-3
    g = big[:, :48]
    dw = train_ops.gemm_tn(g, x)
    assert float((dw - g.float().t() @ x.float()).abs().max()) < 2e-2
    net = torch.nn.Sequential(torch.nn.Linear(48, 64), torch.nn.Conv2d(24, 40, (1, 16), bias=False), torch.nn.Linear(8192, 512), torch.nn.Linear(256, 2)).to(DEV)
    for dt in (torch.bfloat16, torch.float16):
        with torch.autocast('cuda', dtype=dt):
            tg.prepare_shadows(net, dt)
            for p in net.parameters():
                if p.dim() >= 2:
                    img = tg._bf16_t_of(p)
                    assert img is not None and img.dtype == dt and torch.equal(img, p.detach().reshape(p.shape[0], -1).t().to(dt))
                    assert torch.equal(tg._bf16_of(p), p.detach().to(dt))
        tg.release_step_caches()


@pytest.mark.parametrize('m,k,n,bias', [(5000, 64, 32, True), (3000, 256, 2, True), (700, 48, 64, False), (40000, 512, 32, False)])
def test_rows_linear_autograd_uses_the_kernels_and_matches_float64(m, k, n, bias, monkeypatch):
    """train_graph.rows_linear under bf16 autocast: forward, dx, dW, db through the hand-written kernels (the library entry points are made to fail),
    against float64 autograd of the same bf16-rounded operands."""
    import torch.nn.functional as F
    lin = torch.nn.Linear(k, n, bias=bias).to(DEV)
    x = torch.randn((m, k), device=DEV).to(torch.bfloat16).requires_grad_(True)
    gy = torch.randn((m, n), device=DEV).to(torch.bfloat16)

    def boom(*a, **kw):
        raise AssertionError('library GEMM called')
    with torch.autocast('cuda', dtype=torch.bfloat16):
        tg.prepare_shadows(lin, torch.bfloat16)
        monkeypatch.setattr(F, 'linear', boom)
        monkeypatch.setattr(torch, 'bmm', boom)
        y = tg.dense(lin, x)
        y.backward(gy)
    monkeypatch.undo()
    tg.release_step_caches()
    w64 = lin.weight.detach().to(torch.bfloat16).double().requires_grad_(True)
    b64 = lin.bias.detach().double().requires_grad_(True) if bias else None
    x64 = x.detach().double().requires_grad_(True)
    y64 = x64 @ w64.t() + (b64 if bias else 0)
    y64.backward(gy.double())
    assert y.dtype == torch.bfloat16 and float((y.double() - y64).abs().max()) <= 2 ** -7 * float(y64.abs().max())
    assert float((x.grad.double() - x64.grad).abs().max()) <= 2 ** -7 * float(x64.grad.abs().max())
    assert float((lin.weight.grad.double() - w64.grad).abs().max()) <= 1e-4 * float(w64.grad.abs().max()) + 1e-4
    if bias:
        assert float((lin.bias.grad.double() - b64.grad).abs().max()) <= 1e-4 * float(b64.grad.abs().max()) + 1e-4
