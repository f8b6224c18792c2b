"""GPU parity of the FKAConv encoder kernels (C ABI) vs golden fixtures produced by the reference's own modules."""
import numpy as np
import pytest
import torch

from golden_util import load_golden, filled_sd
from ppsurf_amd.encoder import FKAConvParams, ResidualBlockParams, EncoderPlan, LinearParams, gather_max

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
TOL = dict(rtol=0, atol=1e-4)           # the north star's absolute bar, also on the layer / block fixtures


def pm(a):
    """[C,N] channel-first numpy -> point-major device tensor [N,C]."""
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a).T)).to(DEV)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def test_fkaconv_layer_relu_silu_and_k1():
    g = load_golden('fkaconv_layer')
    for act in ('relu', 'silu'):
        p = 'L_{}'.format(act)
        layer = FKAConvParams(filled_sd(p + '.'), p, DEV, act)
        for b in range(2):
            out = layer(pm(g['x'][b]), pm(g['pts'][b]), pm(g['sup'][b]), dev(g['ids'][b]))
            np.testing.assert_allclose(out.cpu().numpy().T, g['out_' + act][b], **TOL)
            out1 = layer(pm(g['xs'][b]), pm(g['sup'][b]), pm(g['pts'][b]), dev(g['ids1'][b]))     # K == 1
            np.testing.assert_allclose(out1.cpu().numpy().T, g['out_k1_' + act][b], **TOL)


def test_residual_block_same_and_downsampling():
    g = load_golden('residual_block')
    same = ResidualBlockParams(filled_sd('RB_same.'), 'RB_same', DEV, 'silu')
    down = ResidualBlockParams(filled_sd('RB_down.'), 'RB_down', DEV, 'silu')
    for b in range(2):
        x, pts, sup = pm(g['x'][b]), pm(g['pts'][b]), pm(g['sup'][b])
        np.testing.assert_allclose(same(x, pts, pts, dev(g['ids_same'][b])).cpu().numpy().T, g['out_same'][b], **TOL)
        np.testing.assert_allclose(down(x, pts, sup, dev(g['ids_down'][b])).cpu().numpy().T, g['out_down'][b], **TOL)


def test_rows_linear_and_gather_max_primitives():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((300, 37)).astype(np.float32)
    y = rng.standard_normal((90, 21)).astype(np.float32)
    idx1 = rng.integers(0, 300, 500)
    idx2 = rng.integers(0, 90, 500)
    w = rng.standard_normal((45, 58)).astype(np.float32)
    b = rng.standard_normal(45).astype(np.float32)
    res = rng.standard_normal((500, 45)).astype(np.float32)
    lin = LinearParams({'L.weight': w[:, :, None], 'L.bias': b}, 'L', DEV)
    out = lin(dev(x), idx1=dev(idx1), in2=dev(y), idx2=dev(idx2), residual=dev(res), relu=True)
    ref = np.maximum(np.concatenate([x[idx1], y[idx2]], axis=1).astype(np.float64) @ w.T.astype(np.float64) + b + res, 0)
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-5, atol=1e-5)
    ids = rng.integers(0, 300, (77, 9))
    assert np.array_equal(gather_max(dev(x), dev(ids)).cpu().numpy(), x[ids].max(axis=1))


@pytest.mark.parametrize('tag', ['small', 'mid'])
def test_fkaconv_network_hidden8(tag):
    g = load_golden('fkaconv_network')
    data = {k[len(tag) + 1:]: v for k, v in g.items() if k.startswith(tag + '_') and '_out_' not in k}
    pts = pm(data['pts'][0])
    sups = [pm(data['support{}'.format(i)][0]) for i in (1, 2, 3, 4)]
    ids = {}
    for k, v in data.items():
        if k.startswith('ids'):
            t = dev(v[0])
            ids[k] = t.reshape(-1) if k in ('ids43', 'ids32', 'ids21', 'ids10') else t
    for name, act, fixed in (('silu_fixed', 'silu', True), ('relu_poco', 'relu', False)):
        p = 'ENC_{}'.format(name)
        plan = EncoderPlan(filled_sd(p + '.'), DEV, prefix=p, act=act, fixed=fixed)
        out = plan.forward(pts, sups, ids).cpu().numpy().T            # [C,N]
        if tag == 'mid':
            out = out[:, ::7]
        np.testing.assert_allclose(out, g['{}_out_{}'.format(tag, name)][0], rtol=0, atol=1e-4)
