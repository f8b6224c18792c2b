"""Host-side driver logic of the PRODUCT (ppsurf_amd) against fixtures recorded from the reference's own functions
(tests/golden/make_golden_r2.py).  Everything here runs the product's code on CPU tensors with the network replaced by a stub,
so no GPU is needed:

  spatial.sampling_quantized (torch-op path)      == reference sampling_quantized, same `random` / torch seeds
  PocoModel.encode_latents (latent_batch 1 and 10) == reference predict_step latent loop, same torch seed
  reconstruct.refine_vertices                      == reference refinement inside export_mesh_and_refine_vertices_region_growing_v3
"""
import contextlib
import io
import random

import numpy as np
import pytest
import torch

from golden_util import load_golden
from golden.cases_r2 import SAMPLING_CASES, bumpy_field
from ppsurf_amd import spatial, reconstruct
from ppsurf_amd.synthetic import make_cloud


def stub_latent(pts_cf, c=8):
    """Same formula as oracle.driver_oracle.stub_latent (the fixture's stand-in for network.get_latent); restated here so that this
    product-side test does not depend on the oracle."""
    freq = torch.arange(1, c + 1, dtype=torch.float32).view(1, c, 1)
    centre = pts_cf.mean(dim=2, keepdim=True)
    return torch.sin(freq * pts_cf[:, 0:1]) + torch.cos(freq * pts_cf[:, 1:2]) * pts_cf[:, 2:3] + centre.sum(dim=1, keepdim=True)


@pytest.mark.parametrize('tag', [c[0] for c in SAMPLING_CASES])
def test_sampling_quantized_torch_path_equals_reference(tag):
    g = load_golden('sampling')
    gen = dict((c[0], c[1]) for c in SAMPLING_CASES)[tag]
    seed = int(g[tag + '_seed'])
    pts = torch.from_numpy(gen().T.copy()).unsqueeze(0)
    random.seed(seed)
    torch.manual_seed(seed)
    sup, ids = spatial.sampling_quantized(pts, ratio=0.25)
    assert np.array_equal(ids[0].numpy(), g[tag + '_ids'])
    assert torch.equal(sup[0], pts[0][:, ids[0]])


def test_sampling_quantized_batch_n_support_and_errors():
    g = load_golden('sampling')
    both = torch.stack([torch.from_numpy(make_cloud(2500, seed=s).T.copy()) for s in (41, 42)])
    random.seed(77)
    torch.manual_seed(77)
    assert np.array_equal(spatial.sampling_quantized(both, n_support=300)[1].numpy(), g['batch_ids'])
    same, ids = spatial.sampling_quantized(both, ratio=1.0)
    assert same is both and torch.equal(ids[0], torch.arange(2500))
    with pytest.raises(ValueError):
        spatial.sampling_quantized(both, n_support=2501)
    sup = torch.zeros(2, 3, 5)
    assert spatial.sampling_quantized(both, support_points=sup, support_points_ids='x') == (sup, 'x')


def _model(n_latent, m, iters, batch):
    from ppsurf_amd.lightning_api import PocoModel
    with contextlib.redirect_stdout(io.StringIO()):
        model = PocoModel(output_names=['x'], in_channels=3, out_channels=2, k=64, lambda_l1=0.0, debug=False, in_file='x.xyz',
                          results_dir='/tmp/x', padding_factor=0.05, name='x', network_latent_size=n_latent, gen_subsample_manifold_iter=iters,
                          gen_subsample_manifold=m, gen_resolution_global=33, rec_batch_size=1000, gen_refine_iter=0, workers=0)
    model.latent_batch = batch
    return model


@pytest.mark.parametrize('batch', [1, 10, 3, 25])
@pytest.mark.parametrize('tag', ['topup', 'exact', 'small'])
def test_latent_loop_equals_reference(tag, batch):
    """Same torch seed -> the same subsets in the same order, the same coverage counts and the same averaged latents as the
    reference's loop (poco_model.py:203-236), for the pass-by-pass form and for subsets drawn ahead and encoded as one batch."""
    g = load_golden('latent_loop')
    n, m, iters, seed = (int(x) for x in g[tag + '_cfg'])
    cloud = torch.from_numpy(make_cloud(n, seed=seed))
    model = _model(8, m, iters, batch)
    trace, sizes = [], []

    def encode(pts_cf, subsets):
        sizes.append(len(subsets))
        return torch.stack([stub_latent(pts_cf[:, ids].unsqueeze(0), 8)[0].t() for ids in subsets])

    torch.manual_seed(seed)
    lat = model.encode_latents(cloud.t().contiguous(), encode_subsets=encode, trace=trace)
    ref = g[tag + '_trace']
    assert len(trace) == ref.shape[0] and all(np.array_equal(a.numpy(), b) for a, b in zip(trace, ref))
    counts = np.zeros(n, dtype=np.float32)
    for ids in trace:
        counts[np.unique(ids.numpy())] += 1
    assert np.array_equal(counts, g[tag + '_counts'])
    np.testing.assert_allclose(lat.numpy(), g[tag + '_latents'], rtol=0, atol=1e-6)
    assert max(sizes) <= batch and (batch == 1 or n < m or max(sizes) > 1)


def test_refine_vertices_equals_reference():
    g = load_golden('refine')

    def occ(q):                                            # what _predict_from_latent computes from the stub network's logits
        d = bumpy_field(q)
        p = torch.softmax(torch.stack([d, torch.zeros_like(d)], dim=0).unsqueeze(0), dim=1)
        return (p[:, 0] - p[:, 1]).squeeze(0)

    verts = torch.from_numpy(g['mc_verts'].astype(np.float64))
    out = reconstruct.refine_vertices(occ, verts, torch.from_numpy(g['volume']), g['step'][()], g['bmin_pad'][()], int(g['refine_iter']))
    assert out.dtype == torch.float64 and np.array_equal(out.numpy(), g['refined'])
    out0 = reconstruct.refine_vertices(occ, verts, torch.from_numpy(g['volume']), g['step'][()], g['bmin_pad'][()], 0)
    assert np.array_equal(out0.numpy(), g['mc_verts'].astype(np.float64) * g['step'][()] + g['bmin_pad'][()])


def test_marching_cubes_twins_agree_on_the_fixture_volume():
    """The device Marching Cubes + clean-up (torch ops) against the numpy twin that produced the fixture's vertices."""
    from ppsurf_amd import mcubes
    g = load_golden('refine')
    v, f = mcubes.marching_cubes_torch(torch.from_numpy(g['volume']), 0.0)
    v, f = mcubes.clean_mesh_torch(v, f, min_component_faces=6)
    assert v.shape[0] == g['mc_verts'].shape[0] and f.shape[0] == g['mc_faces'].shape[0]
    a = np.unique(np.round(v.numpy().astype(np.float32), 5), axis=0)
    b = np.unique(np.round(g['mc_verts'], 5), axis=0)
    np.testing.assert_allclose(a, b, rtol=0, atol=2e-5)


@pytest.mark.parametrize('n,m', [(1000, 100), (1050, 100), (130, 100), (60, 100)])
def test_round_drawn_from_one_permutation_keeps_the_coverage_contract(n, m):
    """PocoModel._draw_round (device / shared-generator streams): the remaining subsets of a coverage round from ONE permutation -- disjoint pieces
    of m valid points, the last one topped up from all points -- and the latent loop built on it covers every point like the pass-by-pass loop
    does (poco_model.py:203-236): counts >= the number of rounds, and == it when m divides n."""
    from ppsurf_amd.lightning_api import PocoModel
    gen = torch.Generator(device='cpu')
    gen.manual_seed(11)
    covered = torch.zeros(n)
    covered[:n // 3] = 1.0                                                   # a third of the cloud already has this round's coverage
    subsets = PocoModel._draw_round(covered, 0, m, gen)
    if n < m:
        assert len(subsets) == 1 and torch.equal(subsets[0], torch.arange(n))
    else:
        valid = n - n // 3
        assert len(subsets) == -(-valid // m) and all(s.shape[0] == m for s in subsets)
        body = torch.cat(subsets)[:valid]
        assert torch.equal(torch.sort(body).values, torch.arange(n // 3, n))                     # every valid point exactly once, no other
        assert not torch.equal(body, torch.arange(n // 3, n))                                    # ... in a shuffled order
    assert PocoModel._draw_round(torch.ones(n), 0, m, gen) == []
    # the loop on per-round drawing (the default on a GPU and in a query-sharded multi-rank run)
    model = _model(8, m, 3, 7)
    trace = []
    cloud = torch.from_numpy(make_cloud(n, seed=3))
    model.latent_per_round = True
    lat = model.encode_latents(cloud.t().contiguous(), trace=trace, encode_subsets=lambda pts_cf, subs: torch.stack([pts_cf[:, i].t().repeat(1, 3)[:, :8] for i in subs]))
    counts = torch.zeros(n)
    for ids in trace:
        counts[torch.unique(ids)] += 1
    assert float(counts.min()) >= 3 and (n % m != 0 or n < m or float(counts.max()) == 3)
    assert torch.isfinite(lat).all()
