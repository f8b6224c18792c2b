"""The driver pieces ON THE DEVICE against the fixtures recorded from the reference's own functions (tests/golden/make_golden_r2.py) and
against the pinned oracle: what tests/test_driver_parity_cpu.py checks on CPU tensors, here on the GPU tensors the product really uses."""
import contextlib
import io
import random

import numpy as np
import pytest
import torch

from golden_util import load_golden
from golden.cases_r2 import bumpy_field
from ppsurf_amd import reconstruct, spatial
from ppsurf_amd.synthetic import make_cloud

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _stub_latent(pts_cf, c=8):
    freq = torch.arange(1, c + 1, dtype=torch.float32, device=pts_cf.device).view(1, c, 1)
    centre = pts_cf.mean(dim=2, keepdim=True)
    return torch.sin(freq * pts_cf[:, 0:1]) + torch.cos(freq * pts_cf[:, 1:2]) * pts_cf[:, 2:3] + centre.sum(dim=1, keepdim=True)


@pytest.mark.parametrize('batch', [1, 3, 10, 25])
@pytest.mark.parametrize('tag', ['exact', 'topup', 'small'])
def test_latent_loop_on_the_device_follows_the_reference_stream(tag, batch):
    """latent_rng='reference' on the GPU: the subset permutations AND the top-up permutation (poco_model.py:217-219) come from torch's CPU
    generator, the stream the reference's recorded run consumed, so the subsets, counts and latents of the reference's loop
    (poco_model.py:203-236) are reproduced exactly on device tensors -- pass by pass and with subsets drawn ahead -- for a cloud that is a
    multiple of the subset size ('exact'), one that needs the top-up branch ('topup') and one smaller than a subset ('small')."""
    from ppsurf_amd.lightning_api import PocoModel
    g = load_golden('latent_loop')
    n, m, iters, seed = (int(x) for x in g[tag + '_cfg'])
    with contextlib.redirect_stdout(io.StringIO()):
        model = PocoModel(output_names=['x'], in_channels=3, out_channels=2, k=64, lambda_l1=0.0, debug=False, in_file='x.xyz', results_dir='/tmp/x',
                          padding_factor=0.05, name='x', network_latent_size=8, gen_subsample_manifold_iter=iters, gen_subsample_manifold=m,
                          gen_resolution_global=33, rec_batch_size=1000, gen_refine_iter=0, workers=0)
    model.latent_batch, model.latent_rng = batch, 'reference'
    cloud = torch.from_numpy(make_cloud(n, seed=seed)).to(DEV)
    trace = []
    torch.manual_seed(seed)
    lat = model.encode_latents(cloud.t().contiguous(), trace=trace,
                               encode_subsets=lambda pts_cf, subsets: torch.stack([_stub_latent(pts_cf[:, i].unsqueeze(0), 8)[0].t() for i in subsets]))
    assert lat.is_cuda and len(trace) == g[tag + '_trace'].shape[0]
    assert all(a.is_cuda and np.array_equal(a.cpu().numpy(), b) for a, b in zip(trace, g[tag + '_trace']))
    counts = np.zeros(n, dtype=np.float32)
    for ids in trace:
        counts[np.unique(ids.cpu().numpy())] += 1
    assert np.array_equal(counts, g[tag + '_counts'])
    np.testing.assert_allclose(lat.cpu().numpy(), g[tag + '_latents'], rtol=0, atol=2e-6)        # sin / cos of the device vs the host


def test_default_device_rng_latent_loop_covers_every_point_equally():
    """The default (device permutations): not the reference's random stream, but the same coverage contract -- every point is encoded
    exactly gen_subsample_manifold_iter times when N is a multiple of the subset size, and the averaged latents are those of the stub."""
    from ppsurf_amd.lightning_api import PocoModel
    with contextlib.redirect_stdout(io.StringIO()):
        model = PocoModel(output_names=['x'], in_channels=3, out_channels=2, k=64, lambda_l1=0.0, debug=False, in_file='x.xyz', results_dir='/tmp/x',
                          padding_factor=0.05, name='x', network_latent_size=8, gen_subsample_manifold_iter=3, gen_subsample_manifold=1000,
                          gen_resolution_global=33, rec_batch_size=1000, gen_refine_iter=0, workers=0)
    cloud = torch.from_numpy(make_cloud(4000, seed=3)).to(DEV)
    trace = []
    model.encode_latents(cloud.t().contiguous(), trace=trace, encode_subsets=lambda p, subs: torch.zeros((len(subs), 1000, 8), device=DEV))
    counts = torch.zeros(4000, device=DEV)
    for ids in trace:
        assert ids.shape[0] == 1000 and ids.unique().shape[0] == 1000
        counts[ids] += 1
    assert len(trace) == 12 and bool((counts == 3).all())


@pytest.mark.parametrize('n', [4300, 2500, 700])
def test_default_device_rng_latent_loop_with_top_up_and_small_clouds(n):
    """The device stream when N is NOT a multiple of the subset size (the last subset of a round is topped up from all points, poco_model.py:217-229)
    and when the cloud is smaller than a subset: subsets of the right size, every point covered at least gen_subsample_manifold_iter times, the
    pieces of a round disjoint before the top-up -- for batches that span rounds."""
    from ppsurf_amd.lightning_api import PocoModel
    m, iters = 1000, 3
    with contextlib.redirect_stdout(io.StringIO()):
        model = PocoModel(output_names=['x'], in_channels=3, out_channels=2, k=64, lambda_l1=0.0, debug=False, in_file='x.xyz', results_dir='/tmp/x',
                          padding_factor=0.05, name='x', network_latent_size=8, gen_subsample_manifold_iter=iters, gen_subsample_manifold=m,
                          gen_resolution_global=33, rec_batch_size=1000, gen_refine_iter=0, workers=0)
    model.latent_batch = 4
    cloud = torch.from_numpy(make_cloud(n, seed=3)).to(DEV)
    trace = []
    lat = model.encode_latents(cloud.t().contiguous(), trace=trace,
                               encode_subsets=lambda p, subs: torch.stack([_stub_latent(p[:, i].unsqueeze(0), 8)[0].t() for i in subs]))
    counts = torch.zeros(n, device=DEV)
    for ids in trace:
        assert ids.shape[0] == min(n, m)
        counts[ids.unique()] += 1
    assert float(counts.min()) >= iters and bool(torch.isfinite(lat).all())
    if n >= m:
        per_round = -(-n // m)
        assert len(trace) <= iters * per_round                                                   # top-ups may finish a round early, never late
        first = torch.cat(trace[:n // m])                                                        # the full pieces of round 0: disjoint valid points
        assert first.unique().shape[0] == first.shape[0]
    else:
        assert len(trace) == iters and all(torch.equal(ids, torch.arange(n, device=DEV)) for ids in trace)


def test_refinement_on_the_device_equals_the_reference():
    g = load_golden('refine')

    def occ(q):
        d = bumpy_field(q)
        p = torch.softmax(torch.stack([d, torch.zeros_like(d)], dim=0).unsqueeze(0), dim=1)
        return (p[:, 0] - p[:, 1]).squeeze(0)

    verts = torch.from_numpy(g['mc_verts'].astype(np.float64)).to(DEV)
    out = reconstruct.refine_vertices(occ, verts, torch.from_numpy(g['volume']).to(DEV), g['step'][()], g['bmin_pad'][()], int(g['refine_iter']))
    # the analytic field (sin / cos / softmax) is evaluated by the device's math library here and by the host's in the fixture: the ten
    # bisection rounds amplify a last-bit difference of the field by at most one bisection interval, 2^-10 of a voxel
    step = float(g['step'])
    assert float(np.abs(out.cpu().numpy() - g['refined']).max()) <= step * 2.0 ** -9
    assert np.median(np.abs(out.cpu().numpy() - g['refined'])) == 0.0


@pytest.mark.parametrize('n,target', [(20000, 5000), (60000, 15000), (10241, 2560), (250000, 10000)])
def test_large_cloud_sampling_path_on_the_device_equals_the_oracle(n, target):
    """Clouds beyond the LDS tables of the one-launch sampling kernel (10240 points) go through the workspace kernel
    (pps_voxel_sample_large_f32) -- no cloud size leaves the device for the torch-op loop: the same selection as the pinned oracle given the
    same rotations and truncation priorities, as a set and ascending."""
    from oracle import driver_oracle as D
    cloud = make_cloud(n, seed=77)
    random.seed(5)
    rots = spatial.draw_rotations()
    prio = torch.from_numpy(np.random.default_rng(2).permutation(n).astype(np.int64))
    _, ids = spatial.sampling_quantized(torch.from_numpy(cloud.T.copy()).unsqueeze(0).to(DEV), n_support=target, _rotations=rots, _priority=prio)
    ref = D.sampling_quantized_ids(cloud, target, rotations=[list(r.numpy()) for r in rots], priority=prio.numpy().astype(np.uint32))
    got = ids[0].cpu().numpy()
    assert got.shape == (target,) and np.all(np.diff(got) > 0)
    assert np.array_equal(got, np.sort(ref))
