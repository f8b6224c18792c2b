"""The oracle's driver restatements (oracle/driver_oracle.py) against the round-2 fixtures recorded from the reference's own
functions (tests/golden/make_golden_r2.py): sampling_quantized, the predict_step latent loop, the refinement loop; plus the
full-size encoder / network forward and PointNetfeat(P=200) fixtures for the network oracle.  CPU only."""
import random

import numpy as np
import pytest
import torch

from golden_util import load_golden, filled_sd, sd_digest
from oracle import driver_oracle as D
from oracle import ppsurf_oracle as O
from ppsurf_amd.synthetic import make_cloud

from golden.cases_r2 import SAMPLING_CASES, bumpy_field, forward_case


@pytest.mark.parametrize('tag', [c[0] for c in SAMPLING_CASES])
def test_sampling_quantized_equals_reference(tag):
    g = load_golden('sampling')
    gen = dict((c[0], c[1]) for c in SAMPLING_CASES)[tag]
    seed = int(g[tag + '_seed'])
    cloud = gen()
    random.seed(seed)
    torch.manual_seed(seed)
    ids, rounds = D.sampling_quantized_ids(cloud, g[tag + '_ids'].shape[0], return_rounds=True)
    assert np.array_equal(ids, g[tag + '_ids'])                       # same ids in the same order
    assert len(rounds) == int(g[tag + '_rounds'])


def test_sampling_batch_and_n_support_form():
    g = load_golden('sampling')
    both = torch.stack([torch.from_numpy(make_cloud(2500, seed=s).T.copy()) for s in (41, 42)])
    random.seed(77)
    torch.manual_seed(77)
    sup, ids = D.sampling_quantized(both, n_support=300)
    assert np.array_equal(ids.numpy(), g['batch_ids'])
    assert torch.equal(sup, torch.gather(both, 2, ids.unsqueeze(1).expand(2, 3, 300)))
    # shortcut and error branch (poco_data_loader.py:79-82,134)
    same, all_ids = D.sampling_quantized(both, ratio=1.0)
    assert same is both and torch.equal(all_ids[1], torch.arange(2500))
    with pytest.raises(ValueError):
        D.sampling_quantized(both, n_support=3000)


def test_priority_truncation_is_a_subset_of_the_last_round():
    cloud = make_cloud(10000, seed=21)
    random.seed(5)
    rots = [D.draw_round_rotations() for _ in range(12)]
    prio = np.random.default_rng(1).permutation(10000).astype(np.uint32)
    ids, rounds = D.sampling_quantized_ids(cloud, 2500, rotations=rots, priority=prio, return_rounds=True)
    full = np.concatenate(rounds[:-1]) if len(rounds) > 1 else np.empty(0, dtype=np.int64)
    assert len(set(ids.tolist())) == 2500 and set(full.tolist()) <= set(ids.tolist())
    last = set(ids.tolist()) - set(full.tolist())
    assert last <= set(rounds[-1].tolist())
    kept = sorted(rounds[-1].tolist(), key=lambda i: (prio[i], i))[:2500 - full.shape[0]]
    assert last == set(kept)


def test_fkaconv_ids_levels_follow_the_reference():
    """get_fkaconv_ids = four sampling levels + 13 tables (poco_data_loader.py:137-209): same seeds -> same support points."""
    g = load_golden('fkaconv_ids')
    cloud = make_cloud(10000, seed=51)
    random.seed(int(g['seed']))
    torch.manual_seed(int(g['seed']))
    cur, sups = torch.from_numpy(cloud.T.copy()).unsqueeze(0), []
    for _ in range(4):
        cur = D.sampling_quantized(cur, ratio=0.25)[0]
        sups.append(cur)
    for i in range(4):
        assert np.array_equal(sups[i].numpy(), g['support{}'.format(i + 1)])
    t = O.fkaconv_ids_from_supports(torch.from_numpy(cloud.T.copy()).unsqueeze(0), sups)
    for k in ('ids44', 'ids34', 'ids43'):
        assert np.array_equal(t[k].numpy(), g[k])


@pytest.mark.parametrize('tag', ['topup', 'exact', 'small'])
def test_latent_loop_equals_reference(tag):
    g = load_golden('latent_loop')
    n, m, iters, seed = (int(x) for x in g[tag + '_cfg'])
    cloud = torch.from_numpy(make_cloud(n, seed=seed))
    torch.manual_seed(seed)
    trace = []
    lat, cnt = D.latent_loop(cloud, lambda p: D.stub_latent(p, 8), 8, m, iters, trace=trace)
    assert len(trace) == g[tag + '_trace'].shape[0]
    assert all(np.array_equal(a.numpy(), b) for a, b in zip(trace, g[tag + '_trace']))
    assert np.array_equal(cnt.numpy(), g[tag + '_counts']) and np.array_equal(lat.numpy(), g[tag + '_latents'])


def test_refinement_equals_reference():
    g = load_golden('refine')

    def occ(q):
        d = bumpy_field(torch.from_numpy(q))
        p = torch.softmax(torch.stack([d, torch.zeros_like(d)], dim=0).unsqueeze(0), dim=1)
        return (p[:, 0] - p[:, 1]).squeeze(0).numpy()

    out = D.refine_vertices(g['mc_verts'].astype(np.float64), g['volume'], occ, g['step'][()], g['bmin_pad'][()], int(g['refine_iter']), 5000)
    assert np.array_equal(out, g['refined'])
    assert np.array_equal(D.refine_vertices(g['mc_verts'].astype(np.float64), g['volume'], occ, g['step'][()], g['bmin_pad'][()], 0),
                          g['mc_verts'].astype(np.float64) * g['step'][()] + g['bmin_pad'][()])


def test_full_size_network_forward_oracle():
    """FKAConvNetwork(hidden=64) at N=10000 + PPSurfNetwork.forward: the oracle against the reference's outputs."""
    g = load_golden('ppsurf_forward')
    sd = filled_sd('', key='ppsurf')
    assert sd_digest(sd) == str(g['digest'])
    data = forward_case(g)
    with torch.no_grad():
        lat = O.fkaconv_network(sd, 'encoder', data, act='silu', fixed=True)
        logits = O.ppsurf_from_latent(sd, dict(data, latents=lat), k=64)
    np.testing.assert_allclose(lat[0, :, ::25].numpy(), g['latents_sub'], rtol=0, atol=6e-5)       # scale 25: 2.4e-6 relative
    np.testing.assert_allclose(logits.numpy(), g['logits'], rtol=0, atol=5e-5)


def test_pointnet_p200_oracle():
    g = load_golden('pointnet_p200')
    sd = filled_sd('PN_p200.', key='PN_p50.')
    assert sd_digest(sd) == str(g['digest'])
    with torch.no_grad():
        feat, trans2 = O.pointnet_feat(sd, 'PN_p200', torch.from_numpy(g['x']))
    np.testing.assert_allclose(feat.numpy(), g['feat'], rtol=0, atol=2e-5)
    np.testing.assert_allclose(trans2[:2].numpy(), g['trans2'], rtol=0, atol=2e-5)
