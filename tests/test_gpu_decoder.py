"""GPU parity of the fused occupancy decoder (C ABI) vs the golden fixture produced by the reference and vs the oracle.
Tolerance: 1e-4 absolute on the fp32 logits (BASELINE.json north_star)."""
import numpy as np
import pytest
import torch

from golden_util import load_golden, filled_sd
from oracle import ppsurf_oracle as O
from ppsurf_amd import ops
from ppsurf_amd.decoder import DecoderPlan
from ppsurf_amd.synthetic import make_cloud, make_band_queries, make_latents
import emulate

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
_plans = {}
DTYPES = ['f32', 'f16x3']          # both decoder dtypes are held to the same 1e-4 bar on the same cases


def plan(dtype='f32'):
    if dtype not in _plans:
        _plans[dtype] = DecoderPlan(filled_sd('', key='ppsurf'), DEV, dtype=dtype)
    return _plans[dtype]


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def test_point_table_both_layouts():
    pl = plan()
    lat = make_latents(256, 1000, seed=3)[0]                      # [256,N] channel-first like data['latents'][0]
    w = emulate.unpack_dense(pl.w['g_w'].cpu().numpy(), 256, 256)
    ref = lat.T.astype(np.float64) @ w.T + pl.w['g_b'].cpu().numpy()
    g_cf = pl.point_table(dev(lat)).cpu().numpy()
    g_pm = pl.point_table(dev(lat.T.copy()).t()).cpu().numpy()    # transposed view of point-major storage
    np.testing.assert_allclose(g_cf, ref, rtol=1e-5, atol=1e-5)
    assert np.array_equal(g_cf, g_pm)


def test_decoder_stages_vs_emulation():
    """Each kernel on its own against the float64 replay of the packed weights (localises a failure)."""
    import ctypes
    from ppsurf_amd import _lib
    pl = plan()
    rng = np.random.default_rng(5)
    cloud = make_cloud(3000, seed=8)
    qry = make_band_queries(cloud, 203, resolution=65, seed=2)     # odd count: exercises the tile tails
    ids = O.knn_point_major(cloud, qry, 64)
    patches = O.normalize_patches(cloud[ids[:, :50]], qry)
    lat = make_latents(256, cloud.shape[0], seed=9)[0]
    w = {k: v.cpu().numpy() for k, v in pl.w.items()}
    ref_logits, ref_trans2 = emulate.decode(w, lat, cloud, qry, ids, patches)
    table = pl.point_table(dev(lat))
    logits, occ = pl.decode(table, dev(cloud), dev(qry), dev(ids), dev(patches))
    torch.cuda.synchronize()
    np.testing.assert_allclose(pl.intermediates(203)['trans2'].cpu().numpy().reshape(-1, 64, 64), ref_trans2, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(logits.cpu().numpy(), ref_logits, rtol=1e-4, atol=1e-4)
    ref_occ = np.tanh((ref_logits[:, 0] - ref_logits[:, 1]) / 2)
    np.testing.assert_allclose(occ.cpu().numpy(), ref_occ, rtol=1e-4, atol=1e-4)


def test_decoder_matches_reference_golden():
    g = load_golden('ppsurf_from_latent')
    pl = plan()
    cloud, qry = g['cloud'], g['query']
    pts = dev(cloud)
    idx = ops.knn_point_major(pts, dev(qry), 64)
    assert np.array_equal(idx.cpu().numpy(), g['proj_ids'][0])
    patches = ops.patch_normalize(pts, dev(qry), idx, 50)
    np.testing.assert_allclose(patches.cpu().numpy(), g['patches'], rtol=1e-5, atol=1e-6)
    table = pl.point_table(dev(make_latents(256, cloud.shape[0], 77)[0]))
    logits, occ = pl.decode(table, pts, dev(qry), idx, patches)
    np.testing.assert_allclose(logits.cpu().numpy(), g['logits'][0].T, rtol=0, atol=1e-4)
    np.testing.assert_allclose(occ.cpu().numpy(), g['occ'], rtol=0, atol=1e-4)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('p,k,q', [(50, 64, 1), (10, 64, 77), (25, 16, 130), (50, 64, 1031), (18, 64, 45), (20, 64, 133), (24, 64, 64), (16, 64, 9)])
def test_decoder_vs_oracle_shapes(p, k, q, dtype):
    """P in the ablation set of configs/ppsurf_*nn.yaml, small k (clamped kNN), ragged query counts."""
    sd = filled_sd('', key='ppsurf')
    pl = plan(dtype)
    n = 1500 if k == 64 else k                                   # k clamps to the number of points (poco_utils.py:259-260)
    cloud = make_cloud(max(n, p), seed=p + k)[:max(n, p)]
    cloud_lat = cloud[:n] if k != 64 else cloud
    qry = make_band_queries(cloud, q, resolution=33, seed=q)
    ids = O.knn_point_major(cloud_lat, qry, min(k, cloud_lat.shape[0]))
    pid = O.knn_point_major(cloud, qry, p)
    patches = O.normalize_patches(cloud[pid], qry)
    lat = make_latents(256, cloud_lat.shape[0], seed=q)
    data = {'latents': torch.from_numpy(lat), 'pts': torch.from_numpy(cloud_lat.T.copy()).unsqueeze(0),
            'pts_query': torch.from_numpy(qry).unsqueeze(0), 'pts_local_ps': torch.from_numpy(patches).unsqueeze(0)}
    ref = O.ppsurf_from_latent(sd, data, k=k)[0].T.numpy()
    table = pl.point_table(dev(lat[0]))
    logits, _ = pl.decode(table, dev(cloud_lat), dev(qry), dev(ids), dev(patches.astype(np.float32)))
    np.testing.assert_allclose(logits.cpu().numpy(), ref, rtol=0, atol=1e-4)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('p,q', [(50, 203), (18, 45), (20, 133), (24, 64), (50, 8), (50, 1), (100, 37), (200, 21)])
def test_packed_pointnet_tiles_match_oracle(p, q, dtype, monkeypatch):
    """Left-over rows of 16/LO queries share one tile (LO = P % 16 in {2,4,8}); forced on for small query counts."""
    monkeypatch.setenv('PPS_PN_FORCE_PACK', '1')
    sd = filled_sd('', key='ppsurf')
    pl = plan(dtype)
    cloud = make_cloud(1500, seed=p)
    qry = make_band_queries(cloud, q, resolution=33, seed=q)
    ids = O.knn_point_major(cloud, qry, 64)
    patches = O.normalize_patches(cloud[O.knn_point_major(cloud, qry, p)], qry).astype(np.float32)
    lat = make_latents(256, cloud.shape[0], seed=q)
    data = {'latents': torch.from_numpy(lat), 'pts': torch.from_numpy(cloud.T.copy()).unsqueeze(0),
            'pts_query': torch.from_numpy(qry).unsqueeze(0), 'pts_local_ps': torch.from_numpy(patches).unsqueeze(0)}
    ref = O.ppsurf_from_latent(sd, data, k=64)[0].T.numpy()
    logits, _ = pl.decode(pl.point_table(dev(lat[0])), dev(cloud), dev(qry), dev(ids), dev(patches))
    np.testing.assert_allclose(logits.cpu().numpy(), ref, rtol=0, atol=1e-4)


@pytest.mark.parametrize('dtype,scale', [('f32', 1.0), ('f32', 20.0), ('f16x3', 1.0), ('f16x3', 20.0)])
def test_decoder_full_chunk_properties(dtype, scale):
    """BASELINE chunk (N=100k, Q=50k, k=64, P=50): finite outputs, permutation equivariance over queries, and a sampled comparison with
    the ORACLE (16 384 queries incl. the 64 largest-|logit| ones, four slices) -- for both decoder dtypes, at latent magnitude 1 and at the magnitude the real encoder produces (x20:
    logits ~ 27, where the absolute 1e-4 bar is hardest)."""
    sd = filled_sd('', key='ppsurf')
    pl = plan(dtype)
    cloud = make_cloud(100_000, seed=42)
    qry = make_band_queries(cloud, 50_000, resolution=257, seed=1)
    pts, qd = dev(cloud), dev(qry)
    lat = make_latents(256, cloud.shape[0], seed=77) * np.float32(scale)
    table = pl.point_table(dev(lat[0]))
    idx = ops.knn_point_major(pts, qd, 64)
    patches = ops.patch_normalize(pts, qd, idx, 50)
    logits, occ = pl.decode(table, pts, qd, idx, patches)
    lg = logits.cpu().numpy()
    assert np.isfinite(lg).all() and (np.abs(occ.cpu().numpy()) <= 1).all()
    perm = torch.randperm(50_000, device=DEV)
    lg2, _ = pl.decode(table, pts, qd[perm].contiguous(), idx[perm].contiguous(), patches[perm].contiguous())
    # each query is independent of its tile neighbours and of its position in the chunk: equal, not close
    assert torch.equal(lg2, logits[perm])
    # 16 384 queries against the ORACLE (a third of the chunk), in four slices of 4096 -- the oracle's [1, 256, Q, 64] tensors are
    # what bounds a slice, not its time: the 64 largest-|logit| queries of the exact-fp32 kernels (where an absolute bar is hardest) + 16 320 random ones
    lg32 = lg if dtype == 'f32' else plan('f32').decode(plan('f32').point_table(dev(lat[0])), pts, qd, idx, patches)[0].cpu().numpy()
    top = np.argsort(-np.abs(lg32).max(axis=1))[:64]
    rest = np.setdiff1d(np.arange(50_000), top)
    sel = np.concatenate([top, np.random.default_rng(1).choice(rest, 16384 - 64, replace=False)])
    worst, peak = 0.0, 0.0
    for part in np.split(sel, 4):
        data = {'latents': torch.from_numpy(lat), 'pts': torch.from_numpy(cloud.T.copy()).unsqueeze(0),
                'pts_query': torch.from_numpy(qry[part]).unsqueeze(0), 'pts_local_ps': patches[torch.from_numpy(part).to(DEV)].cpu().unsqueeze(0)}
        ref = O.ppsurf_from_latent(sd, data, k=64)[0].T.numpy()
        worst, peak = max(worst, float(np.abs(lg[part] - ref).max())), max(peak, float(np.abs(ref).max()))
        np.testing.assert_allclose(lg[part], ref, rtol=0, atol=1e-4)
    print(dtype, 'scale', scale, 'logits |max| {:.1f}, max |dlogit| vs oracle over {} queries {:.2e}'.format(peak, sel.size, worst))
    if dtype == 'f16x3':
        assert pl.range_fallbacks() == 0                       # in range: the fp32 fall-back kernels returned at their gate


@pytest.mark.parametrize('scale', [1.0e4, 1.0e5])
def test_f16x3_range_guard_falls_back_to_fp32(scale):
    """The default dtype splits every activation x = hi + lo with hi = f16(x): |x| must stay below 65504.  Latents scaled until the first hidden
    layer leaves that range: the split-precision kernels raise the guard word and the fp32 kernels queued behind them recompute the chunk on the
    device -- the result IS the fp32 path's (torch.equal), matches the oracle, and the plan counts the fall-back."""
    sd = filled_sd('', key='ppsurf')
    pl, pl32 = plan('f16x3'), plan('f32')
    cloud = make_cloud(20_000, seed=5)
    qry = make_band_queries(cloud, 3000, resolution=129, seed=2)
    pts, qd = dev(cloud), dev(qry)
    lat = make_latents(256, cloud.shape[0], seed=9) * np.float32(scale)
    idx = ops.knn_point_major(pts, qd, 64)
    patches = ops.patch_normalize(pts, qd, idx, 50)
    before = pl.range_fallbacks()
    logits, occ = pl.decode(pl.point_table(dev(lat[0])), pts, qd, idx, patches)
    want, occ32 = pl32.decode(pl32.point_table(dev(lat[0])), pts, qd, idx, patches)
    hmax = float(pl32.point_table(dev(lat[0])).abs().max())
    print('scale', scale, 'max |G| {:.3g}  max |logit| {:.3g}'.format(hmax, float(want.abs().max())))
    assert hmax > 65504.0                                       # the test really leaves the range
    assert pl.range_fallbacks() == before + 1
    assert torch.isfinite(logits).all() and torch.equal(logits, want) and torch.equal(occ, occ32)
    sel = np.arange(0, 3000, 12)
    data = {'latents': torch.from_numpy(lat), 'pts': torch.from_numpy(cloud.T.copy()).unsqueeze(0),
            'pts_query': torch.from_numpy(qry[sel]).unsqueeze(0), 'pts_local_ps': patches[torch.from_numpy(sel).to(DEV)].cpu().unsqueeze(0)}
    ref = O.ppsurf_from_latent(sd, data, k=64)[0].T.numpy()
    got = logits.cpu().numpy()[sel]
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-4 * max(1.0, float(np.abs(ref).max())))      # 1e-4 relative to the (huge) logit scale
    # the next, ordinary chunk runs in split precision again (the guard word is per chunk)
    lat1 = make_latents(256, cloud.shape[0], seed=9)
    l1, _ = pl.decode(pl.point_table(dev(lat1[0])), pts, qd, idx, patches)
    l1_32, _ = pl32.decode(pl32.point_table(dev(lat1[0])), pts, qd, idx, patches)
    assert pl.range_fallbacks() == before + 1 and not torch.equal(l1, l1_32) and float((l1 - l1_32).abs().max()) < 1e-4


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('name,p', [('pointnet', 10), ('pointnet', 50), ('pointnet_p200', 200)])
def test_pointnet_branch_matches_reference_module_fixture(name, p, dtype):
    """The HIP PointNet kernels against the REFERENCE's PointNetfeat outputs (feature STN matrix `trans2` and the attention-pooled
    feature), P = 10 / 50 / 200: the branch's intermediates are read back from the decoder workspace; the reference's
    `feat = att.fc_value(sum_p w_p x_p)` is finished on the host from the pooled vector the kernels hand to the fused tail."""
    g = load_golden(name)
    pre = 'PN_p{}.'.format(p)
    x = g['x'] if name == 'pointnet_p200' else g['p{}_x'.format(p)]
    feat = g['feat'] if name == 'pointnet_p200' else g['p{}_feat'.format(p)]
    trans2 = g['trans2'] if name == 'pointnet_p200' else g['p{}_trans2'.format(p)]
    pn = filled_sd(pre, key='PN_p50.')
    sd = {k: v for k, v in filled_sd('', key='ppsurf').items() if not k.startswith('point_net.')}
    sd.update({'point_net.' + k[len(pre):]: v for k, v in pn.items()})
    from ppsurf_amd.decoder import DecoderPlan
    pl = DecoderPlan(sd, DEV, dtype=dtype)
    q = x.shape[0]
    patches = np.ascontiguousarray(x.transpose(0, 2, 1))                               # [Q,3,P] -> [Q,P,3]
    cloud = make_cloud(500, seed=1)
    qry = cloud[:q].copy()
    ids = O.knn_point_major(cloud, qry, 64)
    pl.decode(pl.point_table(dev(make_latents(256, 500, seed=2)[0])), dev(cloud), dev(qry), dev(ids), dev(patches))
    inter = pl.intermediates(q)
    if dtype == 'f32':                 # (f16x3 keeps the matrix as pre-split hi / lo fragments in the order its consumer reads them: checked through xbar)
        # the kernels hold M = conv1 (with its eval BatchNorm folded) @ trans2: conv1 is composed into the STN's last layer (decoder.py)
        from ppsurf_amd.decoder import _fold_bn, _wb
        w1 = _fold_bn(*_wb(pn, pre + 'conv1'), pn, pre + 'bn1')[0]
        want = np.einsum('oa,qab->qob', w1, trans2.astype(np.float64))
        np.testing.assert_allclose(inter['trans2'][:trans2.shape[0]].cpu().numpy().reshape(-1, 64, 64), want, rtol=0, atol=5e-5)
    # the kernels pool conv2's 128-channel output; conv3 (+ bn3) and att.fc_value follow once per query (composed into the tail by DecoderPlan):
    # finish them here on the host to compare with the reference's pooled feature
    from ppsurf_amd.decoder import _fold_bn, _wb
    w3, b3 = _fold_bn(*_wb(pn, pre + 'conv3'), pn, pre + 'bn3')
    wv = pn[pre + 'att.fc_value.weight'].reshape(256, 256).double()
    z = inter['xbar'].double().cpu() @ torch.from_numpy(w3).t() + torch.from_numpy(b3)
    got = z @ wv.t() + pn[pre + 'att.fc_value.bias'].double()
    np.testing.assert_allclose(got.numpy(), feat, rtol=0, atol=5e-5)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('p', [50, 100, 40])
def test_logits_do_not_depend_on_the_chunking(p, dtype):
    """A query's logits are BIT-identical whether it is decoded in one chunk of 40 000, in two uneven chunks or among other neighbours:
    the PointNet row kernels use full packed rounds for most of a chunk and the same arithmetic with one query per group for the remainder
    (patch_packing mode 2); every other kernel works per query.  This is what makes a query-sharded reconstruction (PPS_SHARD=queries)
    equal -- not close -- to the single-rank one.  P = 50 / 100 are packed patterns (2 / 4 left-over rows), P = 40 is not."""
    from ppsurf_amd.synthetic import network_state_dict
    pl = plan(dtype) if p == 50 else DecoderPlan(network_state_dict('ppsurf', num_pts_local=p), DEV, dtype=dtype)
    cloud = make_cloud(30_000, seed=p)
    qry = make_band_queries(cloud, 40_000, resolution=129, seed=2)
    pts, qd = dev(cloud), dev(qry)
    table = pl.point_table(dev(make_latents(256, cloud.shape[0], seed=9)[0] * np.float32(10.0)))
    idx = ops.KnnBlocks(pts).query(qd, max(64, p))                           # the P nearest in (distance, index) order: the 64 nearest are its prefix
    patches = ops.patch_normalize(pts, qd, idx, p)
    idx = idx[:, :64].contiguous()
    whole, _ = pl.decode(table, pts, qd, idx, patches)
    whole = whole.clone()
    for cut in (17_000, 16_384, 39_999):
        parts = [pl.decode(table, pts, qd[a:b].contiguous(), idx[a:b].contiguous(), patches[a:b].contiguous())[0].clone() for a, b in ((0, cut), (cut, 40_000))]
        assert torch.equal(torch.cat(parts), whole), (p, dtype, cut)
    perm = torch.randperm(40_000, device=DEV)
    shuffled, _ = pl.decode(table, pts, qd[perm].contiguous(), idx[perm].contiguous(), patches[perm].contiguous())
    assert torch.equal(shuffled, whole[perm])


# ---- opt-in split-precision decoder dtype ("f16x3"): same 1e-4 bar as the fp32 path -----------------------------------------
_plan16 = None


def plan16():
    return plan('f16x3')


def test_f16x3_pack_layout_roundtrip():
    """hi + lo of the packed split-precision image reproduce the weights to 2^-21 and follow the documented channel map."""
    from ppsurf_amd.decoder import pack_dense_f16x3
    rng = np.random.default_rng(0)
    w = (rng.standard_normal((64, 256)) * 0.1).astype(np.float32)
    img = pack_dense_f16x3(w).view(np.float16).reshape(4, 8, 2, 64, 8).astype(np.float64)
    rec = np.zeros((64, 256))
    for l in range(64):
        for j in range(8):
            rec[(l & 15)::16, (16 * (j >> 2) + 4 * (l >> 4) + (j & 3))::32] = img[:, :, 0, l, j] + img[:, :, 1, l, j]
    assert np.abs(rec - w).max() <= np.abs(w).max() * 2.0 ** -21
    assert np.array_equal(img[:, :, 0, :, :].astype(np.float16), img[:, :, 0, :, :])          # parts are f16 values


def test_f16x3_reference_golden_logits():
    g = load_golden('ppsurf_from_latent')
    pl = plan16()
    cloud = g['cloud']
    lat = make_latents(256, cloud.shape[0], 77)[0]
    logits, occ = pl.decode(pl.point_table(dev(lat)), dev(cloud), dev(g['query']), dev(g['proj_ids'][0]), dev(g['patches']))
    err = np.abs(logits.cpu().numpy() - g['logits'][0].T).max()
    print('f16x3 vs reference golden: max |dlogit| = {:.2e}'.format(err))
    assert err < 1e-4
    np.testing.assert_allclose(occ.cpu().numpy(), g['occ'], rtol=0, atol=1e-4)


@pytest.mark.parametrize('scale', [1.0, 25.0])
@pytest.mark.parametrize('p,q,k', [(50, 203, 64), (50, 1, 64), (10, 77, 64), (200, 21, 64), (50, 64, 20), (50, 130, 5)])
def test_f16x3_matches_oracle(p, q, k, scale):
    """Split precision against the oracle for ragged query counts, small k (rows beyond k masked), P = 10 / 50 / 200, and latents
    at the magnitude the real encoder produces (|latent| ~ 25 -> logits ~ 30: the absolute 1e-4 bar is hardest there)."""
    sd = filled_sd('', key='ppsurf')
    if p != 50:
        from ppsurf_amd.synthetic import network_state_dict
        sd = network_state_dict('ppsurf', num_pts_local=p)
    pl = DecoderPlan(sd, DEV, dtype='f16x3') if p != 50 else plan16()
    cloud = make_cloud(1500, seed=p + q)
    qry = make_band_queries(cloud, q, resolution=33, seed=q)
    kk = min(k, 64)
    ids = O.knn_point_major(cloud, qry, kk)
    patches = O.normalize_patches(cloud[O.knn_point_major(cloud, qry, p)], qry).astype(np.float32)
    lat = make_latents(256, cloud.shape[0], seed=q) * np.float32(scale)
    data = {'latents': torch.from_numpy(lat), 'pts': torch.from_numpy(cloud.T.copy()).unsqueeze(0),
            'pts_query': torch.from_numpy(qry).unsqueeze(0), 'pts_local_ps': torch.from_numpy(patches).unsqueeze(0)}
    ref = O.ppsurf_from_latent(sd, data, k=kk)[0].T.numpy()
    logits, _ = pl.decode(pl.point_table(dev(lat[0])), dev(cloud), dev(qry), dev(ids), dev(patches))
    np.testing.assert_allclose(logits.cpu().numpy(), ref, rtol=0, atol=1e-4)


def test_f16x3_full_chunk_against_fp32_path():
    """BASELINE chunk (N=100k, Q=50k): the split-precision branch against the fp32 kernels on every query."""
    pl, pl16 = plan(), plan16()
    cloud = make_cloud(100_000, seed=42)
    qry = make_band_queries(cloud, 50_000, resolution=257, seed=1)
    pts, qd = dev(cloud), dev(qry)
    lat = make_latents(256, cloud.shape[0], seed=77) * np.float32(20.0)
    idx = ops.knn_point_major(pts, qd, 64)
    patches = ops.patch_normalize(pts, qd, idx, 50)
    a, _ = pl.decode(pl.point_table(dev(lat[0])), pts, qd, idx, patches)
    b, _ = pl16.decode(pl16.point_table(dev(lat[0])), pts, qd, idx, patches)
    err = float((a - b).abs().max())
    print('f16x3 vs fp32, 50000 queries, logits |max| {:.1f}: max diff {:.2e}'.format(float(a.abs().max()), err))
    assert torch.isfinite(b).all() and err < 1e-4


@pytest.mark.parametrize('dtype', DTYPES)
def test_chunk_lanes_change_nothing(dtype):
    """ChunkPipeline deals long chunk lists to two HIP streams (lanes), each with its own tables, patches and decoder workspace: the logits and
    occupancies are bit-identical to the single-stream loop, whatever the lane count and the chunk sizes, and the range-guard counter still counts."""
    from ppsurf_amd.decoder import ChunkPipeline
    pl = plan(dtype)
    cloud = make_cloud(20_000, seed=21)
    qry = make_band_queries(cloud, 23_000, resolution=129, seed=4)
    pts, qd = dev(cloud), dev(qry)
    table = pl.point_table(dev(make_latents(256, cloud.shape[0], seed=6)[0]))
    cuts = [0, 5000, 9000, 9001, 14000, 19000, 22500, 23000]                  # seven uneven chunks, one of a single query
    chunks = [qd[a:b].contiguous() for a, b in zip(cuts[:-1], cuts[1:])]
    ref = ChunkPipeline(pl, table, pts, pts, 64, 50, same_cloud=True, max_chunk=5000, lanes=1).run(chunks)
    for lanes in (None, 2, 3):
        pipe = ChunkPipeline(pl, table, pts, pts, 64, 50, same_cloud=True, max_chunk=5000, lanes=lanes)
        for _ in range(2):                                                    # the second call reuses the lanes' buffers
            got = pipe.run(chunks)
            torch.cuda.synchronize()
            assert len(got) == len(ref)
            for (lg, oc), (lg0, oc0) in zip(got, ref):
                assert torch.equal(lg, lg0) and torch.equal(oc, oc0), (dtype, lanes)
    short = ChunkPipeline(pl, table, pts, pts, 64, 50, same_cloud=True, max_chunk=5000).run(chunks[:3])      # below LANE_MIN_CHUNKS: one lane
    assert all(torch.equal(a[0], b[0]) for a, b in zip(short, ref[:3]))
