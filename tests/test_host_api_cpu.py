"""CPU tests of the host-side mirror of the reference API: state-dict layout, constructor surface, sampling contract,
Marching Cubes, region growing (against the golden volume produced by the reference driver), error behaviour."""
import json
import os

import numpy as np
import pytest
import torch

from golden_util import GOLDEN, load_golden


def _manifest(key):
    with open(os.path.join(GOLDEN, 'manifest.json')) as f:
        return [(k, tuple(s)) for k, s in json.load(f)[key]]


def test_state_dict_layout_matches_reference_manifest():
    from source.ppsurf_model import PPSurfNetwork
    from source.poco_model import PocoNetwork
    net = PPSurfNetwork(in_channels=3, latent_size=256, out_channels=2, k=64, num_pts_local=50, pointnet_latent_size=256)
    got = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    ref = dict(_manifest('ppsurf'))
    assert len(ref) == 455 and got == ref
    assert sum(p.numel() for p in net.parameters()) == 13_749_111
    poco = PocoNetwork(in_channels=3, latent_size=32, out_channels=2, k=64)
    assert {k: tuple(v.shape) for k, v in poco.state_dict().items()} == dict(_manifest('poco'))


def test_model_constructor_surface_and_errors():
    from source.ppsurf_model import PPSurfModel
    kw = dict(pointnet_latent_size=256, output_names=['imp_surf_sign'], in_channels=3, out_channels=2, k=64, lambda_l1=0.0, debug=False,
              in_file='datasets/abc_minimal/testset.txt', results_dir='results', padding_factor=0.05, name='ppsurf_mini',
              network_latent_size=256, gen_subsample_manifold_iter=10, gen_subsample_manifold=10000, gen_resolution_global=129,
              num_pts_local=50, rec_batch_size=25000, gen_refine_iter=10, workers=8)
    model = PPSurfModel(**kw)
    for hook in ('training_step', 'validation_step', 'test_step', 'predict_step', 'compute_loss', 'calc_metrics', 'on_after_backward'):
        assert callable(getattr(model, hook))
    for sub in ('encoder', 'projection', 'point_net', 'mlp'):
        assert hasattr(model.network, sub)
    assert model.num_pts_local == 50 and model.rec_batch_size == 25000
    with pytest.raises(NotImplementedError, match='batch size > 1 not supported'):
        model.predict_step({'pts_ms': torch.zeros(2, 10, 3), 'pc_file_in': ['a', 'b']}, 0)
    model.train()                                   # train() runs the autograd graph; its kNN / gather ops are HIP-only
    from ppsurf_amd._lib import PpsError
    with pytest.raises(PpsError, match='no CPU'):
        model.network.from_latent({'pts': torch.zeros(1, 3, 4), 'pts_query': torch.zeros(1, 2, 3), 'latents': torch.zeros(1, 256, 4),
                                   'pts_local_ps': torch.zeros(1, 2, 50, 3)})


def test_loss_and_metrics_match_reference_fixture():
    from source.ppsurf_model import PPSurfModel
    from ppsurf_amd.lightning_api import PocoModel
    g = load_golden('shell')
    pred, occ = torch.from_numpy(g['pred']), torch.from_numpy(g['occ'])
    loss, mean, comps = PocoModel.compute_loss(None, pred, {'occ': occ})
    np.testing.assert_allclose(float(loss), float(g['loss']), rtol=1e-6)
    assert tuple(comps.shape) == (1, 3, 50)
    m = PocoModel.calc_metrics(None, pred, {'occ': occ})
    got = [m[k] for k in ('accuracy', 'precision', 'recall', 'f1_score', 'true_pos', 'false_pos', 'false_neg', 'true_neg')]
    np.testing.assert_allclose(got, g['metrics'], rtol=1e-12)
    assert np.isnan(m['abs_dist_rms'])


def test_sampling_quantized_contract():
    from ppsurf_amd.spatial import sampling_quantized
    import random
    random.seed(0)
    torch.manual_seed(0)
    pts = torch.rand(2, 3, 1037) - 0.5
    sup, ids = sampling_quantized(pts, 0.25)
    n = max(1, int(1037 * 0.25))
    assert tuple(sup.shape) == (2, 3, n) and tuple(ids.shape) == (2, n) and ids.dtype == torch.int64
    for b in range(2):
        assert len(set(ids[b].tolist())) == n                                  # exactly n UNIQUE ids
        assert torch.equal(sup[b], pts[b][:, ids[b]])
    # stratification: a voxel-stratified sample covers space more evenly than the cloud itself covers it
    cell = torch.floor((sup[0].t() + 0.5) * 4).long().clamp(0, 3)
    occupied = len(set(map(tuple, cell.tolist())))
    assert occupied >= 60
    same, ids_all = sampling_quantized(pts, 1.0)
    assert same is pts and torch.equal(ids_all[0], torch.arange(1037))
    with pytest.raises(ValueError):
        sampling_quantized(pts, 1.5)
    one, _ = sampling_quantized(pts[:, :, :3], 0.25)
    assert one.shape[2] == 1


def test_marching_cubes_sphere_is_closed_oriented_and_on_grid_edges():
    from ppsurf_amd import mcubes
    n, c, r = 40, 19.5, 12.3
    g = np.mgrid[0:n, 0:n, 0:n].astype(np.float64)
    vol = r - np.sqrt(((g - c) ** 2).sum(0))
    vol[0, 0, 0] = np.nan
    v, f = mcubes.marching_cubes(vol, 0.0)
    e = np.sort(np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]), axis=1)
    u, cnt = np.unique(e, axis=0, return_counts=True)
    assert (cnt == 2).all() and v.shape[0] - u.shape[0] + f.shape[0] == 2       # closed 2-manifold, genus 0
    assert np.abs(np.linalg.norm(v - c, axis=1) - r).max() < 0.02
    a, b, d = v[f[:, 0]] - c, v[f[:, 1]] - c, v[f[:, 2]] - c
    vol6 = np.einsum('ij,ij->i', a, np.cross(b, d)).sum() / 6
    assert abs(vol6 - 4 / 3 * np.pi * r ** 3) / (4 / 3 * np.pi * r ** 3) < 0.01   # outward normals (towards lower values)
    assert (((v - np.floor(v)) > 0).sum(axis=1) <= 1).all()                     # vertices lie on grid edges
    v2, f2 = mcubes.clean_mesh(np.concatenate([v, v[:3] + 100]), np.concatenate([f, [[len(v), len(v) + 1, len(v) + 2]]]))
    assert f2.shape[0] == f.shape[0]                                            # the 1-face component is dropped


def test_region_growing_matches_reference_driver():
    """create_volume on CPU tensors with the analytic field used by make_golden.py == volume of the reference's
    _create_volume (same visited set, same values)."""
    from ppsurf_amd.reconstruct import create_volume
    g = load_golden('create_volume')
    calls = []

    def field(q):
        calls.append(q.shape[0])
        return (0.4 - torch.linalg.norm(q.double(), dim=1)).float()

    vol = create_volume(field, torch.from_numpy(g['pts_ids'].astype(np.int64)), int(g['resolution']), float(g['step']),
                        float(g['bmin_pad']), padding=1, dilation_size=2, out_value=1.0).numpy()
    ref = g['volume']
    assert np.array_equal(np.isnan(vol), np.isnan(ref))
    np.testing.assert_allclose(np.nan_to_num(vol, nan=7.0), np.nan_to_num(ref, nan=7.0), rtol=0, atol=2e-7)
    assert sum(calls) == int((~np.isnan(ref[1:-1, 1:-1, 1:-1])).sum()) or sum(calls) <= int((~np.isnan(ref)).sum())   # no voxel evaluated twice


def test_ply_reader_writer_roundtrip(tmp_path):
    from ppsurf_amd import meshio
    v = np.random.default_rng(0).standard_normal((10, 3)).astype(np.float32)
    f = np.array([[0, 1, 2], [2, 3, 4]])
    p = str(tmp_path / 'm.ply')
    meshio.write_ply_mesh(p, v, f)
    assert np.array_equal(meshio.read_ply_vertices(p).astype(np.float32), v)
    np.save(str(tmp_path / 'c.npy'), v)
    assert np.array_equal(meshio.load_pts(str(tmp_path / 'c.npy')), v)


def test_reference_import_paths_resolve():
    """Everything SURVEY 8(b) lists as a reference surface is importable under the reference's own module paths."""
    import importlib
    for mod, names in (('source.poco_model', ['PocoModel', 'PocoNetwork', 'InterpAttentionKHeadsNet']),
                       ('source.ppsurf_model', ['PPSurfModel', 'PPSurfNetwork']),
                       ('source.poco_data_loader', ['PocoDataModule', 'PocoDataset', 'sampling_quantized', 'get_fkaconv_ids', 'get_proj_ids', 'get_data_poco']),
                       ('source.ppsurf_data_loader', ['PPSurfDataModule', 'PPSurfDataset']),
                       ('source.poco_utils', ['knn', 'export_mesh_and_refine_vertices_region_growing_v3']),
                       ('source.occupancy_data_module', ['in_file_is_dataset', 'get_set_files', 'read_shape_list', 'load_pts']),
                       ('source.base.nn', ['FKAConvLayer', 'ResidualBlock', 'FKAConvNetwork', 'PointNetfeat', 'STN', 'MLP', 'batch_gather', 'max_pool', 'interpolate']),
                       ('source.base.metrics', ['compare_predictions_binary_tensors']),
                       ('source.cli', ['PPSProgressBar', 'PPSProfiler'])):
        m = importlib.import_module(mod)
        for n in names:
            assert hasattr(m, n), '{}.{}'.format(mod, n)


def test_normalize_patches_static_method_matches_reference_fixture():
    from source.ppsurf_data_loader import PPSurfDataset
    g = load_golden('ppsurf_from_latent')
    local = g['cloud'][g['patch_ids']]
    got = PPSurfDataset.normalize_patches(pts_local_ms=local, pts_query_ms=g['query'])
    np.testing.assert_allclose(got, g['patches'], rtol=1e-6, atol=1e-6)
    assert np.abs(np.linalg.norm(got, axis=2).max(axis=1) - 1.0).max() < 1e-5          # farthest point on the unit sphere


def test_device_metrics_equal_the_reference_metrics():
    """The sync-free per-step metrics of the fit loop are the same numbers (NaN where the reference returns NaN)."""
    from ppsurf_amd.lightning_api import binary_metrics_on_device, compare_predictions_binary_tensors
    rng = np.random.default_rng(3)
    cases = [(rng.integers(0, 2, 500), rng.integers(0, 2, 500)), (np.zeros(40), np.zeros(40)), (np.ones(40), np.zeros(40)),
             (np.zeros(40), np.ones(40)), (np.ones(7), np.ones(7))]
    for gt, pr in cases:
        gt, pr = torch.from_numpy(gt.astype(np.int64)), torch.from_numpy(pr.astype(np.float32))
        want = compare_predictions_binary_tensors(gt, pr, None)
        got = binary_metrics_on_device(gt, pr)
        for k in ('accuracy', 'precision', 'recall', 'f1_score'):
            a, b = float(got[k]), float(want[k])
            assert (np.isnan(a) and np.isnan(b)) or abs(a - b) < 1e-6, (k, a, b)


@pytest.mark.skipif(not os.path.isdir('/root/reference/datasets/abc_minimal'), reason='reference dataset only exists in the build container')
def test_readers_on_the_reference_dataset_files():
    """The reference's own abc_minimal files (trimesh-written binary PLY with a comment line, float32 .npy queries / distances,
    set lists) through this package's readers and the reconstruction dataset (source/occupancy_data_module.py:34-71,174-253)."""
    from ppsurf_amd import data
    in_file = '/root/reference/datasets/abc_minimal/testset.txt'
    names = data.read_shape_list(in_file)
    train, val, test = data.get_set_files(in_file)
    assert len(names) == 3 and len(data.read_shape_list(train)) == 7 and len(data.read_shape_list(val)) == 3 and test == in_file
    ds = data.ReconstructionDataset(in_file, 0.05)
    for i in range(len(ds)):
        it = ds[i]
        n = it['pts_ms'].shape[0]
        assert n > 10000 and it['pts_ms'].dtype == torch.float32 and tuple(it['pts_ms'].shape) == (n, 3)
        assert float(it["pts_ms"].abs().max()) < 1.0                       # dataset clouds are already normalised (noisy scans overshoot 0.5)
        assert tuple(it['pts_query_ms'].shape) == (2000, 3) and tuple(it['imp_surf_dist_ms'].shape) == (2000,)
        assert it['pc_file_in'].endswith(names[i] + '.xyz.ply')


def test_accelerator_cpu_is_refused_with_the_replacement_command():
    """BASELINE config 1 as written (`--trainer.accelerator cpu`, configs/ppsurf_mini.yaml + README mini flow, pps.py:27-72): this build has no host
    implementation of the path, and says so BEFORE touching model or data -- naming the command to run instead."""
    from ppsurf_amd import runner
    cfgs = [os.path.join(GOLDEN, 'configs', n) for n in ('poco.yaml', 'ppsurf.yaml', 'ppsurf_mini.yaml')]
    argv = ['pps.py', 'predict', '-c', cfgs[0], '-c', cfgs[1], '-c', cfgs[2], '--model.init_args.gen_resolution_global', '33',
            '--trainer.accelerator', 'cpu', '--trainer.logger', 'False']
    with pytest.raises(RuntimeError) as exc:
        runner.main(argv)
    msg = str(exc.value)
    assert 'no CPU path' in msg and 'trainer.accelerator=cpu' in msg
    want = 'python pps.py predict -c {} -c {} -c {} --model.init_args.gen_resolution_global 33 --trainer.logger False --trainer.accelerator gpu --trainer.devices 1'.format(*cfgs)
    assert want in msg, msg                                   # the same command, on the GPU
    assert 'oracle' in msg and 'INTEGRATION.md' in msg
