"""BASELINE.json configurations at their own sizes and with the reference's REAL configuration files
(tests/golden/configs/*.yaml, copied byte for byte from the reference by tests/golden/make_golden_r2.py) on the real-data
fixture tests/golden/abc_mini4 (four shapes of the reference's datasets/abc_minimal).

  config 1  ppsurf_mini predict, gen_resolution_global=33 (its GPU form: trainer.accelerator stays `gpu`; there is no CPU path)
  config 2  ppsurf_50nn predict, one ABC shape, R=129
  config 3  ppsurf_50nn fit step at B=10 x 10000 points x 2000 queries, fp32 and bf16-mixed
  config 5  ppsurf_200nn chunk: N=250000, P=200, rec_batch_size 25000
"""
import json
import os
import shutil

import numpy as np
import pytest
import torch

from golden_util import GOLDEN
from oracle import ppsurf_oracle as O

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
CFG = os.path.join(GOLDEN, 'configs')


def _stack(*names):
    out = []
    for n in names:
        out += ['-c', os.path.join(CFG, n + '.yaml')]
    return out


@pytest.fixture(scope='module')
def trained(tmp_path_factory):
    """A short fit with the reference's real YAML stack (poco + ppsurf + ppsurf_mini: AdamW / MultiStepLR / 16-mixed / callbacks /
    TensorBoardLogger keys included) on the three training shapes of abc_mini4 -> last.ckpt with learned weights."""
    from ppsurf_amd import runner
    root = tmp_path_factory.mktemp('cfg')
    shutil.copytree(os.path.join(GOLDEN, 'abc_mini4'), root / 'abc')
    cwd = os.getcwd()
    os.chdir(root)
    try:
        runner.main(['pps.py', 'fit'] + _stack('poco', 'ppsurf', 'ppsurf_mini') + [
            '--data.init_args.in_file', str(root / 'abc' / 'testset.txt'), '--data.init_args.batch_size', '3',
            '--data.init_args.manifold_points', '5000', '--trainer.max_epochs', '30', '--trainer.check_val_every_n_epoch', '15',
            '--trainer.precision', 'bf16-mixed', '--lr_scheduler.init_args.milestones', '[22, 27]'])
    finally:
        os.chdir(cwd)
    ckpt = root / 'models' / 'ppsurf_mini' / 'version_0' / 'checkpoints' / 'last.ckpt'
    assert ckpt.is_file()
    return root, str(ckpt)


def _short_fit(tmp_path_factory, tag, epochs, env):
    """the fixture's fit for a few epochs under an environment -> state_dict of last.ckpt"""
    from ppsurf_amd import runner
    root = tmp_path_factory.mktemp(tag)
    shutil.copytree(os.path.join(GOLDEN, 'abc_mini4'), root / 'abc')
    cwd, old = os.getcwd(), {k: os.environ.get(k) for k in env}
    os.chdir(root)
    os.environ.update(env)
    try:
        runner.main(['pps.py', 'fit'] + _stack('poco', 'ppsurf', 'ppsurf_mini') + [
            '--data.init_args.in_file', str(root / 'abc' / 'testset.txt'), '--data.init_args.batch_size', '3',
            '--data.init_args.manifold_points', '5000', '--trainer.max_epochs', str(epochs), '--trainer.check_val_every_n_epoch', '15',
            '--trainer.precision', 'bf16-mixed'])
    finally:
        os.chdir(cwd)
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    return torch.load(root / 'models' / 'ppsurf_mini' / 'version_0' / 'checkpoints' / 'last.ckpt', map_location='cpu')['state_dict']


def test_config1_fit_replayed_as_a_hip_graph_is_the_eager_fit(tmp_path_factory):
    """`pps.py fit` of config 1's stack for 7 epochs (3 eager steps, the recording, 4 replays; dropout active) against the same fit that never
    records (PPS_FIT_GRAPH=norecord: same optimizer class, device-side learning rate, loader thread): every tensor of the checkpoint EQUAL.
    Round 5 found them different in ONE tensor (stn2.fc3.bias, from the second replay on, by amounts that changed with the memory layout of the
    recording): torch's sum over the rows of the [rows, 4096] bf16 gradient inside the replayed graph; the bias gradient now goes through
    pps_col_sum for every width (train_graph._bias_grad)."""
    a = _short_fit(tmp_path_factory, 'replayed', 7, {'PPS_FIT_GRAPH': '1'})
    b = _short_fit(tmp_path_factory, 'norecord', 7, {'PPS_FIT_GRAPH': 'norecord'})
    assert a.keys() == b.keys() and len(a) == 455
    bad = [k for k in a if not torch.equal(a[k], b[k])]
    assert not bad, bad[:5]


def test_config1_ppsurf_mini_predict_real_yaml_stack(trained, capsys):
    """`pps.py predict -c configs/poco.yaml -c configs/ppsurf.yaml -c configs/ppsurf_mini.yaml
    --model.init_args.gen_resolution_global 33` (BASELINE config 1, README 'minimal' flow) with the reference's files unchanged."""
    from ppsurf_amd import runner, meshio
    root, ckpt = trained
    state = torch.load(ckpt, map_location='cpu')
    assert state['pytorch-lightning_version'] == '2.0.0' and len(state['state_dict']) == 455
    model = runner.main(['pps.py', 'predict'] + _stack('poco', 'ppsurf', 'ppsurf_mini') + [
        '--ckpt_path', ckpt, '--data.init_args.in_file', str(root / 'abc' / 'testset.txt'), '--model.init_args.gen_resolution_global', '33',
        '--trainer.logger', 'False', '--trainer.devices', '1', '--model.init_args.results_dir', str(root / 'res1')])
    assert type(model).__name__ == 'PPSurfModel' and model.name == 'ppsurf_mini' and model.rec_batch_size == 25000 and model.k == 64
    assert model.gen_subsample_manifold == 10000 and model.gen_refine_iter == 10 and model.num_pts_local == 50
    names = [l.strip() for l in open(root / 'abc' / 'testset.txt') if l.strip()]
    out = capsys.readouterr().out
    mesh_dir = root / 'res1' / 'ppsurf_mini' / 'abc' / 'meshes'
    done = [n for n in names if (mesh_dir / (n + '.xyz.ply')).is_file()]
    assert len(done) + out.count('No reconstruction for') == len(names) and len(done) >= 1
    for n in done:
        v = meshio.read_ply_vertices(str(mesh_dir / (n + '.xyz.ply')))[:, :3]
        assert v.shape[0] > 100 and np.isfinite(v).all() and np.abs(v).max() < 0.75


def test_consecutive_shapes_of_equal_size_are_decoded_with_their_own_latents(trained):
    """three of the four abc_mini4 clouds have the same N.  Reconstructing [A, B] in one process must give
    the same mesh for B as reconstructing B alone (the per-shape table cache may not survive the shape it was built for)."""
    from ppsurf_amd import runner
    root, ckpt = trained
    names = [l.strip() for l in open(root / 'abc' / 'trainset.txt') if l.strip()]
    from ppsurf_amd import meshio
    sizes = {n: meshio.load_pts(str(root / 'abc' / '04_pts_vis' / (n + '.xyz.ply'))).shape[0] for n in names}
    pair = [n for n in names if list(sizes.values()).count(sizes[n]) >= 2][:2]
    assert len(pair) == 2, sizes

    def run(shapes, tag):
        lst = root / 'abc' / ('pair_{}.txt'.format(tag))
        lst.write_text('\n'.join(shapes) + '\n')
        torch.manual_seed(7)
        import random
        random.seed(7)
        model = runner.main(['pps.py', 'predict'] + _stack('poco', 'ppsurf', 'ppsurf_mini') + [
            '--ckpt_path', ckpt, '--data.init_args.in_file', str(lst), '--model.init_args.gen_resolution_global', '25', '--seed_everything', 'null',
            '--model.init_args.gen_subsample_manifold_iter', '1', '--trainer.logger', 'False', '--model.init_args.results_dir', str(root / ('res_' + tag))])
        return model.last_prediction

    both = run(pair, 'ab')
    alone = run(pair[1:], 'b')
    assert (both is None) == (alone is None)
    if both is not None:
        # the latent loop and the support sampling are stochastic, so the two runs differ in the last digits of the latents;
        # a table of the WRONG shape would move the surface by whole voxels
        va, vb = both[0], alone[0]
        assert abs(va.shape[0] - vb.shape[0]) < 0.1 * vb.shape[0]
        d = np.sqrt(((va[::7, None, :] - vb[None, ::3, :]) ** 2).sum(-1)).min(axis=1)
        # (measured: 0.019 - 0.022 between two runs of the same shape = half a voxel of the R = 25 grid, depending on the dropout stream of the
        # fit that made the checkpoint; a foreign table gives more than a voxel)
        assert np.median(d) < 0.75 / 24, np.median(d)


def test_config2_ppsurf_50nn_predict_one_abc_shape_r129(trained):
    """BASELINE config 2: poco + ppsurf + ppsurf_50nn YAMLs, one ABC shape, gen_resolution_global=129, rec_batch_size 50000."""
    from ppsurf_amd import runner, meshio
    root, ckpt = trained
    names = [l.strip() for l in open(root / 'abc' / 'trainset.txt') if l.strip()]
    one = root / 'abc' / 'one.txt'
    one.write_text(names[0] + '\n')
    model = runner.main(['pps.py', 'predict'] + _stack('poco', 'ppsurf', 'ppsurf_50nn') + [
        '--ckpt_path', ckpt, '--data.init_args.in_file', str(one), '--model.init_args.gen_resolution_global', '129',
        '--trainer.logger', 'False', '--trainer.devices', '1', '--model.init_args.results_dir', str(root / 'res2')])
    assert model.name == 'ppsurf_50nn' and model.rec_batch_size == 50000 and model.gen_resolution_global == 129
    f = root / 'res2' / 'ppsurf_50nn' / 'abc' / 'meshes' / (names[0] + '.xyz.ply')
    assert f.is_file()
    v = meshio.read_ply_vertices(str(f))[:, :3]
    cloud = meshio.read_ply_vertices(str(root / 'abc' / '04_pts_vis' / (names[0] + '.xyz.ply')))[:, :3]
    d = np.sqrt(((cloud[::10, None, :] - v[None, ::2, :]) ** 2).sum(-1)).min(axis=1)
    # a surface at R=129 (voxel 1/128 of the bounding cube) that follows the input cloud of the over-fitted shape
    assert v.shape[0] > 10000 and np.median(d) < 4.0 / 128, (v.shape, np.median(d))


def test_poco_test_subcommand_real_yaml(trained, capsys):
    """`test` with the POCO configuration (no patches: PocoDataset) must not ask for pts_local_ps."""
    from ppsurf_amd import runner
    root, _ = trained
    runner.main(['pps.py', 'test'] + _stack('poco', 'poco_mini') + ['--data.init_args.in_file', str(root / 'abc' / 'testset.txt'),
                                                                    '--trainer.logger', 'False', '--model.init_args.results_dir', str(root / 'res_poco')])
    out = capsys.readouterr().out
    assert out.count('loss ') == 2 and 'Test results (mean): Loss=' in out
    assert (root / 'res_poco' / 'poco_mini' / 'abc' / 'metrics_poco_mini.csv').is_file()


def test_batch_dictionaries_match_the_reference_manifest(trained):
    """Keys / dtypes / shapes of the predict and fit batch dictionaries against the manifest recorded from the reference's own
    datasets (PPSurfReconstructionDataset / PPSurfDataset + default_collate on datasets/abc_minimal, batch_manifest.json)."""
    from ppsurf_amd.data import PPSurfDataModule
    root, _ = trained
    man = json.load(open(os.path.join(GOLDEN, 'batch_manifest.json')))
    dm = PPSurfDataModule(num_pts_local=50, in_file=str(root / 'abc' / 'testset.txt'), workers=0, use_ddp=False, padding_factor=0.05, seed=42,
                          manifold_points=10000, patches_per_shape=2000, do_data_augmentation=False, batch_size=2)
    dm.device = torch.device(DEV)
    pred = next(iter(dm.predict_dataloader()))
    ref = man['predict_batch1']
    assert set(pred.keys()) == set(ref.keys()), set(pred.keys()) ^ set(ref.keys())
    for k, (dt, shape) in ref.items():
        if torch.is_tensor(pred[k]):
            assert str(pred[k].dtype).replace('torch.', '') == dt and pred[k].dim() == len(shape), (k, pred[k].dtype, pred[k].shape, dt, shape)
            assert pred[k].shape[0] == 1 and tuple(pred[k].shape[2:]) == tuple(shape[2:])
        else:
            assert type(pred[k]).__name__ == dt
    dm.trainset = str(root / 'abc' / 'trainset.txt')
    fit = next(iter(dm.train_dataloader()))
    ref = man['fit_batch2']
    missing = set(ref.keys()) - set(k for k in fit.keys() if not k.startswith('_'))
    assert not missing, missing
    for k, (dt, shape) in ref.items():
        if torch.is_tensor(fit[k]):
            assert str(fit[k].dtype).replace('torch.', '') == dt, (k, fit[k].dtype, dt)
            assert tuple(fit[k].shape) == tuple(shape), (k, tuple(fit[k].shape), shape)      # B=2, 10000-point sub-samples, 2000 queries, P=50
        else:
            assert type(fit[k]).__name__ == dt and len(fit[k]) == shape


@pytest.mark.parametrize('precision', ['32', 'bf16-mixed', '16-mixed'])
def test_config3_fit_step_at_full_batch_size(precision):
    """BASELINE config 3 on one GPU: B = 10 shapes x 10000 points, 2000 queries per shape, P = 50 -- in fp32, bf16-mixed (BASELINE's dtype) and
    16-mixed (fp16 autocast + loss scaling: the reference's own default, configs/poco.yaml:10).  The HIP training step against the same
    graph on plain-torch twins of the HIP ops (tests/train_ref_ops.py), like for like: loss within 2e-4 (fp32) / the 16-bit noise floor,
    gradients finite everywhere, no parameter without gradient, and the gradient NORM of every one of the parameter tensors equal to the
    twin step's within the precision's noise (not only the loss)."""
    import train_ref_ops as ref
    import bench_workloads as workloads
    from ppsurf_amd import spatial
    step = workloads.FitStep(batch=10, precision=precision, device=DEV, n_batches=1)
    for m in step.net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    import random
    random.seed(3); torch.manual_seed(3)
    batch = dict(step.batches[0])
    batch['pts_local_ps'] = spatial.get_pts_local_ps_batch([batch['pts_ms'][i] for i in range(10)], batch['pts_query_ms'], 50)
    batch = spatial.get_data_poco(batch)
    assert tuple(batch['pts'].shape) == (10, 3, 10000) and tuple(batch['pts_local_ps'].shape) == (10, 2000, 50, 3) and tuple(batch['ids00'].shape) == (10, 10000, 16)
    sd0 = {k: v.clone() for k, v in step.net.state_dict().items()}
    auto = {'32': None, 'bf16-mixed': torch.bfloat16, '16-mixed': torch.float16}[precision]
    scale = 1024.0 if precision == '16-mixed' else 1.0            # a fixed loss scale (what GradScaler multiplies in), taken out of the norms again
    losses, norms = [], []
    for twins in (False, True):
        step.net.load_state_dict(sd0)
        step.net.zero_grad(set_to_none=True)
        ctx = ref.patched() if twins else __import__('contextlib').nullcontext()
        with ctx, torch.autocast('cuda', dtype=auto or torch.bfloat16, enabled=auto is not None):
            logits = step.net.forward({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()})
            loss = torch.nn.functional.cross_entropy(logits.float(), batch['occ'], reduction='none').mean()
        (loss * scale).backward()
        grads = {k: p.grad for k, p in step.net.named_parameters()}
        assert torch.isfinite(loss) and tuple(logits.shape) == (10, 2, 2000)
        assert all(torch.isfinite(g).all() for g in grads.values() if g is not None)
        losses.append((float(loss.detach()), sorted(k for k, g in grads.items() if g is None)))
        norms.append({k: float(g.double().norm()) / scale for k, g in grads.items()})
    assert losses[0][1] == losses[1][1] == []
    assert abs(losses[0][0] - losses[1][0]) < (2e-4 if precision == '32' else 2e-2), losses
    top = max(norms[1].values())
    rel = {k: abs(norms[0][k] - norms[1][k]) / max(norms[1][k], 1e-3 * top) for k in norms[1]}          # tensors below 1e-3 of the largest norm: absolute
    worst = max(rel, key=rel.get)
    print(precision, 'per-parameter gradient norms, HIP step vs twin step: worst relative difference {:.3e} ({}), median {:.3e}'.format(
        rel[worst], worst, float(np.median(list(rel.values())))))
    # fp32: the two steps differ by summation order only.  16-bit: every activation is rounded to 8 (bf16) / 11 (fp16) bits once per
    # layer in BOTH steps, at different places of the fused / unfused graphs; the norms of 298 tensors agree to a few per cent
    assert rel[worst] < {'32': 2e-3, 'bf16-mixed': 0.3, '16-mixed': 0.1}[precision], (worst, norms[0][worst], norms[1][worst], top)
    assert float(np.median(list(rel.values()))) < {'32': 1e-4, 'bf16-mixed': 2e-2, '16-mixed': 1e-2}[precision]


@pytest.mark.parametrize('precision', ['32', 'bf16-mixed'])
def test_config3_fit_step_at_full_batch_size_against_the_reference(precision):
    """BASELINE config 3 at its full batch size, pinned on the REFERENCE: tests/golden/train_ppsurf_full.npz holds logits, loss,
    buffers and a seeded sample of every parameter gradient of ONE training step of the reference's PPSurfNetwork in train() on 10 shapes x 10 000
    points x 2000 queries (tests/golden/make_golden_train_full.py; outputs from its fp32 run, gradients from its float64 run and -- as the
    reference's own accuracy -- from its fp32 run).  The batch is rebuilt from seeds (tests/golden/cases_full.py) and the id tables checked by
    digest.  fp32: the bars of the small fixture (tests/test_gpu_train.py::test_training_step_gpu).  bf16-mixed -- the dtype of the benched step,
    head chain kernel, fused row layers and side streams included: logits within the 16-bit noise of a 60-layer network, and every gradient tensor
    of significant size within a few per cent of the reference's float64 gradient in direction and length (sampled entries)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    import cases_full as cf
    from golden_util import load_golden
    from test_train_graph_cpu import _check_sigs, _check_samples, _load
    from ppsurf_amd import modules
    import test_train_graph_cpu as T
    g = load_golden('train_ppsurf_full')
    data, occ = cf.full_fit_batch()
    dig = cf.table_digests(data)
    assert [str(k) for k in g['table_names']] == list(dig.keys()) and [str(v) for v in g['table_digests']] == list(dig.values())
    assert cf.digest(occ) == str(g['occ_digest']) and cf.digest(data['pts_local_ps']) == str(g['patches_digest'])
    net = _load(modules.PPSurfNetwork(in_channels=3, latent_size=256, out_channels=2, k=64, num_pts_local=cf.P, pointnet_latent_size=256), '', key='ppsurf',
                digest=g['digest'])
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    net = net.to(DEV)
    batch = {k: v.to(DEV) for k, v in data.items()}
    want_ids = batch.pop('proj_ids')                                # PPSurf recomputes them (ppsurf_model.py:83)
    occ = occ.to(DEV)
    auto = {'32': None, 'bf16-mixed': torch.bfloat16}[precision]
    with torch.autocast('cuda', dtype=auto or torch.bfloat16, enabled=auto is not None):
        logits = net.forward(batch)
        loss = torch.nn.functional.cross_entropy(logits.float(), occ, reduction='none').mean()
    loss.backward()
    assert torch.equal(batch['proj_ids'], want_ids)
    grads = [(k, v.grad.float().cpu()) for k, v in net.named_parameters() if v.grad is not None]
    assert [k for k, p in net.named_parameters() if p.grad is None] == [str(k) for k in g['unused']]
    err_logits = float((logits.detach().float().cpu() - torch.from_numpy(g['logits'])).abs().max())
    err_loss = abs(float(loss.detach()) - float(g['loss']))
    if precision == '32':
        assert err_logits <= 5e-5 * float(np.abs(g['logits']).max()) + 1e-5 and err_loss < 1e-5, (err_logits, err_loss)
        _check_sigs(grads, g['gnames'], g['gsigs'], 1e-2, 'grad', sigs32=g['gsigs32'])
        T_SAMPLE = 1024
        worst = _check_samples_n(grads, g, 1e-2, T_SAMPLE)
        print('fp32 at size: logits {:.2e}, loss {:.2e}; sampled gradient entries worst error / tolerance {:.3f} ({})'.format(err_logits, err_loss, *worst))
        _check_sigs([(k, v.float().cpu()) for k, v in net.named_buffers()], g['bnames'], g['bsigs'], 2e-5, 'buffer')
        return
    # bf16-mixed: direction and length of every sampled gradient against the float64 reference
    names, off = [str(k) for k in g['gnames']], g['gsamp_off']
    named = dict(grads)
    top = max(float(s[2]) for s in g['gsigs'])
    stats = []
    for i, k in enumerate(names):
        sl = slice(int(off[i]), int(off[i + 1]))
        ref = g['gsamp_val'][sl]
        got = named[k].double().reshape(-1)[torch.from_numpy(g['gsamp_idx'][sl])].numpy()
        if float(g['gsigs'][i][2]) < 1e-3 * top or np.linalg.norm(ref) == 0:
            continue                                                # (biases in front of a train-mode BatchNorm: zero gradient, noise in both)
        cos = float(got @ ref / (np.linalg.norm(got) * np.linalg.norm(ref) + 1e-300))
        stats.append((cos, float(np.linalg.norm(got) / np.linalg.norm(ref)), k))
    lo = min(stats)
    ratio = sorted(s[1] for s in stats)
    print('bf16-mixed at size: logits {:.3e} (scale {:.2f}), loss {:.2e}; {} gradient tensors: lowest cosine {:.4f} ({}), length ratio {:.3f} .. {:.3f}'.format(
        err_logits, float(np.abs(g['logits']).max()), err_loss, len(stats), lo[0], lo[2], ratio[0], ratio[-1]))
    # The yardstick is the REFERENCE's own 16-bit step: tests/golden/train_ppsurf_full_bf16ref.npz holds what
    # torch.autocast('cpu', dtype=torch.bfloat16) costs the reference's PPSurfNetwork on this very batch against its own fp32 logits / float64
    # gradients (make_golden_train_full_bf16.py: logits 0.237 of the scale 3.35, loss 4.4e-4, lowest gradient cosine 0.917 -- the same tensor that
    # is lowest here, encoder.resnetb40.shortcut.weight, 390 rows -- median 0.992, length ratio 0.902 .. 1.184).  The build's bf16-mixed step (fused
    # row layers, hand-written dense layers, bf16 storage between layers) may lose at most 1.5 x as much, quantity by quantity; measured on an
    # MI355X it loses LESS than the reference's autocast step: logits 0.164, lowest cosine 0.954, ratio 0.941 .. 1.075.
    y = load_golden('train_ppsurf_full_bf16ref')
    assert str(y['digest']) == str(g['digest'])
    ynames = [str(k) for k in y['gnames']]
    ref = {k: (float(y['cos'][i]), float(y['ratio'][i])) for i, k in enumerate(ynames)}
    ref_used = [ref[k] for _, _, k in stats]                            # the same tensors (above the size threshold, non-zero reference gradient)
    ref_lo = min(c for c, _ in ref_used)
    ref_med = float(np.median([c for c, _ in ref_used]))
    ref_ratio = max(abs(r - 1.0) for _, r in ref_used)
    ours_med = float(np.median([s[0] for s in stats]))
    ours_ratio = max(abs(r - 1.0) for r in ratio)
    F = 1.5
    print('  reference bf16 autocast on the same batch: logits {:.3e}, loss {:.2e}, lowest cosine {:.4f}, median {:.4f}, |ratio - 1| <= {:.3f}'.format(
        float(y['err_logits']), float(y['err_loss']), ref_lo, ref_med, ref_ratio))
    print('  ours / reference: logits {:.2f}, loss {:.2f}, 1 - lowest cosine {:.2f}, 1 - median cosine {:.2f}, |ratio - 1| {:.2f}'.format(
        err_logits / float(y['err_logits']), err_loss / float(y['err_loss']), (1 - lo[0]) / (1 - ref_lo), (1 - ours_med) / (1 - ref_med),
        ours_ratio / ref_ratio))
    assert len(stats) >= 100
    assert err_logits <= F * float(y['err_logits'])
    assert err_loss <= F * float(y['err_loss'])
    assert 1.0 - lo[0] <= F * (1.0 - ref_lo)
    assert 1.0 - ours_med <= F * (1.0 - ref_med)
    assert ours_ratio <= F * ref_ratio
    # tensor by tensor: nowhere more than 1.5 x the reference's own loss of direction plus what two different bf16 roundings of a gradient with
    # 0.92 .. 0.999 cosine to the truth can differ by (1e-2)
    worse = [(k, c, ref[k][0]) for c, _, k in stats if 1.0 - c > F * (1.0 - ref[k][0]) + 1e-2]
    assert not worse, worse


def _check_samples_n(named, g, rtol, n_sample, noise=1e-6):
    """test_train_graph_cpu._check_samples for a fixture with n_sample (not 4096) entries per tensor."""
    named = dict(named)
    names, off = [str(k) for k in g['gnames']], g['gsamp_off']
    floor = noise * max(float(s[2]) for s in g['gsigs'])
    worst = (0.0, None)
    for i, k in enumerate(names):
        sl = slice(int(off[i]), int(off[i + 1]))
        idx, ref = g['gsamp_idx'][sl], g['gsamp_val'][sl]
        t = named[k].detach().double().reshape(-1)
        assert idx.shape[0] == min(n_sample, t.numel())
        got = t[torch.from_numpy(idx)].numpy()
        own = 3 * float(np.abs(g['gsamp_val32'][sl] - ref).max())
        rms = float(g['gsigs'][i][2]) / np.sqrt(t.numel())
        tol = 4 * rtol * max(float(np.abs(ref).max()), rms) + floor + own
        err = float(np.abs(got - ref).max())
        assert err <= tol, 'grad {}: sampled entries differ by {} (tolerance {})'.format(k, err, tol)
        if tol > 0 and err / tol > worst[0]:
            worst = (err / tol, k)
    return worst


@pytest.mark.parametrize('dtype', ['f32', 'f16x3'])
def test_config5_ppsurf_200nn_chunk_at_size(dtype):
    """BASELINE config 5 chunk: N = 250 000 points, P = 200, rec_batch_size = 25 000, k = 64 (configs/ppsurf_200nn.yaml) through the
    product's chunk loop: exact 64-NN and 200-NN tables, finite outputs, permutation equivariance, 1024 queries (incl. the 64 largest-|logit| ones) vs the oracle."""
    from ppsurf_amd.decoder import DecoderPlan, ChunkPipeline
    from ppsurf_amd.synthetic import make_cloud, make_latents, network_state_dict
    import bench_workloads as workloads
    n, p, qn = 250_000, 200, 25_000
    sd = network_state_dict('ppsurf', num_pts_local=p)
    plan = DecoderPlan(sd, DEV, dtype=dtype)
    cloud = make_cloud(n, seed=5)
    lat = make_latents(256, n, seed=6)
    pts = torch.from_numpy(cloud).to(DEV)
    table = plan.point_table(torch.from_numpy(lat[0]).to(DEV))
    chunks, n_band = workloads.band_chunks(cloud, 513, qn, DEV)
    assert n_band > 5_000_000 and len(chunks) > 200                         # the R=513 band of this cloud
    q = chunks[len(chunks) // 2]
    pipe = ChunkPipeline(plan, table, pts, pts, 64, p, same_cloud=True, max_chunk=qn)
    (logits, occ), = pipe.run([q])
    lg = logits.cpu().numpy()
    assert lg.shape == (qn, 2) and np.isfinite(lg).all() and (np.abs(occ.cpu().numpy()) <= 1).all()
    perm = torch.randperm(qn, device=DEV)
    (lg2, _), = pipe.run([q[perm].contiguous()])
    assert torch.equal(lg2, logits[perm])                                    # position in the chunk does not matter, bit for bit
    # 1024 queries against the oracle: the 64 largest-|logit| queries of the exact-fp32 kernels + 960 random ones
    if dtype == 'f32':
        lg32 = logits
    else:
        plan32 = DecoderPlan(sd, DEV, dtype='f32')
        (lg32, _), = ChunkPipeline(plan32, plan32.point_table(torch.from_numpy(lat[0]).to(DEV)), pts, pts, 64, p, same_cloud=True, max_chunk=qn).run([q])
    top = np.argsort(-np.abs(lg32.cpu().numpy()).max(axis=1))[:64]
    sel = np.concatenate([top, np.random.default_rng(2).choice(np.setdiff1d(np.arange(qn), top), 1024 - 64, replace=False)])
    qs = q[torch.from_numpy(sel).to(DEV)].cpu().numpy()
    ids200 = O.knn_point_major(cloud, qs, 200)
    b = (pipe.n - 1) & 1
    # the tables of the LAST run belong to the permuted chunk: compare through the permutation
    inv = torch.empty_like(perm); inv[perm] = torch.arange(qn, device=DEV)
    rows = inv[torch.from_numpy(sel).to(DEV)]
    assert np.array_equal(pipe.pidx[b][rows].cpu().numpy(), ids200)                          # 200-NN patch table: bit-exact
    assert np.array_equal(pipe.idx[b][rows].cpu().numpy(), ids200[:, :64])                  # 64-NN = its prefix
    patches = O.normalize_patches(cloud[ids200], qs).astype(np.float32)
    data = {'latents': torch.from_numpy(lat), 'pts': torch.from_numpy(cloud.T.copy()).unsqueeze(0), 'pts_query': torch.from_numpy(qs).unsqueeze(0),
            'pts_local_ps': torch.from_numpy(patches).unsqueeze(0)}
    ref = O.ppsurf_from_latent(sd, data, k=64)[0].T.numpy()
    np.testing.assert_allclose(lg[sel], ref, rtol=0, atol=1e-4)


def test_f16x3_reconstruction_equals_fp32_reconstruction(trained):
    """The opt-in split-precision decoder end to end: same latents, same driver -> the occupancy of every band voxel agrees with the
    fp32 decoder within 1e-4 and the two meshes coincide (learned weights of the abc_mini4 fit, R = 65)."""
    from ppsurf_amd import reconstruct, meshio
    from ppsurf_amd.lightning_api import PPSurfModel
    import contextlib, io
    root, ckpt = trained
    with contextlib.redirect_stdout(io.StringIO()):
        model = PPSurfModel(pointnet_latent_size=256, output_names=['imp_surf_sign'], in_channels=3, out_channels=2, k=64, lambda_l1=0.0, debug=False,
                            in_file='x.npy', results_dir=str(root / 'r16'), padding_factor=0.05, name='t', network_latent_size=256,
                            gen_subsample_manifold_iter=3, gen_subsample_manifold=10000, gen_resolution_global=65, num_pts_local=50,
                            rec_batch_size=50000, gen_refine_iter=10, workers=1)
    model.load_state_dict(torch.load(ckpt, map_location='cpu')['state_dict'])
    model = model.to(DEV).eval()
    names = [l.strip() for l in open(root / 'abc' / 'trainset.txt') if l.strip()]
    cloud = meshio.load_pts(str(root / 'abc' / '04_pts_vis' / (names[0] + '.xyz.ply')))[:, :3].astype(np.float32)
    pts_cf = torch.from_numpy(cloud).to(DEV).t().contiguous()
    lat = model.encode_latents(pts_cf)
    shape = {'pts': pts_cf.unsqueeze(0), 'latents': lat.t().unsqueeze(0)}
    out = {}
    for dt in ('f32', 'f16x3'):
        model.network.decoder_dtype = dt
        field = reconstruct.OccupancyField(model.network, shape, pts_cf.t().unsqueeze(0), 50000, 50)
        assert field.plan.dtype == dt
        step, bmin_pad, pts_ids = __import__('bench_workloads').grid_geometry(cloud, 65)
        vol = reconstruct.create_volume(field, torch.from_numpy(pts_ids).to(DEV), 65, step, bmin_pad)
        mesh = reconstruct.export_mesh_and_refine_vertices_region_growing_v3(
            network=model.network, latent=shape, pts_raw_ms=pts_cf.t().unsqueeze(0), resolution=65, padding=1, mc_value=0, num_pts=50000,
            num_pts_local=50, input_points=cloud, refine_iter=10, out_value=1)
        out[dt] = (vol.cpu().numpy(), mesh)
    va, vb = out['f32'][0], out['f16x3'][0]
    both = ~np.isnan(va) & ~np.isnan(vb)
    assert (np.isnan(va) != np.isnan(vb)).mean() < 1e-3                      # the same band (a sign flip needs |occ| < 1e-4)
    assert np.abs(va[both] - vb[both]).max() < 1e-4
    ma, mb = out['f32'][1], out['f16x3'][1]
    assert ma is not None and mb is not None and abs(ma[0].shape[0] - mb[0].shape[0]) <= 0.002 * ma[0].shape[0] + 2
    d = np.sqrt(((ma[0][::9, None, :] - mb[0][None, :, :]) ** 2).sum(-1)).min(axis=1)
    assert np.median(d) < 1e-4 and np.percentile(d, 99) < 0.05 / 64


# ---- whole reconstructions at the configurations' own sizes ---------------------------------------------------------------
def _trained_model(root, ckpt, resolution, iters=3):
    from ppsurf_amd.lightning_api import PPSurfModel
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        model = PPSurfModel(pointnet_latent_size=256, output_names=['imp_surf_sign'], in_channels=3, out_channels=2, k=64, lambda_l1=0.0, debug=False,
                            in_file='x.npy', results_dir=str(root / 'rw'), padding_factor=0.05, name='t', network_latent_size=256,
                            gen_subsample_manifold_iter=iters, gen_subsample_manifold=10000, gen_resolution_global=resolution, num_pts_local=50,
                            rec_batch_size=50000, gen_refine_iter=10, workers=1)
    model.load_state_dict(torch.load(ckpt, map_location='cpu')['state_dict'])
    return model.to(DEV).eval()


def _abc_cloud(root, which=0):
    from ppsurf_amd import meshio
    names = [l.strip() for l in open(root / 'abc' / 'trainset.txt') if l.strip()]
    return meshio.load_pts(str(root / 'abc' / '04_pts_vis' / (names[which] + '.xyz.ply')))[:, :3].astype(np.float32)


def _closed_manifold_stats(faces):
    e = np.sort(np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]]), axis=1)
    _, cnt = np.unique(e[:, 0].astype(np.int64) * (int(faces.max()) + 1) + e[:, 1], return_counts=True)
    return float((cnt == 2).mean()), int((cnt == 1).sum()), int((cnt > 2).sum())


def test_config4_unit_of_work_one_abc_shape_at_r257_learned_weights(trained):
    """BASELINE config 4's per-GPU unit of work: ONE whole reconstruction at gen_resolution_global = 257 (latent loop, region growing, Marching
    Cubes, clean-up, 10 refinement rounds; source/poco_utils.py:26-254) of a real ABC shape with the learned checkpoint of the abc_mini4 fit.
    Asserted against a stored band: vertex count, distance of the input cloud to the surface, closedness."""
    from ppsurf_amd import reconstruct
    root, ckpt = trained
    model = _trained_model(root, ckpt, 257)
    cloud = _abc_cloud(root)
    pts_cf = torch.from_numpy(cloud).to(DEV).t().contiguous()
    torch.manual_seed(11)
    lat = model.encode_latents(pts_cf)
    shape = {'pts': pts_cf.unsqueeze(0), 'latents': lat.t().unsqueeze(0)}
    fields = []

    class Field(reconstruct.OccupancyField):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            fields.append(self)

    old, reconstruct.FIELD_CLASS = reconstruct.FIELD_CLASS, Field
    try:
        mesh = reconstruct.export_mesh_and_refine_vertices_region_growing_v3(
            network=model.network, latent=shape, pts_raw_ms=pts_cf.t().unsqueeze(0), resolution=257, padding=1, mc_value=0, num_pts=50000,
            num_pts_local=50, input_points=cloud, refine_iter=10, out_value=1)
    finally:
        reconstruct.FIELD_CLASS = old
    assert mesh is not None
    v, f = mesh
    voxel = float(cloud.max() - cloud.min()) / 256
    sub = cloud[::7]
    d = np.concatenate([np.sqrt(((sub[s:s + 512, None, :] - v[None, ::3, :]) ** 2).sum(-1)).min(axis=1) for s in range(0, sub.shape[0], 512)])
    two, open_e, multi = _closed_manifold_stats(f)
    print('R=257 learned abc shape: {} vertices, {} faces, {} decoder queries, cloud->surface median {:.2f} voxels, p90 {:.2f}; edges with 2 faces {:.4f}, '
          'open {}, >2 {}'.format(v.shape[0], f.shape[0], fields[0].n_queries, np.median(d) / voxel, np.percentile(d, 90) / voxel, two, open_e, multi))
    assert np.isfinite(v).all() and f.min() >= 0 and f.max() < v.shape[0]
    # band: a surface of this CAD part at R = 257 has some 10^5 vertices; the queries are the +-2 band of it plus 10 refinement rounds
    # stored band (measured on an MI355X with this 30-epoch checkpoint: 133 643 vertices, 266 706 faces, 2.31 M decoder queries, cloud -> surface
    # median 3.1 / p90 9.7 voxels, every edge shared by exactly two faces); the fit that makes the checkpoint draws a dropout stream, hence a band
    assert 70_000 < v.shape[0] < 300_000 and 1.95 * v.shape[0] < f.shape[0] < 2.05 * v.shape[0]
    assert 1_200_000 < fields[0].n_queries < 6_000_000
    # the over-fitted shape's surface follows its input cloud (R = 129 with the same checkpoint: within 4 voxels of 1/128; here the voxel is half)
    assert np.median(d) < 5 * voxel and np.percentile(d, 90) < 16 * voxel
    assert two > 0.999 and multi == 0                           # closed 2-manifold (open edges only where the surface leaves the evaluated band)


def test_config4_r257_volume_equals_the_oracle_driver_on_the_same_decoder(trained):
    """BASELINE config 4 at size, pinned on the DRIVER: the R = 257 region-growing volume of a real ABC shape by the product
    (device masks, de-duplicated frontier, ChunkPipeline lanes; ppsurf_amd/reconstruct.py) against oracle.create_volume -- the CPU restatement of
    source/poco_utils.py:178-254, which is pinned to the reference's own `_create_volume` by tests/golden/create_volume.npz -- driven by the SAME
    learned decoder through a second OccupancyField.  35 minutes of CPU decoding are replaced by 6 M GPU queries; what is compared is everything
    else: which voxels are visited in which round, the +-2 dilation, the frontier rule, the borders.  The decoder's kernels are chunk-invariant
    (a query's occupancy does not depend on the batch it is in), so the two volumes are EQUAL, not close -- although the oracle driver evaluates
    2.3 x as many queries (it re-evaluates voxels, poco_utils.py:212-223) in other chunk boundaries."""
    from ppsurf_amd import reconstruct
    import bench_workloads as workloads
    root, ckpt = trained
    model = _trained_model(root, ckpt, 257)
    cloud = _abc_cloud(root)
    pts_cf = torch.from_numpy(cloud).to(DEV).t().contiguous()
    torch.manual_seed(11)
    lat = model.encode_latents(pts_cf)
    shape = {'pts': pts_cf.unsqueeze(0), 'latents': lat.t().unsqueeze(0)}
    step, bmin_pad, pts_ids = workloads.grid_geometry(cloud, 257)
    field = reconstruct.OccupancyField(model.network, shape, pts_cf.t().unsqueeze(0), 50000, 50)
    vol = reconstruct.create_volume(field, torch.from_numpy(pts_ids).to(DEV), 257, step, bmin_pad).cpu().numpy()
    probe = reconstruct.OccupancyField(model.network, shape, pts_cf.t().unsqueeze(0), 50000, 50)

    def eval_occ(q):
        return probe(torch.from_numpy(np.ascontiguousarray(q, dtype=np.float32)).to(DEV)).cpu().numpy()

    ref, n_eval = O.create_volume(eval_occ, pts_ids, 257, step, bmin_pad, 50000)
    seen = ~np.isnan(ref)
    print('R=257 volume: {} voxels evaluated by the product ({} decoder queries), {} by the oracle driver'.format(int(seen.sum()), field.n_queries, n_eval))
    assert np.array_equal(np.isnan(vol), np.isnan(ref))
    assert np.array_equal(vol[seen], ref[seen])
    inner = int(seen[1:-1, 1:-1, 1:-1].sum())
    # (measured on an MI355X: 880 695 visited voxels inside the grid, 880 745 decoder queries, 2 244 531 evaluations by the oracle driver)
    assert 500_000 < inner <= field.n_queries < n_eval and field.n_queries - inner < 0.02 * inner      # every visited voxel decoded once (+ border voxels)


def test_marching_cubes_and_clean_up_on_a_learned_volume_meet_the_specification(trained):
    """f2: the volume of a LEARNED network (abc_mini4 fit) at R = 65 through the device Marching Cubes + clean-up against the independent
    specification of oracle/mesh_oracle.py (vertex set = grid-edge crossings, closed oriented manifold, one cube per face, union-find
    components) -- source/poco_utils.py:95, source/base/mesh.py:7-38."""
    from oracle import mesh_oracle as M
    from ppsurf_amd import reconstruct, mcubes
    root, ckpt = trained
    model = _trained_model(root, ckpt, 65)
    cloud = _abc_cloud(root, 1)
    pts_cf = torch.from_numpy(cloud).to(DEV).t().contiguous()
    lat = model.encode_latents(pts_cf)
    shape = {'pts': pts_cf.unsqueeze(0), 'latents': lat.t().unsqueeze(0)}
    field = reconstruct.OccupancyField(model.network, shape, pts_cf.t().unsqueeze(0), 50000, 50)
    step, bmin_pad, pts_ids = __import__('bench_workloads').grid_geometry(cloud, 65)
    vol = reconstruct.create_volume(field, torch.from_numpy(pts_ids).to(DEV), 65, step, bmin_pad)
    v, f = mcubes.marching_cubes_torch(vol, 0.0)
    assert v.is_cuda
    voln = vol.cpu().numpy()
    dump = os.environ.get('PPS_DUMP_DIR')                          # development aid: keep the volume for a CPU post-mortem
    if dump:
        np.save(os.path.join(dump, 'learned_volume_r65.npy'), voln)
    info = M.check_marching_cubes(v.cpu().numpy(), f.cpu().numpy(), voln, 0.0, require_closed=False)
    assert info['faces'] > 3000 and info['boundary_edges'] < 0.05 * 3 * info['faces']
    v32 = v.to(torch.float32).to(torch.float64)                                # as reconstruct.py hands them to the clean-up (skimage returns float32)
    vc, fc = mcubes.clean_mesh_torch(v32, f, min_component_faces=6)
    M.check_clean_mesh(v32.cpu().numpy(), f.cpu().numpy(), vc.cpu().numpy(), fc.cpu().numpy(), 6)
    print('learned volume R=65: {} faces ({} open edges next to unseen voxels), {} after clean-up'.format(info['faces'], info['boundary_edges'], fc.shape[0]))


RANK_SCRIPT = r'''
import os, sys
sys.path.insert(0, {repo!r}); sys.path.insert(0, os.path.join({repo!r}, 'tests'))
import numpy as np, torch, torch.distributed as dist
from ppsurf_amd import reconstruct, sharding, meshio
from test_gpu_configs import _trained_model
import pathlib
rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
if world > 1:
    dist.init_process_group('gloo', rank=rank, world_size=world)
    sharding.set_query_sharding(True)
root = pathlib.Path({root!r})
model = _trained_model(root, {ckpt!r}, 65)
model.shard_queries = world > 1
cloud = np.load({cloud!r})
pts_cf = torch.from_numpy(cloud).to('cuda:0').t().contiguous()
lat = torch.from_numpy(np.load({lat!r})).to('cuda:0')                      # fixed latents: the latent loop is stochastic
shape = {{'pts': pts_cf.unsqueeze(0), 'latents': lat.t().unsqueeze(0)}}
field = reconstruct.OccupancyField(model.network, shape, pts_cf.t().unsqueeze(0), 50000, 50)
bmin, bmax = cloud.min(), cloud.max(); step = (bmax - bmin) / 64
ids = torch.from_numpy(((cloud - bmin) / step + 1).astype(np.int32).astype(np.int64)).to('cuda:0')
vol = reconstruct.create_volume(field, ids, 65, step, bmin - step).cpu().numpy()
np.save(os.path.join({out!r}, 'vol_w{{}}_r{{}}.npy'.format(world, rank)), vol)
np.save(os.path.join({out!r}, 'nq_w{{}}_r{{}}.npy'.format(world, rank)), np.array([field.n_queries]))
'''


def test_query_sharded_learned_shape_at_r65_equals_the_single_rank_volume(trained, tmp_path):
    """PPS_SHARD=queries on a real shape with learned weights at R = 65: two ranks (gloo, one GPU) decode disjoint query
    ranges of every growth round and exchange them; each rank's volume must EQUAL the single-rank volume."""
    import subprocess, sys
    from golden_util import REPO
    root, ckpt = trained
    model = _trained_model(root, ckpt, 65)
    cloud = _abc_cloud(root, 2)
    lat = model.encode_latents(torch.from_numpy(cloud).to(DEV).t().contiguous())
    np.save(tmp_path / 'cloud.npy', cloud)
    np.save(tmp_path / 'lat.npy', lat.cpu().numpy())
    script = tmp_path / 'run.py'
    script.write_text(RANK_SCRIPT.format(repo=REPO, root=str(root), ckpt=ckpt, cloud=str(tmp_path / 'cloud.npy'), lat=str(tmp_path / 'lat.npy'), out=str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1')
    subprocess.check_call([sys.executable, str(script)], env=env, timeout=900)
    subprocess.check_call([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                           '--master-port', str(29300 + os.getpid() % 300), str(script)], env=env, timeout=900)
    v1 = np.load(tmp_path / 'vol_w1_r0.npy')
    a, b = np.load(tmp_path / 'vol_w2_r0.npy'), np.load(tmp_path / 'vol_w2_r1.npy')
    assert np.array_equal(np.isnan(a), np.isnan(v1)) and np.array_equal(a, b, equal_nan=True)
    assert np.array_equal(a, v1, equal_nan=True)                             # the same kernels on the same queries: equal, not close
    n1 = int(np.load(tmp_path / 'nq_w1_r0.npy')[0])
    n2 = [int(np.load(tmp_path / 'nq_w2_r{}.npy'.format(r))[0]) for r in (0, 1)]
    assert (~np.isnan(v1)).sum() > 50_000 and sum(n2) == n1 and min(n2) > 0.3 * n1


def test_config5_whole_reconstruction_r513_n250k_p200_smoke():
    """BASELINE config 5 end to end on one GPU: a 250 000-point synthetic cloud, P = 200, rec_batch_size 25 000, R = 513 -- latent loop (250
    encoder passes), region growing over the (515)^3 float64 volume, Marching Cubes, clean-up, 10 refinement rounds, every query decoded
    by the real kernels (growth steered by the analytic shape: formula-filled weights describe no surface).  Finite, closed mesh."""
    import bench_workloads as workloads
    model = workloads.make_model(513, 200, 25000, DEV)
    r = workloads.reconstruct_steered(model, 250_000, seed=5, device=DEV, return_mesh=True)
    v, f = r['mesh']
    two, open_e, multi = _closed_manifold_stats(f)
    print('config 5 reconstruction: {:.1f} s, {} decoder queries, {} vertices, {} faces; edges with 2 faces {:.5f}, open {}, >2 {}'.format(
        r['total_s'], r['decoder_queries'], v.shape[0], f.shape[0], two, open_e, multi))
    assert np.isfinite(v).all() and v.shape[0] > 500_000 and f.shape[0] > 1_000_000
    assert r['decoder_queries'] > 10_000_000
    # closed 2-manifold -- except where the band of the cloud's extreme points is clipped by the volume border, whose voxels the driver
    # overwrites with out_value = 1 like the reference (poco_utils.py:248-253): six little sheets between the border and the evaluated
    # band, ~100 open edges each whatever the resolution (the +-2 dilation footprint)
    assert multi == 0 and open_e <= 1200 and two > 0.999
    assert np.abs(np.linalg.norm(v, axis=1) - 0.45).max() < 0.2              # a bumpy sphere of radius 0.45 (synthetic.make_cloud)
