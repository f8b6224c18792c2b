"""Training graph (ppsurf_amd/train_graph.py) against the reference's train-mode fixtures, on the CPU.

The three HIP neighbourhood ops are replaced by their torch twins (tests/train_ref_ops.py); everything else -- layer order,
BatchNorm batch statistics and running-stat updates, InstanceNorm, norm_radius EMA, dropout, the loss -- is the product code.
Outputs / loss / buffers are compared with the reference's fp32 run (<= 5e-5 of the output scale).  Gradients are compared with
the reference's float64 run twice: with this graph evaluated in float64 (<= 1e-7: the graph is the same function) and in fp32
(the precision the product trains in; tolerance stated per test)."""
import numpy as np
import pytest
import torch
from torch import nn

from golden_util import load_golden, filled_sd, sd_digest
from train_ref_ops import patched
from ppsurf_amd import modules, train_graph as tg


def _t(a):
    return torch.from_numpy(np.asarray(a))


def _load(module, prefix, key=None, digest=None):
    sd = filled_sd(prefix, key)
    if digest is not None:
        assert sd_digest(sd) == str(digest)
    module.load_state_dict({k[len(prefix):]: v for k, v in sd.items()})
    return module.train()


def _close(a, b, rtol, what, floor=0.0, flips=False):
    """max |a-b| <= rtol * max(max|b|, floor).  `floor` = scale of the largest gradient of the module: biases in front of a
    train-mode BatchNorm have an analytically ZERO gradient, what is recorded for them is rounding noise of that scale."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    scale = max(np.abs(b).max(), floor, 1e-30)
    err = np.abs(a - b) / scale
    if flips:
        # a ReLU whose pre-activation is within rounding of 0 takes the other branch in one of the two fp32 evaluations
        # (fp64 replay sides with this graph): allow <= 0.5 % of the entries (one flip touches a whole row) to be off by a few percent of the scale
        assert (err > rtol).mean() <= 5e-3 and err.max() <= 5e-2, '{}: {} entries off, max {:.3e}'.format(what, int((err > rtol).sum()), err.max())
        return
    assert err.max() <= rtol, '{}: max err {:.3e} of scale {:.3e}'.format(what, err.max(), scale)


def _sig(t):
    t = t.detach().double().reshape(-1)
    head = torch.zeros(32, dtype=torch.float64)
    head[:min(32, t.numel())] = t[:32]
    return torch.cat([torch.stack([t.sum(), t.abs().sum(), t.norm()]), head]).numpy()


def _check_sigs(named, names, sigs, rtol, what, noise=1e-6, sigs32=None):
    """Signatures [sum, sum|.|, l2, first 32 entries] of every tensor against the reference's float64 evaluation.
    Tolerance per tensor: `rtol` relative, plus an absolute floor of `noise` x the largest l2 norm of the set (biases in front
    of a train-mode BatchNorm have an analytically ZERO gradient), and -- when the fixture also holds the reference's own fp32
    signatures -- 3x the deviation of the reference's fp32 run from its fp64 run (an fp32 backward through ~60 layers is only
    that accurate; this graph is held to the same accuracy as the reference itself)."""
    named = dict(named)
    floor = noise * max(float(s[2]) for s in sigs)
    for i, (k, ref) in enumerate(zip(names, sigs)):
        t = named[str(k)]
        got = _sig(t)
        own = np.abs(sigs32[i] - ref) * 3 if sigs32 is not None else np.zeros_like(ref)
        assert abs(got[1] - ref[1]) <= rtol * ref[1] + floor * np.sqrt(t.numel()) + own[1], '{} {}: sum|.| {} vs {}'.format(what, k, got[1], ref[1])
        assert abs(got[2] - ref[2]) <= rtol * ref[2] + floor + own[2], '{} {}: l2 {} vs {}'.format(what, k, got[2], ref[2])
        rms = ref[2] / np.sqrt(t.numel())
        scale = max(np.abs(ref[3:]).max(), rms)
        err = np.abs(got[3:] - ref[3:])
        tol = 4 * rtol * scale + floor + own[3:].max()
        # ONE of the 32 leading entries may sit up to 10 % of the tensor's scale off: a ReLU whose pre-activation is within fp32 rounding of zero
        # takes the other branch when a sum is evaluated in another order (e.g. another grid of the statistics kernels), and the unit's whole
        # contribution appears in / vanishes from single gradient entries (cf. _close(flips=True); the l2 and sum|.| bars above hold regardless)
        assert (err > tol).sum() <= 1 and err.max() <= max(tol, 0.1 * scale), '{} {}: leading entries differ'.format(what, k)


def _check_samples(named, g, rtol, what, noise=1e-6, use32=True):
    """Elementwise comparison of every parameter gradient at the fixture's seeded sample of <= 4096 flat indices per tensor (`gsamp_*`,
    recorded from the reference's float64 run; tests/golden/make_golden_train.py) -- reaches the tail of every tensor, unlike the
    signatures.  Tolerance per tensor as for the leading entries in _check_sigs: 4 x rtol x the tensor's scale, the noise floor of the
    set, and (fp32 runs) 3 x the largest deviation of the reference's OWN fp32 gradient from its float64 gradient on that sample."""
    named = dict(named)
    names, off = [str(k) for k in g['gnames']], g['gsamp_off']
    assert len(off) == len(names) + 1
    floor = noise * max(float(s[2]) for s in g['gsigs'])
    worst = (0.0, None)
    for i, k in enumerate(names):
        sl = slice(int(off[i]), int(off[i + 1]))
        idx, ref = g['gsamp_idx'][sl], g['gsamp_val'][sl]
        t = named[k].detach().double().reshape(-1)
        assert idx.shape[0] == min(4096, t.numel()) and (idx.max() < t.numel())
        got = t[torch.from_numpy(idx)].cpu().numpy()
        own = 3 * float(np.abs(g['gsamp_val32'][sl] - ref).max()) if use32 else 0.0
        rms = float(g['gsigs'][i][2]) / np.sqrt(t.numel())
        tol = 4 * rtol * max(float(np.abs(ref).max()), rms) + floor + own
        errs = np.abs(got - ref)
        flipped = errs > tol                         # (ReLU flips, see _check_sigs: <= 0.5 % of a tensor's sampled entries, bounded by 10 % of its scale)
        err = float(errs[~flipped].max()) if (~flipped).any() else 0.0
        assert flipped.mean() <= 5e-3 and float(errs.max()) <= max(tol, 0.1 * max(float(np.abs(ref).max()), rms)), \
            '{} {}: sampled entries differ by {} (tolerance {}, worst at flat index {})'.format(what, k, float(errs.max()), tol, int(idx[np.argmax(errs)]))
        if tol > 0 and err / tol > worst[0]:
            worst = (err / tol, k)
    return worst


@pytest.mark.parametrize('act', ['relu', 'silu'])
def test_fkaconv_layer_train(act):
    g = load_golden('train_fkaconv_layer')
    layer = _load(modules.FKAConvLayer(8, 16, 16, activation=nn.SiLU() if act == 'silu' else nn.ReLU()), 'L_{}.'.format(act),
                  digest=g['digest_' + act])
    x = _t(g['x']).transpose(1, 2).contiguous().requires_grad_(True)
    pts, sup = _t(g['pts']).transpose(1, 2).contiguous(), _t(g['sup']).transpose(1, 2).contiguous()
    with patched():
        out = tg.fkaconv_layer(layer, x, pts, sup, _t(g['ids']))
        (out * _t(g['r']).transpose(1, 2)).sum().backward()
    _close(out.detach().transpose(1, 2), g['out_' + act], 2e-5, 'out')
    _close(layer.norm_radius, g['norm_radius_' + act], 1e-6, 'norm_radius')
    _close(x.grad.transpose(1, 2), g['gx_' + act], 1e-4, 'grad x')
    for k, p in layer.named_parameters():
        _close(p.grad, g['g_{}_{}'.format(act, k)], 1e-4, 'grad ' + k)


def test_residual_block_train():
    g = load_golden('train_residual_block')
    blk = _load(modules.ResidualBlock(16, 32, 16, activation=nn.SiLU()), 'RB_down.', digest=g['digest'])
    x = _t(g['x']).transpose(1, 2).contiguous().requires_grad_(True)
    pts, sup = _t(g['pts']).transpose(1, 2).contiguous(), _t(g['sup']).transpose(1, 2).contiguous()
    with patched():
        out = tg.residual_block(blk, x, pts, sup, _t(g['ids']))
        (out * _t(g['r']).transpose(1, 2)).sum().backward()
    _close(out.detach().transpose(1, 2), g['out'], 2e-5, 'out')
    _close(x.grad.transpose(1, 2), g['gx'], 1e-4, 'grad x')
    top = max(np.abs(g['g_' + k]).max() for k, _ in blk.named_parameters())
    for k, p in blk.named_parameters():
        _close(p.grad, g['g_' + k], 2e-4, 'grad ' + k, floor=1e-2 * top)
    for k, b in blk.named_buffers():
        _close(b.double(), g['b_' + k], 1e-5, 'buffer ' + k)


@pytest.mark.parametrize('p', [10, 50])
def test_pointnet_train(p):
    g = load_golden('train_pointnet')
    net = _load(modules.PointNetfeat(net_size_max=256, num_points=p, use_point_stn=False, use_feat_stn=True, output_size=256,
                                     sym_op='att', dim=3), 'PN_p{}.'.format(p), digest=g['digest_p{}'.format(p)])
    x = _t(g['p{}_x'.format(p)]).transpose(1, 2).contiguous().requires_grad_(True)
    with patched():
        feat, trans2 = tg.pointnet(net, x)
        (feat * _t(g['p{}_r'.format(p)])).sum().backward()
    _close(feat.detach(), g['p{}_feat'.format(p)], 2e-5, 'feat')
    _close(trans2[:4].detach(), g['p{}_trans2'.format(p)], 2e-5, 'trans2')
    _close(x.grad.transpose(1, 2), g['p{}_gx'.format(p)], 2e-4, 'grad x')
    _check_sigs([(k, v.grad) for k, v in net.named_parameters()], g['p{}_gnames'.format(p)], g['p{}_gsigs'.format(p)], 2e-4, 'grad')
    _check_sigs([(k, v.float()) for k, v in net.named_buffers()], g['p{}_bnames'.format(p)], g['p{}_bsigs'.format(p)], 1e-5, 'buffer')


@pytest.mark.parametrize('tag,c,cout,k', [('c32', 32, 2, 16), ('c256', 256, 256, 64)])
def test_interp_attention_grads(tag, c, cout, k):
    from ppsurf_amd.synthetic import make_latents
    g = load_golden('train_interp_attention')
    net = _load(modules.InterpAttentionKHeadsNet(c, cout, k), 'IA_{}.'.format(tag), digest=g['digest_' + tag])
    n = g[tag + '_pts'].shape[2]
    lat = _t(make_latents(c, n, seed=c)).transpose(1, 2).contiguous().requires_grad_(True)
    pts, q = _t(g[tag + '_pts']).transpose(1, 2).contiguous(), _t(g[tag + '_query']).transpose(1, 2).contiguous()
    with patched():
        out = tg.interp_attention(net, lat, pts, q, _t(g[tag + '_ids']))
        (out * _t(g[tag + '_r']).transpose(1, 2)).sum().backward()
    _close(out.detach().transpose(1, 2), g[tag + '_out'], 2e-5, 'out')
    _close(lat.grad.transpose(1, 2), g[tag + '_glat'], 1e-4, 'grad latents', flips=True)
    _check_sigs([(k_, v.grad) for k_, v in net.named_parameters()], g[tag + '_gnames'], g[tag + '_gsigs'], 1e-4, 'grad')


def _step_inputs(g, dt=torch.float32):
    data = {k[3:]: _t(v) for k, v in g.items() if k.startswith('in_')}
    data = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in data.items()}
    return data, data['occ']


def _ppsurf(dt):
    g = load_golden('train_ppsurf')
    net = _load(modules.PPSurfNetwork(in_channels=3, latent_size=256, out_channels=2, k=64, num_pts_local=50,
                                      pointnet_latent_size=256), '', key='ppsurf', digest=g['digest']).to(dt)
    for m in net.modules():
        if isinstance(m, nn.Dropout):
            m.p = 0.0
    return g, net


def _poco(dt):
    g = load_golden('train_poco')
    net = _load(modules.PocoNetwork(in_channels=3, latent_size=32, out_channels=2, k=64), 'POCO.', key='poco', digest=g['digest']).to(dt)
    return g, net


def _run_step(forward, net, dt):
    data, occ = _step_inputs(load_golden('train_ppsurf'), dt)
    with patched():
        logits = forward(net, data, data['proj_ids'])
        loss = nn.functional.cross_entropy(logits, occ, reduction='none').mean()
        loss.backward()
    return logits.detach(), float(loss.detach())


@pytest.mark.parametrize('which', ['ppsurf', 'poco'])
def test_training_step_fp32(which):
    """One training step's forward/backward in fp32: logits, loss, updated buffers vs the reference's fp32 run; gradients vs
    its fp64 run within 1 % per tensor (+ the self-calibrating terms of _check_sigs): fp32 summation noise of a ~60-layer
    backward pass -- the exactness of the graph is established by the float64 test below."""
    g, net = (_ppsurf if which == 'ppsurf' else _poco)(torch.float32)
    logits, loss = _run_step(tg.ppsurf_forward if which == 'ppsurf' else tg.poco_forward, net, torch.float32)
    _close(logits, g['logits'], 5e-5, 'logits')
    assert abs(loss - float(g['loss'])) < 1e-5
    assert [k for k, p in net.named_parameters() if p.grad is None] == [str(k) for k in g['unused']]
    _check_sigs([(k, v.grad) for k, v in net.named_parameters()], g['gnames'], g['gsigs'], 1e-2, 'grad', sigs32=g['gsigs32'])
    _check_samples([(k, v.grad) for k, v in net.named_parameters() if v.grad is not None], g, 1e-2, 'grad')
    _check_sigs([(k, v.float()) for k, v in net.named_buffers()], g['bnames'], g['bsigs'], 2e-5, 'buffer')


@pytest.mark.parametrize('which', ['ppsurf', 'poco'])
def test_training_step_fp64_is_the_same_function(which):
    g, net = (_ppsurf if which == 'ppsurf' else _poco)(torch.float64)
    logits, loss = _run_step(tg.ppsurf_forward if which == 'ppsurf' else tg.poco_forward, net, torch.float64)
    _close(logits, g['logits'], 5e-5, 'logits')                      # recorded in fp32
    _check_sigs([(k, v.grad) for k, v in net.named_parameters()], g['gnames'], g['gsigs'], 1e-7, 'grad', noise=1e-10)
    # every sampled entry of every one of the parameter tensors, tails included: the float64 graph IS the reference's function
    _check_samples([(k, v.grad) for k, v in net.named_parameters() if v.grad is not None], g, 1e-7, 'grad', noise=1e-10, use32=False)


def test_ppsurf_training_step_dropout_same_generator():
    """With the CPU generator seeded like the recording run, F.dropout draws the same masks (same shapes, same order)."""
    g = load_golden('train_ppsurf')
    net = _load(modules.PPSurfNetwork(in_channels=3, latent_size=256, out_channels=2, k=64, num_pts_local=50,
                                      pointnet_latent_size=256), '', key='ppsurf')
    data, occ = _step_inputs(g)
    with patched():
        torch.manual_seed(123)
        logits = tg.ppsurf_forward(net, data, data['proj_ids'])
    loss = nn.functional.cross_entropy(logits, occ, reduction='none').mean()
    _close(logits.detach(), g['logits_dropout'], 5e-5, 'logits')
    assert abs(float(loss.detach()) - float(g['loss_dropout'])) < 1e-5


@pytest.mark.parametrize('rows,want', [(100000, 32), (25000, 8), (20000, 16), (6250, 5), (1000000, 64), (1280000, 64), (1560, 0), (40000, 32), (33000, 8),
                                       (32771, 64), (999, 0)])
def test_split_k_slab_choice(rows, want):
    from ppsurf_amd import train_graph
    s = train_graph.splitk_slabs(rows)
    assert s == want
    assert s == 0 or rows // s >= 500


@pytest.mark.parametrize('which', ['ppsurf', 'poco'])
def test_staged_backward_gives_the_same_gradients_and_completes_them_stage_by_stage(which):
    """train_graph.staged(): the encoder cuts the backward pass behind resnetb31 and behind resnetb11 (BackwardStages).  The three stages run the same
    chain rule: the gradients equal those of a single loss.backward() to fp32 re-association (the sum over the consumers of a skip-connected
    activation is taken in another order); and the gradients a stage is said to
    complete (train_graph.parameter_stages: the gradient buckets of a multi-rank fit are exactly these groups, all-reduced between the stages) are
    final when that stage returns -- later stages add nothing to them."""
    g, net = (_ppsurf if which == 'ppsurf' else _poco)(torch.float32)
    forward = tg.ppsurf_forward if which == 'ppsurf' else tg.poco_forward
    data, occ = _step_inputs(load_golden('train_ppsurf'))
    state = {k: v.clone() for k, v in net.state_dict().items()}
    with patched():
        logits = forward(net, data, data['proj_ids'])
        nn.functional.cross_entropy(logits, occ, reduction='none').mean().backward()
    want = {k: (None if p.grad is None else p.grad.clone()) for k, p in net.named_parameters()}
    net.load_state_dict(state)                                  # the forward pass moved norm_radius / running statistics
    for p in net.parameters():
        p.grad = None
    groups = tg.parameter_stages(net)
    assert len(groups) == tg.N_STAGES == 3 and sum(len(x) for x in groups) == sum(1 for p in net.parameters() if p.requires_grad)
    names = {id(p): k for k, p in net.named_parameters()}
    assert all(names[id(p)].startswith(('encoder.cv0.', 'encoder.bn0.', 'encoder.resnetb01', 'encoder.resnetb10', 'encoder.resnetb11')) for p in groups[2])
    assert all(names[id(p)].startswith(('encoder.resnetb2', 'encoder.resnetb3')) for p in groups[1])
    share = sum(p.numel() for p in groups[0]) / sum(p.numel() for x in groups for p in x)
    assert share > 0.75                                          # most gradient bytes are complete after the first stage
    snapshots = []

    def after_stage(k):
        snapshots.append({names[id(p)]: (None if p.grad is None else p.grad.clone()) for x in groups[:k + 1] for p in x})
        if k < 2:                                                # nothing of the later stages has arrived yet
            assert all(p.grad is None for x in groups[k + 1:] for p in x), k

    with patched(), tg.staged() as st:
        logits2 = forward(net, data, data['proj_ids'])
        loss = nn.functional.cross_entropy(logits2, occ, reduction='none').mean()
        assert st.n_stages == 3
        st.backward(loss, after_stage)
    assert torch.equal(logits2, logits)
    worst, gmax = 0.0, max(float(w.abs().max()) for w in want.values() if w is not None)
    for k, p in net.named_parameters():
        assert (p.grad is None) == (want[k] is None), k
        if want[k] is None:
            continue
        if True:
            # behind the last cut nothing changes (the same kernels in the same order: bit-identical on the GPU, tests/test_gpu_train.py; torch's CPU
            # reductions are multi-threaded and differ from run to run by themselves, so the CPU suite compares everything to rounding).  In front of a cut the gradient of a skip-connected activation (x0 .. x3 feed a block AND the up-sampling head) is summed in another
            # order than autograd's single pass sums it: fp32 re-association, a few ulp of the tensor's scale
            # (biases in front of a train-mode BatchNorm have an analytically zero gradient: what is there is rounding noise of the set's scale)
            err = float((p.grad - want[k]).abs().max()) / max(float(want[k].abs().max()), 1e-3 * gmax)
            worst = max(worst, err)
            assert err < 5e-5, (k, err)
    final = {k: (None if p.grad is None else p.grad.clone()) for k, p in net.named_parameters()}
    for snap in snapshots:                                       # a stage's gradients were FINAL when it returned: later stages add nothing
        for k, gk in snap.items():
            assert (gk is None and final[k] is None) or torch.equal(gk, final[k]), k
    # outside staged() the cuts are no-ops (single backward, e.g. the one-GPU whole-step graph)
    assert tg._stages[0] is None and tg._cut(logits)[0] is logits


def test_bias_grad_of_a_host_tensor_is_the_column_sum():
    """train_graph._bias_grad: the HIP reduction takes device tensors only; anything else is torch's sum in fp32 (the torch twins of this suite)."""
    from ppsurf_amd import train_graph
    g = torch.randn(37, 2050).to(torch.bfloat16)
    out = train_graph._bias_grad(g)
    assert out.dtype == torch.float32 and torch.equal(out, g.sum(0, dtype=torch.float32))


def test_the_collector_is_off_during_a_recording_and_back_afterwards():
    """fit._no_gc_during_capture: collects, disables, restores the state it found (also when the body raises)."""
    import gc
    from ppsurf_amd import fit
    assert gc.isenabled()
    with fit._no_gc_during_capture():
        assert not gc.isenabled()
    assert gc.isenabled()
    try:
        with fit._no_gc_during_capture():
            raise RuntimeError('x')
    except RuntimeError:
        pass
    assert gc.isenabled()
    gc.disable()
    try:
        with fit._no_gc_during_capture():
            assert not gc.isenabled()
        assert not gc.isenabled()
    finally:
        gc.enable()


def test_affine_rows_and_sum_rows_on_the_host_are_plain_autograd():
    """train_ops.affine_rows / sum_rows (round 6: the step's row reductions go through the HIP reduction on the device) keep torch semantics on host
    tensors: same value, same gradients as x * scale + shift."""
    from ppsurf_amd import train_ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(37, 12, generator=g, dtype=torch.float64, requires_grad=True)
    sc = torch.randn(12, generator=g, dtype=torch.float64, requires_grad=True)
    sh = torch.randn(12, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(37, 12, generator=g, dtype=torch.float64)
    y = train_ops.affine_rows(x, sc, sh)
    gx, gs, gh = torch.autograd.grad((y * w).sum(), (x, sc, sh))
    x2, sc2, sh2 = (t.detach().clone().requires_grad_(True) for t in (x, sc, sh))
    rx, rs, rh = torch.autograd.grad(((x2 * sc2 + sh2) * w).sum(), (x2, sc2, sh2))
    assert torch.equal(y, x * sc + sh) and torch.allclose(gx, rx) and torch.allclose(gs, rs, atol=1e-12) and torch.allclose(gh, rh, atol=1e-12)
    assert torch.allclose(train_ops.sum_rows(w.float()), w.float().sum(0)) and train_ops.sum_rows(w.float()).dtype == torch.float32


def test_batch_norm_add_relu_off_the_device():
    """train_graph.batch_norm_add_relu (the residual blocks' tail, nn.py:447-450) on host tensors: eval() mode = relu(bn(x) + res) by torch ops on
    the running statistics; train() mode is refused like every train op (the product has no CPU path)."""
    import pytest
    from ppsurf_amd import train_graph, _lib
    torch.manual_seed(0)
    bn = torch.nn.BatchNorm1d(8)
    bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2.0)
    x, res = torch.randn(50, 8), torch.randn(50, 8)
    ref_bn = torch.nn.BatchNorm1d(8)
    ref_bn.load_state_dict(bn.state_dict())
    y = train_graph.batch_norm_add_relu(bn.eval(), x, res)
    assert torch.allclose(y, torch.relu(ref_bn.eval()(x) + res)) and int(bn.num_batches_tracked) == 0
    with pytest.raises(_lib.PpsError, match='no CPU path'):
        train_graph.batch_norm_add_relu(bn.train(), x, res)
