"""BASELINE config 3 at its full batch size, pinned on the reference: ONE training step's forward / loss / backward of the
reference's PPSurfNetwork in train() on 10 shapes x 10 000 points x 2000 queries (tests/golden/cases_full.py), on the CPU.

    python tests/golden/make_golden_train_full.py      # build container only (needs /root/reference, ~35 GB of RAM, ~10 minutes on 8 cores)
                                                       # -> tests/golden/train_ppsurf_full.npz

Stored: logits [10,2,2000], loss and buffer signatures from the fp32 run; parameter-gradient signatures and a seeded sample of <= 1024 entries of
every gradient tensor from a float64 run of the same modules (as in the small fixtures; its two big branches are re-evaluated during backward
with torch.utils.checkpoint so that the pass fits the build container's 62 GB) and from the reference's own fp32 run; digests of the id tables the batch was built
with (the test rebuilds the batch from seeds and checks them).  Dropout is off (p = 0), as in the small fixture train_ppsurf.npz.
The id tables come from the oracle's exact kNN; this script asserts that the reference's own `knn` (source/poco_utils.py:257-273 on the kd-tree
stand-in) returns the same tables on this batch, i.e. that the batch is what the reference's data loader would have produced."""
import copy
import gc
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
import make_golden_train as mt  # noqa: E402
import cases_full as cf  # noqa: E402

mt.SAMPLE = 1024


def main():
    data0, occ = cf.full_fit_batch()
    # the reference's own search on the same clouds: same tables
    for name in ('ids00', 'ids11', 'ids01', 'ids10', 'proj_ids'):
        if name == 'proj_ids':
            ref = mg.ref_knn(data0['pts'], data0['pts_query'], 64)
        else:
            a, b = int(name[3]), int(name[4])
            lv = lambda i: data0['pts'] if i == 0 else data0['support{}'.format(i)]
            ref = mg.ref_knn(lv(a), lv(b), 16 if a <= b else 1)
        assert torch.equal(ref, data0[name]), name + ': the reference kNN disagrees with the oracle tables'
    print('id tables agree with the reference kNN')

    def run(net, dt):
        for p in net.parameters():
            p.grad = None
        for m in net.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        d = {k: (v.to(dt) if v.is_floating_point() else v.clone()) for k, v in data0.items()}
        logits = net.forward(d)
        loss = torch.nn.functional.cross_entropy(logits, occ, reduction='none').mean()
        loss.backward()
        return logits.detach().clone(), loss.detach().clone()

    net = mg.PPSurfNetwork(in_channels=3, latent_size=256, out_channels=2, k=64, num_pts_local=cf.P, pointnet_latent_size=256)
    dg = mt.load_filled_train(net, '')
    logits, loss = run(net, torch.float32)
    print('fp32 step done: loss', float(loss))
    gn, gs32, unused = mt.grad_table(net)
    bn, bs = mt.buffer_table(net)
    gc.collect()
    if os.environ.get('PPS_GOLDEN_FP64', '1') == '1':
        # the float64 pass (the well-defined target, see make_golden_train.py); PPS_GOLDEN_FP64=0 stores the reference's fp32 gradients in both slots
        net64 = copy.deepcopy(net).double()
        for p in net64.parameters():
            p.grad = None
        # the two big branches recomputed during backward (torch.utils.checkpoint, same arithmetic): PointNet (10^6 patch rows) and the
        # interpolation head (1.28 x 10^6 rows x 256 channels): ~30 GB at the peak instead of > 60.  Not the encoder: its layers move norm_radius
        # in place in train() (source/base/nn.py:608-613), a second evaluation would see another radius.
        from torch.utils.checkpoint import checkpoint
        for sub in (net64.point_net, net64.projection):
            orig = sub.forward
            sub.forward = (lambda f: (lambda *a, **k: checkpoint(f, *a, use_reentrant=False, **k)))(orig)
        run(net64, torch.float64)
        print('fp64 step done')
        gn, gs, unused = mt.grad_table(net64)
        samp = mt.grad_samples(net64, net, seed=2026)
    else:
        # gradients of the reference's own fp32 run in both slots: `gsamp_val` == `gsamp_val32`, `gsigs` == `gsigs32`
        gs = gs32
        samp = mt.grad_samples(net, net, seed=2026)
    mg.save('train_ppsurf_full', digest=dg, logits=logits, loss=loss, gnames=gn, gsigs=gs, gsigs32=gs32, unused=unused, bnames=bn, bsigs=bs,
            table_names=np.array(list(cf.table_digests(data0).keys())), table_digests=np.array(list(cf.table_digests(data0).values())),
            occ_digest=np.array(cf.digest(occ)), patches_digest=np.array(cf.digest(data0['pts_local_ps'])), **samp)


if __name__ == '__main__':
    torch.manual_seed(0)
    torch.set_num_threads(8)
    main()
