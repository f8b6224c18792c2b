"""BASELINE config 3 at its full batch size, pinned on the reference: ONE training step's forward / loss / backward of the
reference's PPSurfNetwork in train() on 10 shapes x 10 000 points x 2000 queries (tests/golden/cases_full.py), on the CPU.

    python tests/golden/make_golden_train_full.py      # build container only (needs /root/reference, ~45 GB of RAM for the float64 pass)
                                                       # -> tests/golden/train_ppsurf_full.npz

Stored: logits [10,2,2000], loss and buffer signatures from the fp32 run; parameter-gradient signatures from the float64 run and from the
reference's own fp32 run, a seeded sample of <= 1024 entries of every gradient tensor from both; digests of the id tables the batch was built
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
    net64 = copy.deepcopy(net).double()
    logits, loss = run(net, torch.float32)
    print('fp32 step done: loss', float(loss))
    _, gs32, _ = mt.grad_table(net)
    bn, bs = mt.buffer_table(net)
    gc.collect()
    run(net64, torch.float64)
    print('fp64 step done')
    gn, gs, unused = mt.grad_table(net64)
    samp = mt.grad_samples(net64, net, seed=2026)
    mg.save('train_ppsurf_full', digest=dg, logits=logits, loss=loss, gnames=gn, gsigs=gs, gsigs32=gs32, unused=unused, bnames=bn, bsigs=bs,
            table_names=np.array(list(cf.table_digests(data0).keys())), table_digests=np.array(list(cf.table_digests(data0).values())),
            occ_digest=np.array(cf.digest(occ)), patches_digest=np.array(cf.digest(data0['pts_local_ps'])), **samp)


if __name__ == '__main__':
    torch.manual_seed(0)
    torch.set_num_threads(8)
    main()
