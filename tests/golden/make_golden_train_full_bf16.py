"""The like-for-like yardstick of the bf16-mixed fit step: the REFERENCE's own step on the config-3 batch under
`torch.autocast('cpu', dtype=torch.bfloat16)` -- the precision mode its trainer runs (configs/poco.yaml:10 `precision: 16-mixed`; Lightning turns
that into bf16 autocast on the CPU) -- measured against its own fp32 / float64 step stored in train_ppsurf_full.npz.

    python tests/golden/make_golden_train_full_bf16.py     # build container only (needs /root/reference and train_ppsurf_full.npz)
                                                           # -> tests/golden/train_ppsurf_full_bf16ref.npz  (a few kB)

Stored: what 16-bit autocast arithmetic costs the REFERENCE on this batch -- max |logits - fp32 logits|, |loss - fp32 loss|, and per gradient tensor
the cosine and length ratio of its sampled entries (the 1024 indices of train_ppsurf_full.npz) against the float64 gradient.  The GPU test holds
the build's bf16-mixed step to a small multiple of these numbers instead of constants fitted to its own first run."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
import make_golden_train as mt  # noqa: E402
import cases_full as cf  # noqa: E402


def main():
    g = np.load(os.path.join(HERE, 'train_ppsurf_full.npz'), allow_pickle=False)
    data0, occ = cf.full_fit_batch()
    assert cf.digest(occ) == str(g['occ_digest']) and cf.digest(data0['pts_local_ps']) == str(g['patches_digest'])
    net = mg.PPSurfNetwork(in_channels=3, latent_size=256, out_channels=2, k=64, num_pts_local=cf.P, pointnet_latent_size=256)
    dg = mt.load_filled_train(net, '')
    assert str(dg) == str(g['digest'])
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    d = {k: v.clone() for k, v in data0.items()}
    with torch.autocast('cpu', dtype=torch.bfloat16):
        logits = net.forward(d)
        loss = torch.nn.functional.cross_entropy(logits.float(), occ, reduction='none').mean()
    loss.backward()
    print('reference bf16-autocast step done: loss', float(loss), 'logits dtype', logits.dtype, flush=True)
    err_logits = float((logits.detach().float() - torch.from_numpy(g['logits'])).abs().max())
    err_loss = abs(float(loss.detach()) - float(g['loss']))
    names, off = [str(k) for k in g['gnames']], g['gsamp_off']
    named = {k: p.grad for k, p in net.named_parameters() if p.grad is not None}
    assert [k for k, p in net.named_parameters() if p.grad is not None] == names
    cos, ratio = np.zeros(len(names)), np.zeros(len(names))
    for i, k in enumerate(names):
        sl = slice(int(off[i]), int(off[i + 1]))
        ref = g['gsamp_val'][sl]
        got = named[k].detach().double().reshape(-1)[torch.from_numpy(g['gsamp_idx'][sl])].numpy()
        nr, ng = np.linalg.norm(ref), np.linalg.norm(got)
        cos[i] = float(got @ ref / (ng * nr)) if nr > 0 and ng > 0 else np.nan
        ratio[i] = float(ng / nr) if nr > 0 else np.nan
    mg.save('train_ppsurf_full_bf16ref', digest=dg, err_logits=np.float64(err_logits), err_loss=np.float64(err_loss),
            logits_scale=np.float64(np.abs(g['logits']).max()), gnames=np.array(names), cos=cos, ratio=ratio,
            torch_version=np.array(torch.__version__), autocast=np.array("torch.autocast('cpu', dtype=torch.bfloat16)"))
    top = max(float(s[2]) for s in g['gsigs'])
    big = [i for i in range(len(names)) if float(g['gsigs'][i][2]) >= 1e-3 * top and np.isfinite(cos[i])]
    print('logits {:.4f} of scale {:.2f}, loss {:.2e}; {} tensors above the size threshold: lowest cosine {:.4f} ({}), median {:.4f}, ratio {:.3f} .. {:.3f}'.format(
        err_logits, float(np.abs(g['logits']).max()), err_loss, len(big), min(cos[i] for i in big), names[min(big, key=lambda i: cos[i])],
        float(np.median([cos[i] for i in big])), min(ratio[i] for i in big), max(ratio[i] for i in big)))


if __name__ == '__main__':
    torch.manual_seed(0)
    torch.set_num_threads(8)
    main()
