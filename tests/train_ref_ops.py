"""TEST INFRASTRUCTURE: plain-torch twins of ppsurf_amd.train_ops (gradients by torch autograd).

The product ops are HIP-only.  The CPU suite patches these in so that the surrounding training graph
(ppsurf_amd/train_graph.py) can be checked against the reference's train-mode fixtures without a GPU; the GPU suite
compares the HIP ops (forward and backward) with these twins."""
import contextlib

import torch


def gather_rows(x, idx):
    return x[idx]


def neighbour_max(x, idx):
    return x[idx.reshape(-1)].view(idx.shape[0], idx.shape[1], -1).max(dim=1)[0]


def neighbour_contract(x, idx, g):
    xg = x[idx.reshape(-1)].view(idx.shape[0], idx.shape[1], -1)             # [m,k,c]
    return torch.einsum('mkc,mkt->mct', xg, g).reshape(idx.shape[0], -1)


def fka_geometry(geo, pts, sup, idx, b, m, momentum, owned=False):
    """source/base/nn.py:601-643 in torch ops, on the packed parameter vector of train_graph.pack_geo.
    pts [rows,3], sup [b*m,3], idx [b*m,k] -> (g [b*m,k,16], norm_radius [1])."""
    import torch.nn.functional as F
    k = idx.shape[1]
    radius, alpha, beta, act_id = geo[0], geo[1], geo[2], float(geo[3].detach())
    w1, w2, w3 = geo[4:52].view(16, 3), geo[52:564].view(16, 32), geo[564:1076].view(16, 32)
    in1w, in1b, in2w, in2b = geo[1076:1092], geo[1092:1108], geo[1108:1124], geo[1124:1140]
    act = F.silu if act_id == 2.0 else F.relu
    pn = pts[idx.reshape(-1)].view(b, m, k, 3) - sup.view(b, m, 1, 3)                    # :597,601
    dist = torch.sqrt((pn.detach() ** 2).sum(-1))                                        # :605
    radius = radius.detach()
    if momentum > 0:
        radius = radius * (1 - momentum) + dist.max(2)[0].mean() * momentum              # :608-613
    pn = pn / radius                                                                     # :616
    dw = torch.sigmoid(-alpha * dist + beta)
    s = dw.sum(2, keepdim=True)
    s = s + (s == 0) + 1e-6
    dw = (dw / s * k).unsqueeze(-1)                                                      # :619-624

    def inorm(z, w, bias):                                                               # InstanceNorm2d(affine): stats over (M,K) per shape
        if k == 1:
            return z                                                                     # :627-630,635-638
        mean = z.mean(dim=(1, 2), keepdim=True)
        var = z.var(dim=(1, 2), unbiased=False, keepdim=True)
        return (z - mean) * torch.rsqrt(var + 1e-5) * w + bias

    h = act(inorm(F.linear(pn, w1), in1w, in1b))
    mp = (h * dw).max(dim=2, keepdim=True)[0].expand(-1, -1, k, -1)                      # :631-633
    h = act(inorm(F.linear(torch.cat([h, mp], dim=-1), w2), in2w, in2b))
    mp = (h * dw).max(dim=2, keepdim=True)[0].expand(-1, -1, k, -1)                      # :639-641
    g = act(F.linear(torch.cat([h, mp], dim=-1), w3)) * dw                               # :643
    return g.reshape(b * m, k, 16), radius.reshape(1)


def bn_act(x, weight, bias, running_mean, running_var, momentum, eps, relu):
    import torch.nn.functional as F
    y = F.batch_norm(x, running_mean, running_var, weight, bias, True, momentum, eps)
    return F.relu(y) if relu else y


def bn_supported(rows, c):
    return True


def attn_pool(qy, h):
    att = torch.softmax(qy.float(), dim=1).mean(dim=2)
    return torch.bmm(att.unsqueeze(1).to(h.dtype), h).squeeze(1)


def attn_pool_supported(k, heads, c):
    return True


_NAMES = ('gather_rows', 'neighbour_max', 'neighbour_contract', 'fka_geometry', 'bn_act', 'bn_supported', 'attn_pool', 'attn_pool_supported')


@contextlib.contextmanager
def patched():
    from ppsurf_amd import train_ops
    saved = [getattr(train_ops, n) for n in _NAMES]
    for n in _NAMES:
        setattr(train_ops, n, globals()[n])
    try:
        yield
    finally:
        for n, f in zip(_NAMES, saved):
            setattr(train_ops, n, f)
