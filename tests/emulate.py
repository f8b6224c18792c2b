"""Test-only numpy emulation of what the HIP decoder kernels compute FROM THE PACKED WEIGHT IMAGES.

It inverts the A-operand packing (pps_pack.cpp) and replays the kernel pipeline of ppsurf_amd/csrc/pps_decode.hip in
float64, so the algebra of ppsurf_amd/decoder.py (BatchNorm folding, fc1 split, value/fc8/MLP composition, packing
order, bias offsets) is checked against the oracle on the CPU, before any GPU is involved."""
import numpy as np


def unpack_dense(packed, out, inp):
    ob_n, kb_n = (out + 31) // 32 * 2, (inp + 15) // 16
    p = np.asarray(packed, dtype=np.float64).reshape(ob_n, kb_n, 64, 4)
    w = np.zeros((ob_n * 16, kb_n * 16))
    for l in range(64):
        for s in range(4):
            w[(l & 15)::16, :][:, (4 * (l >> 4) + s)::16] = p[:, :, l, s]
    return w[:out, :inp]


def unpack_xyz(packed, out):
    p = np.asarray(packed, dtype=np.float64).reshape(-1, 64)
    w = np.zeros((p.shape[0] * 16, 4))
    for l in range(64):
        w[(l & 15)::16, l >> 4] = p[:, l]
    return w[:out, :3]


def _split(buf, sizes):
    out, o = [], 0
    for s in sizes:
        out.append(buf[o:o + s]); o += s
    assert o == buf.shape[0]
    return out


def relu(x):
    return np.maximum(x, 0.0)


def softmax(x, axis):
    e = np.exp(x - x.max(axis=axis, keepdims=True))
    return e / e.sum(axis=axis, keepdims=True)


def decode(w, latents_cn, pts, query, idx, patches):
    """w: {name: np.ndarray} packed images of DecoderPlan; latents_cn [256,N]; pts [N,3]; query [Q,3]; idx [Q,k];
    patches [Q,P,3] -> logits [Q,2] (float64)."""
    g_w = unpack_dense(w['g_w'], 256, 256)
    G = latents_cn.T.astype(np.float64) @ g_w.T + w['g_b']
    xyz, fc2, fc3, fcq = _split(w['ip_w'], [1024, 65536, 65536, 16384])
    b2, b3, bq = _split(w['ip_b'].astype(np.float64), [256, 256, 64])
    rel = query[:, None, :].astype(np.float64) - pts[idx]
    h = relu(G[idx] + rel @ unpack_xyz(xyz, 256).T)
    h = relu(h @ unpack_dense(fc2, 256, 256).T + b2)
    h = relu(h @ unpack_dense(fc3, 256, 256).T + b3)
    att = softmax(h @ unpack_dense(fcq, 64, 256).T + bq, axis=1).mean(axis=2)           # [Q,k]
    pooled = (att[:, :, None] * h).sum(axis=1)

    def rows(wbuf, bbuf, x):
        xyz_, c0b, l1, l2, l3 = _split(wbuf, [256, 4096, 4096, 8192, 32768])
        return xyz_, c0b, l1, l2, l3

    xa, c0b, s1, s2, s3 = rows(w['pa_w'], None, None)
    ba = _split(w['pa_b'].astype(np.float64), [64, 64, 64, 128, 256])
    x = patches.astype(np.float64)
    x0 = relu(x @ unpack_xyz(xa, 64).T + ba[0])
    x1 = relu(x0 @ unpack_dense(c0b, 64, 64).T + ba[1])
    t = relu(x1 @ unpack_dense(s1, 64, 64).T + ba[2])
    t = relu(t @ unpack_dense(s2, 128, 64).T + ba[3])
    t = relu(t @ unpack_dense(s3, 256, 128).T + ba[4])
    gmax = t.max(axis=1)
    f1, f2, f3 = _split(w['pb_w'], [32768, 8192, 262144])
    bb = _split(w['pb_b'].astype(np.float64), [128, 64, 4096])
    u = relu(gmax @ unpack_dense(f1, 128, 256).T + bb[0])
    u = relu(u @ unpack_dense(f2, 64, 128).T + bb[1])
    trans2 = (u @ unpack_dense(f3, 4096, 64).T + bb[2]).reshape(-1, 64, 64)
    xc, c0b2, c2 = _split(w['pc_w'], [256, 4096, 8192])
    bc = _split(w['pc_b'].astype(np.float64), [64, 64, 64, 128, 256, 256, 4])
    y0 = relu(x @ unpack_xyz(xc, 64).T + bc[0])
    y1 = relu(y0 @ unpack_dense(c0b2, 64, 64).T + bc[1])
    # `trans2` of the packed images is M = conv1 @ trans2 (conv1 is folded into the STN's last layer by DecoderPlan): one product, then conv1's bias
    y = relu(np.einsum('qab,qpb->qpa', trans2, y1) + bc[2])
    y = relu(y @ unpack_dense(c2, 128, 64).T + bc[3])
    wgt = softmax(y @ bc[5][:128] + bc[6][0], axis=1)          # attention logit from conv3's INPUT: u = W3^T wq, constant wq.b3 + bq
    xbar = (wgt[:, :, None] * y).sum(axis=1)                   # conv3 is behind the pooling, composed into the tail's Wb (decoder.py)
    wa, wb, l2w, l3w = _split(w['tl_w'], [65536, 32768, 65536, 8192])
    bt = _split(w['tl_b'].astype(np.float64), [256, 256, 32])
    hh = relu(pooled @ unpack_dense(wa, 256, 256).T + xbar @ unpack_dense(wb, 256, 128).T + bt[0])
    hh = relu(hh @ unpack_dense(l2w, 256, 256).T + bt[1])
    return hh @ unpack_dense(l3w, 2, 256).T + bt[2][:2], trans2
