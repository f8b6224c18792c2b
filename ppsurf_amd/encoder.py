"""Host side of the FKAConv encoder (source/base/nn.py:420-652, eval mode) on the HIP kernels of csrc/pps_fkaconv.hip.

Activations are kept POINT-MAJOR ([n, C], channel fastest) between kernels so that every neighbour gather reads one
contiguous row; the reference's channel-first [B,C,N] layout only exists at the API boundary (ppsurf_amd/nn.py).
BatchNorm1d (eval) is folded into the preceding 1x1 convolution / FKAConv kernel weights on the host in float64.
"""
import numpy as np
import torch

from . import _lib

BN_EPS = 1e-5
ACT_CODE = {'relu': 1.0, 'silu': 2.0}


def _np64(t):
    if isinstance(t, torch.Tensor):
        t = t.detach().cpu().numpy()
    return np.asarray(t, dtype=np.float64)


def _bn_scale_shift(sd, bn):
    scale = _np64(sd[bn + '.weight']) / np.sqrt(_np64(sd[bn + '.running_var']) + BN_EPS)
    return scale, _np64(sd[bn + '.bias']) - _np64(sd[bn + '.running_mean']) * scale


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


class FKAConvParams:
    """Packed parameters of one FKAConvLayer (+ optional folded BatchNorm and ReLU epilogue)."""

    def __init__(self, sd, p, device, act, bn=None, relu_out=False):
        w = _np64(sd[p + '.cv.weight'])                       # [Cout, Cin, 1, 16]
        self.cout, self.cin = w.shape[0], w.shape[1]
        wd = w[:, :, 0, :].reshape(self.cout, self.cin * 16)                      # W[o][c*16+t]
        bias = None
        if bn is not None:
            scale, shift = _bn_scale_shift(sd, bn)
            wd = wd * scale[:, None]
            bias = shift
        geo = np.zeros(_lib.lib().pps_fkaconv_geo_floats(), dtype=np.float64)
        geo[0] = _np64(sd[p + '.norm_radius']).reshape(-1)[0]
        geo[1] = _np64(sd[p + '.alpha']).reshape(-1)[0]
        geo[2] = _np64(sd[p + '.beta']).reshape(-1)[0]
        geo[3] = ACT_CODE[act]
        o = 4
        for name, n in (('.fc1.weight', 48), ('.fc2.weight', 512), ('.fc3.weight', 512), ('.bn1.weight', 16), ('.bn1.bias', 16),
                        ('.bn2.weight', 16), ('.bn2.bias', 16)):
            v = _np64(sd[p + name]).reshape(-1)
            assert v.shape[0] == n, (p + name, v.shape)
            geo[o:o + n] = v
            o += n
        f = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)
        from .decoder import pack_dense
        self.geo, self.wpack = f(geo), torch.from_numpy(pack_dense(wd)).to(device)
        self.bias = f(bias) if bias is not None else None
        self.act_out = 1 if relu_out else 0

    def __call__(self, x, pts, sup, idx):
        """x [n,cin], pts [n,3], sup [m,3], idx int64 [m,k] -> [m,cout] (all contiguous, same device)."""
        n, m, k = x.shape[0], sup.shape[0], idx.shape[1]
        assert x.shape[1] == self.cin and x.is_contiguous() and pts.is_contiguous() and sup.is_contiguous() and idx.is_contiguous()
        L = _lib.lib()
        out = torch.empty((m, self.cout), dtype=torch.float32, device=x.device)
        ws = torch.empty((L.pps_fkaconv_ws_bytes(m, self.cin),), dtype=torch.uint8, device=x.device)
        _lib.check(L.pps_fkaconv_fwd_f32(x.data_ptr(), pts.data_ptr(), sup.data_ptr(), idx.data_ptr(), n, m, k, self.cin, self.cout,
                                         self.geo.data_ptr(), self.wpack.data_ptr(), self.bias.data_ptr() if self.bias is not None else None,
                                         self.act_out, out.data_ptr(), ws.data_ptr(), _stream(x)), 'pps_fkaconv_fwd_f32')
        return out


    def batch(self, x, pts, sup, idx, b):
        """The same layer for a batch of b equally sized clouds stacked along the rows (x [b*n,cin], sup [b*m,3], idx int64
        [b*m,k] = ROW numbers into x / pts): batched geometry kernels (InstanceNorm statistics per cloud), feature
        aggregation, then the (1,16) convolution with folded BatchNorm / ReLU as one MFMA GEMM over all rows."""
        mt, k = idx.shape
        L = _lib.lib()
        geo_w = self.geo.clone()
        g = torch.empty((mt, k, 16), dtype=torch.float32, device=x.device)
        stat = torch.empty((2, b, 32), dtype=torch.float32, device=x.device)
        ws = torch.empty((L.pps_fka_train_ws_bytes(b, mt // b, k),), dtype=torch.uint8, device=x.device)
        _lib.check(L.pps_fka_geometry_fwd_f32(pts.data_ptr(), sup.data_ptr(), idx.data_ptr(), b, mt // b, k, geo_w.data_ptr(), 0.0,
                                              g.data_ptr(), stat.data_ptr(), ws.data_ptr(), _stream(x)), 'pps_fka_geometry_fwd_f32')
        feat = torch.empty((mt, self.cin * 16), dtype=torch.float32, device=x.device)
        _lib.check(L.pps_neighbour_contract_fwd_f32(x.data_ptr(), idx.data_ptr(), g.data_ptr(), mt, k, self.cin, feat.data_ptr(),
                                                    _stream(x)), 'pps_neighbour_contract_fwd_f32')
        out = torch.empty((mt, self.cout), dtype=torch.float32, device=x.device)
        _lib.check(L.pps_rows_gemm_f32(feat.data_ptr(), None, self.cin * 16, None, None, 0, self.wpack.data_ptr(),
                                       self.bias.data_ptr() if self.bias is not None else None, None, self.act_out, mt, self.cout,
                                       out.data_ptr(), _stream(x)), 'pps_rows_gemm_f32')
        return out


class LinearParams:
    """1x1 convolution (+ folded BatchNorm) as wt [Cin, Cout] + bias."""

    def __init__(self, sd, conv, device, bn=None):
        w = _np64(sd[conv + '.weight'])
        w = w.reshape(w.shape[0], -1)
        b = _np64(sd[conv + '.bias']) if (conv + '.bias') in sd else np.zeros(w.shape[0])
        if bn is not None:
            scale, shift = _bn_scale_shift(sd, bn)
            w = w * scale[:, None]
            b = b * scale + shift
        f = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)
        from .decoder import pack_dense
        self.cin, self.cout = w.shape[1], w.shape[0]
        self.wt, self.bias = f(w.T), f(b)                     # VALU kernel operand (any channel counts)
        self.wpack = torch.from_numpy(pack_dense(w)).to(device)   # MFMA kernel operand (c1, c2 multiples of 16)

    def __call__(self, in1, idx1=None, in2=None, idx2=None, residual=None, relu=False, m=None):
        c1 = in1.shape[1]
        c2 = in2.shape[1] if in2 is not None else 0
        assert c1 + c2 == self.cin, (c1, c2, self.cin)
        if m is None:
            m = idx1.shape[0] if idx1 is not None else in1.shape[0]
        out = torch.empty((m, self.cout), dtype=torch.float32, device=in1.device)
        p = lambda t: t.data_ptr() if t is not None else None
        if c1 % 16 == 0 and c2 % 16 == 0:
            _lib.check(_lib.lib().pps_rows_gemm_f32(in1.data_ptr(), p(idx1), c1, p(in2), p(idx2), c2, self.wpack.data_ptr(),
                                                    self.bias.data_ptr(), p(residual), 1 if relu else 0, m, self.cout, out.data_ptr(),
                                                    _stream(in1)), 'pps_rows_gemm_f32')
        else:
            _lib.check(_lib.lib().pps_rows_linear_f32(in1.data_ptr(), p(idx1), c1, p(in2), p(idx2), c2, self.wt.data_ptr(),
                                                      self.bias.data_ptr(), p(residual), 1 if relu else 0, m, self.cout, out.data_ptr(),
                                                      _stream(in1)), 'pps_rows_linear_f32')
        return out


def gather_max(x, idx):
    """out[m,c] = max_j x[idx[m,j], c]  (nn.py:677-680)."""
    m, k = idx.shape
    out = torch.empty((m, x.shape[1]), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().pps_gather_max_f32(x.data_ptr(), idx.data_ptr(), m, k, x.shape[1], out.data_ptr(), _stream(x)), 'pps_gather_max_f32')
    return out


class ResidualBlockParams:
    def __init__(self, sd, p, device, act):
        self.cv0 = LinearParams(sd, p + '.cv0', device, bn=p + '.bn0')
        self.cv1 = FKAConvParams(sd, p + '.cv1', device, act, bn=p + '.bn1', relu_out=True)
        self.cv2 = LinearParams(sd, p + '.cv2', device, bn=p + '.bn2')
        self.shortcut = LinearParams(sd, p + '.shortcut', device, bn=p + '.bn_shortcut') if (p + '.shortcut.weight') in sd else None

    def __call__(self, x, pts, sup, idx):
        """nn.py:438-450.  x [n,C] -> [m,Cout]."""
        h = self.cv0(x, relu=True)
        h = self.cv1(h, pts, sup, idx)
        sc = self.shortcut(x) if self.shortcut is not None else x
        if sc.shape[0] != sup.shape[0]:
            sc = gather_max(sc, idx)                          # nn.py:445-446
        return self.cv2(h, residual=sc, relu=True)

    def batch(self, x, pts, sup, idx, b):
        h = self.cv0(x, relu=True)
        h = self.cv1.batch(h, pts, sup, idx, b)
        sc = self.shortcut(x) if self.shortcut is not None else x
        if sc.shape[0] != sup.shape[0]:
            sc = gather_max(sc, idx)
        return self.cv2(h, residual=sc, relu=True)


class EncoderPlan:
    """Device-resident, BatchNorm-folded parameters of FKAConvNetwork (segmentation=True, dropout 0)."""

    def __init__(self, sd, device, prefix='encoder', act='silu', fixed=True):
        p = prefix
        self.device = torch.device(device)
        self.fixed = fixed
        self.cv0 = FKAConvParams(sd, p + '.cv0', device, act, bn=p + '.bn0', relu_out=True)
        self.blocks = {n: ResidualBlockParams(sd, '{}.resnetb{}'.format(p, n), device, act)
                       for n in ('01', '10', '11', '20', '21', '30', '31', '40', '41')}
        self.cv5 = LinearParams(sd, p + '.cv5', device, bn=p + '.bn5')
        self.cv3d = LinearParams(sd, p + '.cv3d', device, bn=p + '.bn3d')
        self.cv2d = LinearParams(sd, p + '.cv2d', device, bn=p + '.bn2d')
        self.cv1d = LinearParams(sd, p + '.cv1d', device, bn=p + '.bn1d')
        self.cv0d = LinearParams(sd, p + '.cv0d', device, bn=p + '.bn0d')
        self.fcout = LinearParams(sd, p + '.fcout', device)

    def forward(self, pts, supports, ids):
        """nn.py:508-554 (spectral part) for ONE cloud.  pts [n,3]; supports = [s1..s4] ([ni,3]); ids: dict of int64
        tables 'ids00'.. [m,k] and 'ids43','ids32','ids21','ids10' [m] -> latents [n, out] point-major."""
        s1, s2, s3, s4 = supports
        b = self.blocks
        x = torch.ones_like(pts)                              # nn.py:517: constant input features
        x0 = self.cv0(x, pts, pts, ids['ids00'])
        x0 = b['01'](x0, pts, pts, ids['ids00'])
        x1 = b['10'](x0, pts, s1, ids['ids01'])
        x1 = b['11'](x1, s1, s1, ids['ids11'])
        x2 = b['20'](x1, s1, s2, ids['ids12'])
        x2 = b['21'](x2, s2, s2, ids['ids22'])
        x3 = b['30'](x2, s2, s3, ids['ids23'])
        x3 = b['31'](x3, s3, s3, ids['ids33'])
        x4 = b['40'](x3, s3, s4, ids['ids34'])
        x4 = b['41'](x4, s4, s4, ids['ids44'])
        if self.fixed:
            n4 = x4.shape[0]
            allrows = torch.arange(n4, dtype=torch.int64, device=x4.device).view(1, n4)
            x5 = gather_max(x4, allrows)                      # global max over the coarsest level (nn.py:531)
            zeros = torch.zeros((n4,), dtype=torch.int64, device=x4.device)
            x4d = self.cv5(x4, in2=x5, idx2=zeros, relu=True)   # cat([x4, x5]) -> cv5 -> bn5 -> relu (nn.py:532)
        else:
            x4d = x4                                          # POCO discards cv5 (nn.py:533-534)
        x3d = self.cv3d(x4d, idx1=ids['ids43'], in2=x3, relu=True)      # nearest up-sampling + skip (nn.py:536-537)
        x2d = self.cv2d(x3d, idx1=ids['ids32'], in2=x2, relu=True)
        x1d = self.cv1d(x2d, idx1=ids['ids21'], in2=x1, relu=True)
        xo = self.cv0d(x1d, idx1=ids['ids10'], in2=x0, relu=True)
        return self.fcout(xo)                                 # dropout p=0 (nn.py:547-548)

    def forward_batch(self, levels, ids, b):
        """The same for a batch of b equally sized clouds, everything stacked along the rows: levels = [pts, s1..s4] with
        shapes [b*n_l,3]; ids: tables of ROW numbers (the batch offsets b_i*n_l already added), 'ids00'.. int64 [b*m,k],
        up-sampling tables int64 [b*n_fine] -> latents [b*n, out].  All launches cover the whole batch: a single 10k-point
        cloud cannot fill 256 CUs, ten of them (one coverage wave of the latent loop) can."""
        pts, s1, s2, s3, s4 = levels
        bl = self.blocks
        x = torch.ones_like(pts)
        x0 = self.cv0.batch(x, pts, pts, ids['ids00'], b)
        x0 = bl['01'].batch(x0, pts, pts, ids['ids00'], b)
        x1 = bl['10'].batch(x0, pts, s1, ids['ids01'], b)
        x1 = bl['11'].batch(x1, s1, s1, ids['ids11'], b)
        x2 = bl['20'].batch(x1, s1, s2, ids['ids12'], b)
        x2 = bl['21'].batch(x2, s2, s2, ids['ids22'], b)
        x3 = bl['30'].batch(x2, s2, s3, ids['ids23'], b)
        x3 = bl['31'].batch(x3, s3, s3, ids['ids33'], b)
        x4 = bl['40'].batch(x3, s3, s4, ids['ids34'], b)
        x4 = bl['41'].batch(x4, s4, s4, ids['ids44'], b)
        if self.fixed:
            n4 = x4.shape[0] // b
            rows = torch.arange(b * n4, dtype=torch.int64, device=x4.device)
            x5 = gather_max(x4, rows.view(b, n4))             # per-cloud global max over the coarsest level (nn.py:531)
            x4d = self.cv5(x4, in2=x5, idx2=rows // n4, relu=True)
        else:
            x4d = x4
        x3d = self.cv3d(x4d, idx1=ids['ids43'], in2=x3, relu=True)
        x2d = self.cv2d(x3d, idx1=ids['ids32'], in2=x2, relu=True)
        x1d = self.cv1d(x2d, idx1=ids['ids21'], in2=x1, relu=True)
        xo = self.cv0d(x1d, idx1=ids['ids10'], in2=x0, relu=True)
        return self.fcout(xo)
