"""Network modules with the reference's attribute / state-dict layout, executed by the HIP kernels.

Mirrors (names, constructor arguments, tensor conventions):
  FKAConvLayer, ResidualBlock, FKAConvNetwork  -> source/base/nn.py:420-652
  AttentionPoco, STN, PointNetfeat, MLP        -> source/base/nn.py:72-96,133-190,255-417
  InterpAttentionKHeadsNet                     -> source/poco_model.py:362-419
  PocoNetwork / PPSurfNetwork                  -> source/poco_model.py:332-359, source/ppsurf_model.py:39-117
  batch_gather / max_pool / interpolate        -> source/base/nn.py:655-697

The torch.nn layers created here only HOLD parameters (so `state_dict()` matches the reference's 455 entries and its
checkpoints load unchanged); they are never called.  forward() builds a *plan* (BatchNorm-folded, packed weights on the
device; ppsurf_amd/encoder.py, ppsurf_amd/decoder.py), cached until a parameter changes, and launches the kernels.
That is eval() mode.  In train() mode forward() runs the autograd graph of ppsurf_amd/train_graph.py instead (batch-statistics
BatchNorm, dropout, norm_radius EMA; HIP gather/scatter ops with hand-written backward) on the same parameter tensors.
"""
import typing

import torch
from torch import nn

from . import spatial, train_graph
from .decoder import DecoderPlan, PocoDecoderPlan
from .encoder import EncoderPlan, FKAConvParams, ResidualBlockParams, gather_max

try:                                                    # Lightning is optional (absent in the build image)
    from pytorch_lightning import LightningModule as _Base
except Exception:                                       # pragma: no cover
    _Base = nn.Module


def count_parameters(model):
    return sum(p.numel() for p in model.parameters() if p.requires_grad)


def _params_version(module: nn.Module):
    """Changes whenever a parameter/buffer is modified in place or replaced, or the device / mode changes."""
    return tuple((id(t), t._version, t.device) for t in list(module.parameters()) + list(module.buffers())) + (module.training,)


def _sd(module):
    return {k: v.detach() for k, v in module.state_dict().items()}


def _pm(t):
    """[B,C,N] -> contiguous point-major [B,N,C]."""
    return t.transpose(1, 2).contiguous()


# ----------------------------------------------------------------------------------------------------------------
# gathers with the reference's channel-first signature
# ----------------------------------------------------------------------------------------------------------------
def batch_gather(data: torch.Tensor, dim: int, index: torch.Tensor):
    """data [B,C,N], index [B,M,K] -> [B,C,M,K] (nn.py:655-674, dim=2).  Plain indexing: API parity only, the kernels
    never materialise this tensor."""
    assert dim == 2
    b, c, _ = data.shape
    _, m, k = index.shape
    return torch.gather(data, 2, index.reshape(b, 1, m * k).expand(b, c, m * k)).reshape(b, c, m, k)


def max_pool(data: torch.Tensor, indices: torch.Tensor) -> torch.Tensor:
    """[B,C,N], [B,M,K] -> [B,C,M] through pps_gather_max_f32."""
    out = [gather_max(data[b].t().contiguous(), indices[b].contiguous()).t() for b in range(data.shape[0])]
    return torch.stack(out, dim=0)


def interpolate(x, neighbors_indices, method='mean'):
    """nn.py:684-697; negative ids are set to 0 IN PLACE like the reference."""
    neighbors_indices[neighbors_indices < 0] = 0
    g = batch_gather(x, 2, neighbors_indices)
    return g.mean(-1) if neighbors_indices.shape[-1] > 1 else g.squeeze(-1)


# ----------------------------------------------------------------------------------------------------------------
# encoder
# ----------------------------------------------------------------------------------------------------------------
def _act_name(activation):
    return 'silu' if isinstance(activation, nn.SiLU) else 'relu'


class FKAConvLayer(_Base):
    def __init__(self, in_channels, out_channels, kernel_size=16, bias=False, dim=3, activation=nn.ReLU()):
        super().__init__()
        self.in_channels, self.out_channels, self.kernel_size, self.bias, self.dim = in_channels, out_channels, kernel_size, bias, dim
        self.cv = nn.Conv2d(in_channels, out_channels, (1, kernel_size), bias=bias)
        self.norm_radius_momentum = 0.1
        self.register_buffer('norm_radius', torch.ones(1))
        self.alpha = nn.Parameter(torch.ones(1))
        self.beta = nn.Parameter(torch.ones(1))
        self.fc1 = nn.Conv2d(dim, kernel_size, 1, bias=False)
        self.fc2 = nn.Conv2d(2 * kernel_size, kernel_size, 1, bias=False)
        self.fc3 = nn.Conv2d(2 * kernel_size, kernel_size, 1, bias=False)
        self.bn1 = nn.InstanceNorm2d(kernel_size, affine=True)
        self.bn2 = nn.InstanceNorm2d(kernel_size, affine=True)
        self.activation = activation
        self._plan = None

    def forward(self, x, pts, support_points, neighbors_indices):
        """x [B,Cin,N], pts [B,3,N], support_points [B,3,M], neighbors_indices [B,M,K] -> [B,Cout,M]."""
        if x is None:
            return None
        if self.training:
            return train_graph.fkaconv_layer(self, _pm(x), _pm(pts), _pm(support_points), neighbors_indices).transpose(1, 2)
        ver = _params_version(self)
        if self._plan is None or self._plan[0] != ver:
            sd = {'L.' + k: v for k, v in _sd(self).items()}
            self._plan = (ver, FKAConvParams(sd, 'L', x.device, _act_name(self.activation)))
        layer = self._plan[1]
        out = [layer(x[b].t().contiguous(), pts[b].t().contiguous(), support_points[b].t().contiguous(),
                     neighbors_indices[b].contiguous()).t() for b in range(x.shape[0])]
        return torch.stack(out, dim=0)


class ResidualBlock(_Base):
    def __init__(self, in_channels, out_channels, kernel_size, activation=nn.ReLU()):
        super().__init__()
        half = in_channels // 2
        self.cv0, self.bn0 = nn.Conv1d(in_channels, half, 1), nn.BatchNorm1d(half)
        self.cv1, self.bn1 = FKAConvLayer(half, half, kernel_size, activation=activation), nn.BatchNorm1d(half)
        self.cv2, self.bn2 = nn.Conv1d(half, out_channels, 1), nn.BatchNorm1d(out_channels)
        self.activation = nn.ReLU(inplace=True)
        same = in_channels == out_channels
        self.shortcut = nn.Identity() if same else nn.Conv1d(in_channels, out_channels, 1)
        self.bn_shortcut = nn.Identity() if same else nn.BatchNorm1d(out_channels)
        self._act = _act_name(activation)
        self._plan = None

    def forward(self, x, pts, support_points, neighbors_indices):
        if self.training:
            return train_graph.residual_block(self, _pm(x), _pm(pts), _pm(support_points), neighbors_indices).transpose(1, 2)
        ver = _params_version(self)
        if self._plan is None or self._plan[0] != ver:
            sd = {'R.' + k: v for k, v in _sd(self).items()}
            self._plan = (ver, ResidualBlockParams(sd, 'R', x.device, self._act))
        blk = self._plan[1]
        out = [blk(x[b].t().contiguous(), pts[b].t().contiguous(), support_points[b].t().contiguous(),
                   neighbors_indices[b].contiguous()).t() for b in range(x.shape[0])]
        return torch.stack(out, dim=0)


class FKAConvNetwork(_Base):
    def __init__(self, in_channels, out_channels, segmentation=False, hidden=64, dropout=0.5, last_layer_additional_size=None,
                 fix_support_number=False, activation=nn.ReLU(), x4d_bug_fixed=False):
        super().__init__()
        if not segmentation or last_layer_additional_size is not None:
            raise NotImplementedError('only the segmentation head used by POCO / PPSurf is implemented')
        self.fixed = x4d_bug_fixed
        self.lcp_preprocess = True
        self.segmentation = segmentation
        self.fix_support_point_number = fix_support_number
        self.kernel_size = 16
        h = hidden
        self.cv0, self.bn0 = FKAConvLayer(in_channels, h, 16, activation=activation), nn.BatchNorm1d(h)
        for name, cin, cout in (('01', h, h), ('10', h, 2 * h), ('11', 2 * h, 2 * h), ('20', 2 * h, 4 * h), ('21', 4 * h, 4 * h),
                                ('30', 4 * h, 8 * h), ('31', 8 * h, 8 * h), ('40', 8 * h, 16 * h), ('41', 16 * h, 16 * h)):
            setattr(self, 'resnetb' + name, ResidualBlock(cin, cout, self.kernel_size, activation=activation))
        for name, cin, cout in (('5', 32 * h, 16 * h), ('3d', 24 * h, 8 * h), ('2d', 12 * h, 4 * h), ('1d', 6 * h, 2 * h), ('0d', 3 * h, h)):
            setattr(self, 'cv' + name, nn.Conv1d(cin, cout, 1))
            setattr(self, 'bn' + name, nn.BatchNorm1d(cout))
        self.fcout = nn.Conv1d(h, out_channels, 1)
        self.dropout = nn.Dropout(dropout)
        self.activation = nn.ReLU()
        self._act = _act_name(activation)
        self._plan = None

    def plan(self, device):
        ver = _params_version(self)
        if self._plan is None or self._plan[0] != ver or self._plan[1].device != torch.device(device):
            sd = {'E.' + k: v for k, v in _sd(self).items()}
            self._plan = (ver, EncoderPlan(sd, device, prefix='E', act=self._act, fixed=self.fixed))
        return self._plan[1]

    def forward_point_major(self, data, b=0):
        """Encoder pass of batch item b; returns the latents POINT-MAJOR [N, C] (no transposes on the hot path)."""
        plan = self.plan(data['pts'].device)
        pm = lambda t: t[b].t().contiguous()
        ids = {}
        for k, v in data.items():
            if k.startswith('ids') and torch.is_tensor(v):
                t = v[b] if v.dim() == 3 else v
                ids[k] = t.reshape(-1).contiguous() if k in ('ids43', 'ids32', 'ids21', 'ids10') else t.contiguous()
        cached = data.get('_levels_point_major')
        if cached is not None:                                    # levels already point-major (spatial.get_fkaconv_ids)
            return plan.forward(cached[b][0], cached[b][1:], ids)
        sups = [pm(data['support{}'.format(i)]) for i in (1, 2, 3, 4)]
        return plan.forward(pm(data['pts']), sups, ids)

    def forward_batch_point_major(self, data):
        """Eval-mode encoder pass of ALL batch items in batched launches (equally sized clouds): -> latents [B, N, C] point-major.
        data as produced by spatial.get_fkaconv_ids ([B,3,N] supports, [B,M,K] tables of per-cloud indices)."""
        assert not self.training
        plan = self.plan(data['pts'].device)
        b = data['pts'].shape[0]
        flat = lambda t: t.transpose(1, 2).reshape(-1, 3).contiguous().float()
        levels = [flat(data['pts'])] + [flat(data['support{}'.format(i)]) for i in (1, 2, 3, 4)]
        sizes = [lv.shape[0] // b for lv in levels]
        ids = {}
        for a in range(5):
            for c in (a - 1, a, a + 1):
                key = 'ids{}{}'.format(a, c)
                if key in data and torch.is_tensor(data[key]):
                    t = data[key]                                        # rows of level c, indices into level a
                    off = (torch.arange(b, device=t.device) * sizes[a]).view(b, 1, 1)
                    t = (t + off).reshape(b * t.shape[1], t.shape[2])
                    ids[key] = t.reshape(-1).contiguous() if c == a - 1 else t.contiguous()
        return plan.forward_batch(levels, ids, b).view(b, sizes[0], -1)

    def forward(self, data, spectral_only=False):
        """nn.py:508-554.  data['pts'] [B,3,N] (+ supports / ids unless spectral_only=False) -> [B,C,N]."""
        if not spectral_only:
            for key, value in spatial.get_fkaconv_ids(data).items():
                data[key] = value
        if self.training:
            return train_graph.encoder(self, data).transpose(1, 2)
        if self.dropout.p != 0:
            raise NotImplementedError('encoder dropout != 0 is not used by POCO / PPSurf')
        if data['pts'].shape[0] > 1:                      # several clouds (validation batches): batched launches
            return self.forward_batch_point_major(data).transpose(1, 2)
        out = [self.forward_point_major(data, b).t() for b in range(data['pts'].shape[0])]
        return torch.stack(out, dim=0)          # [B,C,N] as transposed views of point-major storage


# ----------------------------------------------------------------------------------------------------------------
# decoder parameter holders
# ----------------------------------------------------------------------------------------------------------------
class AttentionPoco(_Base):
    def __init__(self, net_size_max=1024, reduce=True):
        super().__init__()
        self.fc_query = nn.Conv2d(net_size_max, 1, 1)
        self.fc_value = nn.Conv2d(net_size_max, net_size_max, 1)
        self.reduce = reduce


class STN(_Base):
    def __init__(self, net_size_max=1024, num_scales=1, num_points=500, dim=3, sym_op='max'):
        super().__init__()
        if num_scales != 1:
            raise NotImplementedError('num_scales > 1 is not used by PPSurf')
        self.net_size_max, self.dim, self.sym_op, self.num_scales, self.num_points = net_size_max, dim, sym_op, num_scales, num_points
        self.conv1, self.conv2, self.conv3 = nn.Conv1d(dim, 64, 1), nn.Conv1d(64, 128, 1), nn.Conv1d(128, net_size_max, 1)
        self.mp1 = nn.MaxPool1d(num_points)
        self.fc1 = nn.Linear(net_size_max, net_size_max // 2)
        self.fc2 = nn.Linear(net_size_max // 2, net_size_max // 4)
        self.fc3 = nn.Linear(net_size_max // 4, dim * dim)
        self.bn1, self.bn2, self.bn3 = nn.BatchNorm1d(64), nn.BatchNorm1d(128), nn.BatchNorm1d(net_size_max)
        self.bn4, self.bn5 = nn.BatchNorm1d(net_size_max // 2), nn.BatchNorm1d(net_size_max // 4)


class PointNetfeat(_Base):
    def __init__(self, net_size_max=1024, num_scales=1, num_points=500, polar=False, use_point_stn=True, use_feat_stn=True,
                 output_size=100, sym_op='max', dim=3):
        super().__init__()
        if use_point_stn or not use_feat_stn or sym_op != 'att' or num_scales != 1 or polar or dim != 3:
            raise NotImplementedError('the HIP PointNet implements the PPSurf configuration only: use_point_stn=False, '
                                      'use_feat_stn=True, sym_op="att", num_scales=1 (source/ppsurf_model.py:52-53)')
        self.net_size_max, self.num_points, self.num_scales, self.polar = net_size_max, num_points, num_scales, polar
        self.use_point_stn, self.use_feat_stn, self.sym_op, self.output_size, self.dim = use_point_stn, use_feat_stn, sym_op, output_size, dim
        self.stn2 = STN(net_size_max=net_size_max, num_scales=num_scales, num_points=num_points, dim=64, sym_op=sym_op)
        self.conv0a, self.conv0b = nn.Conv1d(dim, 64, 1), nn.Conv1d(64, 64, 1)
        self.bn0a, self.bn0b = nn.BatchNorm1d(64), nn.BatchNorm1d(64)
        self.conv1, self.conv2, self.conv3 = nn.Conv1d(64, 64, 1), nn.Conv1d(64, 128, 1), nn.Conv1d(128, output_size, 1)
        self.bn1, self.bn2, self.bn3 = nn.BatchNorm1d(64), nn.BatchNorm1d(128), nn.BatchNorm1d(output_size)
        self.att = AttentionPoco(output_size)


class MLP(_Base):
    def __init__(self, input_size: int, output_size: int, num_layers: int, halving_size=True, final_bn_act=False,
                 final_layer_norm=False, activation=nn.ReLU, norm=nn.BatchNorm1d, fc_layer=nn.Linear, dropout=0.0):
        super().__init__()
        if final_bn_act or final_layer_norm:
            raise NotImplementedError('final_bn_act / final_layer_norm are not used by PPSurf')
        self.num_layers = num_layers
        sizes = [int(input_size / (2 ** i)) if halving_size else input_size for i in range(num_layers)]
        layers = [nn.Sequential(fc_layer(sizes[i], sizes[i + 1]), norm(sizes[i + 1]), activation(), nn.Dropout(dropout))
                  for i in range(num_layers - 1)]
        layers.append(nn.Sequential(fc_layer(sizes[-1], output_size)))
        self.layers = nn.Sequential(*layers)


class InterpAttentionKHeadsNet(nn.Module):
    def __init__(self, latent_size, out_channels, k=16):
        super().__init__()
        print('InterpNet - Simple - K={}'.format(k))
        self.fc1 = nn.Conv2d(latent_size + 3, latent_size, 1)
        self.fc2 = nn.Conv2d(latent_size, latent_size, 1)
        self.fc3 = nn.Conv2d(latent_size, latent_size, 1)
        self.fc8 = nn.Conv1d(latent_size, out_channels, 1)
        self.fc_query = nn.Conv2d(latent_size, 64, 1)
        self.fc_value = nn.Conv2d(latent_size, latent_size, 1)
        self.k = k
        self.activation = nn.ReLU()


# ----------------------------------------------------------------------------------------------------------------
# networks
# ----------------------------------------------------------------------------------------------------------------
def _channel_first(t):
    return t if t.shape[1] == 3 else t.transpose(1, 2)


def _cached_point_table(net, latents_b, plan):
    """One-entry cache of the per-point table G of `latents_b`.  The entry KEEPS the latent tensor (and the plan): while it
    is alive the allocator cannot hand its storage to the next shape's latents, so `same storage + same _version` really
    means "same values" (a pointer-only key matched the recycled block of the previous, equally sized shape)."""
    ent = net._table
    if (ent is not None and ent[1] is plan and ent[0]._version == ent[2] and ent[0].data_ptr() == latents_b.data_ptr()
            and ent[0].shape == latents_b.shape and ent[0].stride() == latents_b.stride()
            and ent[0].untyped_storage().data_ptr() == latents_b.untyped_storage().data_ptr()):
        return ent[3]
    net._table = None                                                 # drop the old table before allocating the new one
    table = plan.point_table(latents_b)
    net._table = (latents_b, plan, latents_b._version, table)
    return table


class PPSurfNetwork(_Base):
    def __init__(self, in_channels, latent_size, out_channels, k, num_pts_local, pointnet_latent_size):
        super().__init__()
        self.latent_size = latent_size
        self.encoder = FKAConvNetwork(in_channels, latent_size, segmentation=True, dropout=0, activation=nn.SiLU(), x4d_bug_fixed=True)
        self.projection = InterpAttentionKHeadsNet(latent_size, latent_size, k)
        self.point_net = PointNetfeat(net_size_max=pointnet_latent_size, num_points=num_pts_local, use_point_stn=False,
                                      use_feat_stn=True, output_size=latent_size, sym_op='att', dim=3)
        self.mlp = MLP(input_size=latent_size, output_size=out_channels, num_layers=3, halving_size=False, dropout=0.3)
        self.lcp_preprocess = True
        self.activation = nn.ReLU()
        self._dec = None
        self._table = None
        for name in ('encoder', 'projection', 'point_net', 'mlp'):
            print('Network -- {} -- {} parameters'.format('backbone' if name == 'encoder' else name, count_parameters(getattr(self, name))))

    # -- plans / caches ------------------------------------------------------------------------------------------
    def decoder_plan(self, device) -> DecoderPlan:
        mods = nn.ModuleList([self.projection, self.point_net, self.mlp])
        # decoder_dtype: 'f16x3' (split precision, the default of decoder.DecoderPlan) or 'f32' (exact fp32 MFMA); unset, the environment
        # variable PPS_DECODER_DTYPE decides, then the default
        ver = _params_version(mods) + (self.training, getattr(self, 'decoder_dtype', None))
        if self._dec is None or self._dec[0] != ver or self._dec[1].device != torch.device(device):
            sd = {k: v for k, v in _sd(self).items() if not k.startswith('encoder.')}
            self._dec = (ver, DecoderPlan(sd, device, dtype=getattr(self, 'decoder_dtype', None)))
            self._table = None
        return self._dec[1]

    def point_table(self, latents_b, plan):
        """Per-point table G = fc1_latent(latents)+b1 of one batch item, cached while the caller keeps passing the
        same latent tensor (the reference re-reads data['latents'] for every query chunk, poco_utils.py:220-223)."""
        return _cached_point_table(self, latents_b, plan)

    # -- reference API -------------------------------------------------------------------------------------------
    def forward(self, data):
        if self.training:
            train_graph._prepare(self, data)             # bf16 images of all parameters for this step (one multi-tensor copy)
            train_graph.start_pointnet(self, data)       # PointNet needs only the patches: on its side stream, beside the encoder
        data['latents'] = self.encoder.forward(data, spectral_only=True)
        return self.from_latent(data)

    def get_latent(self, data):
        data['latents'] = self.encoder.forward(data, spectral_only=False)
        data['proj_correction'] = None
        return data

    def from_latent(self, data: typing.Dict[str, torch.Tensor]):
        """source/ppsurf_model.py:82-117: data{latents [B,C,N], pts [B,3,N], pts_query [B,Q,3]|[B,3,Q], pts_local_ps [B,Q,P,3]}
        -> logits [B,2,Q]; sets data['proj_ids'] (int64 [B,Q,k]) like the reference (has_proj_ids=False, :83)."""
        pts = _channel_first(data['pts'])
        dev = pts.device
        ptq = _channel_first(data['pts_query'].to(dev))
        if self.training:
            k = min(self.projection.k, pts.shape[2])
            have = data.get('proj_ids')
            if not (torch.is_tensor(have) and tuple(have.shape) == (pts.shape[0], ptq.shape[2], k)):
                # the reference always recomputes them here with self.k (ppsurf_model.py:83); the dataset's table
                # (get_data_poco, k = 64) is the same search whenever the sizes agree, so it is reused
                with torch.no_grad():
                    data['proj_ids'] = spatial.knn(pts, ptq, k)
            return train_graph.ppsurf_from_latent(self, _pm(data['latents']), data, data['proj_ids'])
        plan = self.decoder_plan(dev)
        k = min(self.projection.k, pts.shape[2])
        if pts.shape[0] > 1:
            # several shapes (validation batches): one kNN launch and one decoder call over the rows of all shapes, the
            # per-point tables stacked and the neighbour ids offset into the stack
            b, n, q = pts.shape[0], pts.shape[2], ptq.shape[2]
            ids = spatial.knn(pts, ptq, k)                                            # [B,Q,k]
            flat = (ids + (torch.arange(b, device=dev) * n).view(b, 1, 1)).reshape(b * q, k).contiguous()
            table = torch.cat([plan.point_table(data['latents'][i]) for i in range(b)])
            pts_pm = pts.transpose(1, 2).reshape(b * n, 3).contiguous().float()
            q_pm = ptq.transpose(1, 2).reshape(b * q, 3).contiguous().float()
            patches = data['pts_local_ps'].to(dev).reshape(b * q, -1, 3).contiguous().float()
            lg, _ = plan.decode(table, pts_pm, q_pm, flat, patches, want_occ=False)
            data['proj_ids'] = ids
            return lg.view(b, q, 2).transpose(1, 2)
        logits, ids_all = [], []
        for b in range(pts.shape[0]):
            pts_pm = pts[b].t().contiguous().float()
            q_pm = ptq[b].t().contiguous().float()
            from . import ops
            idx = ops.knn_point_major(pts_pm, q_pm, k)
            table = self.point_table(data['latents'][b], plan)
            patches = data['pts_local_ps'][b].to(dev).contiguous().float()
            lg, _ = plan.decode(table, pts_pm, q_pm, idx, patches, want_occ=False)
            logits.append(lg.t())
            ids_all.append(idx)
        data['proj_ids'] = torch.stack(ids_all, dim=0)
        return torch.stack(logits, dim=0)


class PocoNetwork(_Base):
    """source/poco_model.py:332-359: FKAConv encoder (ReLU, cv5 discarded) + the interpolation-attention head straight to logits."""

    def __init__(self, in_channels, latent_size, out_channels, k):
        super().__init__()
        self.encoder = FKAConvNetwork(in_channels, latent_size, segmentation=True, dropout=0, x4d_bug_fixed=False)
        self.projection = InterpAttentionKHeadsNet(latent_size, out_channels, k)
        self.lcp_preprocess = True
        self._dec = None
        self._table = None
        print('Network -- backbone -- {} parameters'.format(count_parameters(self.encoder)))
        print('Network -- projection -- {} parameters'.format(count_parameters(self.projection)))

    def decoder_plan(self, device) -> PocoDecoderPlan:
        ver = _params_version(self.projection) + (self.training, getattr(self, 'decoder_dtype', None))
        if self._dec is None or self._dec[0] != ver or self._dec[1].device != torch.device(device):
            self._dec = (ver, PocoDecoderPlan({'projection.' + k: v for k, v in _sd(self.projection).items()}, device,
                                              dtype=getattr(self, 'decoder_dtype', None)))
            self._table = None
        return self._dec[1]

    def point_table(self, latents_b, plan):
        return _cached_point_table(self, latents_b, plan)

    def get_latent(self, data):
        data['latents'] = self.encoder.forward(data, spectral_only=False)
        data['proj_correction'] = None
        return data

    def forward(self, data):
        """poco_model.py:345-349: encoder with the precomputed tables, projection with the PRECOMPUTED proj_ids."""
        if self.training:
            train_graph._prepare(self, data)             # bf16 images of all parameters for this step (one multi-tensor copy)
        data['latents'] = self.encoder.forward(data, spectral_only=True)
        return self._project(data, has_proj_ids=True)

    def from_latent(self, data):
        """poco_model.py:357-359: proj_ids are recomputed (has_proj_ids defaults to False, :381-387)."""
        return self._project(data, has_proj_ids=False)

    def _project(self, data, has_proj_ids):
        from . import ops
        pts = _channel_first(data['pts'])
        dev = pts.device
        ptq = _channel_first(data['pts_query'].to(dev))
        if self.training:
            if not has_proj_ids:
                with torch.no_grad():
                    data['proj_ids'] = spatial.knn(pts, ptq, self.projection.k)
            return train_graph.interp_attention(self.projection, _pm(data['latents']), _pm(pts), _pm(ptq), data['proj_ids']).transpose(1, 2)
        plan = self.decoder_plan(dev)
        k = min(self.projection.k, pts.shape[2])
        if pts.shape[0] > 1:                               # validation batches: all shapes stacked along the rows, one decoder call
            b, n, q = pts.shape[0], pts.shape[2], ptq.shape[2]
            ids = data['proj_ids'] if has_proj_ids else spatial.knn(pts, ptq, k)
            flat = (ids + (torch.arange(b, device=dev) * n).view(b, 1, 1)).reshape(b * q, -1).contiguous()
            table = torch.cat([plan.point_table(data['latents'][i]) for i in range(b)])
            lg = plan.decode(table, pts.transpose(1, 2).reshape(b * n, 3).contiguous().float(),
                             ptq.transpose(1, 2).reshape(b * q, 3).contiguous().float(), flat)
            data['proj_ids'] = ids
            return lg.view(b, q, -1).transpose(1, 2)
        logits, ids_all = [], []
        for b in range(pts.shape[0]):
            pts_pm, q_pm = pts[b].t().contiguous().float(), ptq[b].t().contiguous().float()
            idx = data['proj_ids'][b].contiguous() if has_proj_ids else ops.knn_point_major(pts_pm, q_pm, k)
            logits.append(plan.decode(self.point_table(data['latents'][b], plan), pts_pm, q_pm, idx).t())
            ids_all.append(idx)
        if not has_proj_ids:
            data['proj_ids'] = torch.stack(ids_all, dim=0)
        return torch.stack(logits, dim=0)
