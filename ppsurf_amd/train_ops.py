"""Autograd ops of the training step backed by the HIP kernels of csrc/pps_train.hip (forward AND backward hand-written).

    gather_rows(x [n,c], idx [r])                     -> [r,c]          (latent gather, nearest up-sampling)
    neighbour_max(x [n,c], idx [m,k])                 -> [m,c]          (nn.py:677-680 max_pool)
    neighbour_contract(x [n,c], idx [m,k], g [m,k,16])-> [m, c*16]      (nn.py:598,647-649 FKAConv feature aggregation)
    fka_geometry(geo [1140], pts, sup, idx, b, m, momentum) -> (g [b*m,k,16], norm_radius')   (nn.py:601-643, pps_fka_train.hip)
    bn_act(x [rows,c], weight, bias, running_mean, running_var, momentum, eps, relu) -> [rows,c]   (train-mode BatchNorm1d + ReLU)
    attn_pool(qy [Q,k,H], h [Q,k,C], relu_h) / query_attn_pool(y3, wq, bq, k)          (attention pooling of the interpolation head)
  bf16 row layers on STORED activations (Act = raw tensor + per-channel (scale, shift) + relu flag, applied by the consumer on load;
  pps_rows_train.hip):
    rows_layer(act, w, b, bn, relu) -> Act        rows3_layer(x [rows,3], w, b, bn) -> Act        act_max(act, groups, p) -> [groups, C]
    patch_transform(act, t_raw [Q,64,64], p)      patch_attn(h [Q,k,256], v [256])               head_input(table, ids, pts, query, k, wx)

Device tensors only: there is no CPU implementation in the product (tests/train_ref_ops.py holds the torch twins the CPU
suite patches in to check the surrounding graph).  Backward scatter-adds are atomics-free and bit-reproducible: the id table
is sorted once (`csr`, cached per table and step) and each target row sums its contributions in a fixed order.
All ops compute in fp32; gather_rows, neighbour_contract (matrix-pipe shapes), bn_act and the attention ops read and write 16-bit activations as they
are, the others up-cast them.
"""
import os

import torch

from . import _lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


LOW = (torch.bfloat16, torch.float16)          # 16-bit storage types of the fused training ops (trainer.precision bf16-mixed / 16-mixed)


def _code(dt):
    """dtype argument of the C entries: 1 bfloat16, 2 IEEE half."""
    return 1 if dt == torch.bfloat16 else 2


def _low(t, like=None):
    """t in a 16-bit storage type: as it is if it has one, else the type of `like` / the autocast type / bfloat16."""
    if t.dtype in LOW:
        return t.contiguous()
    if like is not None and like.dtype in LOW:
        return t.to(like.dtype).contiguous()
    dt = torch.get_autocast_dtype('cuda') if torch.is_autocast_enabled('cuda') else torch.bfloat16
    return t.to(dt if dt in LOW else torch.bfloat16).contiguous()


def _need_cuda(*tensors):
    for t in tensors:
        if not t.is_cuda:
            raise _lib.PpsError('ppsurf_amd.train_ops runs on the GPU only (got a {} tensor): there is no CPU path'.format(t.device))


_csr_cache = {}


def csr(idx_flat: torch.Tensor, n: int):
    """(order, offsets) of a flat id table: entries stably sorted by target row, offsets int64 [n+1].  Cached per table."""
    key = (idx_flat.data_ptr(), idx_flat._version, idx_flat.numel(), n)
    hit = _csr_cache.get(key)                       # the cached entry keeps the table alive, so the address cannot be reused
    if hit is not None:
        return hit[1], hit[2]
    order, offsets = csr_build(idx_flat, n)
    if len(_csr_cache) > 64:
        _csr_cache.clear()
    _csr_cache[key] = (idx_flat, order, offsets)
    return order, offsets


def csr_register(idx_flat, n, order, offsets):
    """Enter a CSR built elsewhere (with the batch, train_graph.table_extras) so that the backward pass finds it."""
    _csr_cache[(idx_flat.data_ptr(), idx_flat._version, idx_flat.numel(), n)] = (idx_flat, order, offsets)


def csr_build(idx_flat: torch.Tensor, n: int):
    """(order, offsets) of a flat id table, not cached: entries stably sorted by target row (ascending entry number inside a row), offsets int64
    [n+1] -- a counting sort on the device (pps_csr_build: count, scan, fill, rank; csrc/pps_csr.hip), the same arrays torch.sort(stable) +
    torch.searchsorted returned until round 5 (tests/test_gpu_train.py::test_csr_build_*)."""
    return csr_build_table(idx_flat.reshape(-1), 0, 0, n, False, want_flat=False)[1:]


def csr_build_table(ids: torch.Tensor, per_item: int, rows_per_item: int, rows: int, clamp_negative: bool, want_flat: bool = True):
    """(flat, order, offsets) of an id table [B, M, K] of a fit batch in one call: flat row numbers ids + item * rows_per_item (per_item = M * K
    entries per batch item; -1 -> row 0 with clamp_negative), and their CSR.  per_item = 0: `ids` are flat rows already."""
    _need_cuda(ids)
    if ids.dtype != torch.int64:
        raise _lib.PpsError('csr_build: id tables are int64')
    ids = ids.contiguous()
    entries = ids.numel()
    L = _lib.lib()
    dev = ids.device
    flat = torch.empty((entries,), dtype=torch.int64, device=dev) if want_flat else None
    order = torch.empty((entries,), dtype=torch.int64, device=dev)
    offsets = torch.empty((rows + 1,), dtype=torch.int64, device=dev)
    nbytes = L.pps_csr_ws_bytes(entries, rows)
    ws = torch.empty(((nbytes + 7) // 8,), dtype=torch.int64, device=dev)
    _lib.check(L.pps_csr_build(ids.data_ptr(), entries, int(per_item), int(rows_per_item), int(rows), 1 if clamp_negative else 0,
                               flat.data_ptr() if flat is not None else None, order.data_ptr(), offsets.data_ptr(), ws.data_ptr(), nbytes, _stream()),
               'pps_csr_build')
    return flat, order, offsets


def clear_cache():
    _csr_cache.clear()


class _GatherRows(torch.autograd.Function):
    """Row gather; a pure copy, so bf16 rows travel as they are (viewed as half as many 4-byte words).  The backward
    (segmented sum over the rows pointing at each source row) accumulates in fp32."""

    @staticmethod
    def forward(ctx, x, idx):
        _need_cuda(x, idx)
        if x.dtype not in (torch.float32,) + LOW or (x.dtype in LOW and x.shape[1] % 2):
            x = x.float()
        x = x.contiguous()
        idx = idx.contiguous()
        out = torch.empty((idx.numel(), x.shape[1]), device=x.device, dtype=x.dtype)
        words = x.shape[1] if x.dtype == torch.float32 else x.shape[1] // 2
        _lib.check(_lib.lib().pps_gather_rows_f32(x.data_ptr(), idx.data_ptr(), idx.numel(), words, out.data_ptr(), _stream()),
                   'pps_gather_rows_f32')
        ctx.save_for_backward(idx)
        ctx.n = x.shape[0]
        ctx.dtype = x.dtype
        return out

    @staticmethod
    def backward(ctx, dout):
        idx, = ctx.saved_tensors
        if idx.numel() == 0:
            return torch.zeros((ctx.n, dout.shape[1]), device=dout.device, dtype=ctx.dtype), None
        order, offsets = csr(idx, ctx.n)
        dout = dout.contiguous()
        dx = torch.empty((ctx.n, dout.shape[1]), device=dout.device, dtype=torch.float32)
        if dout.dtype in LOW and dout.shape[1] % 4 == 0:
            _lib.check(_lib.lib().pps_segment_sum_rows_16(dout.data_ptr(), order.data_ptr(), offsets.data_ptr(), ctx.n, dout.shape[1],
                                                          _code(dout.dtype), dx.data_ptr(), _stream()), 'pps_segment_sum_rows_16')
        else:
            dout = dout.float()
            _lib.check(_lib.lib().pps_segment_sum_rows_f32(dout.data_ptr(), order.data_ptr(), offsets.data_ptr(), ctx.n, dout.shape[1],
                                                           dx.data_ptr(), _stream()), 'pps_segment_sum_rows_f32')
        return dx.to(ctx.dtype), None


class _HeadInput(torch.autograd.Function):
    """h1[(q,j)] = table[ids[q,j]] + wx (query[q] - pts[ids[q,j]]): gather, neighbour offset, 3 -> C layer and sum of the interpolation head's
    first layer in one pass (pps_head_input_fwd); backward: d table by the segmented sum of the gather, d wx by pps_head_input_dwx."""

    @staticmethod
    def forward(ctx, table, ids, pts, query, k, wx):
        _need_cuda(table, ids, pts, query, wx)
        L = _lib.lib()
        table = _low(table)
        ids = ids.contiguous()
        pts32, q32 = pts.detach().float().contiguous(), query.detach().float().contiguous()
        wx32 = wx.detach().float().contiguous()
        nq, c = q32.shape[0], table.shape[1]
        h1 = torch.empty((nq * k, c), device=table.device, dtype=table.dtype)
        _lib.check(L.pps_head_input_fwd(table.data_ptr(), ids.data_ptr(), pts32.data_ptr(), q32.data_ptr(), nq, k, c, _code(table.dtype), wx32.data_ptr(),
                                        h1.data_ptr(), _stream()), 'pps_head_input_fwd')
        ctx.save_for_backward(ids, pts32, q32)
        ctx.meta = (table.shape[0], k, c, wx.dtype, wx.shape, table.dtype)
        return h1

    @staticmethod
    def backward(ctx, dh1):
        ids, pts32, q32 = ctx.saved_tensors
        n, k, c, wdt, wshape, dt = ctx.meta
        L = _lib.lib()
        dh1 = dh1.to(dt).contiguous()
        dtable = dwx = None
        if ctx.needs_input_grad[0]:
            order, offsets = csr(ids, n)
            dt32 = torch.empty((n, c), device=dh1.device, dtype=torch.float32)
            _lib.check(L.pps_segment_sum_rows_16(dh1.data_ptr(), order.data_ptr(), offsets.data_ptr(), n, c, _code(dt), dt32.data_ptr(), _stream()),
                       'pps_segment_sum_rows_16')
            dtable = dt32.to(dt)
        if ctx.needs_input_grad[5]:
            dwx = torch.empty((c, 3), device=dh1.device, dtype=torch.float32)
            ws = torch.empty((L.pps_head_input_ws_bytes(c),), device=dh1.device, dtype=torch.uint8)
            _lib.check(L.pps_head_input_dwx(dh1.data_ptr(), ids.data_ptr(), pts32.data_ptr(), q32.data_ptr(), q32.shape[0], k, c, _code(dt), dwx.data_ptr(),
                                            ws.data_ptr(), _stream()), 'pps_head_input_dwx')
            dwx = dwx.reshape(wshape).to(wdt)
        return dtable, None, None, None, None, dwx


def head_input_supported(c):
    return c % 8 == 0 and 256 % (c // 8) == 0


def head_input(table, ids, pts, query, k, wx):
    """table [N, C] (bf16), ids [Q*k] rows of table / pts, pts [N, 3], query [Q, 3], wx [C, 3] -> h1 [Q*k, C] bf16 (before its ReLU)."""
    return _HeadInput.apply(table, ids, pts, query, k, wx)


class _NeighbourMax(torch.autograd.Function):
    """x fp32, or 16-bit storage (kept: the maximum of stored values is one of them; the incoming gradient is read in that type, summed in fp32
    and rounded once -- the values an fp32 op between two casts produces, without the four cast kernels)."""

    @staticmethod
    def forward(ctx, x, idx):
        _need_cuda(x, idx)
        if x.dtype not in LOW:
            x = x.float()
        x = x.contiguous()
        idx = idx.contiguous()
        m, k = idx.shape
        c = x.shape[1]
        out = torch.empty((m, c), device=x.device, dtype=x.dtype)
        arg = torch.empty((m, c), device=x.device, dtype=torch.int32)
        if x.dtype in LOW:
            _lib.check(_lib.lib().pps_gather_max_arg_16(x.data_ptr(), idx.data_ptr(), m, k, c, _code(x.dtype), out.data_ptr(), arg.data_ptr(),
                                                        _stream()), 'pps_gather_max_arg_16')
        else:
            _lib.check(_lib.lib().pps_gather_max_arg_f32(x.data_ptr(), idx.data_ptr(), m, k, c, out.data_ptr(), arg.data_ptr(), _stream()),
                       'pps_gather_max_arg_f32')
        ctx.save_for_backward(idx, arg)
        ctx.n = x.shape[0]
        ctx.dtype = x.dtype
        return out

    @staticmethod
    def backward(ctx, dout):
        idx, arg = ctx.saved_tensors
        order, offsets = csr(idx.view(-1), ctx.n)
        dout = dout.contiguous().to(ctx.dtype)
        c = dout.shape[1]
        dx = torch.empty((ctx.n, c), device=dout.device, dtype=ctx.dtype)
        if ctx.dtype in LOW:
            _lib.check(_lib.lib().pps_gather_max_bwd_16(dout.data_ptr(), arg.data_ptr(), order.data_ptr(), offsets.data_ptr(), ctx.n,
                                                        idx.shape[1], c, _code(ctx.dtype), dx.data_ptr(), _stream()), 'pps_gather_max_bwd_16')
        else:
            _lib.check(_lib.lib().pps_gather_max_bwd_f32(dout.data_ptr(), arg.data_ptr(), order.data_ptr(), offsets.data_ptr(), ctx.n,
                                                         idx.shape[1], c, dx.data_ptr(), _stream()), 'pps_gather_max_bwd_f32')
        return dx, None


class _NeighbourContract(torch.autograd.Function):
    """x [n,c] in fp32 or, when the matrix-pipe kernels take the shape (c % 16 == 0, k <= 16), in the 16-bit type it arrives in: the output has x's
    type and the incoming gradient is read in it (no cast kernels around the op in an autocast step); g, dg and the arithmetic are fp32."""

    @staticmethod
    def forward(ctx, x, idx, g):
        _need_cuda(x, idx, g)
        m, k = idx.shape
        c = x.shape[1]
        low = x.dtype in LOW and bool(_lib.lib().pps_neighbour_contract_16_supported(k, c))
        if not low:
            x = x.float()
        x, idx, g = x.contiguous(), idx.contiguous(), g.float().contiguous()
        if g.shape != (m, k, 16):
            raise ValueError('neighbour_contract: g must be [m, k, 16], got {}'.format(tuple(g.shape)))
        out = torch.empty((m, c * 16), device=x.device, dtype=x.dtype)
        _lib.check(_lib.lib().pps_neighbour_contract_fwd(x.data_ptr(), idx.data_ptr(), g.data_ptr(), m, k, c, _code(x.dtype) if low else 0,
                                                         out.data_ptr(), _stream()), 'pps_neighbour_contract_fwd')
        ctx.save_for_backward(x, idx, g)
        ctx.low = low
        return out

    @staticmethod
    def backward(ctx, dout):
        x, idx, g = ctx.saved_tensors
        m, k = idx.shape
        n, c = x.shape
        dout = dout.to(x.dtype).contiguous()
        need_x, _, need_g = ctx.needs_input_grad
        dxg = torch.empty((m * k, c), device=x.device, dtype=torch.float32) if need_x else None
        dg = torch.empty((m, k, 16), device=x.device, dtype=torch.float32) if need_g else None
        if need_x or need_g:
            _lib.check(_lib.lib().pps_neighbour_contract_bwd(x.data_ptr(), idx.data_ptr(), g.data_ptr(), dout.data_ptr(), m, k, c,
                                                             _code(x.dtype) if ctx.low else 0, dxg.data_ptr() if need_x else None,
                                                             dg.data_ptr() if need_g else None, _stream()), 'pps_neighbour_contract_bwd')
        dx = None
        if need_x:
            order, offsets = csr(idx.view(-1), n)
            dx = torch.empty((n, c), device=x.device, dtype=torch.float32)
            _lib.check(_lib.lib().pps_segment_sum_rows_f32(dxg.data_ptr(), order.data_ptr(), offsets.data_ptr(), n, c, dx.data_ptr(),
                                                           _stream()), 'pps_segment_sum_rows_f32')
        return dx, None, dg


GEO_FLOATS = 1140          # pps_fkaconv_geo_floats(): radius, alpha, beta, act, fc1 [16,3], fc2 [16,32], fc3 [16,32], IN1 w/b, IN2 w/b


class _FkaGeometry(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, geo, pts, sup, idx, b, m, momentum, owned=False):
        _need_cuda(geo, pts, sup, idx)
        pts, sup, idx = pts.contiguous(), sup.contiguous(), idx.contiguous()
        k = idx.shape[1]
        if geo.numel() != GEO_FLOATS or idx.shape[0] != b * m or sup.shape[0] != b * m:
            raise ValueError('fka_geometry: inconsistent sizes')
        # the kernel moves entry 0 (norm_radius) in place: on a copy, unless the caller hands over a vector nobody else reads (pack_geo's)
        geo_w = geo.detach() if owned and geo.is_contiguous() and geo.dtype == torch.float32 else geo.detach().clone().contiguous()
        g = torch.empty((b * m, k, 16), device=pts.device, dtype=torch.float32)
        stat = torch.empty((2, b, 32), device=pts.device, dtype=torch.float32)
        ws = torch.empty((_lib.lib().pps_fka_train_ws_bytes(b, m, k),), device=pts.device, dtype=torch.uint8)
        _lib.check(_lib.lib().pps_fka_geometry_fwd_f32(pts.data_ptr(), sup.data_ptr(), idx.data_ptr(), b, m, k, geo_w.data_ptr(),
                                                       float(momentum), g.data_ptr(), stat.data_ptr(), ws.data_ptr(), _stream()),
                   'pps_fka_geometry_fwd_f32')
        ctx.save_for_backward(pts, sup, idx, geo_w, stat)
        ctx.dims = (b, m, k)
        radius = geo_w[0:1]                                     # (a view: the caller copies it into the layer's buffer)
        ctx.mark_non_differentiable(radius)
        ctx.set_materialize_grads(False)                        # no zero tensor for the gradient of radius that nobody computes
        return g, radius

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, dg, _dradius):
        if dg is None:
            return None, None, None, None, None, None, None, None
        pts, sup, idx, geo_w, stat = ctx.saved_tensors
        b, m, k = ctx.dims
        dg = dg.contiguous().float()
        dgeo = torch.empty((GEO_FLOATS,), device=dg.device, dtype=torch.float32)
        ws = torch.empty((_lib.lib().pps_fka_train_ws_bytes(b, m, k),), device=dg.device, dtype=torch.uint8)
        _lib.check(_lib.lib().pps_fka_geometry_bwd_f32(pts.data_ptr(), sup.data_ptr(), idx.data_ptr(), b, m, k, geo_w.data_ptr(),
                                                       stat.data_ptr(), dg.data_ptr(), dgeo.data_ptr(), ws.data_ptr(), _stream()),
                   'pps_fka_geometry_bwd_f32')
        return dgeo, None, None, None, None, None, None, None


def fka_geometry(geo, pts, sup, idx, b, m, momentum, owned=False):
    """geo: packed small parameters of the layer (differentiable); pts [rows,3], sup [b*m,3], idx int64 [b*m,k] rows of pts.
    momentum > 0: train() -- norm_radius is first moved towards the mean neighbourhood radius, the new value is used and
    returned.  owned: geo is a temporary of the caller's (its entry 0 may be overwritten with the new norm_radius) -- saves the copy.
    -> (g [b*m,k,16], norm_radius [1])."""
    return _FkaGeometry.apply(geo, pts, sup, idx, b, m, momentum, owned)


class _BnAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps, relu):
        _need_cuda(x, weight, bias)
        if x.dtype not in (torch.float32,) + LOW:
            x = x.float()
        x = x.contiguous()
        rows, c = x.shape
        w32, b32 = weight.detach().float().contiguous(), bias.detach().float().contiguous()
        y = torch.empty_like(x)
        save = torch.empty((2, c), device=x.device, dtype=torch.float32)
        nbytes = _lib.lib().pps_bn_train_ws_bytes(rows, c)
        if nbytes == 0 and rows > 0:
            raise ValueError('bn_act: unsupported shape [{}, {}]'.format(rows, c))
        ws = torch.empty((max(nbytes, 1),), device=x.device, dtype=torch.uint8)
        _lib.check(_lib.lib().pps_bn_train_fwd(x.data_ptr(), rows, c, _code(x.dtype) if x.dtype in LOW else 0, w32.data_ptr(), b32.data_ptr(),
                                               running_mean.data_ptr() if running_mean is not None else None,
                                               running_var.data_ptr() if running_var is not None else None, float(momentum), float(eps),
                                               int(bool(relu)), y.data_ptr(), save.data_ptr(), ws.data_ptr(), _stream()), 'pps_bn_train_fwd')
        ctx.save_for_backward(x, w32, b32, save)
        ctx.relu = bool(relu)
        ctx.dtypes = (weight.dtype, bias.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w32, b32, save = ctx.saved_tensors
        rows, c = x.shape
        dy = dy.to(x.dtype).contiguous()
        dx = torch.empty_like(x)
        dgamma = torch.empty((c,), device=x.device, dtype=torch.float32)
        dbeta = torch.empty((c,), device=x.device, dtype=torch.float32)
        ws = torch.empty((max(_lib.lib().pps_bn_train_ws_bytes(rows, c), 1),), device=x.device, dtype=torch.uint8)
        _lib.check(_lib.lib().pps_bn_train_bwd(x.data_ptr(), dy.data_ptr(), rows, c, _code(x.dtype) if x.dtype in LOW else 0, w32.data_ptr(),
                                               b32.data_ptr(), save.data_ptr(), int(ctx.relu), dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                                               ws.data_ptr(), _stream()), 'pps_bn_train_bwd')
        return dx, dgamma.to(ctx.dtypes[0]), dbeta.to(ctx.dtypes[1]), None, None, None, None, None


class _BnAddRelu(torch.autograd.Function):
    """relu(BatchNorm(x) + res) in train(): the tail of a residual block inside the BatchNorm's apply pass (pps_bn_add_relu_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, x, res, weight, bias, running_mean, running_var, momentum, eps):
        _need_cuda(x, res, weight, bias)
        if x.dtype not in (torch.float32,) + LOW:
            x = x.float()
        x = x.contiguous()
        res = res.to(x.dtype).contiguous()
        rows, c = x.shape
        w32, b32 = weight.detach().float().contiguous(), bias.detach().float().contiguous()
        y = torch.empty_like(x)
        save = torch.empty((2, c), device=x.device, dtype=torch.float32)
        nbytes = _lib.lib().pps_bn_train_ws_bytes(rows, c)
        if nbytes == 0 and rows > 0:
            raise ValueError('bn_add_relu: unsupported shape [{}, {}]'.format(rows, c))
        ws = torch.empty((max(nbytes, 1),), device=x.device, dtype=torch.uint8)
        _lib.check(_lib.lib().pps_bn_add_relu_fwd(x.data_ptr(), res.data_ptr(), rows, c, _code(x.dtype) if x.dtype in LOW else 0, w32.data_ptr(), b32.data_ptr(),
                                                  running_mean.data_ptr() if running_mean is not None else None,
                                                  running_var.data_ptr() if running_var is not None else None, float(momentum), float(eps),
                                                  y.data_ptr(), save.data_ptr(), ws.data_ptr(), _stream()), 'pps_bn_add_relu_fwd')
        ctx.save_for_backward(x, res, w32, b32, save)
        ctx.dtypes = (weight.dtype, bias.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, res, w32, b32, save = ctx.saved_tensors
        rows, c = x.shape
        dy = dy.to(x.dtype).contiguous()
        dx, dres = torch.empty_like(x), torch.empty_like(x)
        dgamma = torch.empty((c,), device=x.device, dtype=torch.float32)
        dbeta = torch.empty((c,), device=x.device, dtype=torch.float32)
        ws = torch.empty((max(_lib.lib().pps_bn_train_ws_bytes(rows, c), 1),), device=x.device, dtype=torch.uint8)
        _lib.check(_lib.lib().pps_bn_add_relu_bwd(x.data_ptr(), res.data_ptr(), dy.data_ptr(), rows, c, _code(x.dtype) if x.dtype in LOW else 0,
                                                  w32.data_ptr(), b32.data_ptr(), save.data_ptr(), dx.data_ptr(), dres.data_ptr(), dgamma.data_ptr(),
                                                  dbeta.data_ptr(), ws.data_ptr(), _stream()), 'pps_bn_add_relu_bwd')
        return dx, dres, dgamma.to(ctx.dtypes[0]), dbeta.to(ctx.dtypes[1]), None, None, None, None


def bn_add_relu(x, res, weight, bias, running_mean, running_var, momentum, eps):
    """relu(BatchNorm1d(x) + res) over the rows of x [rows, c] in train() mode (batch statistics, running statistics updated in place)."""
    return _BnAddRelu.apply(x, res, weight, bias, running_mean, running_var, momentum, eps)


def col_sum(x):
    """x [rows, c] -> fp32 [c] column sums (the bias gradient of a row layer) by the HIP reduction; None if the shape is not one it takes
    (the caller then uses torch's sum).  x may be a block of columns of a wider contiguous tensor (unit column stride, row stride and first column
    multiples of 4): summed where it lies.  No autograd: called from backward passes."""
    if not x.is_cuda or x.dim() != 2 or x.dtype not in (torch.float32,) + LOW:
        return None
    rows, c = x.shape
    ld = x.stride(0) if rows > 1 else c
    if rows < 1 or (c > 1 and x.stride(1) != 1) or ld < c or ld % 4 or x.data_ptr() % (4 * x.element_size()):
        return None
    nbytes = _lib.lib().pps_bn_train_ws_bytes(rows, c)
    if nbytes == 0:
        return None
    out = torch.empty((c,), device=x.device, dtype=torch.float32)
    ws = torch.empty((nbytes,), device=x.device, dtype=torch.uint8)
    _lib.check(_lib.lib().pps_col_sum_strided(x.data_ptr(), rows, c, ld, _code(x.dtype) if x.dtype in LOW else 0, out.data_ptr(), ws.data_ptr(),
                                              _stream()), 'pps_col_sum_strided')
    return out


def sum_rows(x):
    """x [rows, c] -> [c] sums over the rows in fp32: the HIP reduction wherever it takes the shape (any c: blocks of <= 1024 columns, narrow tensors
    padded to 4 columns), torch's sum only for what is left (non-contiguous / other types).  The step's own reductions go through here so that a
    replayed HIP graph runs the same reduction kernels as the eager step (train_graph._bias_grad: torch's sum of a [rows, 4096] tensor did not
    replay like it ran eagerly; ADVICE r5 asked for the rest of the class to follow)."""
    if x.dim() == 2 and x.is_cuda and x.dtype in (torch.float32,) + LOW:
        x = x.contiguous()
        rows, c = x.shape
        if c % 4:
            pad = 4 - c % 4
            return sum_rows(torch.cat([x, x.new_zeros((rows, pad))], dim=1))[:c]
        if c <= 1024:
            out = col_sum(x)
            if out is not None:
                return out
            if c < 64:                                   # column counts the kernel's thread layout does not divide: pad to the next power of two
                width = 4
                while width < c:
                    width *= 2
                if width != c:
                    return sum_rows(torch.cat([x, x.new_zeros((rows, width - c))], dim=1))[:c]
        else:
            parts = [col_sum(x[:, i:min(i + 1024, c)]) for i in range(0, c, 1024)]        # column blocks summed where they lie
            if all(q is not None for q in parts):
                return torch.cat(parts)
    return x.sum(0, dtype=torch.float32)


class _AffineRows(torch.autograd.Function):
    """x [rows, c] * scale [c] + shift [c] (fp32) with the parameter gradients summed over the rows by the HIP reduction (autograd's own broadcast
    backward is torch.sum: see sum_rows)."""

    @staticmethod
    def forward(ctx, x, scale, shift):
        ctx.save_for_backward(x, scale)
        return x * scale + shift

    @staticmethod
    def backward(ctx, g):
        x, scale = ctx.saved_tensors
        g = g.float()
        dx = g * scale if ctx.needs_input_grad[0] else None
        dscale = sum_rows(g * x).to(scale.dtype) if ctx.needs_input_grad[1] else None
        dshift = sum_rows(g).to(scale.dtype) if ctx.needs_input_grad[2] else None
        return dx, dscale, dshift


def affine_rows(x, scale, shift):
    return _AffineRows.apply(x, scale, shift)


def gemm_supported(x, k):
    """The hand-written dense-layer kernels (csrc/pps_gemm_train.hip) take this operand: device tensor in a 16-bit storage type, contraction length a
    multiple of 8."""
    return x.is_cuda and x.dtype in LOW and k % 8 == 0 and k >= 8


def gemm_nt(x, w, bias=None, out_f32=False):
    """x [M, K] @ w [N, K]^T (+ bias [N] fp32) -> [M, N] in x's 16-bit type (or fp32): pps_gemm_nt_16.  No autograd (called from autograd functions)."""
    _need_cuda(x, w)
    assert x.dim() == 2 and w.dim() == 2 and x.shape[1] == w.shape[1] and x.dtype == w.dtype and x.dtype in LOW
    if x.stride(1) != 1 or x.stride(0) % 8 or x.data_ptr() % 16:
        x = x.contiguous()
    if w.stride(1) != 1 or w.stride(0) % 8 or w.data_ptr() % 16:
        w = w.contiguous()
    m, k = x.shape
    n = w.shape[0]
    y = torch.empty((m, n), device=x.device, dtype=torch.float32 if out_f32 else x.dtype)
    b32 = None if bias is None else bias.detach().float().contiguous()
    _lib.check(_lib.lib().pps_gemm_nt_16(x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0), None if b32 is None else b32.data_ptr(), y.data_ptr(), n,
                                         m, n, k, _code(x.dtype), int(bool(out_f32)), _stream()), 'pps_gemm_nt_16')
    return y


def gemm_tn(g, x):
    """g [M, N]^T @ x [M, K] -> fp32 [N, K] (the weight gradient of a row layer: contraction over the rows): pps_gemm_tn_16.  N, K multiples of 8."""
    _need_cuda(g, x)
    assert g.dim() == 2 and x.dim() == 2 and g.shape[0] == x.shape[0] and g.dtype == x.dtype and g.dtype in LOW
    if g.stride(1) != 1 or g.stride(0) % 8 or g.data_ptr() % 16:
        g = g.contiguous()
    if x.stride(1) != 1 or x.stride(0) % 8 or x.data_ptr() % 16:
        x = x.contiguous()
    m, n = g.shape
    k = x.shape[1]
    L = _lib.lib()
    dw = torch.empty((n, k), device=g.device, dtype=torch.float32)
    ws = torch.empty((max(int(L.pps_gemm_tn_ws_bytes(m, n, k)), 16),), device=g.device, dtype=torch.uint8)
    _lib.check(L.pps_gemm_tn_16(g.data_ptr(), g.stride(0), x.data_ptr(), x.stride(0), m, n, k, _code(g.dtype), dw.data_ptr(), ws.data_ptr(), _stream()),
               'pps_gemm_tn_16')
    return dw


class _AttnPool(torch.autograd.Function):
    """pooled[q] = sum_j mean_h softmax_j(qy[q,j,h]) * h[q,j]  (poco_model.py:412-414 in the pooled form): one HIP kernel forward, one
    backward; fp32 or bf16 storage, fp32 arithmetic; the softmax is recomputed in backward, only the two inputs are saved."""

    @staticmethod
    def forward(ctx, qy, h, relu_h=False):
        _need_cuda(qy, h)
        ctx.relu_h = int(bool(relu_h))
        dt = h.dtype if h.dtype in (torch.float32,) + LOW else torch.float32
        qy, h = qy.to(dt).contiguous(), h.to(dt).contiguous()
        q, k, heads = qy.shape
        c = h.shape[2]
        pooled = torch.empty((q, c), device=h.device, dtype=dt)
        _lib.check(_lib.lib().pps_attn_pool_fwd(qy.data_ptr(), h.data_ptr(), q, k, heads, c, _code(dt) if dt in LOW else 0, ctx.relu_h, pooled.data_ptr(), _stream()),
                   'pps_attn_pool_fwd')
        ctx.save_for_backward(qy, h)
        return pooled

    @staticmethod
    def backward(ctx, dpooled):
        qy, h = ctx.saved_tensors
        q, k, heads = qy.shape
        c = h.shape[2]
        dpooled = dpooled.to(h.dtype).contiguous()
        dqy, dh = torch.empty_like(qy), torch.empty_like(h)
        _lib.check(_lib.lib().pps_attn_pool_bwd(qy.data_ptr(), h.data_ptr(), dpooled.data_ptr(), q, k, heads, c, _code(h.dtype) if h.dtype in LOW else 0,
                                                ctx.relu_h, dqy.data_ptr(), dh.data_ptr(), _stream()), 'pps_attn_pool_bwd')
        return dqy, dh, None


class Act:
    """A layer output that is NOT stored in its activated form: `raw` [rows, C] bf16 plus the per-channel `affine` [2, C] (scale, shift;
    None = identity) and `relu` that the consumer applies while loading -- act = relu?(raw * scale + shift)."""
    __slots__ = ('raw', 'affine', 'relu')

    def __init__(self, raw, affine=None, relu=False):
        self.raw, self.affine, self.relu = raw, affine, relu

    def materialize(self):
        if self.affine is None:
            return torch.relu(self.raw) if self.relu else self.raw
        z = self.raw.float() * self.affine[0] + self.affine[1]
        return (torch.relu(z) if self.relu else z).to(self.raw.dtype)


class _RowsLayer(torch.autograd.Function):
    """y = act(x) W^T + b over rows with the input's BatchNorm + ReLU applied on load and the output's batch statistics taken on store
    (pps_rows_train.hip).  Returns (y bf16, out_affine [2, cout] or None)."""

    @staticmethod
    def forward(ctx, x, in_affine, in_relu, w, b, gamma, beta, running_mean, running_var, momentum, eps):
        _need_cuda(x, w)
        L = _lib.lib()
        x = _low(x)
        rows, cin = x.shape
        cout = w.shape[0]
        w32 = w.detach().float().contiguous()
        b32 = None if b is None else b.detach().float().contiguous()
        aff = None if in_affine is None else in_affine.detach().float().contiguous()
        bn = gamma is not None
        g32 = gamma.detach().float().contiguous() if bn else None
        be32 = beta.detach().float().contiguous() if bn else None
        y = torch.empty((rows, cout), device=x.device, dtype=x.dtype)
        out_affine = torch.empty((2, cout), device=x.device, dtype=torch.float32) if bn else None
        save = torch.empty((2, cout), device=x.device, dtype=torch.float32) if bn else None
        ws = torch.empty((L.pps_rows_layer_ws_bytes(cin, cout),), device=x.device, dtype=torch.uint8)
        ptr = lambda t: None if t is None else t.data_ptr()
        _lib.check(L.pps_rows_layer_fwd(x.data_ptr(), rows, cin, _code(x.dtype), ptr(aff), None if aff is None else aff.data_ptr() + 4 * cin, int(bool(in_relu)),
                                        w32.data_ptr(), ptr(b32), cout, y.data_ptr(), ptr(g32), ptr(be32), ptr(running_mean), ptr(running_var),
                                        float(momentum or 0.0), float(eps or 0.0), ptr(out_affine), ptr(save), ws.data_ptr(), _stream()),
                   'pps_rows_layer_fwd')
        ctx.save_for_backward(x, aff, w32, g32, save, y)
        ctx.meta = (bool(in_relu), b is not None, bn, w.dtype, None if b is None else b.dtype)
        return y, out_affine

    @staticmethod
    def backward(ctx, gy, g_affine):
        x, aff, w32, g32, save, y = ctx.saved_tensors
        in_relu, has_b, bn, wdt, bdt = ctx.meta
        L = _lib.lib()
        rows, cin = x.shape
        cout = w32.shape[0]
        dev = x.device
        gy = gy.to(x.dtype).contiguous()
        if bn:
            g_affine = torch.zeros((2, cout), device=dev, dtype=torch.float32) if g_affine is None else g_affine.float().contiguous()
        need_dx, need_daff = ctx.needs_input_grad[0], ctx.needs_input_grad[1] and aff is not None
        dx = torch.empty_like(x) if (need_dx or need_daff) else None
        d_in = torch.empty((2, cin), device=dev, dtype=torch.float32) if need_daff else None
        dw = torch.empty((cout, cin), device=dev, dtype=torch.float32)
        db = torch.empty((cout,), device=dev, dtype=torch.float32) if has_b else None
        dgamma = torch.empty((cout,), device=dev, dtype=torch.float32) if bn else None
        dbeta = torch.empty((cout,), device=dev, dtype=torch.float32) if bn else None
        ws = torch.empty((L.pps_rows_layer_ws_bytes(cin, cout),), device=dev, dtype=torch.uint8)
        ptr = lambda t: None if t is None else t.data_ptr()
        _lib.check(L.pps_rows_layer_bwd(x.data_ptr(), y.data_ptr(), gy.data_ptr(), rows, cin, cout, _code(x.dtype), ptr(aff),
                                        None if aff is None else aff.data_ptr() + 4 * cin, int(in_relu), w32.data_ptr(), ptr(g32), ptr(save),
                                        ptr(g_affine) if bn else None, ptr(dx), None, ptr(d_in), dw.data_ptr(), ptr(db), ptr(dgamma), ptr(dbeta),
                                        ws.data_ptr(), _stream()), 'pps_rows_layer_bwd')
        return (dx if need_dx else None, d_in, None, dw.to(wdt), None if db is None else db.to(bdt), dgamma, dbeta, None, None, None, None)


class _RowsLayerMax(torch.autograd.Function):
    """rows_layer (with BatchNorm statistics) followed by act_max over the p rows of every group, as ONE node: the raw output is stored (its
    own backward needs it), but the gradient that comes back is one value per (group, channel) in the winning row -- pps_rows_layer_bwd_pooled
    rebuilds those rows on load, so the [rows, cout] gradient tensor (98 % zeros at p = 50) is neither written nor read (three passes over
    512 MB per step for the STN of PointNet).  Returns the activated maxima [groups, cout] fp32."""

    @staticmethod
    def forward(ctx, x, in_affine, in_relu, w, b, gamma, beta, running_mean, running_var, momentum, eps, relu, groups, p):
        _need_cuda(x, w)
        L = _lib.lib()
        x = _low(x)
        rows, cin = x.shape
        cout = w.shape[0]
        dev = x.device
        w32 = w.detach().float().contiguous()
        b32 = None if b is None else b.detach().float().contiguous()
        aff = None if in_affine is None else in_affine.detach().float().contiguous()
        g32, be32 = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        y = torch.empty((rows, cout), device=dev, dtype=x.dtype)
        out_affine = torch.empty((2, cout), device=dev, dtype=torch.float32)
        save = torch.empty((2, cout), device=dev, dtype=torch.float32)
        ws = torch.empty((L.pps_rows_layer_ws_bytes(cin, cout),), device=dev, dtype=torch.uint8)
        ptr = lambda t: None if t is None else t.data_ptr()
        _lib.check(L.pps_rows_layer_fwd(x.data_ptr(), rows, cin, _code(x.dtype), ptr(aff), None if aff is None else aff.data_ptr() + 4 * cin, int(bool(in_relu)),
                                        w32.data_ptr(), ptr(b32), cout, y.data_ptr(), g32.data_ptr(), be32.data_ptr(), ptr(running_mean), ptr(running_var),
                                        float(momentum or 0.0), float(eps or 0.0), out_affine.data_ptr(), save.data_ptr(), ws.data_ptr(), _stream()),
                   'pps_rows_layer_fwd')
        mx, mn = torch.empty((groups, cout), device=dev), torch.empty((groups, cout), device=dev)
        amx, amn = torch.empty((groups, cout), device=dev, dtype=torch.int32), torch.empty((groups, cout), device=dev, dtype=torch.int32)
        _lib.check(L.pps_rows_extrema_16(y.data_ptr(), groups, p, cout, _code(y.dtype), mx.data_ptr(), mn.data_ptr(), amx.data_ptr(), amn.data_ptr(), _stream()),
                   'pps_rows_extrema_16')
        scale, shift = out_affine[0], out_affine[1]
        up = scale >= 0
        ext, arg = torch.where(up, mx, mn), torch.where(up, amx, amn).to(torch.uint8)
        out = ext * scale + shift
        live = None
        if relu:
            live = out > 0
            out = torch.relu(out)
        ctx.save_for_backward(x, aff, w32, g32, save, y, ext, arg, scale.clone(), live)
        ctx.meta = (bool(in_relu), b is not None, w.dtype, None if b is None else b.dtype, int(p))
        return out

    @staticmethod
    def backward(ctx, dout):
        x, aff, w32, g32, save, y, ext, arg, scale, live = ctx.saved_tensors
        in_relu, has_b, wdt, bdt, p = ctx.meta
        L = _lib.lib()
        rows, cin = x.shape
        cout = w32.shape[0]
        dev = x.device
        d = dout.float()
        if live is not None:
            d = d * live
        g_affine = torch.stack([sum_rows(d * ext), sum_rows(d)]).contiguous()        # gradient of (scale, shift) of this layer's BatchNorm
        gval = (d * scale).to(x.dtype).contiguous()                                    # gradient of the raw output, per (group, channel)
        need_dx, need_daff = ctx.needs_input_grad[0], ctx.needs_input_grad[1] and aff is not None
        dx = torch.empty_like(x) if (need_dx or need_daff) else None
        d_in = torch.empty((2, cin), device=dev, dtype=torch.float32) if need_daff else None
        dw = torch.empty((cout, cin), device=dev, dtype=torch.float32)
        db = torch.empty((cout,), device=dev, dtype=torch.float32) if has_b else None
        dgamma, dbeta = torch.empty((cout,), device=dev, dtype=torch.float32), torch.empty((cout,), device=dev, dtype=torch.float32)
        ws = torch.empty((L.pps_rows_layer_ws_bytes(cin, cout),), device=dev, dtype=torch.uint8)
        ptr = lambda t: None if t is None else t.data_ptr()
        _lib.check(L.pps_rows_layer_bwd_pooled(x.data_ptr(), y.data_ptr(), gval.data_ptr(), arg.data_ptr(), p, rows, cin, cout, _code(x.dtype), ptr(aff),
                                               None if aff is None else aff.data_ptr() + 4 * cin, int(in_relu), w32.data_ptr(), g32.data_ptr(), save.data_ptr(),
                                               g_affine.data_ptr(), ptr(dx), ptr(d_in), dw.data_ptr(), ptr(db), dgamma.data_ptr(), dbeta.data_ptr(),
                                               ws.data_ptr(), _stream()), 'pps_rows_layer_bwd_pooled')
        return (dx if need_dx else None, d_in, None, dw.to(wdt), None if db is None else db.to(bdt), dgamma, dbeta, None, None, None, None, None, None, None)


class _Rows3Layer(torch.autograd.Function):
    """conv0a of PointNet in train(): x [rows, 3] fp32 -> (y [rows, 64] bf16, out_affine [2, 64] of the BatchNorm on y); x gets no gradient."""

    @staticmethod
    def forward(ctx, x, w, b, gamma, beta, running_mean, running_var, momentum, eps):
        _need_cuda(x, w)
        L = _lib.lib()
        x = x.detach().float().contiguous()
        rows = x.shape[0]
        w32 = w.detach().float().contiguous()
        b32 = None if b is None else b.detach().float().contiguous()
        g32, be32 = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        dt = torch.get_autocast_dtype('cuda') if torch.is_autocast_enabled('cuda') else torch.bfloat16
        y = torch.empty((rows, 64), device=x.device, dtype=dt if dt in LOW else torch.bfloat16)
        aff, save = torch.empty((2, 64), device=x.device), torch.empty((2, 64), device=x.device)
        ws = torch.empty((L.pps_rows3_ws_bytes(),), device=x.device, dtype=torch.uint8)
        ptr = lambda t: None if t is None else t.data_ptr()
        _lib.check(L.pps_rows3_fwd(x.data_ptr(), rows, w32.data_ptr(), ptr(b32), _code(y.dtype), y.data_ptr(), g32.data_ptr(), be32.data_ptr(), ptr(running_mean),
                                   ptr(running_var), float(momentum), float(eps), aff.data_ptr(), save.data_ptr(), ws.data_ptr(), _stream()), 'pps_rows3_fwd')
        ctx.save_for_backward(x, y, g32, save)
        ctx.meta = (b is not None, w.dtype, None if b is None else b.dtype)
        return y, aff

    @staticmethod
    def backward(ctx, gy, g_aff):
        x, y, g32, save = ctx.saved_tensors
        has_b, wdt, bdt = ctx.meta
        L = _lib.lib()
        dev = x.device
        gy = gy.to(y.dtype).contiguous()
        g_aff = torch.zeros((2, 64), device=dev) if g_aff is None else g_aff.float().contiguous()
        dw = torch.empty((64, 3), device=dev)
        db = torch.empty((64,), device=dev) if has_b else None
        dgamma, dbeta = torch.empty((64,), device=dev), torch.empty((64,), device=dev)
        ws = torch.empty((L.pps_rows3_ws_bytes(),), device=dev, dtype=torch.uint8)
        _lib.check(L.pps_rows3_bwd(x.data_ptr(), y.data_ptr(), gy.data_ptr(), x.shape[0], _code(y.dtype), g32.data_ptr(), save.data_ptr(), g_aff.data_ptr(), dw.data_ptr(),
                                   None if db is None else db.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), ws.data_ptr(), _stream()), 'pps_rows3_bwd')
        return None, dw.to(wdt), None if db is None else db.to(bdt), dgamma, dbeta, None, None, None, None


def rows3_layer(x, w, b, bn, relu=True):
    """x [rows, 3] -> Act(raw [rows, 64] bf16, affine of bn (train() statistics), relu)."""
    y, aff = _Rows3Layer.apply(x, w, b, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps)
    return Act(y, aff, relu)


class _PatchTransform(torch.autograd.Function):
    """out[q, i, :] = act(x)[q, i, :] (T[q] + I)^T for groups of p <= 64 rows (PointNet's feature transform, nn.py:330-331): x [Q*p, 64] stored
    activation, t [Q, 64, 64] the RAW output of the STN's fc3 (the identity is added inside) -> [Q*p, 64] bf16."""

    @staticmethod
    def forward(ctx, x, affine, relu, t, p):
        _need_cuda(x, t)
        L = _lib.lib()
        x = _low(x)
        t16 = t.to(x.dtype).contiguous()
        nq = t16.shape[0]
        aff = None if affine is None else affine.detach().float().contiguous()
        out = torch.empty_like(x)
        _lib.check(L.pps_patch_transform_fwd(x.data_ptr(), None if aff is None else aff.data_ptr(), None if aff is None else aff.data_ptr() + 256,
                                             int(bool(relu)), t16.data_ptr(), 1, nq, p, _code(x.dtype), out.data_ptr(), _stream()), 'pps_patch_transform_fwd')
        ctx.save_for_backward(x, aff, t16)
        ctx.meta = (bool(relu), p, t.dtype)
        return out

    @staticmethod
    def backward(ctx, g):
        x, aff, t16 = ctx.saved_tensors
        relu, p, tdt = ctx.meta
        L = _lib.lib()
        nq = t16.shape[0]
        g = g.to(x.dtype).contiguous()
        dx, dt = torch.empty_like(x), torch.empty_like(t16)
        need_daff = aff is not None and ctx.needs_input_grad[1]
        daff = torch.empty((2, 64), device=x.device) if need_daff else None
        ws = torch.empty((L.pps_patch_transform_ws_bytes(),), device=x.device, dtype=torch.uint8)
        _lib.check(L.pps_patch_transform_bwd(x.data_ptr(), None if aff is None else aff.data_ptr(), None if aff is None else aff.data_ptr() + 256,
                                             int(relu), t16.data_ptr(), 1, g.data_ptr(), nq, p, _code(x.dtype), dx.data_ptr(), dt.data_ptr(),
                                             None if daff is None else daff.data_ptr(), ws.data_ptr(), _stream()), 'pps_patch_transform_bwd')
        return dx, daff, None, dt.to(tdt), None


def patch_transform_supported(p, c):
    return 1 <= p <= 64 and c == 64


def patch_transform(act, t_raw, p):
    """act: Act with 64 channels; t_raw [Q, 64, 64] (fc3 output, identity not yet added) -> [Q*p, 64] bf16."""
    aff = act.affine
    if aff is None and act.relu:
        aff = torch.cat([torch.ones((1, 64), device=act.raw.device), torch.zeros((1, 64), device=act.raw.device)])
    return _PatchTransform.apply(act.raw, aff, act.relu, t_raw, p)


class _PatchAttn(torch.autograd.Function):
    """pooled[q] = sum_j softmax_j(h[q,j] . v) h[q,j] over the k <= 64 rows of every group, h [Q, k, 256] bf16 -> [Q, 256] fp32: logits, softmax
    and pooling in one kernel each way, h read once (pps_attn_train.hip)."""

    @staticmethod
    def forward(ctx, h, v):
        _need_cuda(h, v)
        h = _low(h)
        v32 = v.detach().float().contiguous()
        q, k, c = h.shape
        pooled = torch.empty((q, c), device=h.device, dtype=torch.float32)
        _lib.check(_lib.lib().pps_patch_attn_fwd(h.data_ptr(), v32.data_ptr(), q, k, c, _code(h.dtype), pooled.data_ptr(), _stream()), 'pps_patch_attn_fwd')
        ctx.save_for_backward(h, v32)
        ctx.vdtype = v.dtype
        return pooled

    @staticmethod
    def backward(ctx, dpooled):
        h, v32 = ctx.saved_tensors
        q, k, c = h.shape
        L = _lib.lib()
        dpooled = dpooled.float().contiguous()
        dh = torch.empty_like(h)
        part = torch.empty((L.pps_patch_attn_partials(q), c), device=h.device, dtype=torch.float32)
        _lib.check(L.pps_patch_attn_bwd(h.data_ptr(), v32.data_ptr(), dpooled.data_ptr(), q, k, c, _code(h.dtype), dh.data_ptr(), part.data_ptr(), _stream()),
                   'pps_patch_attn_bwd')
        return dh, sum_rows(part).to(ctx.vdtype)


def patch_attn_supported(k, c):
    return 1 <= k <= 64 and c == 256


def patch_attn(h, v):
    return _PatchAttn.apply(h, v)


class _ActMax(torch.autograd.Function):
    """max over the p rows of every group of act(raw) = relu?(raw * scale + shift) WITHOUT the activated tensor: the activation is monotone
    per channel, so the extremum of the raw rows (max where scale >= 0, min otherwise) is activated instead (pps_rows_extrema_bf16)."""

    @staticmethod
    def forward(ctx, raw, affine, relu, groups, p):
        _need_cuda(raw)
        raw = _low(raw)
        c = raw.shape[1]
        dev = raw.device
        mx, mn = torch.empty((groups, c), device=dev), torch.empty((groups, c), device=dev)
        amx, amn = torch.empty((groups, c), device=dev, dtype=torch.int32), torch.empty((groups, c), device=dev, dtype=torch.int32)
        _lib.check(_lib.lib().pps_rows_extrema_16(raw.data_ptr(), groups, p, c, _code(raw.dtype), mx.data_ptr(), mn.data_ptr(), amx.data_ptr(), amn.data_ptr(),
                                                  _stream()), 'pps_rows_extrema_16')
        if affine is None:
            ext, arg, out = mx, amx, mx
            scale = None
        else:
            scale, shift = affine[0].float(), affine[1].float()
            up = scale >= 0
            ext, arg = torch.where(up, mx, mn), torch.where(up, amx, amn)
            out = ext * scale + shift
        live = None
        if relu:
            live = out > 0
            out = torch.relu(out)
        ctx.save_for_backward(ext, arg, scale, live)
        ctx.meta = (groups, p, c, affine is not None, raw.dtype)
        return out

    @staticmethod
    def backward(ctx, dout):
        ext, arg, scale, live = ctx.saved_tensors
        groups, p, c, has_aff, rdt = ctx.meta
        d = dout.float()
        if live is not None:
            d = d * live
        daff = torch.stack([sum_rows(d * ext), sum_rows(d)]) if has_aff and ctx.needs_input_grad[1] else None
        dval = d * scale if has_aff else d
        draw = torch.zeros((groups, p, c), device=dout.device, dtype=rdt)
        draw.scatter_(1, arg.long().unsqueeze(1), dval.to(rdt).unsqueeze(1))
        return draw.view(groups * p, c), daff, None, None, None


def act_max(act, groups, p):
    """[groups * p, C] stored activation -> [groups, C] fp32: max over the p rows of every group of the ACTIVATED values."""
    return _ActMax.apply(act.raw, act.affine, act.relu, groups, p)


def rows_layer_supported(rows, cin, cout):
    return rows >= 1 and bool(_lib.lib().pps_rows_layer_supported(int(cin), int(cout)))


def rows_layer_max_supported(rows, cin, cout, groups, p):
    return rows == groups * p and bool(_lib.lib().pps_rows_layer_pooled_supported(int(cin), int(cout), int(p)))


def rows_layer_max(act, w, b, bn, relu, groups, p):
    """act_max(rows_layer(act, w, b, bn, relu), groups, p) as one autograd node (see _RowsLayerMax): -> [groups, cout] fp32."""
    return _RowsLayerMax.apply(act.raw, act.affine, act.relu, w, b, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps,
                               bool(relu), int(groups), int(p))


def rows_layer(act, w, b, bn=None, relu=False):
    """act: Act (input in its stored form); w [cout, cin], b [cout] or None; bn: BatchNorm1d holder whose batch statistics are taken on the
    output (train mode), or None.  -> Act(raw output, affine of bn or None, relu)."""
    if bn is not None:
        y, aff = _RowsLayer.apply(act.raw, act.affine, act.relu, w, b, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps)
    else:
        y, aff = _RowsLayer.apply(act.raw, act.affine, act.relu, w, b, None, None, None, None, None, None)
    return Act(y, aff, relu)


class _RowsLayerPatchAttn(torch.autograd.Function):
    """rows_layer with BatchNorm statistics (conv3 / bn3 of PointNet) followed by the single-head attention pooling over the p rows of every group
    (AttentionPoco, source/base/nn.py:84-96, on the RAW layer output: logit = raw . (w_q * scale) + const, and softmax ignores the constant) as
    ONE node.  The raw output is stored (both backward passes need it), but its gradient -- a[q, j] dpooled[q, :] + dl[q, j] v, rank two per group
    -- is never a tensor: pps_patch_attn_bwd_weights returns (a, dl) per row and pps_rows_layer_bwd_rank2 rebuilds the rows on load in the
    input-gradient and the weight-gradient kernel (1 M x 256 bf16 = 512 MB per step: written once and read twice otherwise).
    Returns (pooled raw rows [groups, cout] fp32, out_affine [2, cout]): the caller applies the BatchNorm's affine to the pooled rows (the
    weights of a group sum to 1)."""

    @staticmethod
    def forward(ctx, x, in_affine, in_relu, w, b, gamma, beta, running_mean, running_var, momentum, eps, wq, groups, p):
        _need_cuda(x, w, wq)
        L = _lib.lib()
        x = _low(x)
        rows, cin = x.shape
        cout = w.shape[0]
        dev = x.device
        w32 = w.detach().float().contiguous()
        b32 = None if b is None else b.detach().float().contiguous()
        aff = None if in_affine is None else in_affine.detach().float().contiguous()
        g32, be32 = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        y = torch.empty((rows, cout), device=dev, dtype=x.dtype)
        out_affine = torch.empty((2, cout), device=dev, dtype=torch.float32)
        save = torch.empty((2, cout), device=dev, dtype=torch.float32)
        ws = torch.empty((L.pps_rows_layer_ws_bytes(cin, cout),), device=dev, dtype=torch.uint8)
        ptr = lambda t: None if t is None else t.data_ptr()
        _lib.check(L.pps_rows_layer_fwd(x.data_ptr(), rows, cin, _code(x.dtype), ptr(aff), None if aff is None else aff.data_ptr() + 4 * cin, int(bool(in_relu)),
                                        w32.data_ptr(), ptr(b32), cout, y.data_ptr(), g32.data_ptr(), be32.data_ptr(), ptr(running_mean), ptr(running_var),
                                        float(momentum or 0.0), float(eps or 0.0), out_affine.data_ptr(), save.data_ptr(), ws.data_ptr(), _stream()),
                   'pps_rows_layer_fwd')
        wq32 = wq.detach().float().reshape(-1).contiguous()
        v = wq32 * out_affine[0]                                                     # the logit's weights on the raw output
        pooled = torch.empty((groups, cout), device=dev, dtype=torch.float32)
        _lib.check(L.pps_patch_attn_fwd(y.data_ptr(), v.data_ptr(), groups, p, cout, _code(y.dtype), pooled.data_ptr(), _stream()), 'pps_patch_attn_fwd')
        ctx.save_for_backward(x, aff, w32, g32, save, y, v, wq32, out_affine)
        ctx.meta = (bool(in_relu), b is not None, w.dtype, None if b is None else b.dtype, wq.dtype, tuple(wq.shape), groups, p)
        return pooled, out_affine

    @staticmethod
    def backward(ctx, dpooled, g_affine):
        x, aff, w32, g32, save, y, v, wq32, out_affine = ctx.saved_tensors
        in_relu, has_b, wdt, bdt, qdt, qshape, groups, p = ctx.meta
        L = _lib.lib()
        rows, cin = x.shape
        cout = w32.shape[0]
        dev = x.device
        st = _stream()
        f32e = lambda *shape: torch.empty(shape, device=dev, dtype=torch.float32)
        dpooled = torch.zeros((groups, cout), device=dev, dtype=torch.float32) if dpooled is None else dpooled.float().contiguous()
        a, dl = f32e(rows), f32e(rows)
        part = f32e(L.pps_patch_attn_partials(groups), cout)
        _lib.check(L.pps_patch_attn_bwd_weights(y.data_ptr(), v.data_ptr(), dpooled.data_ptr(), groups, p, cout, _code(y.dtype), a.data_ptr(), dl.data_ptr(),
                                                part.data_ptr(), st), 'pps_patch_attn_bwd_weights')
        dv = sum_rows(part)
        # v = w_q * scale: d w_q = dv * scale, and the scale of the layer's own BatchNorm gets dv * w_q on top of what its consumers hand back
        dwq = (dv * out_affine[0]).reshape(qshape).to(qdt)
        g_affine = torch.zeros((2, cout), device=dev, dtype=torch.float32) if g_affine is None else g_affine.float().clone()
        g_affine[0] += dv * wq32
        need_dx, need_daff = ctx.needs_input_grad[0], ctx.needs_input_grad[1] and aff is not None
        dx = torch.empty_like(x) if (need_dx or need_daff) else None
        d_in = f32e(2, cin) if need_daff else None
        dw = f32e(cout, cin)
        db = f32e(cout) if has_b else None
        dgamma, dbeta = f32e(cout), f32e(cout)
        ws = torch.empty((L.pps_rows_layer_ws_bytes(cin, cout),), device=dev, dtype=torch.uint8)
        ptr = lambda t: None if t is None else t.data_ptr()
        _lib.check(L.pps_rows_layer_bwd_rank2(x.data_ptr(), y.data_ptr(), a.data_ptr(), dl.data_ptr(), dpooled.data_ptr(), v.data_ptr(), p, rows, cin, cout,
                                              _code(x.dtype), ptr(aff), None if aff is None else aff.data_ptr() + 4 * cin, int(in_relu), w32.data_ptr(),
                                              g32.data_ptr(), save.data_ptr(), g_affine.data_ptr(), ptr(dx), ptr(d_in), dw.data_ptr(), ptr(db),
                                              dgamma.data_ptr(), dbeta.data_ptr(), ws.data_ptr(), st), 'pps_rows_layer_bwd_rank2')
        return (dx if need_dx else None, d_in, None, dw.to(wdt), None if db is None else db.to(bdt), dgamma, dbeta, None, None, None, None, dwq, None, None)


def rows_layer_patch_attn_supported(cin, cout, p, rows):
    return bool(_lib.lib().pps_rows_layer_pooled_supported(cin, cout, p)) and patch_attn_supported(p, cout) and rows % p == 0 and rows * p < 1 << 32


def rows_layer_patch_attn(act, w, b, bn, wq, groups, p):
    """act: Act; w [cout, cin]; bn: BatchNorm1d holder (train() statistics on the output); wq: weight of the attention's fc_query [1, cout].
    -> (pooled RAW rows [groups, cout] fp32, affine [2, cout] of bn)."""
    return _RowsLayerPatchAttn.apply(act.raw, act.affine, act.relu, w, b, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps, wq,
                                     groups, p)


def _query_attn_bwd(L, y3, qy, dpooled, wq32, k, dwq, dbq, ws, st):
    """Backward of (fc_query, attention pooling) on the stored raw output y3 of fc3 -> d y3 [Q*k, 256]; dwq / dbq are filled.  y3 has two
    consumers.  The pooling's gradient relu'(y3) * a[q, j] * dpooled[q, c] is one multiply per element, so it is NOT stored: pps_attn_pool_bwd_weights
    returns a [Q, k] and the input-gradient kernel of fc_query rebuilds the product where it adds the two gradients (pps_rows_layer_bwd_attn) --
    the [Q*k, 256] tensor (655 MB at 50 x 2000 x 64 rows) is neither written nor read.  PPS_ATTN_GRAD=stored: the pooling gradient goes through
    memory (pps_attn_pool_bwd + dx_add; also taken when rows * k does not fit the kernel's 32-bit row / k)."""
    rows, c = y3.shape
    heads = wq32.shape[0]
    code = _code(y3.dtype)
    ptr = lambda t: None if t is None else t.data_ptr()
    dqy, dy3 = torch.empty_like(qy), torch.empty_like(y3)
    if os.environ.get('PPS_ATTN_GRAD', 'rebuilt') != 'stored' and rows * k < 1 << 32:
        a = torch.empty((rows,), device=y3.device, dtype=torch.float32)
        _lib.check(L.pps_attn_pool_bwd_weights(qy.data_ptr(), y3.data_ptr(), dpooled.data_ptr(), rows // k, k, heads, c, code, 1, dqy.data_ptr(), a.data_ptr(),
                                               st), 'pps_attn_pool_bwd_weights')
        _lib.check(L.pps_rows_layer_bwd_attn(y3.data_ptr(), dqy.data_ptr(), rows, c, heads, code, wq32.data_ptr(), a.data_ptr(), dpooled.data_ptr(), k,
                                             dy3.data_ptr(), dwq.data_ptr(), ptr(dbq), ws.data_ptr(), st), 'pps_rows_layer_bwd_attn')
        return dy3
    _lib.check(L.pps_attn_pool_bwd(qy.data_ptr(), y3.data_ptr(), dpooled.data_ptr(), rows // k, k, heads, c, code, 1, dqy.data_ptr(), dy3.data_ptr(), st),
               'pps_attn_pool_bwd')
    _lib.check(L.pps_rows_layer_bwd(y3.data_ptr(), qy.data_ptr(), dqy.data_ptr(), rows, c, heads, code, None, None, 1, wq32.data_ptr(), None, None, None,
                                    dy3.data_ptr(), dy3.data_ptr(), None, dwq.data_ptr(), ptr(dbq), None, None, ws.data_ptr(), st), 'pps_rows_layer_bwd')
    return dy3


class _QueryAttnPool(torch.autograd.Function):
    """The end of the interpolation head on the stored (pre-ReLU) output y3 of fc3 [Q*k, 256]:  qy = fc_query(relu(y3)),  pooled[q] = sum_j mean_h
    softmax_j(qy) relu(y3)[q, j]  as ONE autograd node: y3 has two consumers, and their two gradients are summed inside the input-gradient
    kernel of fc_query (dx_add) instead of by a separate pass over [Q*k, 256]."""

    @staticmethod
    def forward(ctx, y3, wq, bq, k):
        _need_cuda(y3, wq)
        L = _lib.lib()
        y3 = _low(y3)
        rows, c = y3.shape
        heads = wq.shape[0]
        w32 = wq.detach().float().contiguous()
        b32 = None if bq is None else bq.detach().float().contiguous()
        qy = torch.empty((rows, heads), device=y3.device, dtype=y3.dtype)
        ws = torch.empty((L.pps_rows_layer_ws_bytes(c, heads),), device=y3.device, dtype=torch.uint8)
        ptr = lambda t: None if t is None else t.data_ptr()
        _lib.check(L.pps_rows_layer_fwd(y3.data_ptr(), rows, c, _code(y3.dtype), None, None, 1, w32.data_ptr(), ptr(b32), heads, qy.data_ptr(),
                                        None, None, None, None, 0.0, 0.0, None, None, ws.data_ptr(), _stream()), 'pps_rows_layer_fwd')
        pooled = torch.empty((rows // k, c), device=y3.device, dtype=y3.dtype)
        _lib.check(L.pps_attn_pool_fwd(qy.data_ptr(), y3.data_ptr(), rows // k, k, heads, c, _code(y3.dtype), 1, pooled.data_ptr(), _stream()), 'pps_attn_pool_fwd')
        ctx.save_for_backward(y3, w32, qy)
        ctx.meta = (k, bq is not None, wq.dtype, None if bq is None else bq.dtype)
        return pooled

    @staticmethod
    def backward(ctx, dpooled):
        y3, w32, qy = ctx.saved_tensors
        k, has_b, wdt, bdt = ctx.meta
        L = _lib.lib()
        rows, c = y3.shape
        heads = w32.shape[0]
        dev = y3.device
        dpooled = dpooled.to(y3.dtype).contiguous()
        dw = torch.empty((heads, c), device=dev, dtype=torch.float32)
        db = torch.empty((heads,), device=dev, dtype=torch.float32) if has_b else None
        ws = torch.empty((L.pps_rows_layer_ws_bytes(c, heads),), device=dev, dtype=torch.uint8)
        dy3 = _query_attn_bwd(L, y3, qy, dpooled, w32, k, dw, db, ws, _stream())
        return dy3, dw.to(wdt), None if db is None else db.to(bdt), None


def query_attn_pool(y3, wq, bq, k):
    """y3 [Q*k, 256] stored BEFORE its ReLU, fc_query weights [heads, 256] -> pooled [Q, 256] bf16."""
    return _QueryAttnPool.apply(y3, wq, bq, k)


class _HeadChain(torch.autograd.Function):
    """head_input -> fc2 -> fc3 -> fc_query -> attention pooling of the interpolation head as ONE autograd node whose forward is the fused chain
    kernel (pps_head_chain_fwd: the rows pass through the three layers in registers, the raw outputs the backward needs are written once) followed
    by pps_attn_pool_fwd.  The backward pass is the sequence of backward entries the separate nodes (_QueryAttnPool, _RowsLayer x 2, _HeadInput)
    run, on the same saved tensors."""

    @staticmethod
    def forward(ctx, table, ids, pts, query, k, wx, w2, b2, w3, b3, wq, bq):
        _need_cuda(table, ids, pts, query, wx, w2, w3, wq)
        L = _lib.lib()
        table = _low(table)
        ids = ids.contiguous()
        f32 = lambda t: None if t is None else t.detach().float().contiguous()
        pts32, q32, wx32, w2_32, w3_32, wq32 = f32(pts), f32(query), f32(wx), f32(w2), f32(w3), f32(wq)
        b2_32, b3_32, bq32 = f32(b2), f32(b3), f32(bq)
        nq, c, heads = q32.shape[0], table.shape[1], wq32.shape[0]
        dev, dt = table.device, table.dtype
        rows = nq * k
        pad = (rows + 255) // 256 * 256               # the kernel writes whole 256-row units (unconditional stores): padded storage, views of the rows
        h1, y2, y3 = (torch.empty((pad, c), device=dev, dtype=dt)[:rows] for _ in range(3))
        qy = torch.empty((pad, heads), device=dev, dtype=dt)[:rows]
        ws = torch.empty((L.pps_head_chain_ws_bytes(),), device=dev, dtype=torch.uint8)
        ptr = lambda t: None if t is None else t.data_ptr()
        _lib.check(L.pps_head_chain_fwd(table.data_ptr(), ids.data_ptr(), pts32.data_ptr(), q32.data_ptr(), nq, k, _code(dt), wx32.data_ptr(), w2_32.data_ptr(),
                                        ptr(b2_32), w3_32.data_ptr(), ptr(b3_32), wq32.data_ptr(), ptr(bq32), h1.data_ptr(), y2.data_ptr(), y3.data_ptr(),
                                        qy.data_ptr(), ws.data_ptr(), _stream()), 'pps_head_chain_fwd')
        pooled = torch.empty((nq, c), device=dev, dtype=dt)
        _lib.check(L.pps_attn_pool_fwd(qy.data_ptr(), y3.data_ptr(), nq, k, heads, c, _code(dt), 1, pooled.data_ptr(), _stream()), 'pps_attn_pool_fwd')
        ctx.save_for_backward(ids, pts32, q32, w2_32, w3_32, wq32, h1, y2, y3, qy)
        ctx.meta = (table.shape[0], k, wx.dtype, tuple(wx.shape), w2.dtype, w3.dtype, wq.dtype, None if b2 is None else b2.dtype,
                    None if b3 is None else b3.dtype, None if bq is None else bq.dtype)
        return pooled

    @staticmethod
    def backward(ctx, dpooled):
        ids, pts32, q32, w2_32, w3_32, wq32, h1, y2, y3, qy = ctx.saved_tensors
        n, k, wxdt, wxshape, w2dt, w3dt, wqdt, b2dt, b3dt, bqdt = ctx.meta
        L = _lib.lib()
        rows, c = y3.shape
        heads = wq32.shape[0]
        dev, dt = y3.device, y3.dtype
        code = _code(dt)
        st = _stream()
        ptr = lambda t: None if t is None else t.data_ptr()
        dpooled = dpooled.to(dt).contiguous()
        # attention pooling + fc_query: the two gradients of y3 are summed inside fc_query's input-gradient kernel
        f32e = lambda *shape: torch.empty(shape, device=dev, dtype=torch.float32)
        dwq, dbq = f32e(heads, c), (f32e(heads) if bqdt is not None else None)
        ws = torch.empty((L.pps_rows_layer_ws_bytes(c, c),), device=dev, dtype=torch.uint8)        # (the largest of the three layers)
        dy3 = _query_attn_bwd(L, y3, qy, dpooled, wq32, k, dwq, dbq, ws, st)
        # fc3
        dy2 = torch.empty_like(y2)
        dw3, db3 = f32e(c, c), (f32e(c) if b3dt is not None else None)
        _lib.check(L.pps_rows_layer_bwd(y2.data_ptr(), y3.data_ptr(), dy3.data_ptr(), rows, c, c, code, None, None, 1, w3_32.data_ptr(), None, None, None,
                                        dy2.data_ptr(), None, None, dw3.data_ptr(), ptr(db3), None, None, ws.data_ptr(), st), 'pps_rows_layer_bwd')
        del dy3
        # fc2
        dh1 = torch.empty_like(h1)
        dw2, db2 = f32e(c, c), (f32e(c) if b2dt is not None else None)
        _lib.check(L.pps_rows_layer_bwd(h1.data_ptr(), y2.data_ptr(), dy2.data_ptr(), rows, c, c, code, None, None, 1, w2_32.data_ptr(), None, None, None,
                                        dh1.data_ptr(), None, None, dw2.data_ptr(), ptr(db2), None, None, ws.data_ptr(), st), 'pps_rows_layer_bwd')
        del dy2
        # head input: d table by the segmented sum of the gather, d wx
        dtable = dwx = None
        if ctx.needs_input_grad[0]:
            order, offsets = csr(ids, n)
            dt32 = f32e(n, c)
            _lib.check(L.pps_segment_sum_rows_16(dh1.data_ptr(), order.data_ptr(), offsets.data_ptr(), n, c, code, dt32.data_ptr(), st), 'pps_segment_sum_rows_16')
            dtable = dt32.to(dt)
        if ctx.needs_input_grad[5]:
            dwx = f32e(c, 3)
            ws2 = torch.empty((L.pps_head_input_ws_bytes(c),), device=dev, dtype=torch.uint8)
            _lib.check(L.pps_head_input_dwx(dh1.data_ptr(), ids.data_ptr(), pts32.data_ptr(), q32.data_ptr(), q32.shape[0], k, c, code, dwx.data_ptr(),
                                            ws2.data_ptr(), st), 'pps_head_input_dwx')
            dwx = dwx.reshape(wxshape).to(wxdt)
        cast = lambda t, d: None if (t is None or d is None) else t.to(d)
        return (dtable, None, None, None, None, dwx, dw2.to(w2dt), cast(db2, b2dt), dw3.to(w3dt), cast(db3, b3dt), dwq.to(wqdt), cast(dbq, bqdt))


def head_chain_supported(c, heads, k):
    return c == 256 and heads == 64 and 1 <= k <= 64


_head_chain_checked = {}


def head_chain_trusted(dtype):
    """First-use self-check of the one-kernel head chain, once per process and storage type: a fixed synthetic case (2003 queries x 64
    neighbours on a 10 000-row table; a partial last row unit) through pps_head_chain_fwd and through the separate launches it replaces
    (pps_head_input_fwd + pps_rows_layer_fwd x 3): h1 must be EQUAL, y2 / y3 / qy within four units of the storage type's last place on the
    tensor's scale (the accumulation order inside an MFMA may move a value across a rounding boundary), the chain itself equal on a second launch.  False -> the
    caller keeps to the separate launches (also hand-written HIP kernels) and a warning says so once.  Not run while a stream is being captured
    (the eager warm-up steps of a fit come first); PPS_HEAD_CHAIN_CHECK=0 skips it."""
    hit = _head_chain_checked.get(dtype)
    if hit is not None:
        return hit
    if os.environ.get('PPS_HEAD_CHAIN_CHECK', '1') == '0' or torch.cuda.is_current_stream_capturing():
        return True
    dev = torch.device('cuda', torch.cuda.current_device())
    g = torch.Generator().manual_seed(20260930)
    r = lambda *sh, scale=1.0: (torch.randn(*sh, generator=g) * scale).to(dev)
    nq, k, n = 2003, 64, 10000
    table = r(n, 256).to(dtype)
    ids = torch.randint(0, n, (nq * k,), generator=g).to(dev)
    pts, query = (torch.rand(n, 3, generator=g) - 0.5).to(dev), (torch.rand(nq, 3, generator=g) - 0.5).to(dev)
    wx, w2, w3, wq = r(256, 3, scale=0.5), r(256, 256, scale=1 / 16), r(256, 256, scale=1 / 16), r(64, 256, scale=1 / 8)
    b2, b3, bq = r(256, scale=0.1), r(256, scale=0.1), r(64, scale=0.1)
    L = _lib.lib()
    rows = nq * k
    pad = (rows + 255) // 256 * 256

    def chain():
        h1, y2, y3 = (torch.zeros((pad, 256), device=dev, dtype=dtype) for _ in range(3))
        qy = torch.zeros((pad, 64), device=dev, dtype=dtype)
        ws = torch.empty((L.pps_head_chain_ws_bytes(),), device=dev, dtype=torch.uint8)
        _lib.check(L.pps_head_chain_fwd(table.data_ptr(), ids.data_ptr(), pts.data_ptr(), query.data_ptr(), nq, k, _code(dtype), wx.data_ptr(), w2.data_ptr(),
                                        b2.data_ptr(), w3.data_ptr(), b3.data_ptr(), wq.data_ptr(), bq.data_ptr(), h1.data_ptr(), y2.data_ptr(),
                                        y3.data_ptr(), qy.data_ptr(), ws.data_ptr(), _stream()), 'pps_head_chain_fwd')
        return h1[:rows], y2[:rows], y3[:rows], qy[:rows]

    with torch.no_grad(), torch.autocast('cuda', dtype=dtype):
        a = chain()
        b = chain()
        h1 = head_input(table, ids, pts, query, k, wx)
        y2 = rows_layer(Act(h1, None, True), w2, b2, None, True)
        y3 = rows_layer(y2, w3, b3, None, True)
        qy = rows_layer(y3, wq, bq, None, True)
    ref = (h1.view(rows, 256), y2.raw.view(rows, 256), y3.raw.view(rows, 256), qy.raw.view(rows, 64))
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    ok = all(torch.equal(x, y) for x, y in zip(a, b)) and torch.equal(a[0], ref[0])
    for x, y in zip(a[1:], ref[1:]):                 # (a value that crosses a rounding boundary in one layer moves the next layer's sums by a few ulp of
        ok = ok and float((x.float() - y.float()).abs().max()) <= 4.0 * ulp * float(y.float().abs().max())      # their TERMS: the bar is on the tensor's scale)
    _head_chain_checked[dtype] = ok
    if not ok:
        print('ppsurf_amd: the one-kernel head chain FAILED its first-use self-check ({}); the interpolation head runs as separate launches in this '
              'process (PPS_HEAD_CHAIN=0 selects that form outright)'.format(dtype))
    return ok


def head_chain(table, ids, pts, query, k, wx, fc2, fc3, fc_query):
    """pooled [Q, 256] of the interpolation head from the per-point table: fc2 / fc3 / fc_query are (weight [out, 256], bias or None) pairs."""
    return _HeadChain.apply(table, ids, pts, query, k, wx, fc2[0], fc2[1], fc3[0], fc3[1], fc_query[0], fc_query[1])


def attn_pool_supported(k, heads, c):
    return 1 <= k <= 64 and 1 <= heads <= 64 and c <= 256


def attn_pool(qy, h, relu_h=False):
    """qy [Q,k,64] attention logits, h [Q,k,C] -> pooled [Q,C] (dtype of h).  relu_h: h is the stored PRE-activation, relu(h) is pooled."""
    return _AttnPool.apply(qy, h, relu_h)


def bn_supported(rows, c):
    return c % 4 == 0 and c <= 1024 and 256 % (c // 4) == 0


def bn_act(x, weight, bias, running_mean, running_var, momentum, eps, relu):
    """Train-mode BatchNorm1d over the rows of x [rows, c] (batch statistics, running statistics updated in place) with an
    optional fused ReLU; x fp32 or bf16 (kept), statistics and gradients of the affine parameters in fp32."""
    return _BnAct.apply(x, weight, bias, running_mean, running_var, momentum, eps, relu)


def gather_rows(x, idx):
    return _GatherRows.apply(x, idx)


def neighbour_max(x, idx):
    """A maximum of 16-bit values is one of them: the result goes back to the type it came in (the op computes in fp32), so that the residual
    sum it feeds stays a same-type addition (a bf16 + fp32 addition runs ATen's slow mixed-type kernel and promotes everything downstream)."""
    return _NeighbourMax.apply(x, idx)                    # 16-bit storage in, the same storage out (pps_gather_max_arg_16): no cast kernels


def neighbour_contract(x, idx, g):
    return _NeighbourContract.apply(x, idx, g)
