"""Train-mode forward of the networks as an autograd graph (fit / training_step path).

The inference path (ppsurf_amd/decoder.py, encoder.py) folds eval-mode BatchNorm into packed weights and never leaves
registers; training cannot do that: every BatchNorm needs statistics over the whole batch before the next layer starts, the
activations are needed again by the backward pass, and the weights change every step.  So the training step is organised
differently: POINT-MAJOR activations `[rows, C]` that stay resident in HBM (a fit batch is ~3 GB of activations on a 288 GB
part), dense layers as plain library GEMMs over all rows of the batch at once, and the neighbourhood work -- kNN tables, patch
search, gathers, per-neighbourhood pooling -- as HIP kernels with hand-written backward (ppsurf_amd/train_ops.py).  The id
tables are built on the device inside the step (spatial.get_data_poco), not by DataLoader worker processes.

The parameter holders of ppsurf_amd/modules.py are used directly (same state-dict names as the reference, so the optimizer
classes named in configs/poco.yaml work on `model.parameters()` and checkpoints stay interchangeable).

Semantics restated from the reference, train() mode:
  FKAConvLayer       source/base/nn.py:592-652  (norm_radius EMA :608-613, InstanceNorm skipped when K == 1 :627-638)
  ResidualBlock      source/base/nn.py:438-450
  FKAConvNetwork     source/base/nn.py:508-554  (x4d_bug_fixed :531-534)
  InterpAttention    source/poco_model.py:381-419
  STN / PointNetfeat source/base/nn.py:162-190, 305-373 ; AttentionPoco :84-96 ; MLP :376-417
  networks           source/ppsurf_model.py:70-117, source/poco_model.py:345-359
Parity: tests/test_train_graph_cpu.py and tests/test_gpu_train.py against tests/golden/train_*.npz (outputs, loss, updated
buffers and parameter gradients recorded from the reference itself).
"""
import weakref

import torch
import torch.nn.functional as F

from . import train_ops


# ---------------------------------------------------------------------------------------------------------------------
# small layers on [rows, C]
# ---------------------------------------------------------------------------------------------------------------------
def _w2d(layer):
    w = layer.weight
    return w.reshape(w.shape[0], -1)


SPLITK_MIN_ROWS = 32768
SPLITK_SLABS = 64            # row slabs of the split-K weight gradient (64 / 128 / 256 measured within 3 % of each other).  NOT more than 64: with 128
                             # or 256 batches the library's strided-batched bf16 GEMM faults when the step is replayed as a HIP graph (memory
                             # aperture violation on replay, round-2 probe fit_graph_matrix.py, git history; 32 and 64 replay correctly)
SPLITK_EXACT = (64, 32, 16, 8, 50, 40, 25, 20, 10, 5, 4)      # slab counts tried in this order: the first that divides the rows into slabs of
SPLITK_SLAB_ROWS = 1000                                        # at least this many rows (no ragged tail -> no second GEMM + add for it)


def splitk_slabs(rows):
    """Slab count of the split-K weight gradient for `rows` rows, or 0 for one plain GEMM.  The encoder's levels (100 000 / 25 000 / 6 250 rows
    per batch of 10) and the 20 000 query rows of the MLP run 3-6x faster split than as ONE library GEMM with a 256 x 256 (or smaller)
    result, which occupies a handful of CUs (round-2 probe time_dw_small.py, git history: 25 000 x 128 -> 64: 96 us whole, 16 us as 8 slabs)."""
    for s in SPLITK_EXACT:
        if rows % s == 0 and rows // s >= SPLITK_SLAB_ROWS:
            return s
    return SPLITK_SLABS if rows >= SPLITK_MIN_ROWS else 0


# bf16 images of the fp32 master parameters, refreshed once per forward pass by ONE launch (prepare_shadows -> pps_cast_pieces): the library GEMMs
# of the encoder otherwise cast every weight and bias with a launch of its own (~140 four-microsecond kernels per step)
_shadow = {}                 # id(parameter) -> bf16 tensor of the same shape (persistent storage)
_shadow_live = [False]
_shadow_ver = {}             # id(parameter) -> parameter._version when its image was taken (an optimizer step in between makes the image stale)


_cast_tables = {}            # (device, dtype, pointers of every (parameter, image) pair) -> (key, device table of pieces, number of pieces).  Entries
                             # are never replaced or dropped: a pps_cast_pieces launch recorded into a HIP graph reads its table by ADDRESS on every
                             # replay, so a table must outlive every graph that may hold it (80 KB per parameter set)
CAST_PIECE = 4096


def _cast_all(params, dtype):
    """All images in ONE launch (pps_cast_pieces) through a device table of (source, image, count) pieces; the table holds raw pointers and is
    rebuilt when a parameter or an image moved.  False (the caller copies with torch): a table would have to be uploaded while a graph is being
    captured, non-contiguous tensors, parameters on several devices."""
    code = {torch.bfloat16: 1, torch.float16: 2}.get(dtype)
    if code is None or len({p.device for p in params}) != 1:
        return False
    if any(not p.is_contiguous() or not _shadow[id(p)].is_contiguous() for p in params):
        return False
    from . import _lib
    import numpy as np
    dev = params[0].device
    key = tuple((p.data_ptr(), _shadow[id(p)].data_ptr(), p.numel()) for p in params)
    cached = _cast_tables.get((dev, dtype, key))
    if cached is None:
        if torch.cuda.is_current_stream_capturing():
            return False
        assert _lib.lib().pps_cast_piece_bytes() == 24
        rows = [(sp + 4 * off, dp + 2 * off, min(CAST_PIECE, n - off)) for sp, dp, n in key for off in range(0, n, CAST_PIECE)]
        cached = (key, torch.from_numpy(np.asarray(rows, dtype=np.int64)).to(dev), len(rows))
        _cast_tables[(dev, dtype, key)] = cached
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().pps_cast_pieces(cached[1].data_ptr(), cached[2], code, torch.cuda.current_stream(dev).cuda_stream), 'pps_cast_pieces')
    return True


_shadow_t = {}               # id(parameter) -> 16-bit image of the parameter as a [K, N] matrix (N = shape[0]): the A operand of the input-gradient product
_tr_tables = {}              # like _cast_tables, for pps_transpose_cast_pieces


def _transpose_all(params, dtype):
    """Transposed images of every matrix-shaped parameter in ONE launch (pps_transpose_cast_pieces).  False: not possible now (see _cast_all)."""
    code = {torch.bfloat16: 1, torch.float16: 2}.get(dtype)
    mats = [p for p in params if p.dim() >= 2]
    if code is None or not mats or len({p.device for p in mats}) != 1 or any(not p.is_contiguous() for p in mats):
        return False
    from . import _lib
    import numpy as np
    dev = mats[0].device
    key = tuple((p.data_ptr(), _shadow_t[id(p)].data_ptr(), p.shape[0], p.numel() // p.shape[0]) for p in mats)
    cached = _tr_tables.get((dev, dtype, key))
    if cached is None:
        if torch.cuda.is_current_stream_capturing():
            return False
        assert _lib.lib().pps_transpose_entry_bytes() == 32
        rows, tile0 = [], 0
        for sp, dp, n, k in key:
            rows.append((sp, dp, n | (k << 32), tile0))                       # {src, dst, int32 n, int32 k, int64 tile0}, little endian
            tile0 += ((n + 31) // 32) * ((k + 31) // 32)
        cached = (key, torch.from_numpy(np.asarray(rows, dtype=np.int64)).to(dev), len(rows), tile0)
        _tr_tables[(dev, dtype, key)] = cached
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().pps_transpose_cast_pieces(cached[1].data_ptr(), cached[2], cached[3], code, torch.cuda.current_stream(dev).cuda_stream),
                   'pps_transpose_cast_pieces')
    return True


def _drop_images(pid):
    _shadow.pop(pid, None)
    _shadow_t.pop(pid, None)
    _shadow_ver.pop(pid, None)


def prepare_shadows(module, dtype=torch.bfloat16):
    """Call at the start of an autocast forward pass in train(): parameters -> persistent images in the autocast type (bf16 / fp16)."""
    params = [p for p in module.parameters() if p.is_cuda and p.dtype == torch.float32]
    if not params:
        return
    with torch.no_grad():
        for p in params:
            t = _shadow.get(id(p))
            if t is None or t.shape != p.shape or t.device != p.device or t.dtype != dtype:
                if t is None:
                    # the images die with their parameter (every model created in a process used to leave its 16-bit images -- 55 MB for
                    # PPSurf with the transposed ones -- behind for good).  A recorded step reads images by address, but it also holds its model.
                    weakref.finalize(p, _drop_images, id(p))
                _shadow[id(p)] = torch.empty_like(p, dtype=dtype)
        if not _cast_all(params, dtype):
            torch._foreach_copy_([_shadow[id(p)] for p in params], params)
        for p in params:
            if p.dim() >= 2:
                t = _shadow_t.get(id(p))
                shape = (p.numel() // p.shape[0], p.shape[0])
                if t is None or tuple(t.shape) != shape or t.device != p.device or t.dtype != dtype:
                    _shadow_t[id(p)] = torch.empty(shape, device=p.device, dtype=dtype)
        if not _transpose_all(params, dtype):
            for p in params:
                if p.dim() >= 2:
                    _shadow_t[id(p)].copy_(p.reshape(p.shape[0], -1).t())
    for p in params:
        _shadow_ver[id(p)] = p._version
    _shadow_live[0] = True


def _bf16_of(param):
    """The 16-bit image of a parameter if prepare_shadows() took it during THIS forward pass in the autocast type in force, else None
    (a flag left set by a pass that never reached release_step_caches(), or an image older than the parameter, must not be used)."""
    if not _shadow_live[0] or param is None or not torch.is_autocast_enabled('cuda'):
        return None
    t = _shadow.get(id(param))
    if t is None or t.dtype != torch.get_autocast_dtype('cuda') or _shadow_ver.get(id(param)) != param._version:
        return None
    return t


def _bf16_t_of(param):
    """The transposed 16-bit image [K, N] of a matrix-shaped parameter, under the same conditions as _bf16_of."""
    if _bf16_of(param) is None:
        return None
    return _shadow_t.get(id(param))


_mm_fp32_out = [True]


def _mm_f32(a, b):
    """a @ b for bf16 operands with an fp32 result written by the GEMM itself (no separate cast of the weight gradient)."""
    if _mm_fp32_out[0] and a.is_cuda and a.dtype in train_ops.LOW:
        try:
            return torch.mm(a, b, out_dtype=torch.float32)
        except (TypeError, RuntimeError, NotImplementedError):
            _mm_fp32_out[0] = False
    return (a @ b).float()


def _bias_grad(g):
    """Column sums of a gradient [rows, n] in fp32 by the HIP reduction for ANY n (train_ops.sum_rows: the kernel takes <= 1024 columns, wider layers
    -- the 4096 outputs of the STN's last layer -- go through it in column blocks, the 2-column output layer padded to 4).  NOT torch's sum: inside
    a replayed HIP graph `g.sum(0, dtype=fp32)` of that [rows, 4096] bf16 tensor returned values that depended on the memory layout of the
    recording from the second replay on (the config-1 fit: one tensor of the 455 in the checkpoint, stn2.fc3.bias, differed between two builds whose
    eager fits are bit-identical; profiles/NOTES_r5.md section 3) -- the replayed step was not the eager step."""
    return train_ops.sum_rows(g)


class _RowsLinear(torch.autograd.Function):
    """y = x W^T + b for x [rows, K] (rows = all points / patch points / neighbours of the batch).

    Device tensors under 16-bit autocast: the three products are the hand-written kernels of csrc/pps_gemm_train.hip for ANY layer shape
    (train_ops.gemm_nt for the forward and -- with the transposed weight image -- the input gradient, train_ops.gemm_tn for the weight gradient,
    whose contraction runs over the rows: row slabs with fp32 partials summed in a fixed order), the bias gradient is the column-sum kernel.
    Until round 4 these were library GEMMs (hipBLASLt behind F.linear / mm / a split-K bmm); they remain only for fp32 / float64 tensors
    (trainer.precision 32, the CPU suite's torch twins) and for K that is not a multiple of 8 (the 3-channel offset layer of the unfused head)."""

    @staticmethod
    def forward(ctx, x, w, b, wc=None, bc=None, wt=None):
        dev = x.device.type
        dt = torch.get_autocast_dtype(dev) if torch.is_autocast_enabled(dev) else x.dtype
        xc = x.to(dt)
        wc = wc if (wc is not None and wc.dtype == dt) else w.to(dt)            # wc / bc: 16-bit images of the step (prepare_shadows)
        n, k = wc.shape
        ctx.own = train_ops.gemm_supported(xc, k) and xc.dim() == 2 and xc.shape[0] > 0      # (no rows: F.linear and its plain backward)
        if ctx.own:
            npad = (n + 7) // 8 * 8                                             # (the 256 -> 2 output layer: zero rows up to a multiple of 8)
            if npad != n:
                wc = torch.cat([wc, wc.new_zeros((npad - n, k))])
                wt = None
            bias = None if b is None else (b.detach().float() if npad == n else torch.cat([b.detach().float(), b.new_zeros(npad - n, dtype=torch.float32)]))
            y = train_ops.gemm_nt(xc, wc, bias)
            if npad != n:
                y = y[:, :n].contiguous()
            ctx.save_for_backward(xc, wc, wt if (wt is not None and wt.dtype == dt) else None)
            ctx.meta = (x.dtype, w.dtype, None if b is None else b.dtype, n)
            return y
        bc = None if b is None else (bc if (bc is not None and bc.dtype == dt) else b.to(dt))
        with torch.autocast(dev, enabled=False):
            y = F.linear(xc, wc, bc)
        ctx.save_for_backward(xc, wc, None)
        ctx.meta = (x.dtype, w.dtype, None if b is None else b.dtype, n)
        return y

    @staticmethod
    def backward(ctx, g):
        xc, wc, wt = ctx.saved_tensors
        xdt, wdt, bdt, n = ctx.meta
        g = g.to(xc.dtype).contiguous()
        dx = dw = db = None
        if ctx.own:
            npad = wc.shape[0]
            gp = g if npad == n else F.pad(g, (0, npad - n))
            if ctx.needs_input_grad[0]:
                dx = train_ops.gemm_nt(gp, wt if wt is not None else wc.t().contiguous()).to(xdt)      # dx = g W: NT product with the [K, N] image
            if ctx.needs_input_grad[1]:
                dw = train_ops.gemm_tn(gp, xc)[:n].to(wdt)
            if bdt is not None and ctx.needs_input_grad[2]:
                db = _bias_grad(gp)[:n].to(bdt)
            return dx, dw, db, None, None, None
        acc = torch.float64 if xc.dtype == torch.float64 else torch.float32
        if ctx.needs_input_grad[0]:
            dx = (g @ wc).to(xdt)
        if ctx.needs_input_grad[1]:
            rows = g.shape[0]
            s = splitk_slabs(rows) if xc.dtype in train_ops.LOW else (SPLITK_SLABS if rows >= SPLITK_MIN_ROWS else 0)
            if s == 0:
                dw = _mm_f32(g.t(), xc) if acc == torch.float32 else g.t() @ xc
            else:
                rs = rows // s
                main = rs * s
                dw = torch.bmm(g[:main].view(s, rs, -1).transpose(1, 2), xc[:main].view(s, rs, -1)).sum(0, dtype=acc)
                if main < rows:
                    dw = dw + (g[main:].t() @ xc[main:]).to(acc)
            dw = dw.to(wdt)
        if bdt is not None and ctx.needs_input_grad[2]:
            db = train_ops.col_sum(g) if acc == torch.float32 else None
            db = (g.sum(0, dtype=acc) if db is None else db).to(bdt)
        return dx, dw, db, None, None, None


def _own_gemm(x, k):
    dev = x.device.type
    return (x.is_cuda and x.dim() == 2 and torch.is_autocast_enabled(dev) and torch.get_autocast_dtype(dev) in train_ops.LOW and k % 8 == 0 and k >= 8
            and x.shape[0] > 0)


ROWS_LAYER_MIN = 16384       # rows from which a 64 / 128 / 256-channel dense layer goes through the LDS-resident-weight row kernels (pps_rows_train.hip)


def rows_linear(x, w, b=None, wc=None, bc=None, wt=None):
    if (x.dim() == 2 and x.shape[0] >= ROWS_LAYER_MIN and _own_gemm(x, w.shape[1]) and train_ops.rows_layer_supported(x.shape[0], w.shape[1], w.shape[0])):
        # the whole weight matrix fits the LDS of a CU: rows stream past it once (forward), and the backward pass is ONE call (input gradient,
        # weight gradient from row slabs, bias gradient) -- the generic NT kernel re-reads the rows once per 64-channel block
        return train_ops.rows_layer(train_ops.Act(x), w, b).raw
    if x.dim() == 2 and (x.shape[0] >= SPLITK_MIN_ROWS or wc is not None or _own_gemm(x, w.shape[1])):
        return _RowsLinear.apply(x, w, b, wc, bc, wt)
    return F.linear(x, w, b)


def dense(layer, x):
    """1x1 Conv1d / Conv2d / Linear holder applied to rows."""
    wc = _bf16_of(layer.weight)
    return rows_linear(x, _w2d(layer), layer.bias, None if wc is None else wc.reshape(wc.shape[0], -1), _bf16_of(layer.bias), _bf16_t_of(layer.weight))


_counters = []
_radii = ([], [])            # (norm_radius buffers, their new values) of the FKAConv layers evaluated since the last flush


def _count_batch(bn):
    """num_batches_tracked += 1 of a BatchNorm in train(): collected and applied by ONE multi-tensor kernel at the end of the forward pass
    (one 4-byte kernel per BatchNorm otherwise: 60 launches per step)."""
    if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
        _counters.append(bn.num_batches_tracked)


def flush_batch_counters():
    if _counters:
        torch._foreach_add_(list(_counters), 1)
        _counters.clear()
    if _radii[0]:
        with torch.no_grad():                                   # IN PLACE: a replayed HIP graph reads and writes the buffers' own storage
            torch._foreach_copy_(list(_radii[0]), list(_radii[1]))
        _radii[0].clear()
        _radii[1].clear()


def _store_radius(layer, radius):
    """norm_radius <- the value the geometry kernel moved it to (nn.py:614-620, train()): collected like the BatchNorm counters and written by ONE
    multi-tensor copy when the outermost graph function returns (a 4-byte device memcpy per FKAConv layer otherwise: 10 per step, 5 us each on
    the step's queue); at once when called outside the graph functions."""
    value = radius.detach().reshape(layer.norm_radius.shape)
    if _depth[0] > 0:
        _radii[0].append(layer.norm_radius)
        _radii[1].append(value)
    else:
        with torch.no_grad():
            layer.norm_radius.copy_(value)


_depth = [0]


def _counted(fn):
    """The outermost graph function that returns applies the collected counter updates (the module forwards of ppsurf_amd/modules.py and
    the tests enter the graph at different levels)."""
    import functools

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        _depth[0] += 1
        try:
            return fn(*args, **kwargs)
        finally:
            _depth[0] -= 1
            if _depth[0] == 0:
                flush_batch_counters()
    return wrapper


def batch_norm(bn, x, relu=False):
    """BatchNorm1d holder on [rows, C] (+ fused ReLU): batch statistics + running-stat update in train() through the fused HIP
    op (2 + 2 streaming passes instead of torch's 7, train_ops.bn_act), running statistics in eval()."""
    _count_batch(bn)
    if bn.training and train_ops.bn_supported(x.shape[0], x.shape[1]):
        return train_ops.bn_act(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps, relu)
    y = F.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, bn.training, bn.momentum, bn.eps)
    return F.relu(y) if relu else y


def batch_norm_add_relu(bn, x, res):
    """relu(bn(x) + res) on [rows, C]: the tail of a residual block (source/base/nn.py:448-450).  In train() on the device the sum and the ReLU ride
    in the BatchNorm's apply pass (train_ops.bn_add_relu: two passes over the block's output less forward, one less backward, three launches
    less per block); otherwise the three separate ops."""
    if (bn.training and x.is_cuda and res.shape == x.shape and res.dtype == x.dtype and train_ops.bn_supported(x.shape[0], x.shape[1])
            and x.dtype in (torch.float32,) + train_ops.LOW and _os.environ.get('PPS_BN_ADD_RELU', '1') != '0'):
        _count_batch(bn)
        return train_ops.bn_add_relu(x, res, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps)
    return F.relu(batch_norm(bn, x) + res)


_flat_cache = {}


def _flat_ids(ids, n, up=False):
    """[B, M, K] ids into a batch item -> flat row numbers into [B*n, C].  Memoised per table (a table serves several
    layers, and the CSR used by the backward kernels is cached on the flat tensor).  up: nearest up-sampling table, -1 -> row 0."""
    key = (ids.data_ptr(), ids._version, tuple(ids.shape), n, up)
    hit = _flat_cache.get(key)
    if hit is not None:
        return hit[1]
    b = ids.shape[0]
    if up:
        ids = torch.where(ids > -1, ids, torch.zeros_like(ids))
    flat = (ids + torch.arange(b, device=ids.device).view(b, 1, 1) * n).reshape(-1)
    if len(_flat_cache) > 64:
        _flat_cache.clear()
    _flat_cache[key] = (ids, flat)
    return flat


UP_TABLES = ('ids43', 'ids32', 'ids21', 'ids10')          # nearest up-sampling tables (coarse level -> fine level, K = 1, -1 = none)


def table_extras(data):
    """Flat row numbers and CSR (entries sorted by target row) of every id table of a fit batch, as extra batch entries
    `tables_flat_<name>`, `tables_order_<name>`, `tables_offsets_<name>`: built with the batch -- on the loader's side stream -- instead of
    inside the backward pass (14 radix sorts per step), and ordinary input tensors of the step, so that a HIP-graph replay of the step
    contains no sort."""
    pts = data['pts']
    b = pts.shape[0]
    sizes = [pts.shape[2]] + [data['support{}'.format(a)].shape[2] for a in range(1, 5)]
    out = {}
    for name, ids in list(data.items()):
        if name == 'proj_ids':
            src = 0
        elif len(name) == 5 and name.startswith('ids') and name[3:].isdigit():
            src = int(name[3])
        else:
            continue
        if not torch.is_tensor(ids) or ids.dim() != 3:
            continue
        n = sizes[src]
        # flat row numbers (-1 -> row 0 for the up-sampling tables) and their CSR in one device call (a counting sort, pps_csr.hip): four
        # elementwise torch launches, a stable sort and a searchsorted per table until round 5
        flat, order, offsets = train_ops.csr_build_table(ids, ids.shape[1] * ids.shape[2], n, b * n, name in UP_TABLES)
        out['tables_flat_' + name], out['tables_order_' + name], out['tables_offsets_' + name] = flat, order, offsets
    return out


def register_tables(data):
    """Make the tables of table_extras() known to _flat_ids / train_ops.csr for this step (host-side dictionary entries only)."""
    for key, flat in data.items():
        if not key.startswith('tables_flat_'):
            continue
        name = key[len('tables_flat_'):]
        ids = data[name]
        order, offsets = data['tables_order_' + name], data['tables_offsets_' + name]
        rows = offsets.numel() - 1
        n = rows // ids.shape[0]
        _flat_cache[(ids.data_ptr(), ids._version, tuple(ids.shape), n, name in UP_TABLES)] = (ids, flat)
        train_ops.csr_register(flat, rows, order, offsets)


def release_step_caches():
    """Drop the id-table caches (call once per optimisation step, after backward)."""
    _flat_cache.clear()
    train_ops.clear_cache()
    _shadow_live[0] = False
    _early.clear()


# ---------------------------------------------------------------------------------------------------------------------
# independent branches of the step on side streams  [EXPERIMENT, off by default: PPS_FIT_STREAMS, see side_streams_on]
# ---------------------------------------------------------------------------------------------------------------------
# The step is a long chain of small kernels (the encoder: ~600 launches of 5-30 us on a few hundred workgroups) next to a few heavy streaming ones
# (PointNet on 10^6 patch rows).  Two branches do not depend on the encoder's features at all:
#   * PointNet (source/base/nn.py:305-373) reads only the patches;
#   * the FKAConv geometry branch of every layer (nn.py:601-643) reads only positions, id tables and its 1140 small parameters.
# Forked onto side streams they run beside the feature chain -- and so do their BACKWARD passes: autograd runs a node's backward on the stream of its
# forward and orders the streams with events, so the three geometry backward passes of a layer (which nothing downstream waits for) and PointNet's
# backward leave the critical path as well.  Recorded into a HIP graph the forks and joins become edges of the graph.
# Measured and NOT adopted (side_streams_on): the product keeps the step on one stream.
import os as _os

_side_streams = {}


def side_streams_on(t, branch='pointnet'):
    """PPS_FIT_STREAMS: 0 (default: one stream) | 1 (both branches forked) | pointnet | geometry.  The forks are an EXPERIMENT that stays switched off
    (measurements in profiles/NOTES_r5.md section 3): with the geometry kernels at their resident occupancy (5 / 3 workgroups per CU) forking them
    beside the feature chain makes the config-3 step SLOWER (21.4 ms against ~19.7 on one stream: they fill the chip and the twenty joins cost 10-25 us
    each); PointNet alone was the fastest configuration (19.3 ms) but a bf16-mixed step recorded with only that fork does not replay like the eager
    step (loss off by 0.07 after four replays; unexplained), so it is not usable."""
    mode = _os.environ.get('PPS_FIT_STREAMS', '0')
    # not under staged(): the multi-rank step records every backward stage as its own graph in ONE memory pool replayed in a fixed order
    # (fit.StagedStep); with a forked branch inside, replays did not reproduce the eager steps bit for bit
    # (tests/test_gpu_train.py::test_staged_step_overlap_structure_replay_equals_eager), so that step stays on one stream
    return t.is_cuda and torch.is_grad_enabled() and (mode == '1' or mode == branch) and _stages[0] is None


def _side(dev, name):
    key = (dev, name)
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device=dev)
    return _side_streams[key]


class Forked:
    """with Forked(dev, 'pointnet') as f: ... ops on the side stream ...;  f.join(*outputs) on the consumer's stream before the first use."""

    def __init__(self, dev, name):
        self.side = _side(dev, name)
        self.main = torch.cuda.current_stream(dev)
        self.ctx = None

    def __enter__(self):
        self.side.wait_stream(self.main)
        self.ctx = torch.cuda.stream(self.side)
        self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        self.ctx.__exit__(*exc)
        self.done = torch.cuda.Event()
        self.done.record(self.side)
        return False

    def join(self, *tensors):
        cur = torch.cuda.current_stream(self.side.device)
        cur.wait_event(self.done)
        for t in tensors:
            if torch.is_tensor(t):
                t.record_stream(cur)                 # allocated on the side stream, read here: not to be recycled under the reader
        return tensors[0] if len(tensors) == 1 else tensors


_early = {}


def start_pointnet(net, data):
    """PointNet's forward pass of this step on its side stream, BEFORE the encoder is queued (PPSurfNetwork.forward in train()): the result is
    picked up by ppsurf_from_latent.  No-op on one stream."""
    _early.clear()
    pl = data.get('pts_local_ps')
    if not (torch.is_tensor(pl) and side_streams_on(pl)) or pl.dim() != 4:
        return
    b, q = pl.shape[0], pl.shape[1]
    with Forked(pl.device, 'pointnet') as f:
        feat, _ = pointnet(net.point_net, pl.reshape(b * q, pl.shape[2], 3), need_trans=False)
    _early['pn'] = (pl, f, feat)


# ---------------------------------------------------------------------------------------------------------------------
# backward in stages (gradient all-reduce overlapped with the rest of backward, also when the step is replayed as HIP graphs)
# ---------------------------------------------------------------------------------------------------------------------
class BackwardStages:
    """The backward pass cut into consecutive pieces at tensors of the forward pass ("cuts"), so that whatever runs BETWEEN two pieces -- the
    all-reduce of the gradients the previous piece completed -- overlaps the next piece.  Lightning DDP, which the reference trains under
    (configs/device_server.yaml:2, source/base/mp.py:85-91), gets that overlap from autograd hooks; hooks do not fire in a replayed HIP graph, so
    here every piece can be recorded as its own graph and the collectives are issued between the replays (ppsurf_amd/fit.py).

    cut(*tensors) replaces each tensor by a detached leaf for everything that follows in the forward pass and remembers the pair.
    Stage 0 = loss.backward(): gradients of everything behind the last cut, and of that cut's leaves.  Stage k = torch.autograd.backward(originals
    of the k-th cut from the end, their leaves' gradients): the same chain rule, evaluated in the same order by the same kernels -- gradients are
    bit-identical to a single backward pass (tests/test_train_graph_cpu.py::test_staged_backward_*)."""

    def __init__(self):
        self.cuts = []

    def cut(self, tensors):
        leaves = [t.detach().requires_grad_(True) if t.requires_grad else t for t in tensors]
        self.cuts.append((list(tensors), leaves))
        return leaves

    @property
    def n_stages(self):
        return len(self.cuts) + 1

    def run_stage(self, k, loss=None):
        if k == 0:
            loss.backward()
            return
        orig, leaves = self.cuts[len(self.cuts) - k]
        pairs = [(o, l.grad) for o, l in zip(orig, leaves) if o.requires_grad and l.grad is not None]
        if pairs:
            torch.autograd.backward([o for o, _ in pairs], [g for _, g in pairs])

    def backward(self, loss, after_stage=None):
        try:
            for k in range(self.n_stages):
                self.run_stage(k, loss)
                if after_stage is not None:
                    after_stage(k)
        finally:
            self.release()

    def release(self):
        """Drop the cut tensors (and with them the autograd graph of the step).  A graph that outlives its step keeps the parameters' AccumulateGrad
        nodes -- and the stream they were created on -- alive; a later HIP-graph capture on another stream would then fork that stream into the
        capture (hipStreamEndCapture faults on the unjoined branch: tools/dbg notes in profiles/NOTES_r5.md)."""
        self.cuts = []


_stages = [None]


class staged:
    """with staged() as st: ... forward ...; st.backward(loss)   -- the encoder places its cuts while this is active."""

    def __enter__(self):
        self.prev = _stages[0]
        _stages[0] = BackwardStages()
        return _stages[0]

    def __exit__(self, *exc):
        _stages[0] = self.prev
        return False


def _cut(*tensors):
    st = _stages[0]
    if st is None or not torch.is_grad_enabled():
        return tensors
    return st.cut(tensors)


N_STAGES = 3
# parameters by the backward stage that completes their gradient (module names of the encoder, source/base/nn.py:455-506): the cuts of encoder()
# sit behind resnetb31 and behind resnetb11.  Stage 0 -- decoder, the encoder's up-sampling head and its two coarsest levels -- holds 82 % of the
# 13.75 M parameters and is complete when 70 % of the backward pass (the fine levels: 100 000 and 25 000 rows per batch of 10) is still to run.
_STAGE_PREFIX = {2: ('cv0.', 'bn0.', 'resnetb01.', 'resnetb10.', 'resnetb11.'), 1: ('resnetb20.', 'resnetb21.', 'resnetb30.', 'resnetb31.')}


def parameter_stages(module):
    """[[parameters of stage 0], [stage 1], [stage 2]] of a model holding the encoder as `network.encoder` (PocoModel / PPSurfModel) or
    `encoder` (a network): the encoder's parameters are found through the MODULE (identity, not a substring of their names) and sorted by the
    block they belong to; every other parameter is stage 0.  Each group in reverse registration order (roughly the order backward produces them).
    A stage without trainable parameters (frozen encoder levels) stays in the list as an EMPTY group: sharding.GradBuckets keeps it as an empty
    bucket, so the stage <-> bucket correspondence fit.StagedStep relies on survives."""
    groups = [[] for _ in range(N_STAGES)]
    enc = getattr(getattr(module, 'network', module), 'encoder', None)
    stage_of = {}
    if isinstance(enc, torch.nn.Module):
        for name, p in enc.named_parameters():
            for k, prefixes in _STAGE_PREFIX.items():
                if name.startswith(prefixes):
                    stage_of[id(p)] = k
    for _, p in module.named_parameters():
        if p.requires_grad:
            groups[stage_of.get(id(p), 0)].append(p)
    return [list(reversed(g)) for g in groups]


# ---------------------------------------------------------------------------------------------------------------------
# encoder
# ---------------------------------------------------------------------------------------------------------------------
def pack_geo(layer):
    """The layer's small parameters as ONE differentiable vector in the layout of the HIP kernels (pps_fka_common.h):
    [norm_radius, alpha, beta, activation (1 relu / 2 silu), fc1 [16,3], fc2 [16,32], fc3 [16,32], bn1 w, bn1 b, bn2 w, bn2 b]."""
    act = 2.0 if isinstance(layer.activation, torch.nn.SiLU) else 1.0
    cached = getattr(layer, '_pps_act_code', None)          # (value, device constant) made once, in the eager steps before any recording: no fill kernel per layer and step
    if cached is None or cached[0] != act or cached[1].device != layer.alpha.device or cached[1].dtype != layer.alpha.dtype:
        cached = (act, torch.full((1,), act, dtype=layer.alpha.dtype, device=layer.alpha.device))
        layer._pps_act_code = cached
    head = cached[1]
    parts = [layer.norm_radius.detach().reshape(1), layer.alpha.reshape(1), layer.beta.reshape(1), head, layer.fc1.weight.reshape(-1),
             layer.fc2.weight.reshape(-1), layer.fc3.weight.reshape(-1), layer.bn1.weight, layer.bn1.bias, layer.bn2.weight, layer.bn2.bias]
    return torch.cat([p.to(layer.alpha.dtype) for p in parts])


def fka_geometry_of(layer, pts, sup, ids):
    """The [B*M, K, 16] kernel-weighting matrix of one FKAConv layer (nn.py:601-643) incl. the norm_radius EMA of train(): positions, ids and the
    layer's small parameters only -- no features."""
    b, n = pts.shape[0], pts.shape[1]
    m, k = ids.shape[1], ids.shape[2]
    flat = _flat_ids(ids, n).view(b * m, k)
    momentum = layer.norm_radius_momentum if layer.training else 0.0
    g, radius = train_ops.fka_geometry(pack_geo(layer), pts.reshape(b * n, 3), sup.reshape(b * m, 3), flat, b, m, momentum, owned=True)
    if layer.training:
        _store_radius(layer, radius)
    return g


def fkaconv_layer(layer, x, pts, sup, ids, geo=None):
    """x [B,N,Cin], pts [B,N,3], sup [B,M,3], ids int64 [B,M,K] -> [B,M,Cout].
    Geometry branch (nn.py:601-643, incl. the norm_radius EMA in train()) and feature aggregation (:647-649) are HIP ops with
    hand-written backward; the (1,16) convolution is one GEMM over all support points of the batch.  geo: (Forked, g) of a geometry branch that
    was started on the side stream (encoder()), else it is evaluated here."""
    b, n, cin = x.shape
    m, k = ids.shape[1], ids.shape[2]
    flat = _flat_ids(ids, n).view(b * m, k)
    g = fka_geometry_of(layer, pts, sup, ids) if geo is None else geo[0].join(geo[1])
    feat = train_ops.neighbour_contract(x.reshape(b * n, cin), flat, g)                  # [B*M, Cin*16]
    wc = _bf16_of(layer.cv.weight)
    return rows_linear(feat, _w2d(layer.cv), None, None if wc is None else wc.reshape(wc.shape[0], -1), None,
                       _bf16_t_of(layer.cv.weight)).view(b, m, -1)   # Conv2d (1,16): (c,t) -> c*16+t


@_counted
def residual_block(blk, x, pts, sup, ids, geo=None):
    """[B,N,Cin] -> [B,M,Cout]."""
    b, n, cin = x.shape
    m = ids.shape[1]
    h = batch_norm(blk.bn0, dense(blk.cv0, x.reshape(b * n, cin)), relu=True).view(b, n, -1)
    h = fkaconv_layer(blk.cv1, h, pts, sup, ids, geo)
    h = batch_norm(blk.bn1, h.reshape(b * m, -1), relu=True)
    h = dense(blk.cv2, h)
    sc = x.reshape(b * n, cin)
    if not isinstance(blk.shortcut, torch.nn.Identity):
        sc = batch_norm(blk.bn_shortcut, dense(blk.shortcut, sc))
    if n != m:
        sc = train_ops.neighbour_max(sc, _flat_ids(ids, n).view(b * m, -1))
    return batch_norm_add_relu(blk.bn2, h, sc).view(b, m, -1)                     # relu(bn2(h) + shortcut): one op (nn.py:447-450)


def _upsample(x, ids_up, n_coarse):
    """nearest-neighbour interpolation (nn.py:684-697 with K == 1): x [B,Nc,C], ids_up [B,Nf,1] -> [B,Nf,C]."""
    b, nf = ids_up.shape[0], ids_up.shape[1]
    return train_ops.gather_rows(x.reshape(b * n_coarse, -1), _flat_ids(ids_up, n_coarse, up=True)).view(b, nf, -1)


@_counted
def encoder(enc, data):
    """FKAConvNetwork.forward(data, spectral_only=True): data['pts'] [B,3,N] + supports / id tables -> latents [B,N,C]."""
    pm = lambda t: t.transpose(1, 2).contiguous()
    pts = pm(data['pts'])
    s1, s2, s3, s4 = (pm(data['support{}'.format(i)]) for i in (1, 2, 3, 4))
    b = pts.shape[0]
    x = torch.ones_like(pts)                                                             # input features are all-ones (:517)
    # the geometry branches of all ten layers, one after the other on a side stream (they need no features); each layer joins its own
    levels = {'cv0': (enc.cv0, pts, pts, 'ids00'), 'b01': (enc.resnetb01.cv1, pts, pts, 'ids00'), 'b10': (enc.resnetb10.cv1, pts, s1, 'ids01'),
              'b11': (enc.resnetb11.cv1, s1, s1, 'ids11'), 'b20': (enc.resnetb20.cv1, s1, s2, 'ids12'), 'b21': (enc.resnetb21.cv1, s2, s2, 'ids22'),
              'b30': (enc.resnetb30.cv1, s2, s3, 'ids23'), 'b31': (enc.resnetb31.cv1, s3, s3, 'ids33'), 'b40': (enc.resnetb40.cv1, s3, s4, 'ids34'),
              'b41': (enc.resnetb41.cv1, s4, s4, 'ids44')}
    geo = dict.fromkeys(levels)
    if side_streams_on(pts, 'geometry'):
        for name, (layer, p_in, p_out, table) in levels.items():
            with Forked(pts.device, 'geometry') as f:
                g = fka_geometry_of(layer, p_in, p_out, data[table])
            geo[name] = (f, g)
    x0 = fkaconv_layer(enc.cv0, x, pts, pts, data['ids00'], geo['cv0'])
    x0 = batch_norm(enc.bn0, x0.reshape(b * pts.shape[1], -1), relu=True).view(b, pts.shape[1], -1)
    x0 = residual_block(enc.resnetb01, x0, pts, pts, data['ids00'], geo['b01'])
    x1 = residual_block(enc.resnetb10, x0, pts, s1, data['ids01'], geo['b10'])
    x1 = residual_block(enc.resnetb11, x1, s1, s1, data['ids11'], geo['b11'])
    x0, x1 = _cut(x0, x1)                                     # backward stage 2 = everything above (BackwardStages; no-op unless staged() is active)
    x2 = residual_block(enc.resnetb20, x1, s1, s2, data['ids12'], geo['b20'])
    x2 = residual_block(enc.resnetb21, x2, s2, s2, data['ids22'], geo['b21'])
    x3 = residual_block(enc.resnetb30, x2, s2, s3, data['ids23'], geo['b30'])
    x3 = residual_block(enc.resnetb31, x3, s3, s3, data['ids33'], geo['b31'])
    x0, x1, x2, x3 = _cut(x0, x1, x2, x3)                     # backward stage 1 = the two levels above; stage 0 = everything below
    x4 = residual_block(enc.resnetb40, x3, s3, s4, data['ids34'], geo['b40'])
    x4 = residual_block(enc.resnetb41, x4, s4, s4, data['ids44'], geo['b41'])

    def head(cv, bn, coarse, ids_up, skip):
        z = torch.cat([_upsample(coarse, ids_up, coarse.shape[1]), skip], dim=-1)
        return batch_norm(bn, dense(cv, z.reshape(-1, z.shape[-1])), relu=True).view(b, skip.shape[1], -1)

    x4d = x4
    if enc.fixed or enc.training:                                                        # :531-534
        # without x4d_bug_fixed (POCO) the reference still evaluates cv5/bn5 and throws the result away: in train() that
        # updates bn5's running statistics, which end up in the checkpoint
        x5 = x4.max(dim=1, keepdim=True)[0].expand_as(x4)
        z = torch.cat([x4, x5], dim=-1)
        z = batch_norm(enc.bn5, dense(enc.cv5, z.reshape(-1, z.shape[-1])), relu=True).view(b, x4.shape[1], -1)
        if enc.fixed:
            x4d = z
    x3d = head(enc.cv3d, enc.bn3d, x4d, data['ids43'], x3)
    x2d = head(enc.cv2d, enc.bn2d, x3d, data['ids32'], x2)
    x1d = head(enc.cv1d, enc.bn1d, x2d, data['ids21'], x1)
    xo = head(enc.cv0d, enc.bn0d, x1d, data['ids10'], x0)
    xo = enc.dropout(xo)
    return dense(enc.fcout, xo.reshape(-1, xo.shape[-1])).view(b, pts.shape[1], -1)


# The interpolation head's three layers as ONE kernel (csrc/pps_head_chain_impl.h): 0.36 ms off the config-3 step (0.60 against 0.97 ms for the
# separate launches), same stored tensors (h1 bit-identical to pps_head_input_fwd, the layers to the tolerance of tests/test_gpu_head_chain.py) and
# run-to-run identical.  PPS_HEAD_CHAIN=0: one launch per layer (the form it is compared with).
HEAD_CHAIN = _os.environ.get('PPS_HEAD_CHAIN', '1') != '0'
FUSED_ROWS = _os.environ.get('PPS_FUSED_ROWS', '1') != '0'            # False: every row layer through the separate ops (library GEMM + fused BatchNorm op), e.g. to compare


def fused_rows_ok(x, *bns):
    """The fused row layers (train_ops.rows_layer: 16-bit storage, BatchNorm statistics of train()) apply: device tensor, bf16 / fp16 autocast, every
    BatchNorm in train() with a momentum."""
    return (x.is_cuda and torch.is_autocast_enabled('cuda') and torch.get_autocast_dtype('cuda') in train_ops.LOW
            and all(b.training and b.track_running_stats and b.momentum is not None for b in bns))


# ---------------------------------------------------------------------------------------------------------------------
# decoder
# ---------------------------------------------------------------------------------------------------------------------
@_counted
def interp_attention(proj, latents, pts, query, ids, last_layer=True):
    """latents [B,N,C], pts [B,N,3], query [B,Q,3], ids [B,Q,k] -> [B,Q,Cout] (poco_model.py:400-417)."""
    b, n, c = latents.shape
    q, k = ids.shape[1], ids.shape[2]
    flat = _flat_ids(ids, n)
    # fc1 is linear in [latent ; q - p]: its latent part is evaluated once per POINT (B*N rows instead of B*Q*k) and gathered --
    # the per-point table G of the inference kernels (identity 1); autograd differentiates this form
    w1 = _w2d(proj.fc1)
    w1c = _bf16_of(proj.fc1.weight)
    w1c = None if w1c is None else w1c.reshape(w1c.shape[0], -1)
    table = rows_linear(latents.reshape(b * n, c), w1[:, :c], proj.fc1.bias, None if w1c is None else w1c[:, :c].contiguous(),
                        _bf16_of(proj.fc1.bias))                                         # [B*N, C]
    if (FUSED_ROWS and fused_rows_ok(latents) and train_ops.attn_pool_supported(k, _w2d(proj.fc_query).shape[0], 256)
            and all(train_ops.rows_layer_supported(b * q * k, *reversed(_w2d(l).shape)) for l in (proj.fc2, proj.fc3, proj.fc_query))
            and train_ops.head_input_supported(w1.shape[0])):
        # fc2, fc3, fc_query as fused row layers (pps_rows_train.hip): each stores its RAW output once, the ReLU is applied by the consumer on
        # load (and masks the gradient on the way back), so no activated [B*Q*k, 256] tensor is written or read
        if (HEAD_CHAIN and train_ops.head_chain_supported(w1.shape[0], _w2d(proj.fc_query).shape[0], k)
                and train_ops.head_chain_trusted(table.dtype if table.dtype in train_ops.LOW else torch.bfloat16)):
            # the three layers in ONE kernel: a wave carries its rows through them in registers (csrc/pps_head_chain_impl.h)
            pooled = train_ops.head_chain(table, flat, pts.reshape(b * n, 3), query.reshape(b * q, 3), k, w1[:, c:], (_w2d(proj.fc2), proj.fc2.bias),
                                          (_w2d(proj.fc3), proj.fc3.bias), (_w2d(proj.fc_query), proj.fc_query.bias))
        else:
            h1 = train_ops.head_input(table, flat, pts.reshape(b * n, 3), query.reshape(b * q, 3), k, w1[:, c:])
            y2 = train_ops.rows_layer(train_ops.Act(h1, None, True), _w2d(proj.fc2), proj.fc2.bias, None, True)
            y3 = train_ops.rows_layer(y2, _w2d(proj.fc3), proj.fc3.bias, None, True)
            pooled = train_ops.query_attn_pool(y3.raw, _w2d(proj.fc_query), proj.fc_query.bias, k)      # fc_query + attention pooling: one node
        out = dense(proj.fc_value, pooled)
        if last_layer:
            out = dense(proj.fc8, out)
        return out.view(b, q, -1)
    rel = (query.unsqueeze(2) - pts.reshape(b * n, 3)[flat].view(b, q, k, 3)).reshape(-1, 3)     # query minus neighbour
    h = F.relu(train_ops.gather_rows(table, flat) + rows_linear(rel, w1[:, c:]).to(table.dtype))
    h = F.relu(dense(proj.fc2, h))
    h = F.relu(dense(proj.fc3, h))
    # sum_j a_j (W_v h_j + b_v) = W_v (sum_j a_j h_j) + b_v because the weights of a query sum to 1 (:412-414): fc_value runs on
    # Q rows instead of Q*k (same identity as the inference kernel); autograd differentiates the pooled form
    qy = dense(proj.fc_query, h).view(b * q, k, -1)
    if h.is_cuda and train_ops.attn_pool_supported(k, qy.shape[2], h.shape[1]):
        pooled = train_ops.attn_pool(qy, h.view(b * q, k, -1))          # softmax over neighbours, mean of heads, pooling: one HIP op
    else:
        att = torch.softmax(qy, dim=1).mean(dim=2)
        pooled = torch.bmm(att.unsqueeze(1).to(h.dtype), h.view(b * q, k, -1)).squeeze(1)
    out = dense(proj.fc_value, pooled)
    if last_layer:
        out = dense(proj.fc8, out)
    return out.view(b, q, -1)


@_counted
def stn(t, h, nq, p):
    """h [nq*p, dim] -> [nq, dim, dim]."""
    d = t.dim
    z = batch_norm(t.bn1, dense(t.conv1, h), relu=True)
    z = batch_norm(t.bn2, dense(t.conv2, z), relu=True)
    z = batch_norm(t.bn3, dense(t.conv3, z), relu=True)
    z = z.view(nq, p, -1).max(dim=1)[0]
    z = batch_norm(t.bn4, dense(t.fc1, z), relu=True)
    z = batch_norm(t.bn5, dense(t.fc2, z), relu=True)
    z = dense(t.fc3, z) + torch.eye(d, dtype=z.dtype, device=z.device).reshape(1, d * d)
    return z.view(nq, d, d)


def _layer(act, conv, bn, relu):
    """conv (1x1 / Linear holder) -> bn (train() statistics) on a stored activation, as ONE op: train_ops.Act in, Act out."""
    if bn is not None:
        _count_batch(bn)
    return train_ops.rows_layer(act, _w2d(conv), conv.bias, bn, relu)


def _stn_fused(t, act, nq, p):
    """stn() on a stored activation: the three row layers and the max over the patch never write an activated [nq*p, C] tensor."""
    d = t.dim
    z = _layer(act, t.conv1, t.bn1, True)
    z = _layer(z, t.conv2, t.bn2, True)
    w3 = _w2d(t.conv3)
    if train_ops.rows_layer_max_supported(z.raw.shape[0], w3.shape[1], w3.shape[0], nq, p):
        _count_batch(t.bn3)
        z = train_ops.rows_layer_max(z, w3, t.conv3.bias, t.bn3, True, nq, p)      # conv3 + bn3 + relu + max over the patch: the gradient goes back per patch
    else:
        z = _layer(z, t.conv3, t.bn3, True)
        z = train_ops.act_max(z, nq, p)
    z = batch_norm(t.bn4, dense(t.fc1, z), relu=True)
    z = batch_norm(t.bn5, dense(t.fc2, z), relu=True)
    return dense(t.fc3, z).view(nq, d, d)                  # WITHOUT the identity (:188): the feature transform adds it on load


def _pointnet_fused(pn, patches, need_trans=True):
    """pointnet() with the 64..256-channel row layers as fused ops (pps_rows_train.hip): per layer the raw output is written once and read
    by its consumers, which apply BatchNorm + ReLU on load."""
    nq, p, _ = patches.shape
    _count_batch(pn.bn0a)
    l0a = train_ops.rows3_layer(patches.reshape(nq * p, 3), _w2d(pn.conv0a), pn.conv0a.bias, pn.bn0a, True)   # 3 input channels: not an MFMA shape
    l0b = _layer(l0a, pn.conv0b, pn.bn0b, True)
    t_raw = _stn_fused(pn.stn2, l0b, nq, p)
    h = train_ops.patch_transform(l0b, t_raw, p)                     # (t_raw + I) applied to every patch point: trans2 @ x (:330-331)
    d = pn.stn2.dim
    trans2 = t_raw.detach().float() + torch.eye(d, device=h.device).view(1, d, d) if need_trans else None
    z = _layer(train_ops.Act(h), pn.conv1, pn.bn1, True)
    z = _layer(z, pn.conv2, pn.bn2, True)
    # AttentionPoco (nn.py:84-96) on h = raw * scale + shift (no ReLU after bn3): the logit is linear in raw, and the pooled row of h is the
    # affine image of the pooled raw row because the weights of a patch sum to 1
    wq = _w2d(pn.att.fc_query).float()
    w3 = _w2d(pn.conv3)
    if (_os.environ.get('PPS_PATCH_ATTN_GRAD', 'rebuilt') != 'stored'
            and train_ops.rows_layer_patch_attn_supported(w3.shape[1], w3.shape[0], p, nq * p)):
        # conv3 / bn3 and the pooling as one node: the gradient of conv3's raw output has rank two per patch and is rebuilt where it is read
        _count_batch(pn.bn3)
        pooled, aff3 = train_ops.rows_layer_patch_attn(z, w3, pn.conv3.bias, pn.bn3, wq, nq, p)
        scale, shift = aff3[0], aff3[1]
    else:
        z = _layer(z, pn.conv3, pn.bn3, False)
        scale, shift = z.affine[0], z.affine[1]
        pooled = train_ops.patch_attn(z.raw.view(nq, p, -1), (wq * scale).reshape(-1))
    # (softmax ignores the constant w_q . shift + b_q; it stays in the graph with weight 0 so that fc_query.bias gets its zero gradient, not None)
    const = (wq * shift).sum() + pn.att.fc_query.bias.float().sum()
    pooled = (train_ops.affine_rows(pooled, scale, shift) + 0.0 * const).to(z.raw.dtype)      # (the row sums of its backward by the HIP reduction)
    return dense(pn.att.fc_value, pooled), trans2


@_counted
def pointnet(pn, patches, need_trans=True):
    """patches [Q', P, 3] -> (feat [Q', C], trans2 [Q', 64, 64] or None if not need_trans)."""
    nq, p, _ = patches.shape
    t = pn.stn2
    if (FUSED_ROWS and fused_rows_ok(patches, pn.bn0a, pn.bn0b, pn.bn1, pn.bn2, pn.bn3, t.bn1, t.bn2, t.bn3)
            and train_ops.patch_attn_supported(p, _w2d(pn.conv3).shape[0]) and train_ops.patch_transform_supported(p, t.dim)
            and tuple(_w2d(pn.conv0a).shape) == (64, 3)):
        return _pointnet_fused(pn, patches, need_trans)
    h = batch_norm(pn.bn0a, dense(pn.conv0a, patches.reshape(nq * p, 3)), relu=True)
    h = batch_norm(pn.bn0b, dense(pn.conv0b, h), relu=True)
    trans2 = stn(pn.stn2, h, nq, p)
    h = torch.bmm(h.view(nq, p, -1), trans2.transpose(1, 2)).reshape(nq * p, -1)         # channel-first: trans2 @ x
    h = batch_norm(pn.bn1, dense(pn.conv1, h), relu=True)
    h = batch_norm(pn.bn2, dense(pn.conv2, h), relu=True)
    h = batch_norm(pn.bn3, dense(pn.conv3, h))
    logit = dense(pn.att.fc_query, h).view(nq, p, 1)
    if h.is_cuda and train_ops.attn_pool_supported(p, 1, h.shape[1]):
        pooled = train_ops.attn_pool(logit, h.view(nq, p, -1))                           # softmax over the patch + pooling: one HIP op (P <= 64)
    else:
        w = torch.softmax(logit.view(nq, p), dim=1)
        pooled = torch.bmm(w.unsqueeze(1).to(h.dtype), h.view(nq, p, -1)).squeeze(1)      # pool first: the weights sum to 1
    return dense(pn.att.fc_value, pooled), trans2


@_counted
def mlp(m, x):
    for i, block in enumerate(m.layers):
        x = dense(block[0], x)
        if i < len(m.layers) - 1:
            x = block[3](batch_norm(block[1], x, relu=True))                                # Dropout holder, active in train()
    return x


def _point_major(t):
    """[B,3,N] or [B,N,3] -> [B,N,3]."""
    return t.transpose(1, 2) if t.shape[1] == 3 and t.shape[2] != 3 else t


@_counted
def ppsurf_from_latent(net, latents, data, proj_ids):
    """latents [B,N,C] point-major; data{pts, pts_query, pts_local_ps [B,Q,P,3]}; proj_ids [B,Q,k] -> logits [B,2,Q]."""
    pts = _point_major(data['pts']).contiguous()
    query = _point_major(data['pts_query']).contiguous()
    b, q = query.shape[0], query.shape[1]
    feat_proj = interp_attention(net.projection, latents, pts, query, proj_ids)
    pl = data['pts_local_ps']
    early = _early.pop('pn', None)
    if early is not None and early[0] is pl:
        feat_pn = early[1].join(early[2])                    # started before the encoder on its side stream (start_pointnet)
    else:
        feat_pn, _ = pointnet(net.point_net, pl.reshape(b * q, pl.shape[2], 3), need_trans=False)
    out = mlp(net.mlp, feat_proj.reshape(b * q, -1) + feat_pn)
    return out.view(b, q, -1).transpose(1, 2)


def _prepare(net, data):
    x = data['pts']
    if x.is_cuda and net.training:
        register_tables(data)
        if torch.is_autocast_enabled('cuda') and torch.get_autocast_dtype('cuda') in train_ops.LOW:
            prepare_shadows(net, torch.get_autocast_dtype('cuda'))


@_counted
def ppsurf_forward(net, data, proj_ids):
    _prepare(net, data)
    start_pointnet(net, data)
    return ppsurf_from_latent(net, encoder(net.encoder, data), data, proj_ids)


@_counted
def poco_forward(net, data, proj_ids):
    _prepare(net, data)
    pts = _point_major(data['pts']).contiguous()
    query = _point_major(data['pts_query']).contiguous()
    return interp_attention(net.projection, encoder(net.encoder, data), pts, query, proj_ids).transpose(1, 2)
