"""Thin torch-facing wrappers of the spatial-query entry points of the C ABI (device tensors in, device tensors out)."""
import torch

from . import _lib


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def knn_point_major(pts: torch.Tensor, query: torch.Tensor, k: int, return_d2: bool = False, out: torch.Tensor = None):
    """pts [n,3], query [m,3] float32 contiguous on the GPU -> int64 [m,k] (sorted by (d2, index)); k <= min(n, 64).
    `out`: optional preallocated int64 [m,k] (stream-pipelined callers)."""
    if not (pts.is_cuda and query.is_cuda):
        raise _lib.PpsError('pps_knn_f32 needs device tensors; there is no CPU fallback')
    pts = pts.contiguous().float()
    query = query.contiguous().float()
    m = query.shape[0]
    idx = out if out is not None else torch.empty((m, k), dtype=torch.int64, device=pts.device)
    assert idx.dtype == torch.int64 and idx.is_contiguous() and tuple(idx.shape) == (m, k)
    d2 = torch.empty((m, k), dtype=torch.float32, device=pts.device) if return_d2 else None
    _lib.check(_lib.lib().pps_knn_f32(pts.data_ptr(), pts.shape[0], query.data_ptr(), m, int(k), idx.data_ptr(),
                                      d2.data_ptr() if return_d2 else None, _stream(pts)), 'pps_knn_f32')
    return (idx, d2) if return_d2 else idx


def knn_batch_point_major(pts_list, query_list, k_list):
    """Up to 64 independent searches (k <= 64 each) per launch through pps_knn_multi_f32: the shapes of a fit batch, or the
    tables of one encoder pass.  pts_list[i] [n_i,3], query_list[i] [m_i,3] contiguous float32 on the GPU -> list of int64 [m_i,k_i]."""
    import ctypes
    outs = []
    for s in range(0, len(pts_list), 64):
        ps, qs, ks = pts_list[s:s + 64], query_list[s:s + 64], k_list[s:s + 64]
        nt = len(ps)
        P, I64, I = ctypes.c_void_p * nt, ctypes.c_int64 * nt, ctypes.c_int * nt
        pts, qry, out, ns, ms, kk = P(), P(), P(), I64(), I64(), I()
        for t in range(nt):
            if not (ps[t].is_cuda and qs[t].is_cuda):
                raise _lib.PpsError('pps_knn_multi_f32 needs device tensors; there is no CPU fallback')
            o = torch.empty((qs[t].shape[0], ks[t]), dtype=torch.int64, device=ps[t].device)
            outs.append(o)
            pts[t], qry[t], out[t], ns[t], ms[t], kk[t] = ps[t].data_ptr(), qs[t].data_ptr(), o.data_ptr(), ps[t].shape[0], qs[t].shape[0], ks[t]
        _lib.check(_lib.lib().pps_knn_multi_f32(nt, pts, ns, qry, ms, kk, out, _stream(ps[0])), 'pps_knn_multi_f32')
    return outs


def _morton_spread(v):                                   # 10 bits -> every third bit
    v = (v | (v << 16)) & 0x030000FF
    v = (v | (v << 8)) & 0x0300F00F
    v = (v | (v << 4)) & 0x030C30C3
    return (v | (v << 2)) & 0x09249249


class BlockedLevel:
    """One level of a BATCH of equally sized clouds [B,n,3] arranged for pps_knn_blocked_batch_f32 with batched torch ops (one sort for
    all clouds): per cloud Morton order, blocks of 64 points (the tail repeats the last point, orig = -1), boxes, boxes of the full blocks."""

    def __init__(self, pts_b: torch.Tensor):
        b, n = pts_b.shape[0], pts_b.shape[1]
        lo = pts_b.amin(dim=1, keepdim=True)
        span = (pts_b.amax(dim=1, keepdim=True) - lo).amax(dim=2, keepdim=True).clamp_min(1e-20)
        cell = ((pts_b - lo) / span * 1023.0).to(torch.int64).clamp_(0, 1023)
        code = _morton_spread(cell[..., 0]) | (_morton_spread(cell[..., 1]) << 1) | (_morton_spread(cell[..., 2]) << 2)
        order = torch.sort(code, dim=1, stable=True)[1]
        self.b, self.n, self.nb = b, n, (n + 63) // 64
        pad = self.nb * 64 - n
        sp = torch.gather(pts_b, 1, order.unsqueeze(-1).expand(b, n, 3))
        self.pts = torch.cat([sp, sp[:, -1:].expand(b, pad, 3)], dim=1).contiguous()
        self.orig = torch.cat([order.to(torch.int32), torch.full((b, pad), -1, dtype=torch.int32, device=pts_b.device)], dim=1).contiguous()
        blk = self.pts.view(b, self.nb, 64, 3)
        self.bbox = torch.cat([blk.amin(dim=2), blk.amax(dim=2)], dim=2).contiguous()
        self.n_win = n // 64
        self.win = self.bbox[:, :self.n_win].contiguous() if self.n_win > 0 else None


def knn_blocked_batch(kinds):
    """kinds: list of (points BlockedLevel, queries BlockedLevel | tensor [B,m,3], k) -> list of int64 [B,m,k] (per-cloud original indices),
    one launch of pps_knn_blocked_batch_f32 per 8 kinds."""
    import ctypes
    outs = []
    for s in range(0, len(kinds), 8):
        part = kinds[s:s + 8]
        nt = len(part)
        P, I64, I = ctypes.c_void_p * nt, ctypes.c_int64 * nt, ctypes.c_int * nt
        pts, orig, bbox, win, qry, qorig, out = P(), P(), P(), P(), P(), P(), P()
        nb, nwin, qstride, ms, ks = I64(), I64(), I64(), I64(), I()
        keep = []
        for t, (pl, ql, k) in enumerate(part):
            if isinstance(ql, BlockedLevel):
                q_t, qo_t, m, stride = ql.pts, ql.orig, ql.n, ql.nb * 64 * 3
            else:
                q_t = ql.contiguous().float()
                qo_t, m, stride = None, q_t.shape[1], q_t.shape[1] * 3
            k = min(int(k), pl.n)
            o = torch.empty((pl.b, m, k), dtype=torch.int64, device=pl.pts.device)
            keep.append(q_t)
            outs.append(o)
            pts[t], orig[t], bbox[t], win[t] = pl.pts.data_ptr(), pl.orig.data_ptr(), pl.bbox.data_ptr(), pl.win.data_ptr() if pl.win is not None else None
            qry[t], qorig[t], out[t] = q_t.data_ptr(), qo_t.data_ptr() if qo_t is not None else None, o.data_ptr()
            nb[t], nwin[t], qstride[t], ms[t], ks[t] = pl.nb, pl.n_win, stride, m, k
        _lib.check(_lib.lib().pps_knn_blocked_batch_f32(nt, part[0][0].b, pts, orig, bbox, nb, win, nwin, qry, qorig, qstride, ms, ks, out,
                                                        _stream(part[0][0].pts)), 'pps_knn_blocked_batch_f32')
    return outs


def patch_normalize(raw: torch.Tensor, query: torch.Tensor, idx: torch.Tensor, p: int, out: torch.Tensor = None) -> torch.Tensor:
    """raw [n,3], query [q,3], idx int64 [q,>=p] -> patches [q,p,3] in patch space (ppsurf_data_loader.py:91-123)."""
    raw = raw.contiguous().float()
    query = query.contiguous().float()
    assert idx.dtype == torch.int64 and idx.stride(1) == 1
    q = query.shape[0]
    if out is None:
        out = torch.empty((q, p, 3), dtype=torch.float32, device=raw.device)
    assert out.is_contiguous() and tuple(out.shape) == (q, p, 3)
    _lib.check(_lib.lib().pps_patch_normalize_f32(raw.data_ptr(), query.data_ptr(), idx.data_ptr(), idx.stride(0), q, int(p),
                                                  out.data_ptr(), _stream(raw)), 'pps_patch_normalize_f32')
    return out


class KnnBlocks:
    """A cloud arranged for pps_knn_blocked_f32: points sorted along a Morton curve, cut into blocks of 64 with bounding
    boxes.  Built once per cloud with a handful of torch device ops; queries return the ORIGINAL indices, bit-identical
    to knn_point_major."""

    def __init__(self, pts: torch.Tensor):
        if not pts.is_cuda:
            raise _lib.PpsError('KnnBlocks needs a device tensor; there is no CPU fallback')
        pts = pts.contiguous().float()
        n = pts.shape[0]
        lo = pts.min(dim=0)[0]
        span = (pts.max(dim=0)[0] - lo).max().clamp_min(1e-20)
        cell = ((pts - lo) / span * 1023.0).to(torch.int64).clamp_(0, 1023)

        def spread(v):                                   # 10 bits -> every third bit
            v = (v | (v << 16)) & 0x030000FF
            v = (v | (v << 8)) & 0x0300F00F
            v = (v | (v << 4)) & 0x030C30C3
            return (v | (v << 2)) & 0x09249249
        code = spread(cell[:, 0]) | (spread(cell[:, 1]) << 1) | (spread(cell[:, 2]) << 2)
        order = torch.sort(code, stable=True)[1]
        nb = (n + 63) // 64
        pad = nb * 64 - n
        sorted_pts = pts[order]
        self.n, self.nb = n, nb
        self.pts = torch.cat([sorted_pts, sorted_pts[-1:].expand(pad, 3)], dim=0).contiguous()
        self.orig = torch.cat([order.to(torch.int32), torch.full((pad,), -1, dtype=torch.int32, device=pts.device)]).contiguous()
        blk = self.pts.view(nb, 64, 3)                   # padding repeats the last valid point: boxes stay tight
        self.bbox = torch.cat([blk.min(dim=1)[0], blk.max(dim=1)[0]], dim=1).contiguous()
        # second level: boxes of 64 consecutive blocks (the last group padded with its own last box)
        ng = (nb + 63) // 64
        bb = torch.cat([self.bbox, self.bbox[-1:].expand(ng * 64 - nb, 6)], dim=0).view(ng, 64, 6)
        self.gbox = torch.cat([bb[:, :, :3].min(dim=1)[0], bb[:, :, 3:].max(dim=1)[0]], dim=1).contiguous()
        self._win = {}
        self.groups = True                                # tests switch the second level off to compare

    def _windows(self, r: int):
        """Boxes of r consecutive FULL blocks (>= 64*r points each): upper bounds of the k-th distance for k <= 64*r."""
        if r not in self._win:
            nfull = self.n // 64
            if nfull < r:
                self._win[r] = None
            else:
                lo, hi = self.bbox[:nfull, :3], self.bbox[:nfull, 3:]
                wl = lo.unfold(0, r, 1).min(dim=2)[0]
                wh = hi.unfold(0, r, 1).max(dim=2)[0]
                self._win[r] = torch.cat([wl, wh], dim=1).contiguous()
        return self._win[r]

    def query(self, query: torch.Tensor, k: int, return_d2: bool = False, out: torch.Tensor = None):
        query = query.contiguous().float()
        m = query.shape[0]
        idx = out if out is not None else torch.empty((m, k), dtype=torch.int64, device=query.device)
        assert idx.dtype == torch.int64 and idx.is_contiguous() and tuple(idx.shape) == (m, k)
        d2 = torch.empty((m, k), dtype=torch.float32, device=query.device) if return_d2 else None
        win = self._windows((int(k) + 63) // 64) if 1 <= int(k) <= 256 else None
        _lib.check(_lib.lib().pps_knn_blocked_groups_f32(self.pts.data_ptr(), self.orig.data_ptr(), self.bbox.data_ptr(), self.nb, self.n,
                                                         win.data_ptr() if win is not None else None, win.shape[0] if win is not None else 0,
                                                         self.gbox.data_ptr() if self.groups else None, query.data_ptr(), m, int(k), idx.data_ptr(),
                                                         d2.data_ptr() if return_d2 else None, _stream(query)), 'pps_knn_blocked_groups_f32')
        return (idx, d2) if return_d2 else idx


# ---- region-growing driver on byte masks (csrc/pps_grow.hip) ---------------------------------------------------------------------------------
def dilate_box(mask: torch.Tensor, r: int) -> torch.Tensor:
    """Binary dilation of a bool volume [nx,ny,nz] on the device with the box [-r, r]^3, clipped at the border (source/poco_utils.py:181-196)."""
    assert mask.dtype == torch.bool and mask.dim() == 3 and mask.is_cuda
    src = mask.contiguous()
    dst, tmp = torch.empty_like(src), torch.empty_like(src)
    _lib.check(_lib.lib().pps_dilate_box_u8(src.data_ptr(), dst.data_ptr(), tmp.data_ptr(), src.shape[0], src.shape[1], src.shape[2], int(r), _stream(src)),
               'pps_dilate_box_u8')
    return dst


def grow_frontier(volume: torch.Tensor, neg: torch.Tensor, pos: torch.Tensor, to_see: torch.Tensor) -> torch.Tensor:
    """to_see & ((neg & volume >= 0) | (pos & volume <= 0)) in one pass (source/poco_utils.py:245-246); volume float64, masks bool, same shape."""
    assert volume.dtype == torch.float64 and volume.is_cuda and volume.is_contiguous()
    for m in (neg, pos, to_see):
        assert m.dtype == torch.bool and m.shape == volume.shape and m.is_contiguous()
    out = torch.empty_like(to_see)
    _lib.check(_lib.lib().pps_grow_frontier_f64(volume.data_ptr(), neg.data_ptr(), pos.data_ptr(), to_see.data_ptr(), out.data_ptr(), volume.numel(),
                                                _stream(volume)), 'pps_grow_frontier_f64')
    return out


def grow_band_todo(volume: torch.Tensor, band: torch.Tensor) -> torch.Tensor:
    """band & isnan(volume) in one pass."""
    assert volume.dtype == torch.float64 and volume.is_cuda and volume.is_contiguous() and band.dtype == torch.bool and band.is_contiguous()
    out = torch.empty_like(band)
    _lib.check(_lib.lib().pps_grow_band_todo_f64(volume.data_ptr(), band.data_ptr(), out.data_ptr(), volume.numel(), _stream(volume)), 'pps_grow_band_todo_f64')
    return out


def marching_cubes(volume: torch.Tensor, level: float):
    """Iso-surface of a float64 device volume [nx,ny,nz] (NaN = unseen) -> (verts float64 [V,3] in index space, faces int64 [F,3]): the four
    streaming passes of csrc/pps_mc.hip with two block-level prefix sums in between (replaces skimage.measure.marching_cubes,
    source/poco_utils.py:95-96).  One host synchronisation (the three totals size the outputs)."""
    import ctypes
    from . import mcubes
    assert volume.dtype == torch.float64 and volume.dim() == 3 and volume.is_cuda and volume.is_contiguous()
    L = _lib.lib()
    nx, ny, nz = volume.shape
    dev = volume.device
    tri, ntri, amb, tun_index, tun_cand = mcubes.device_tables(dev)
    nbc, nbe = L.pps_mc_cube_blocks(nx, ny, nz), L.pps_mc_edge_blocks(nx, ny, nz)
    if nbc < 0:
        raise ValueError('volume of shape {} is outside the range of pps_mc'.format(tuple(volume.shape)))
    nedge = 3 * nx * ny * nz
    flags = torch.empty(nedge, dtype=torch.uint8, device=dev)
    counts = torch.empty(2 * nbc + nbe, dtype=torch.int32, device=dev)
    bt, bc, bv = counts[:nbc], counts[nbc:2 * nbc], counts[2 * nbc:]
    st = _stream(volume)
    lvl = ctypes.c_double(float(level))
    _lib.check(L.pps_mc_count_f64(volume.data_ptr(), nx, ny, nz, lvl, tri.data_ptr(), mcubes.TABLE_WIDTH, ntri.data_ptr(), amb.data_ptr(),
                                  tun_index.data_ptr(), tun_cand.data_ptr(), flags.data_ptr(), bt.data_ptr(), bc.data_ptr(), bv.data_ptr(), st), 'pps_mc_count_f64')
    inc = [torch.cumsum(x, 0, dtype=torch.int64) for x in (bt, bc, bv)]
    n_tri, n_cen, n_vert = [int(v) for v in torch.stack([i[-1] for i in inc]).tolist()]
    verts = torch.empty((n_vert + n_cen, 3), dtype=torch.float64, device=dev)
    faces = torch.empty((n_tri, 3), dtype=torch.int64, device=dev)
    if n_tri == 0:
        return verts[:0], faces
    off = [(i - x).contiguous() for i, x in zip(inc, (bt, bc, bv))]
    vidx = torch.empty(nedge, dtype=torch.int32, device=dev)
    _lib.check(L.pps_mc_emit_f64(volume.data_ptr(), nx, ny, nz, lvl, tri.data_ptr(), mcubes.TABLE_WIDTH, ntri.data_ptr(), amb.data_ptr(), tun_index.data_ptr(),
                                 tun_cand.data_ptr(), flags.data_ptr(), off[0].data_ptr(), off[1].data_ptr(), off[2].data_ptr(), n_vert, vidx.data_ptr(), verts.data_ptr(), faces.data_ptr(), st),
               'pps_mc_emit_f64')
    return verts, faces


def mesh_small_components(faces: torch.Tensor, nv: int, k: int):
    """bool [F]: faces of face-connected components with at most k faces (csrc/pps_mesh.hip: one hash-table pass over the edges + one bounded walk
    per face; replaces trimesh's split + filter, source/base/mesh.py:22-38).  faces int64 [F,3] on the device."""
    assert faces.is_cuda and faces.dtype == torch.int64 and faces.dim() == 2 and faces.shape[1] == 3
    L = _lib.lib()
    nf = faces.shape[0]
    small = torch.empty(nf, dtype=torch.uint8, device=faces.device)
    if nf == 0:
        return small.bool()
    faces = faces.contiguous()
    ws = torch.empty(L.pps_mesh_components_ws_bytes(nf), dtype=torch.uint8, device=faces.device)
    _lib.check(L.pps_mesh_small_components(faces.data_ptr(), nf, int(nv), int(k), small.data_ptr(), ws.data_ptr(), _stream(faces)), 'pps_mesh_small_components')
    return small.bool()


def mesh_corner_weld(verts: torch.Tensor, digits: int = 8):
    """(remap int64 [V], hot bool [V], merged: int) of a mesh welded by grid-edge key, vertices float64 [V,3] in index space: the vertices that share
    a rounded position on a grid corner are merged into the smallest id (trimesh merge_vertices, source/base/mesh.py:9, restricted to where it can
    act).  One host read (the count decides whether the faces have to be touched at all)."""
    assert verts.is_cuda and verts.dtype == torch.float64 and verts.dim() == 2 and verts.shape[1] == 3
    L = _lib.lib()
    nv = verts.shape[0]
    dev = verts.device
    verts = verts.contiguous()
    remap = torch.empty(nv, dtype=torch.int64, device=dev)
    hot = torch.empty(nv, dtype=torch.uint8, device=dev)
    counters = torch.empty(2, dtype=torch.int32, device=dev)
    ws = torch.empty(L.pps_mesh_weld_ws_bytes(nv), dtype=torch.uint8, device=dev)
    _lib.check(L.pps_mesh_corner_weld(verts.data_ptr(), nv, int(digits), remap.data_ptr(), hot.data_ptr(), counters.data_ptr(), ws.data_ptr(), _stream(verts)),
               'pps_mesh_corner_weld')
    merged, out_of_range = counters.tolist()
    if out_of_range:
        raise _lib.PpsError('pps_mesh_corner_weld: vertex coordinates outside [0, 524287] (not a mesh in grid index space)')
    return remap, hot, int(merged)


def mesh_face_filter(faces: torch.Tensor, hot: torch.Tensor):
    """bool [F] keep: not degenerate and, among the faces around a hot vertex, the first of its vertex triple (trimesh remove_degenerate_faces /
    remove_duplicate_faces, source/base/mesh.py:12-17, after a corner weld)."""
    assert faces.is_cuda and faces.dtype == torch.int64 and hot.dtype == torch.uint8
    L = _lib.lib()
    nf = faces.shape[0]
    keep = torch.empty(nf, dtype=torch.uint8, device=faces.device)
    if nf == 0:
        return keep.bool()
    faces = faces.contiguous()
    ws = torch.empty(L.pps_mesh_face_filter_ws_bytes(nf), dtype=torch.uint8, device=faces.device)
    _lib.check(L.pps_mesh_face_filter(faces.data_ptr(), nf, hot.data_ptr(), keep.data_ptr(), ws.data_ptr(), _stream(faces)), 'pps_mesh_face_filter')
    return keep.bool()
