"""Iso-surface driver around the HIP decoder: region-growing volume, Marching Cubes, vertex refinement.

Mirrors source/poco_utils.py:26-254 `export_mesh_and_refine_vertices_region_growing_v3` / `_create_volume`:
same grid geometry (scalar min/max of the cloud, step = (max-min)/(R-1), padding 1), same +-2 dilation band, same frontier
rule, same float64 volume with NaN = unseen and `out_value` borders, same 10 bisection rounds.

What is different by design (MI355X-first):
  * the cloud, the latent table, the query lists and the volume live on the GPU; per chunk there is no host round trip
    (the reference goes device -> host -> device for the kd-tree and the patches, poco_utils.py:67-72,261-273);
  * one 64-NN search per chunk serves both the interpolation ids and the 50-NN patches (same cloud in predict:
    source/occupancy_data_module.py:227-253, manifold_points=None);
  * dilation is a separable max-filter on the device instead of a Python loop over points;
  * already evaluated voxels of the band are NOT re-evaluated in later growth rounds (the reference re-evaluates them,
    `volume[mask] = z`, poco_utils.py:232); the decoder is deterministic, so the volume is identical.
"""
import typing

import numpy as np
import torch

from . import ops, mcubes, sharding


def _dilate(mask: torch.Tensor, r: int) -> torch.Tensor:
    """Binary dilation with a (2r+1)^3 box, clipped at the volume border (poco_utils.py:181-196).  Device masks: the HIP byte-mask kernel
    (pps_dilate_box_u8; 12 calls per R = 257 shape, 1.1 ms each as max_pool3d over a float copy in round 2).  Masks on the host only occur in the
    CPU tests of the driver logic (stub decoder); they take the torch op."""
    if mask.is_cuda:
        return ops.dilate_box(mask, r)
    m = mask[None, None].float()
    m = torch.nn.functional.max_pool3d(m, kernel_size=2 * r + 1, stride=1, padding=r)
    return m[0, 0] > 0


CHUNK_MULT = 4


class OccupancyField:
    """occ(q) for arbitrary query points of ONE shape on the GPU, chunked by rec_batch_size.
    PPSurf networks (with `point_net`): kNN + patches + fused decoder.  POCO networks: kNN + projection head."""

    def __init__(self, network, latent: dict, pts_raw_ms: torch.Tensor, num_pts: int, num_pts_local: typing.Optional[int]):
        self.net = network
        pts = latent['pts']
        pts = pts if pts.shape[1] == 3 else pts.transpose(1, 2)
        self.dev = pts.device
        self.pts = pts[0].t().contiguous().float()                        # [N,3]
        self.plan = network.decoder_plan(self.dev)
        self.table = network.point_table(latent['latents'][0], self.plan)
        self.k = min(network.projection.k, self.pts.shape[0])
        # `num_pts` (rec_batch_size) exists in the reference to bound memory; queries are independent, so decoding CHUNK_MULT of them per
        # launch sequence gives identical results with fewer, better filled launches (16-query tiles over 2048 wave slots: 12.33 -> 12.03 ms
        # per 50000 queries at 4x, round-3 probe subchunk_probe.py (git history); 3.3 GB of scratch instead of 0.8 GB)
        self.chunk = int(num_pts) * CHUNK_MULT
        self.n_queries = 0
        self.ppsurf = hasattr(network, 'point_net')
        if self.ppsurf:
            if num_pts_local is None:
                raise ValueError('PPSurf networks need num_pts_local')
            self.raw = pts_raw_ms[0].to(self.dev).contiguous().float() if pts_raw_ms is not None else self.pts
            same = self.raw.shape == self.pts.shape and bool(torch.equal(self.raw, self.pts))
            from .decoder import ChunkPipeline
            self.pipe = ChunkPipeline(self.plan, self.table, self.pts, self.raw, self.k, num_pts_local, same, self.chunk)
        else:
            self.blocks = ops.KnnBlocks(self.pts)

    @torch.no_grad()
    def __call__(self, queries: torch.Tensor) -> torch.Tensor:
        """queries [q,3] float32 on the device -> occ [q] float32 (= softmax(logits)[0] - softmax(logits)[1], poco_utils.py:78-81)."""
        if queries.shape[0] == 0:
            return torch.empty((0,), device=self.dev)
        chunks = [queries[s:s + self.chunk].contiguous() for s in range(0, queries.shape[0], self.chunk)]
        self.n_queries += queries.shape[0]
        if self.ppsurf:
            return torch.cat([occ for _, occ in self.pipe.run(chunks, want_occ=True)])
        out = []
        for q in chunks:
            pr = torch.softmax(self.plan.decode(self.table, self.pts, q, self.blocks.query(q, self.k)), dim=1)
            out.append(pr[:, 0] - pr[:, 1])
        return torch.cat(out)


FIELD_CLASS = OccupancyField          # measurement workloads substitute a subclass (bench_workloads.py)


def create_volume(field, pts_ids: torch.Tensor, resolution: int, step: float, bmin_pad: float, padding=1, dilation_size=2,
                  out_value=1.0, progress=None):
    """Region growing (poco_utils.py:178-254).  pts_ids int64 [n,3] voxel ids of the input points (device).
    Returns the float64 volume (device) with NaN = never evaluated."""
    dev = pts_ids.device
    n = resolution + 2 * padding
    volume = torch.full((n, n, n), float('nan'), dtype=torch.float64, device=dev)
    to_see = torch.ones((n, n, n), dtype=torch.bool, device=dev)
    it = 0
    while pts_ids.shape[0] > 0:
        seeds = torch.zeros((n, n, n), dtype=torch.bool, device=dev)
        seeds[pts_ids[:, 0], pts_ids[:, 1], pts_ids[:, 2]] = True
        band = _dilate(seeds, dilation_size)
        todo = ops.grow_band_todo(volume, band) if volume.is_cuda else band & torch.isnan(volume)      # skip voxels already evaluated
        coords = torch.nonzero(todo)
        if coords.shape[0] > 0:
            q = coords.to(torch.float32) * np.float32(step) + np.float32(bmin_pad)      # :212-213 (float32 arithmetic)
            volume[todo] = sharding.sharded_map(field, q).to(torch.float64)    # query blocks sharded over the ranks, if any
        to_see[pts_ids[:, 0], pts_ids[:, 1], pts_ids[:, 2]] = False
        v = volume[pts_ids[:, 0], pts_ids[:, 1], pts_ids[:, 2]]
        neg_seeds = torch.zeros_like(seeds)
        pos_seeds = torch.zeros_like(seeds)
        s = pts_ids[v <= 0]
        neg_seeds[s[:, 0], s[:, 1], s[:, 2]] = True
        s = pts_ids[v >= 0]
        pos_seeds[s[:, 0], s[:, 1], s[:, 2]] = True
        neg, pos = _dilate(neg_seeds, dilation_size), _dilate(pos_seeds, dilation_size)
        new_mask = ops.grow_frontier(volume, neg, pos, to_see) if volume.is_cuda else (neg & (volume >= 0) & to_see) | (pos & (volume <= 0) & to_see)
        pts_ids = torch.nonzero(new_mask)
        it += 1
        if progress is not None:
            progress('occ_batch round {}'.format(it))
    p = padding
    volume[:p] = out_value; volume[-p:] = out_value
    volume[:, :p] = out_value; volume[:, -p:] = out_value
    volume[:, :, :p] = out_value; volume[:, :, -p:] = out_value
    return volume


def refine_vertices(eval_occ, verts: torch.Tensor, volume: torch.Tensor, step, bmin_pad, refine_iter: int, progress=None) -> torch.Tensor:
    """Bisection of the Marching-Cubes edge vertices (poco_utils.py:111-168) on the device of `verts`.
    verts [V,3] float64 in GRID coordinates; volume float64 with NaN = unseen; eval_occ(points float32 [q,3]) -> occ [q];
    step / bmin_pad: numpy float32 scalars of the grid geometry.  Returns the vertices in model space, float64 [V,3].
    Same arithmetic as the reference (float32 corner coordinates and queries, float64 volume values and vertex array; the
    vertices are written back once at the end instead of after every round): tests/test_driver_parity_cpu.py compares it with
    the reference's own output (tests/golden/refine.npz)."""
    step32, bmin32 = np.float32(step), np.float32(bmin_pad)
    if refine_iter <= 0 or verts.shape[0] == 0:
        return verts * step32 + bmin32
    frac = (verts - torch.floor(verts)) > 0
    nfrac = frac.sum(dim=1)
    sel = torch.nonzero((nfrac > 0) & (nfrac < 2))[:, 0]               # vertices on a grid edge (:115-117)
    v = verts[sel]
    v1i = torch.floor(v).to(torch.int64)
    v2i = v1i + frac[sel].to(torch.int64)
    p1 = volume[v1i[:, 0], v1i[:, 1], v1i[:, 2]]
    p2 = volume[v2i[:, 0], v2i[:, 1], v2i[:, 2]]
    ok = ~torch.isnan(p1) & ~torch.isnan(p2)                           # :131-137
    sel, v, v1i, v2i, p1, p2 = sel[ok], v[ok], v1i[ok], v2i[ok], p1[ok].clone(), p2[ok].clone()
    v1 = v1i.to(torch.float32) * step32 + bmin32
    v2 = v2i.to(torch.float32) * step32 + bmin32
    verts = verts * step32 + bmin32
    vq = (v * step32 + bmin32).to(torch.float32)
    for it in range(refine_iter):                                       # :146-165
        pr = eval_occ(vq).to(torch.float64)
        m1 = (pr * p1) > 0
        v1[m1] = vq[m1]; p1[m1] = pr[m1]
        m2 = (pr * p2) > 0
        v2[m2] = vq[m2]; p2[m2] = pr[m2]
        vq = (v2 + v1) / 2
        if progress is not None:
            progress('refine iter {}'.format(it))
    verts[sel] = vq.to(verts.dtype)
    return verts


def export_mesh_and_refine_vertices_region_growing_v3(network, latent: dict, pts_raw_ms, resolution: int, padding=0, mc_value=0,
                                                      num_pts=50000, num_pts_local=None, refine_iter=10, input_points=None,
                                                      out_value=np.nan, dilation_size=2, prog_bar=None, pc_file_in: str = 'unknown'):
    """poco_utils.py:26-175.  Returns (vertices float32 [V,3], faces int64 [F,3]) in model space, or None when the occupancy
    never crosses `mc_value`.  (The reference wraps the same arrays into a trimesh.Trimesh; ppsurf_amd.meshio writes PLY.)"""
    if latent['pts'].shape[0] != 1:
        raise ValueError('Reconstruction must be done with batch size = 0!')     # message kept from poco_utils.py:50
    progress = None
    if prog_bar is not None and getattr(prog_bar, 'predict_progress_bar', None) is not None:
        progress = lambda s: prog_bar.predict_progress_bar.set_postfix_str('{}, {}'.format(pc_file_in[-24:], s), refresh=True)
    field = FIELD_CLASS(network, latent, pts_raw_ms, num_pts, num_pts_local)
    dev = field.dev
    input_points = np.asarray(input_points)
    bmin, bmax = input_points.min(), input_points.max()
    step = (bmax - bmin) / (resolution - 1)
    bmin_pad = bmin - padding * step
    pts_ids = torch.from_numpy(((input_points - bmin) / step + padding).astype(np.int32).astype(np.int64)).to(dev)
    volume = create_volume(field, pts_ids, resolution, step, bmin_pad, padding, dilation_size, out_value, progress)

    seen = volume[~torch.isnan(volume)]
    if not (float(seen.max()) > mc_value > float(seen.min())):
        return None
    # Marching Cubes (HIP kernels, csrc/pps_mc.hip), clean-up and the bisection refinement stay on the device (poco_utils.py:96-168)
    verts, faces = mcubes.marching_cubes_torch(volume, mc_value)
    verts = verts.to(torch.float32).to(torch.float64)            # skimage returns float32 vertices, trimesh stores them as float64
    verts, faces = mcubes.clean_mesh_torch(verts, faces, min_component_faces=6, welded=True, grid_coords=True)
    verts = refine_vertices(lambda q: sharding.sharded_map(field, q), verts, volume, step, bmin_pad, refine_iter, progress)
    verts, faces = mcubes.clean_mesh_torch(verts, faces, min_component_faces=6, welded=True, grid_coords=False)
    return verts.to(torch.float32).cpu().numpy(), faces.cpu().numpy()
