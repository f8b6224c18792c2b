"""Marching Cubes and mesh clean-up in numpy (host post-processing of the occupancy volume).

Stands in for `skimage.measure.marching_cubes` (source/poco_utils.py:96) and the trimesh clean-up of
source/base/mesh.py:7-38 (merge vertices, drop degenerate / duplicate faces, drop connected components with <= 6 faces);
neither package exists in the build image and the reference pins no output for them ("parity unpinned").

The 256-case triangle table is DERIVED at import instead of being typed in: for every corner-sign pattern the crossing
points of each cube face are joined by segments (an ambiguous face always cuts off its inside corners, a rule that only
depends on the face's own corners, so neighbouring cubes agree and the surface has no cracks), the segments chain into
closed loops, and each loop is fan-triangulated.  Vertices sit on grid edges (one fractional coordinate), which is what
the bisection refinement of the reference relies on (poco_utils.py:111-119).
"""
import numpy as np

# corner c has offset (c & 1, (c >> 1) & 1, (c >> 2) & 1); an edge is a pair of corners differing in one bit
_CORNERS = np.array([[c & 1, (c >> 1) & 1, (c >> 2) & 1] for c in range(8)], dtype=np.int64)
_EDGES = [(a, a | (1 << ax)) for ax in range(3) for a in range(8) if not a & (1 << ax)]          # 12 edges
_EDGE_ID = {e: i for i, e in enumerate(_EDGES)}
_EDGE_AXIS = np.array([int(np.log2(b - a)) for a, b in _EDGES], dtype=np.int64)
_EDGE_ORIGIN = _CORNERS[[a for a, _ in _EDGES]]


def _face_cycles():
    """For each of the 6 faces: its 4 corners in counter-clockwise order seen from OUTSIDE the cube."""
    faces = []
    for ax in range(3):
        u, v = (ax + 1) % 3, (ax + 2) % 3
        for side in (0, 1):
            def corner(a, b):
                c = [0, 0, 0]
                c[ax], c[u], c[v] = side, a, b
                return c[0] | (c[1] << 1) | (c[2] << 2)
            cyc = [corner(0, 0), corner(1, 0), corner(1, 1), corner(0, 1)]     # CCW around +ax
            faces.append(cyc if side == 1 else cyc[::-1])
    return faces


def _triangulate(loop, edge_faces):
    """Triangles (as index triples in loop order) of the polygon `loop` of cube-edge ids WITHOUT a diagonal that lies in a cube face: two
    crossing points on edges of one face that are not joined by that face's own segment would give a triangle edge flush with the face, and
    the cube across the face may draw the same chord -- four triangles on one edge, a non-manifold mesh (found by oracle/mesh_oracle.py on a
    learned volume; a plain fan from loop[0] has such chords in 18 of the 256 cases).  All triangulations of the <= 12-gon are enumerated
    (Catalan numbers) and the first one without an in-face chord is taken; every case has one."""
    n = len(loop)

    def chord_ok(i, j):
        if (j - i) % n in (1, n - 1):
            return True                                           # a polygon side: the face segment itself
        return not (edge_faces[loop[i]] & edge_faces[loop[j]])

    from functools import lru_cache

    @lru_cache(maxsize=None)
    def best(i, j):                                               # triangulations of the sub-polygon i..j (chord i-j given): (bad chords, triangles)
        if j - i < 2:
            return 0, ()
        out = None
        for k in range(i + 1, j):
            bi, ti = best(i, k)
            bj, tj = best(k, j)
            cost = bi + bj + (0 if chord_ok(i, k) else 1) + (0 if chord_ok(k, j) else 1)
            if out is None or cost < out[0]:
                out = (cost, ti + tj + ((i, k, j),))
        return out

    cost, tris = best(0, n - 1)
    assert cost == 0, 'no triangulation without an in-face chord for loop {}'.format(loop)
    return [(loop[a], loop[b], loop[c]) for a, b, c in tris]


def _build_table():
    faces = _face_cycles()
    edge_faces = {i: frozenset(fi for fi, cyc in enumerate(faces) if a in cyc and b in cyc) for i, (a, b) in enumerate(_EDGES)}
    table = []
    for case in range(256):
        inside = [(case >> c) & 1 for c in range(8)]
        nxt = {}                                        # directed segments between edge ids (inside on the left, seen from outside)
        for cyc in faces:
            for i in range(4):
                c0, c1, c2 = cyc[i - 1], cyc[i], cyc[(i + 1) % 4]
                # walking CCW: entering the inside region on edge (c0->c1) and leaving it on a later edge
                if not inside[c0] and inside[c1]:
                    j = i
                    while inside[cyc[(j + 1) % 4]] and (j + 1) % 4 != (i - 1) % 4:
                        j += 1
                    e_in = _EDGE_ID[tuple(sorted((c0, c1)))]
                    e_out = _EDGE_ID[tuple(sorted((cyc[j % 4], cyc[(j + 1) % 4])))]
                    nxt[e_out] = e_in
        tris, seen = [], set()
        for start in sorted(nxt):
            if start in seen:
                continue
            loop, e = [], start
            while e not in seen:
                seen.add(e)
                loop.append(e)
                e = nxt[e]
            for a, b, c in _triangulate(loop, edge_faces):
                tris.append((a, c, b))                              # winding: normals point towards LOWER values
        table.append(tris)
    width = max(len(t) for t in table)
    out = np.full((256, width, 3), -1, dtype=np.int64)
    for i, t in enumerate(table):
        if t:
            out[i, :len(t)] = np.array(t)
    return out


_TRI_TABLE = _build_table()


def marching_cubes(volume: np.ndarray, level: float = 0.0):
    """volume [X,Y,Z] (NaN = unknown: cubes touching a NaN are skipped) -> (verts float64 [V,3] in index space,
    faces int64 [F,3]).  'Inside' is value > level; triangles are oriented with normals towards lower values."""
    vol = np.asarray(volume, dtype=np.float64)
    nx, ny, nz = vol.shape
    corner_vals = [vol[dx:nx - 1 + dx, dy:ny - 1 + dy, dz:nz - 1 + dz] for dx, dy, dz in _CORNERS]
    finite = np.ones(corner_vals[0].shape, dtype=bool)
    case = np.zeros(corner_vals[0].shape, dtype=np.int64)
    for c, v in enumerate(corner_vals):
        finite &= ~np.isnan(v)
        with np.errstate(invalid='ignore'):
            case |= (v > level).astype(np.int64) << c
    active = finite & (case != 0) & (case != 255)
    cx, cy, cz = np.nonzero(active)
    if cx.size == 0:
        return np.zeros((0, 3)), np.zeros((0, 3), dtype=np.int64)
    tris = _TRI_TABLE[case[cx, cy, cz]]                             # [n, W, 3] local edge ids
    valid = tris[:, :, 0] >= 0
    cube_of = np.broadcast_to(np.arange(cx.size)[:, None], valid.shape)[valid]
    e = tris[valid]                                                  # [T,3]
    base = np.stack([cx, cy, cz], axis=1)[cube_of]                   # [T,3]
    origin = base[:, None, :] + _EDGE_ORIGIN[e]                      # [T,3(verts),3]
    axis = _EDGE_AXIS[e]
    key = ((origin[..., 0] * ny + origin[..., 1]) * nz + origin[..., 2]) * 3 + axis
    ukey, faces = np.unique(key.reshape(-1), return_inverse=True)
    faces = faces.reshape(-1, 3)
    ax = ukey % 3
    lin = ukey // 3
    o = np.stack([lin // (ny * nz), (lin // nz) % ny, lin % nz], axis=1)
    o2 = o.copy()
    o2[np.arange(o.shape[0]), ax] += 1
    va = vol[o[:, 0], o[:, 1], o[:, 2]]
    vb = vol[o2[:, 0], o2[:, 1], o2[:, 2]]
    t = (level - va) / (vb - va)
    verts = o.astype(np.float64)
    verts[np.arange(o.shape[0]), ax] += t
    return verts, faces.astype(np.int64)


def clean_mesh(verts: np.ndarray, faces: np.ndarray, min_component_faces=6, digits=8):
    """merge vertices by position, drop degenerate and duplicate faces, keep connected components with MORE than
    `min_component_faces` faces (mesh.py:7-38), drop unreferenced vertices."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    if faces.shape[0] == 0:
        return verts, faces
    _, first, inv = np.unique(np.round(verts, digits), axis=0, return_index=True, return_inverse=True)
    verts = verts[first]
    faces = inv.reshape(-1)[faces]
    ok = (faces[:, 0] != faces[:, 1]) & (faces[:, 1] != faces[:, 2]) & (faces[:, 0] != faces[:, 2])
    faces = faces[ok]
    _, keep = np.unique(np.sort(faces, axis=1), axis=0, return_index=True)
    faces = faces[np.sort(keep)]
    if faces.shape[0] and min_component_faces is not None:
        nf = faces.shape[0]
        edges = np.sort(np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]]), axis=1)
        owner = np.tile(np.arange(nf), 3)
        order = np.lexsort((edges[:, 1], edges[:, 0]))
        es, ow = edges[order], owner[order]
        same = (es[1:] == es[:-1]).all(axis=1)
        a, b = ow[:-1][same], ow[1:][same]
        graph = coo_matrix((np.ones(a.shape[0]), (a, b)), shape=(nf, nf))
        _, label = connected_components(graph, directed=False)
        size = np.bincount(label)
        faces = faces[size[label] > min_component_faces]
    used, inv = np.unique(faces.reshape(-1), return_inverse=True)
    return verts[used], inv.reshape(-1, 3).astype(np.int64)


# ---------------------------------------------------------------------------------------------------------------------
# torch twins (run on the device that holds the volume; identical results to the numpy functions above)
# ---------------------------------------------------------------------------------------------------------------------
def marching_cubes_torch(volume, level: float = 0.0):
    """Same as marching_cubes for a torch tensor on any device -> (verts float64 [V,3], faces int64 [F,3]) on that device."""
    import torch
    vol = volume.to(torch.float64)
    dev = vol.device
    nx, ny, nz = vol.shape
    corners = _CORNERS.tolist()
    finite = None
    case = torch.zeros((nx - 1, ny - 1, nz - 1), dtype=torch.int64, device=dev)
    for c, (dx, dy, dz) in enumerate(corners):
        v = vol[dx:nx - 1 + dx, dy:ny - 1 + dy, dz:nz - 1 + dz]
        ok = ~torch.isnan(v)
        finite = ok if finite is None else (finite & ok)
        case |= (v > level).to(torch.int64) << c
    active = finite & (case != 0) & (case != 255)
    cubes = torch.nonzero(active)
    if cubes.shape[0] == 0:
        return torch.zeros((0, 3), dtype=torch.float64, device=dev), torch.zeros((0, 3), dtype=torch.int64, device=dev)
    table = torch.from_numpy(_TRI_TABLE).to(dev)
    tris = table[case[cubes[:, 0], cubes[:, 1], cubes[:, 2]]]          # [n, W, 3]
    valid = tris[:, :, 0] >= 0
    cube_of = torch.arange(cubes.shape[0], device=dev)[:, None].expand_as(valid)[valid]
    e = tris[valid]                                                     # [T,3]
    origin = cubes[cube_of][:, None, :] + torch.from_numpy(_EDGE_ORIGIN).to(dev)[e]
    axis = torch.from_numpy(_EDGE_AXIS).to(dev)[e]
    key = ((origin[..., 0] * ny + origin[..., 1]) * nz + origin[..., 2]) * 3 + axis
    ukey, faces = torch.unique(key.reshape(-1), return_inverse=True)
    faces = faces.reshape(-1, 3)
    ax = ukey % 3
    lin = torch.div(ukey, 3, rounding_mode='floor')
    o = torch.stack([torch.div(lin, ny * nz, rounding_mode='floor'), torch.div(lin, nz, rounding_mode='floor') % ny, lin % nz], dim=1)
    ar = torch.arange(o.shape[0], device=dev)
    o2 = o.clone()
    o2[ar, ax] += 1
    va = vol[o[:, 0], o[:, 1], o[:, 2]]
    vb = vol[o2[:, 0], o2[:, 1], o2[:, 2]]
    verts = o.to(torch.float64)
    verts[ar, ax] += (level - va) / (vb - va)
    return verts, faces


def _small_component_faces(a, b, nf, k):
    """bool [nf]: faces that lie in a connected component of at most k faces; (a, b) = pairs of faces sharing an edge.
    EXACT without iterating to convergence: a component of <= k faces has diameter <= k - 1, so after k rounds of min-label propagation (each
    round one hop) all its faces carry the component's smallest face id -- its label class is the whole component and no adjacency leaves the
    class.  Conversely a label class that no adjacency leaves is a union of whole components, hence the component itself; so
    "class size <= k and closed" holds exactly for the faces of small components, while the unconverged classes inside large components are
    either open or larger than k.  k rounds, two scatters each, no host synchronisation (round 2 iterated pointer jumping to a fixed point --
    ~75 rounds with a device-to-host comparison each at R = 257 -- and counted sizes with a histogram kernel: 17 of the 18 ms of a clean-up)."""
    import torch
    dev = a.device
    label = torch.arange(nf, device=dev)
    for _ in range(max(int(k), 1)):
        new = label.clone()
        new.scatter_reduce_(0, a, label[b], reduce='amin')
        new.scatter_reduce_(0, b, label[a], reduce='amin')
        label = new
    size = torch.zeros(nf, dtype=torch.int64, device=dev).scatter_add_(0, label, torch.ones(nf, dtype=torch.int64, device=dev))
    la, lb = label[a], label[b]
    diff = la != lb
    is_open = torch.zeros(nf, dtype=torch.bool, device=dev)
    is_open[la[diff]] = True
    is_open[lb[diff]] = True
    return (size[label] <= k) & ~is_open[label]


def clean_mesh_torch(verts, faces, min_component_faces=6, digits=8):
    """Same as clean_mesh on torch tensors (small components by a fixed number of label-propagation rounds, _small_component_faces)."""
    import torch
    if faces.shape[0] == 0:
        return verts, faces
    dev = verts.device
    scale = 10.0 ** digits
    _, inv = torch.unique(torch.round(verts * scale), dim=0, return_inverse=True)
    first = torch.full((int(inv.max()) + 1,), verts.shape[0], dtype=torch.int64, device=dev)
    first.scatter_reduce_(0, inv, torch.arange(verts.shape[0], device=dev), reduce='amin')
    verts = verts[first]
    faces = inv[faces]
    ok = (faces[:, 0] != faces[:, 1]) & (faces[:, 1] != faces[:, 2]) & (faces[:, 0] != faces[:, 2])
    faces = faces[ok]
    srt = torch.sort(faces, dim=1)[0]
    nv = verts.shape[0]
    fkey = (srt[:, 0] * nv + srt[:, 1]) * nv + srt[:, 2] if nv < 2_000_000 else None
    if fkey is not None:
        _, finv = torch.unique(fkey, return_inverse=True)
        keep = torch.full((int(finv.max()) + 1,), faces.shape[0], dtype=torch.int64, device=dev)
        keep.scatter_reduce_(0, finv, torch.arange(faces.shape[0], device=dev), reduce='amin')
        faces = faces[torch.sort(keep)[0]]
    if faces.shape[0] and min_component_faces is not None:
        nf = faces.shape[0]
        e = torch.cat([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]])
        e = torch.sort(e, dim=1)[0]
        ekey = e[:, 0] * nv + e[:, 1]
        owner = torch.arange(nf, device=dev).repeat(3)
        skey, order = torch.sort(ekey, stable=True)
        ow = owner[order]
        same = skey[1:] == skey[:-1]
        a, b = ow[:-1][same], ow[1:][same]
        faces = faces[~_small_component_faces(a, b, nf, int(min_component_faces))]
    used = torch.zeros(verts.shape[0], dtype=torch.bool, device=dev)           # compaction of the referenced vertices, order kept
    used[faces.reshape(-1)] = True
    remap = torch.cumsum(used, 0) - 1
    return verts[used], remap[faces]
