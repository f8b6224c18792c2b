"""Marching Cubes and mesh clean-up (post-processing of the occupancy volume): HIP kernels on the device, numpy twins as the CPU reference.

Stands in for `skimage.measure.marching_cubes` (source/poco_utils.py:96, Lewiner's variant) and the trimesh clean-up of
source/base/mesh.py:7-38 (merge vertices, drop degenerate / duplicate faces, drop connected components with <= 6 faces);
neither package exists in the build image and the reference pins no output for them ("parity unpinned").

The triangle table is DERIVED at import instead of being typed in.  For every corner-sign pattern (256) and every resolution of the pattern's
AMBIGUOUS faces (a face whose inside corners are diagonal: up to 6 faces, one bit each) the crossing points of each cube face are joined by
segments, the segments chain into closed loops, and each loop is triangulated without in-face chords.  How an ambiguous face is resolved is not a
table convention but a property of the DATA: the asymptotic decider of Nielson & Hamann (the sign of the bilinear interpolant at the face's
saddle point, f00 f11 - f10 f01 against the level), which is what Lewiner et al. (2003), "Efficient implementation of Marching Cubes' cases with
topological guarantees" -- the method behind skimage's default -- use for their face tests: the inside corners of the face are joined through the
face when the saddle value is inside, cut off separately when it is outside.  Both cubes that share a face evaluate the same four values in the same
order, so they agree and the surface has no cracks.  Interior ambiguities (Lewiner's cases 4, 6, 7, 10, 12, 13 with a tunnel through the cube) are NOT
resolved: every loop is closed with a disc, which is Lewiner's topology whenever his interior test finds no tunnel.
Vertices sit on grid edges (one fractional coordinate), which is what the bisection refinement of the reference relies on (poco_utils.py:111-119).
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


CENTER = 12          # pseudo edge id of the extra vertex inside the cube


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
    if cost > 0:
        # every triangulation of this polygon has an in-face chord (long loops that only exist for some resolutions of ambiguous faces): a fan
        # around an extra vertex INSIDE the cube, like the 13th vertex of Lewiner's tables (id CENTER; its position is the mean of the loop's
        # crossing points)
        return [(CENTER, loop[i], loop[(i + 1) % n]) for i in range(n)]
    return [(loop[a], loop[b], loop[c]) for a, b, c in tris]


def _face_uv_corners():
    """For each of the 6 faces (order of _face_cycles: axis 0 low, axis 0 high, axis 1 low, ...): the cube corners at (u, v) = (0,0), (1,0), (0,1),
    (1,1) with u = (axis + 1) % 3, v = (axis + 2) % 3 -- GLOBAL axis order, so the two cubes sharing a face read its corners in the same order."""
    out = []
    for ax in range(3):
        u, v = (ax + 1) % 3, (ax + 2) % 3
        for side in (0, 1):
            def corner(a, b):
                c = [0, 0, 0]
                c[ax], c[u], c[v] = side, a, b
                return c[0] | (c[1] << 1) | (c[2] << 2)
            out.append([corner(0, 0), corner(1, 0), corner(0, 1), corner(1, 1)])
    return np.array(out, dtype=np.int64)


_FACE_UV = _face_uv_corners()


def _ambiguous_mask(case):
    """bit f set: face f of corner pattern `case` has its two inside corners on a diagonal."""
    m = 0
    for f, (c00, c10, c01, c11) in enumerate(_FACE_UV):
        i00, i10, i01, i11 = [(case >> c) & 1 for c in (c00, c10, c01, c11)]
        if i00 == i11 and i10 == i01 and i00 != i10:
            m |= 1 << f
    return m


def _build_table():
    """-> (tri int8 [256 * 64, W, 3] cube-edge ids (-1 = none), ntri uint8 [256 * 64], amb uint8 [256]); row case * 64 + dec, dec = one bit per
    face: 1 = the inside corners of that (ambiguous) face are JOINED through the face.  Rows whose dec has bits outside amb[case] are copies of
    the row with those bits cleared, so a lookup may mask or not."""
    faces = _face_cycles()
    edge_faces = {i: frozenset(fi for fi, cyc in enumerate(faces) if a in cyc and b in cyc) for i, (a, b) in enumerate(_EDGES)}
    rows = {}
    amb = np.zeros(256, dtype=np.uint8)
    for case in range(256):
        amb[case] = _ambiguous_mask(case)
        inside = [(case >> c) & 1 for c in range(8)]
        for dec in range(64):
            if dec & ~int(amb[case]):
                continue
            nxt = {}                                    # directed segments between edge ids (inside on the left, seen from outside)
            for fi, cyc in enumerate(faces):
                enters, leaves = [], []                 # walking the face boundary counter-clockwise: edges where the inside region is entered / left
                for i in range(4):
                    c0, c1 = cyc[i], cyc[(i + 1) % 4]
                    if inside[c0] != inside[c1]:
                        (enters if inside[c1] else leaves).append((i, _EDGE_ID[tuple(sorted((c0, c1)))]))
                if len(enters) == 1:
                    nxt[leaves[0][1]] = enters[0][1]
                elif len(enters) == 2:
                    # ambiguous face.  Each inside corner lies between an 'enter' edge and the next 'leave' edge.
                    pair = {}
                    for ie, ee in enters:
                        il, el = min(leaves, key=lambda t: (t[0] - ie) % 4)          # the leave edge that follows this enter edge
                        pair[ee] = el
                    if (dec >> fi) & 1:
                        # joined: the inside region is a band through the face; a segment runs from each leave edge to the NEXT enter edge
                        # (cutting off the outside corner between them)
                        for il, el in leaves:
                            ie, ee = min(enters, key=lambda t: (t[0] - il) % 4)
                            nxt[el] = ee
                    else:
                        for ee, el in pair.items():                                  # separated: every inside corner is cut off by itself
                            nxt[el] = ee
            per_loop, seen = [], set()
            for start in sorted(nxt):
                if start in seen:
                    continue
                loop, e = [], start
                while e not in seen:
                    seen.add(e)
                    loop.append(e)
                    e = nxt[e]
                per_loop.append([(a, c, b) for a, b, c in _triangulate(loop, edge_faces)])      # winding: normals point towards LOWER values
            # a fan around the centre vertex (at most one per cube: such loops have >= 8 of the 12 edges) is listed FIRST in its row: the kernels
            # read "this cube has a centre vertex" off the row's first entry
            per_loop.sort(key=lambda t: 0 if t[0][0] == CENTER else 1)
            assert sum(1 for t in per_loop if t[0][0] == CENTER) <= 1
            rows[(case, dec)] = [t for lp in per_loop for t in lp]
    width = max(len(t) for t in rows.values())
    tri = np.full((256 * 64, width, 3), -1, dtype=np.int8)
    ntri = np.zeros(256 * 64, dtype=np.uint8)
    for case in range(256):
        for dec in range(64):
            t = rows[(case, dec & int(amb[case]))]
            ntri[case * 64 + dec] = len(t)
            if t:
                tri[case * 64 + dec, :len(t)] = np.array(t, dtype=np.int8)
    return tri, ntri, amb


_TRI_TABLE, _NTRI, _AMB = _build_table()
TABLE_WIDTH = int(_TRI_TABLE.shape[1])


def face_decisions(corner_vals, level):
    """corner_vals: sequence of 8 arrays (values at cube corner c, offsets (c & 1, (c >> 1) & 1, (c >> 2) & 1)) -> int64 array of 6 decision
    bits: bit f = 1 when the inside corners of face f are joined through the face by the asymptotic decider.  For the corners of a face in
    global (u, v) order, d = (f00 - L)(f11 - L) - (f10 - L)(f01 - L) has the sign of the saddle value when (0,0), (1,1) are the inside pair and the
    opposite sign when (1,0), (0,1) are; d == 0 counts as outside (separated).  The bit is only meaningful for ambiguous faces (mask _AMB)."""
    dec = None
    for f, (c00, c10, c01, c11) in enumerate(_FACE_UV):
        a, b, c, d = [corner_vals[i] - level for i in (c00, c10, c01, c11)]
        det = a * d - b * c
        joined = np.where(a > 0, det > 0, det < 0)
        bit = joined.astype(np.int64) << f
        dec = bit if dec is None else dec | bit
    return dec


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
    with np.errstate(invalid='ignore'):
        dec = face_decisions([v[cx, cy, cz] for v in corner_vals], level) & _AMB[case[cx, cy, cz]].astype(np.int64)
    tris = _TRI_TABLE[case[cx, cy, cz] * 64 + dec].astype(np.int64)   # [n, W, 3] local edge ids
    valid = tris[:, :, 0] >= 0
    cube_of = np.broadcast_to(np.arange(cx.size)[:, None], valid.shape)[valid]
    e = tris[valid]                                                  # [T,3] cube-edge ids, CENTER = the extra vertex inside the cube
    base = np.stack([cx, cy, cz], axis=1)[cube_of]                   # [T,3]
    is_c = e == CENTER
    ee = np.where(is_c, 0, e)
    origin = base[:, None, :] + _EDGE_ORIGIN[ee]                     # [T,3(verts),3]
    axis = _EDGE_AXIS[ee]
    n_edges = nx * ny * nz * 3
    cube_lin = (base[:, 0] * (ny - 1) + base[:, 1]) * (nz - 1) + base[:, 2]
    # vertex keys: grid edges first (voxel-major, axis-minor), then the centre vertices in cube order
    key = np.where(is_c, n_edges + cube_lin[:, None], ((origin[..., 0] * ny + origin[..., 1]) * nz + origin[..., 2]) * 3 + axis)
    ukey, faces = np.unique(key.reshape(-1), return_inverse=True)
    faces = faces.reshape(-1, 3)
    ne = int(np.searchsorted(ukey, n_edges))
    ek = ukey[:ne]
    ax = ek % 3
    lin = ek // 3
    o = np.stack([lin // (ny * nz), (lin // nz) % ny, lin % nz], axis=1)
    o2 = o.copy()
    o2[np.arange(o.shape[0]), ax] += 1
    va = vol[o[:, 0], o[:, 1], o[:, 2]]
    vb = vol[o2[:, 0], o2[:, 1], o2[:, 2]]
    t = (level - va) / (vb - va)
    verts = np.zeros((ukey.shape[0], 3), dtype=np.float64)
    verts[:ne] = o
    verts[np.arange(ne), ax] += t
    if ne < ukey.shape[0]:
        # a centre vertex = mean of the crossing points of its loop: the fan (CENTER, v_i, v_i+1) lists every loop vertex once in its 2nd column;
        # summed in fan order, then divided (the kernel does the same)
        fan = is_c[:, 0]
        cidx, vidx = faces[fan, 0], faces[fan, 1]
        acc = np.zeros((ukey.shape[0], 3)); cnt = np.zeros(ukey.shape[0])
        for j in np.argsort(cidx, kind='stable'):                      # fan triangles of one cube are consecutive and in table order
            acc[cidx[j]] += verts[vidx[j]]; cnt[cidx[j]] += 1
        verts[ne:] = acc[ne:] / cnt[ne:, None]
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
_DEV_TABLES = {}


def device_tables(dev):
    """(tri int8 [16384, W, 3], ntri uint8 [16384], amb uint8 [256]) on `dev` (uploaded once)."""
    import torch
    t = _DEV_TABLES.get(str(dev))
    if t is None:
        t = (torch.from_numpy(_TRI_TABLE).to(dev).contiguous(), torch.from_numpy(_NTRI).to(dev), torch.from_numpy(_AMB).to(dev))
        _DEV_TABLES[str(dev)] = t
    return t


def marching_cubes_torch(volume, level: float = 0.0):
    """marching_cubes for a torch tensor -> (verts float64 [V,3], faces int64 [F,3]) on the tensor's device.  Device tensors run the HIP kernels
    (csrc/pps_mc.hip through ops.marching_cubes: classify + count, two block-level prefix sums, emit vertices welded by grid-edge key, emit faces);
    host tensors the numpy function above.  Same vertices in the same order and the same faces in the same order on both paths."""
    import torch
    if volume.is_cuda:
        from . import ops
        return ops.marching_cubes(volume.to(torch.float64).contiguous(), float(level))
    v, f = marching_cubes(volume.detach().numpy(), level)
    return torch.from_numpy(v), torch.from_numpy(f)


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


def clean_mesh_torch(verts, faces, min_component_faces=6, digits=8, welded=False, grid_coords=True):
    """Same as clean_mesh on torch tensors (small components by a fixed number of label-propagation rounds, _small_component_faces).
    welded=True: the caller passes a mesh whose vertices are welded by GRID-EDGE KEY (marching_cubes_torch, also after the bisection refinement,
    which moves a vertex along its own edge only).  Two such vertices can share a position (to `digits` digits) only where their grid edges meet:
    at a grid corner.  With grid_coords (vertices in index space) the candidates are found exactly -- the vertices within 10^-digits of a corner --
    and merging by position, degenerate and duplicate faces are restricted to them and the faces around them instead of three sorts over the
    whole mesh (3 of the 4 ms of a clean-up at R = 257).  The result is the same mesh; only the vertex ORDER differs from the general path (which
    sorts by position).  welded without grid_coords (model space, after the refinement): the merge is skipped -- a refined vertex stays at least
    step / 2^iterations away from the end points of its edge, everything that could coincide was merged by the first clean-up."""
    import torch
    if faces.shape[0] == 0:
        return verts, faces
    dev = verts.device
    scale = 10.0 ** digits
    nv = verts.shape[0]
    skip = False
    if welded and grid_coords:
        # the only vertices that can share a position: those that sit (after rounding) ON a grid corner -- a handful per mesh (float32 vertex
        # coordinates round a crossing within ~1e-5 of a corner onto it).  Merge exactly those, by position, and touch only the faces around them.
        skip = True
        near = ((verts - torch.round(verts)).abs() <= 1.0 / scale).all(dim=1)
        cand = torch.nonzero(near)[:, 0]
        if cand.shape[0] > 1:
            _, inv = torch.unique(torch.round(verts[cand] * scale), dim=0, return_inverse=True)
            first = torch.full((int(inv.max()) + 1,), nv, dtype=torch.int64, device=dev)
            first.scatter_reduce_(0, inv, cand, reduce='amin')
            rep = first[inv]
            if bool((rep != cand).any()):
                remap = torch.arange(nv, device=dev)
                remap[cand] = rep
                faces = remap[faces]
                faces = faces[(faces[:, 0] != faces[:, 1]) & (faces[:, 1] != faces[:, 2]) & (faces[:, 0] != faces[:, 2])]
                hot = torch.zeros(nv, dtype=torch.bool, device=dev)
                hot[rep[rep != cand]] = True                                # vertices something was merged into
                tf = torch.nonzero(hot[faces].any(dim=1))[:, 0]             # duplicate faces can only be among the faces around them
                if tf.shape[0] > 1:
                    srt = torch.sort(faces[tf], dim=1)[0]
                    _, finv = torch.unique(srt, dim=0, return_inverse=True)
                    keep1 = torch.full((int(finv.max()) + 1,), tf.shape[0], dtype=torch.int64, device=dev)
                    keep1.scatter_reduce_(0, finv, torch.arange(tf.shape[0], device=dev), reduce='amin')
                    drop = torch.ones(tf.shape[0], dtype=torch.bool, device=dev)
                    drop[keep1] = False
                    keep = torch.ones(faces.shape[0], dtype=torch.bool, device=dev)
                    keep[tf[drop]] = False
                    faces = faces[keep]
    elif welded:
        # model space after refinement: see the docstring (the first clean-up has merged what could coincide).  The argument needs every refined
        # vertex to stay off the end points of its edge; a vertex whose end-point values were NaN is left unrefined, and float32 midpoints can round
        # onto an end point for tiny steps / many iterations.  Cheap guard: if any face has two corners at one rounded position, the
        # general path below merges and drops as the reference's second clean-up does (source/base/mesh.py:7-20).
        pos = torch.round(verts * scale)
        p0, p1, p2 = pos[faces[:, 0]], pos[faces[:, 1]], pos[faces[:, 2]]
        degenerate = ((p0 == p1).all(dim=1) | (p1 == p2).all(dim=1) | (p0 == p2).all(dim=1)).any()
        skip = not bool(degenerate)
    if not skip:
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
