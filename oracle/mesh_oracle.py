"""TEST INFRASTRUCTURE ONLY -- never imported by the product (ppsurf_amd/, source/, pps.py, bench.py's timed region).

Independent SPECIFICATION of iso-surface extraction and mesh clean-up, the two steps of the reference's reconstruction that live in
third-party packages absent from this image (parity of the implementations themselves is unpinned):

  * `skimage.measure.marching_cubes(volume, level)`           called at source/poco_utils.py:95-96
  * trimesh `merge_vertices / remove_degenerate_faces / remove_duplicate_faces` and `graph.connected_components(face_adjacency)`
                                                               called at source/base/mesh.py:7-38 (components with <= 6 faces dropped)

Nothing here uses a triangle table or shares code with ppsurf_amd/mcubes.py: the functions state PROPERTIES that any correct Marching-Cubes
mesh of a volume must have, by plain enumeration over the grid, and a union-find restatement of the component filter:

  edge_crossings            the vertex set: one vertex per grid edge whose two (finite) end values straddle the level, at the linear
                            interpolation position -- restricted to edges of at least one cube with eight finite corners (cubes that touch
                            an unseen = NaN voxel produce no triangles: skimage emits NaN vertices there, which the reference's
                            clean_simple_inplace -> remove_infinite_values removes again)
  check_marching_cubes      vertices with >= 2 integer coordinates (on grid edges) == edge_crossings as a multiset; any OTHER vertex lies strictly inside
                            one all-finite cube, at the mean of the vertices of its link (the extra vertex of Lewiner's tables: skimage's default
                            method emits such vertices too, which is why the reference's refinement selects vertices with exactly one fractional
                            coordinate, source/poco_utils.py:115-117); every face inside ONE grid cube; the faces form a closed, consistently
                            oriented 2-manifold (every directed edge once, its reverse once) wherever the surface does not run into unseen
                            voxels; normals point from values > level towards values <= level
  check_face_decider        AMBIGUOUS cube faces (the two inside corners on a diagonal) are resolved by the asymptotic decider -- the sign of the
                            bilinear interpolant at the face's saddle point (Nielson & Hamann 1991, "The asymptotic decider"; the face tests of
                            Lewiner, Lopes, Vieira, Tavares 2003, "Efficient implementation of Marching Cubes' cases with topological guarantees",
                            the algorithm behind skimage.measure.marching_cubes): the mesh segments lying in such a face cut off the two OUTSIDE
                            corners when the saddle value is inside (> level) and the two INSIDE corners otherwise.  Two meshers that both satisfy
                            this agree on every face segment, i.e. on the surface's topology up to what happens strictly inside a cube
  check_cube_topology       [round 5] INTERIOR ambiguity (Chernyaev 1995; Lewiner et al. 2003 section 4, `test_interior`): two same-side groups of corners
                            that no face joins may be connected THROUGH the cube by the trilinear interpolant.  Per cube: the mesh patch has one
                            boundary loop per pair of surface regions it separates and consists of discs -- except exactly one ANNULUS (tube) per pair
                            of corner groups that the interior test (restated below from the papers, checked against the densely sampled trilinear
                            interpolant) joins.  A mesher that closes every loop with a disc (rounds 1-4) is REJECTED
  components_union_find     face components over shared (undirected) edges by union-find
  check_clean_mesh          the cleaned mesh == input with vertices merged by position, degenerate and duplicate faces dropped, components
                            of <= min faces dropped, unreferenced vertices dropped (compared as sets of face-corner positions)
"""
import numpy as np


def _cube_all_finite(vol):
    fin = np.isfinite(vol)
    nx, ny, nz = vol.shape
    ok = np.ones((nx - 1, ny - 1, nz - 1), dtype=bool)
    for dx in (0, 1):
        for dy in (0, 1):
            for dz in (0, 1):
                ok &= fin[dx:nx - 1 + dx, dy:ny - 1 + dy, dz:nz - 1 + dz]
    return ok


def edge_crossings(volume, level=0.0):
    """-> (positions float64 [V,3] sorted lexicographically, origins int64 [V,3], axes int64 [V]) of all level crossings on grid edges that
    belong to at least one all-finite cube.  'Inside' is value > level (a value equal to the level is outside: the crossing sits ON that
    corner)."""
    vol = np.asarray(volume, dtype=np.float64)
    shape = np.array(vol.shape)
    cube_ok = _cube_all_finite(vol)
    pos, org, axs = [], [], []
    for ax in range(3):
        u, v = (ax + 1) % 3, (ax + 2) % 3
        lo = [slice(None)] * 3
        hi = [slice(None)] * 3
        lo[ax], hi[ax] = slice(0, -1), slice(1, None)
        a, b = vol[tuple(lo)], vol[tuple(hi)]                 # end values of every edge along `ax`, indexed by the edge's lower corner
        with np.errstate(invalid='ignore'):
            straddle = np.isfinite(a) & np.isfinite(b) & ((a > level) != (b > level))
        # an edge (origin o, axis ax) belongs to the (up to) four cubes with corner o - du*e_u - dv*e_v, du, dv in {0, 1}
        o = np.stack(np.nonzero(straddle), axis=1)
        if o.shape[0] == 0:
            continue
        keep = np.zeros(o.shape[0], dtype=bool)
        for du in (0, 1):
            for dv in (0, 1):
                c = o.copy()
                c[:, u] -= du
                c[:, v] -= dv
                inside = (c >= 0).all(axis=1) & (c < shape - 1).all(axis=1)
                ci = np.where(inside[:, None], c, 0)
                keep |= inside & cube_ok[ci[:, 0], ci[:, 1], ci[:, 2]]
        o = o[keep]
        va, vb = a[o[:, 0], o[:, 1], o[:, 2]], b[o[:, 0], o[:, 1], o[:, 2]]
        p = o.astype(np.float64)
        p[:, ax] += (level - va) / (vb - va)
        pos.append(p); org.append(o); axs.append(np.full(o.shape[0], ax, dtype=np.int64))
    if not pos:
        return np.zeros((0, 3)), np.zeros((0, 3), dtype=np.int64), np.zeros((0,), dtype=np.int64)
    pos, org, axs = np.concatenate(pos), np.concatenate(org), np.concatenate(axs)
    order = np.lexsort((pos[:, 2], pos[:, 1], pos[:, 0]))
    return pos[order], org[order], axs[order]


def _trilinear(vol, p):
    i = np.clip(np.floor(p).astype(np.int64), 0, np.array(vol.shape) - 2)
    f = p - i
    out = np.zeros(p.shape[0])
    for dx in (0, 1):
        for dy in (0, 1):
            for dz in (0, 1):
                w = (f[:, 0] if dx else 1 - f[:, 0]) * (f[:, 1] if dy else 1 - f[:, 1]) * (f[:, 2] if dz else 1 - f[:, 2])
                out += w * vol[i[:, 0] + dx, i[:, 1] + dy, i[:, 2] + dz]
    return out


def check_marching_cubes(verts, faces, volume, level=0.0, atol=1e-9, require_closed=True):
    """Raises AssertionError with a description if (verts, faces) is not a Marching-Cubes mesh of `volume` in the sense of the module
    docstring.  verts float [V,3] in index space, faces int [F,3].  Returns a dict of counts."""
    vol = np.asarray(volume, dtype=np.float64)
    verts = np.asarray(verts, dtype=np.float64)
    faces = np.asarray(faces, dtype=np.int64)
    want, _, _ = edge_crossings(vol, level)
    all_verts = verts
    # 1. vertex multiset.  A vertex on a grid edge has at least two integer coordinates; the others must be cube-interior extra vertices (1b)
    n_int = (verts == np.floor(verts)).sum(axis=1)
    interior = n_int < 2
    if interior.any():
        _check_interior_vertices(all_verts, faces, interior, vol)
    verts = all_verts[~interior]
    assert verts.shape[0] == want.shape[0], 'vertex count {} != {} level crossings on edges of finite cubes'.format(verts.shape[0], want.shape[0])
    got = verts[np.lexsort((verts[:, 2], verts[:, 1], verts[:, 0]))]
    if want.shape[0]:
        # lexicographic order is not stable under rounding noise in x: compare through nearest partners of equal rank in a coarser sort
        err = np.abs(got - want).max()
        if err > atol:
            key = lambda a: np.lexsort((np.round(a[:, 2], 6), np.round(a[:, 1], 6), np.round(a[:, 0], 6)))
            err = np.abs(verts[key(verts)] - want[key(want)]).max()
        assert err <= atol, 'vertex positions differ from the linear-interpolation crossings by {}'.format(err)
    verts = all_verts
    if faces.shape[0] == 0:
        assert want.shape[0] == 0
        return {'vertices': 0, 'faces': 0, 'boundary_edges': 0}
    assert faces.min() >= 0 and faces.max() < verts.shape[0]
    assert np.unique(faces.reshape(-1)).shape[0] == verts.shape[0], 'unreferenced vertices'
    # 2. every face inside one grid cube (a vertex on a cube corner / edge belongs to all adjacent cubes: closed unit boxes)
    tri = verts[faces]                                                   # [F,3,3]
    # an integer cube origin c with c <= v <= c + 1 for the three corners exists iff ceil(max - 1) <= floor(min), per axis
    assert (np.ceil(tri.max(axis=1) - 1.0 - 1e-12) <= np.floor(tri.min(axis=1) + 1e-12)).all(), 'a face spans more than one grid cube'
    # 3. closed, consistently oriented 2-manifold: each directed edge at most once, and its reverse present
    e = np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]])
    assert (e[:, 0] != e[:, 1]).all(), 'degenerate face'
    nv = verts.shape[0]
    key = e[:, 0] * nv + e[:, 1]
    uniq, cnt = np.unique(key, return_counts=True)
    assert (cnt == 1).all(), 'a directed edge is used by {} faces: inconsistent orientation or a non-manifold edge'.format(int(cnt.max()))
    rev = e[:, 1] * nv + e[:, 0]
    boundary = ~np.isin(rev, uniq)
    nb = int(boundary.sum())
    if nb:
        # an open edge is only legitimate (a) where the surface leaves the grid: both end points on one outer face of the volume -- not
        # counted -- or (b) next to an unseen voxel: the cube across it touches a NaN
        ends = verts[e[boundary]]                                          # [nb,2,3]
        top = np.array(vol.shape, dtype=np.float64) - 1.0
        on_face = (((ends[:, 0] == 0) & (ends[:, 1] == 0)) | ((ends[:, 0] == top) & (ends[:, 1] == top))).any(axis=1)
        boundary[np.nonzero(boundary)[0][on_face]] = False
        nb = int(boundary.sum())
    if nb:
        mid = verts[e[boundary]].mean(axis=1)
        c = np.floor(mid).astype(np.int64)
        near_nan = np.zeros(nb, dtype=bool)
        for dx in (-1, 0, 1, 2):
            for dy in (-1, 0, 1, 2):
                for dz in (-1, 0, 1, 2):
                    p = np.clip(c + [dx, dy, dz], 0, np.array(vol.shape) - 1)
                    near_nan |= ~np.isfinite(vol[p[:, 0], p[:, 1], p[:, 2]])
        assert near_nan.all(), '{} open edges away from any unseen voxel'.format(int((~near_nan).sum()))
        assert not require_closed, '{} open edges (surface runs into unseen voxels)'.format(nb)
    # 4. orientation: the normal points from inside (value > level) to outside.  Step 3 makes the orientation CONSISTENT within every
    # edge-connected component, so one bit per component remains: probe the field a little off each face along its normal and take the
    # component's majority (a trilinear probe of a rough field misjudges single faces, never most faces of a component)
    n = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    area = np.linalg.norm(n, axis=1)
    big = area > 1e-3 * np.median(area[area > 0]) if (area > 0).any() else area > 0
    nn = n / np.maximum(area, 1e-300)[:, None]
    cen = tri.mean(axis=1)
    # faces whose probe would read an unseen voxel do not vote (the trilinear field is undefined there)
    ci = np.clip(np.floor(cen).astype(np.int64), 0, np.array(vol.shape) - 2)
    for dx in (0, 1):
        for dy in (0, 1):
            for dz in (0, 1):
                big &= np.isfinite(vol[ci[:, 0] + dx, ci[:, 1] + dy, ci[:, 2] + dz])
    fin = np.nan_to_num(vol, nan=level)
    # probe step: 0.05 voxel, but never more than a fraction of the face's own size (a one-voxel blob whose value barely crosses the level is an
    # octahedron of 1e-3 voxel: a fixed step would probe the far side of it)
    eps = np.minimum(0.05, 0.2 * np.sqrt(area))[:, None]
    d = _trilinear(fin, cen + eps * nn) - _trilinear(fin, cen - eps * nn)
    wrong = big & (d > 1e-12)
    assert wrong.sum() <= 0.05 * max(1, big.sum()), '{} of {} faces are oriented towards HIGHER values'.format(int(wrong.sum()), int(big.sum()))
    if wrong.any():
        lab = components_union_find(faces)
        _, inv = np.unique(lab, return_inverse=True)
        bad = np.bincount(inv, weights=wrong.astype(np.float64)) > 0.5 * np.maximum(np.bincount(inv, weights=big.astype(np.float64)), 1)
        assert not bad.any(), '{} components are oriented towards HIGHER values'.format(int(bad.sum()))
    amb = check_face_decider(verts, faces, vol, level)
    topo = check_cube_topology(verts, faces, vol, level)        # interior ambiguity: tubes exactly where the interior test joins two corner groups
    return {'vertices': int(verts.shape[0]), 'faces': int(faces.shape[0]), 'boundary_edges': nb, 'interior_vertices': int(interior.sum()),
            'ambiguous_faces': amb, 'multi_loop_cubes': topo['multi_loop'], 'tunnels': topo['tunnels']}


def _check_interior_vertices(verts, faces, interior, vol):
    """1b: a vertex that is not on a grid edge lies strictly inside ONE cube with eight finite corners, all its faces lie in that cube, its link is
    a single closed cycle of grid-edge vertices, and it sits at their mean."""
    cube_ok = _cube_all_finite(vol)
    ids = np.nonzero(interior)[0]
    p = verts[ids]
    c = np.floor(p).astype(np.int64)
    assert ((p > c) & (p < c + 1)).all(), 'an extra vertex lies on a cube face, not strictly inside a cube'
    assert (c >= 0).all() and (c < np.array(vol.shape) - 1).all() and cube_ok[c[:, 0], c[:, 1], c[:, 2]].all(), 'an extra vertex in a cube with an unseen corner'
    assert np.unique((c[:, 0] * vol.shape[1] + c[:, 1]) * vol.shape[2] + c[:, 2]).shape[0] == ids.shape[0], 'two extra vertices in one cube'
    where = {int(v): k for k, v in enumerate(ids)}
    link = [[] for _ in ids]
    for f in faces[interior[faces].any(axis=1)]:
        hub = [v for v in f if interior[v]]
        assert len(hub) == 1, 'a face joins two extra vertices'
        i = list(f).index(hub[0])
        link[where[int(hub[0])]].append((int(f[(i + 1) % 3]), int(f[(i + 2) % 3])))
    for k, segs in enumerate(link):
        assert len(segs) >= 3, 'an extra vertex with fewer than three faces'
        nxt = dict(segs)
        assert len(nxt) == len(segs), 'the link of an extra vertex is not a simple cycle'
        start, e, n = segs[0][0], segs[0][0], 0
        while True:
            e = nxt[e]
            n += 1
            if e == start or n > len(segs):
                break
        assert n == len(segs), 'the link of an extra vertex is not one closed cycle'
        ring = verts[[a for a, _ in segs]]
        assert ((ring >= c[k]) & (ring <= c[k] + 1)).all(), 'the fan of an extra vertex leaves its cube'
        assert np.abs(ring.mean(axis=0) - p[k]).max() <= 1e-9, 'an extra vertex is not at the mean of its link'


def check_face_decider(verts, faces, volume, level=0.0):
    """Every mesh segment that lies in an AMBIGUOUS grid face follows the asymptotic decider (module docstring).  Plain enumeration: a mesh edge lies
    in the grid face (axis ax, plane x_ax = c, cell (u0, v0)) when both end points have x_ax == c and fall into that unit square; if one end point
    sits on a u-edge of the square and the other on a v-edge, the segment cuts off the corner the two grid edges share.  For an ambiguous face the
    corner that is cut off must be OUTSIDE (value <= level) exactly when  (f00 - L)(f11 - L) - (f10 - L)(f01 - L)  has the sign that puts the
    saddle value inside.  Returns the number of ambiguous faces checked.  Segments with an end point ON a grid corner (a value exactly at the
    level) are skipped: the corner belongs to several grid edges and the face's segment structure is degenerate there."""
    vol = np.asarray(volume, dtype=np.float64)
    verts = np.asarray(verts, dtype=np.float64)
    faces = np.asarray(faces, dtype=np.int64)
    if faces.shape[0] == 0:
        return 0
    e = np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]])
    # a FACE SEGMENT is an edge the two cubes on either side of the grid face share: its (at most two) triangles lie in different cubes, or it has
    # one triangle only (volume border / next to an unseen voxel).  An edge whose two triangles lie in the SAME cube is interior to that cube's
    # patch even when it happens to be flush with a cube face (a bridge of a tube, see check_cube_topology; Lewiner's tables have such chords in
    # case 7.4.2) and says nothing about how the face was resolved.
    cube = np.floor(verts[faces].mean(axis=1)).astype(np.int64)
    cube_id = (cube[:, 0] * (vol.shape[1] + 1) + cube[:, 1]) * (vol.shape[2] + 1) + cube[:, 2]
    owner = np.tile(cube_id, 3)
    es = np.sort(e, axis=1)
    order = np.lexsort((owner, es[:, 1], es[:, 0]))
    es, owner = es[order], owner[order]
    new_edge = np.ones(es.shape[0], dtype=bool)
    new_edge[1:] = (es[1:] != es[:-1]).any(axis=1)
    group = np.cumsum(new_edge) - 1
    count = np.bincount(group)
    lo = np.full(group.max() + 1, np.iinfo(np.int64).max); hi = np.full(group.max() + 1, -1)
    np.minimum.at(lo, group, owner); np.maximum.at(hi, group, owner)
    segment = (count == 1) | (lo != hi)
    e = es[new_edge][segment]
    pa, pb = verts[e[:, 0]], verts[e[:, 1]]
    shape = np.array(vol.shape)
    checked = set()
    for ax in range(3):
        u, v = (ax + 1) % 3, (ax + 2) % 3
        flat = (pa[:, ax] == pb[:, ax]) & (pa[:, ax] == np.floor(pa[:, ax]))
        a, b = pa[flat], pb[flat]
        # one end point on a u-directed grid edge (v integer, u fractional), the other on a v-directed one
        a_on_u = (a[:, v] == np.floor(a[:, v])) & (a[:, u] != np.floor(a[:, u]))
        a_on_v = (a[:, u] == np.floor(a[:, u])) & (a[:, v] != np.floor(a[:, v]))
        b_on_u = (b[:, v] == np.floor(b[:, v])) & (b[:, u] != np.floor(b[:, u]))
        b_on_v = (b[:, u] == np.floor(b[:, u])) & (b[:, v] != np.floor(b[:, v]))
        sel = (a_on_u & b_on_v) | (a_on_v & b_on_u)
        a, b = a[sel], b[sel]
        swap = ~(a_on_u & b_on_v)[sel]
        pu = np.where(swap[:, None], b, a)                    # the end point on the u-directed edge: its v coordinate is the corner's v
        pv = np.where(swap[:, None], a, b)                    # the end point on the v-directed edge: its u coordinate is the corner's u
        cu, cv, cp = pv[:, u].astype(np.int64), pu[:, v].astype(np.int64), pu[:, ax].astype(np.int64)
        u0, v0 = np.floor(pu[:, u]).astype(np.int64), np.floor(pv[:, v]).astype(np.int64)
        for i in range(cu.shape[0]):
            if not (u0[i] <= cu[i] <= u0[i] + 1 and v0[i] <= cv[i] <= v0[i] + 1) or u0[i] + 1 >= shape[u] or v0[i] + 1 >= shape[v]:
                continue                                       # end points in different cells: not a segment of one grid face
            idx = [0, 0, 0]
            idx[ax] = cp[i]

            def val(du, dv):
                idx[u], idx[v] = u0[i] + du, v0[i] + dv
                return vol[idx[0], idx[1], idx[2]]
            f00, f10, f01, f11 = val(0, 0), val(1, 0), val(0, 1), val(1, 1)
            if not np.isfinite([f00, f10, f01, f11]).all():
                continue
            i00, i10, i01, i11 = f00 > level, f10 > level, f01 > level, f11 > level
            if not (i00 == i11 and i10 == i01 and i00 != i10):
                continue                                       # not ambiguous
            det = (f00 - level) * (f11 - level) - (f10 - level) * (f01 - level)
            saddle_inside = det > 0 if i00 else det < 0
            corner_inside = bool(vol[tuple(_set(idx, u, cu[i], v, cv[i]))] > level)
            assert corner_inside != saddle_inside, ('ambiguous face (axis {}, plane {}, cell {},{}): the segment cuts off an {} corner but the saddle '
                                                    'value is {}'.format(ax, cp[i], u0[i], v0[i], 'inside' if corner_inside else 'outside',
                                                                         'inside' if saddle_inside else 'outside'))
            checked.add((ax, int(cp[i]), int(u0[i]), int(v0[i])))
    return len(checked)


def _set(idx, u, cu, v, cv):
    out = list(idx)
    out[u], out[v] = int(cu), int(cv)
    return out


def components_union_find(faces):
    """Label of the connected component of every face.  Two faces are adjacent iff they share an UNDIRECTED edge that belongs to EXACTLY TWO
    faces: source/base/mesh.py:27 builds its components from trimesh's `face_adjacency`, which pairs the faces of the edges that occur twice
    (`group_rows(edges_sorted, require_count=2)` [ext: trimesh is not in the image]) -- an edge shared by three or more faces (non-manifold)
    joins nothing, so pieces that touch only along such an edge stay separate components (rounds 3-5 joined all owners of an edge).
    Sequential union-find with path halving, no sparse-graph library."""
    faces = np.asarray(faces, dtype=np.int64)
    nf = faces.shape[0]
    parent = list(range(nf))

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x

    owners = {}
    for f in range(nf):
        a, b, c = faces[f]
        for u, v in ((a, b), (b, c), (c, a)):
            owners.setdefault((int(u), int(v)) if u < v else (int(v), int(u)), []).append(f)
    for fs in owners.values():
        if len(fs) != 2:
            continue
        ra, rb = find(fs[0]), find(fs[1])
        if ra != rb:
            parent[max(ra, rb)] = min(ra, rb)
    return np.array([find(f) for f in range(nf)], dtype=np.int64)


def clean_mesh_spec(verts, faces, min_component_faces=6, digits=8):
    """The clean-up as a set: a sorted array [F', 9] of the three corner POSITIONS (rounded to `digits`) of every surviving face, rotated so
    that the lexicographically smallest corner comes first (orientation preserved), plus the sorted surviving vertex positions."""
    verts = np.round(np.asarray(verts, dtype=np.float64), digits)
    faces = np.asarray(faces, dtype=np.int64)
    # merge by position
    uniq, inv = np.unique(verts, axis=0, return_inverse=True)
    f = inv.reshape(-1)[faces]
    f = f[(f[:, 0] != f[:, 1]) & (f[:, 1] != f[:, 2]) & (f[:, 0] != f[:, 2])]
    seen, keep = set(), []
    for i, t in enumerate(map(tuple, np.sort(f, axis=1))):                # duplicate faces: same three vertices in any order
        if t not in seen:
            seen.add(t)
            keep.append(i)
    f = f[keep]
    if f.shape[0] and min_component_faces is not None:
        lab = components_union_find(f)
        size = np.bincount(lab, minlength=f.shape[0])
        f = f[size[lab] > min_component_faces]
    return _canonical_faces(uniq, f), np.unique(uniq[np.unique(f.reshape(-1))], axis=0) if f.shape[0] else np.zeros((0, 3))


def _canonical_faces(verts, faces):
    if faces.shape[0] == 0:
        return np.zeros((0, 9))
    tri = verts[faces]                                                     # [F,3,3]
    # rotate each face so that its smallest corner (lexicographic) is first
    order = np.lexsort((tri[..., 2], tri[..., 1], tri[..., 0]), axis=1)[:, 0]
    idx = (order[:, None] + np.arange(3)[None, :]) % 3
    tri = np.take_along_axis(tri, idx[:, :, None], axis=1).reshape(-1, 9)
    return tri[np.lexsort(tri.T[::-1])]


def check_clean_mesh(verts_in, faces_in, verts_out, faces_out, min_component_faces=6, digits=8):
    """The product's cleaned mesh (verts_out, faces_out) must be exactly the specification's, as sets of oriented faces given by corner
    positions, with no unreferenced, duplicate-position vertices left."""
    want_faces, want_verts = clean_mesh_spec(verts_in, faces_in, min_component_faces, digits)
    vo = np.round(np.asarray(verts_out, dtype=np.float64), digits)
    fo = np.asarray(faces_out, dtype=np.int64)
    got_faces = _canonical_faces(vo, fo)
    assert got_faces.shape == want_faces.shape, 'cleaned mesh has {} faces, specification {}'.format(got_faces.shape[0], want_faces.shape[0])
    assert np.array_equal(got_faces, want_faces), 'cleaned faces differ from the specification'
    assert np.unique(vo, axis=0).shape[0] == vo.shape[0], 'vertices with equal positions survive the merge'
    if fo.shape[0]:
        assert np.unique(fo.reshape(-1)).shape[0] == vo.shape[0], 'unreferenced vertices survive'
    assert np.array_equal(np.unique(vo, axis=0), want_verts)
    return {'faces': int(fo.shape[0]), 'vertices': int(vo.shape[0])}


# ---------------------------------------------------------------------------------------------------------------------------------------------
# Interior ambiguity (round 5).  Reference: skimage.measure.marching_cubes (Lewiner) at source/poco_utils.py:95-96 resolves, beyond the ambiguous
# FACES, the cubes in which two same-sign groups of corners that no face joins are connected THROUGH the cube by the trilinear interpolant
# (Chernyaev 1995 "Marching Cubes 33", cases 4, 6, 7, 10, 12, 13; Lewiner et al. 2003 section 4, `test_interior`).  Two independent statements:
#   trilinear_corner_groups   NUMERICAL GROUND TRUTH: the trilinear interpolant of the cube sampled on a dense lattice, connected components of
#                             {F > level} and of {F <= level} inside the closed cube; which corners share a component
#   interior_corner_groups    the ANALYTIC test of the papers in its general form (plane sweep along every cube axis, the asymptotic decider of the
#                             plane at the height where A C - B D is extremal), restated here with plain scalar loops; nothing is shared with
#                             ppsurf_amd/mcubes.py
#   check_cube_topology       the mesh patch inside every cube has one boundary loop per pair of surface regions it separates, and consists of
#                             discs except for exactly one ANNULUS per pair of groups the interior test joins
# ---------------------------------------------------------------------------------------------------------------------------------------------
_CUBE_CORNERS = [(c & 1, (c >> 1) & 1, (c >> 2) & 1) for c in range(8)]          # corner c of a cube: offsets along x, y, z


def _partition(labels, members):
    """frozenset of frozensets: the corners in `members` grouped by label."""
    groups = {}
    for c in members:
        groups.setdefault(labels[c], set()).add(c)
    return frozenset(frozenset(g) for g in groups.values())


def trilinear_corner_groups(vals, level=0.0, n=64):
    """vals: the 8 corner values of one cube (corner c at offsets (c & 1, (c >> 1) & 1, (c >> 2) & 1)).  The trilinear interpolant is sampled at the
    (n + 1)^3 lattice points of the closed cube; lattice neighbours (6-connectivity) of equal side are joined.  -> (partition of the inside corners,
    partition of the outside corners) by connected component.  A connection thinner than 1 / n is missed: callers use cubes with a margin."""
    from scipy import ndimage
    t = np.linspace(0.0, 1.0, n + 1)
    x, y, z = np.meshgrid(t, t, t, indexing='ij')
    f = np.zeros_like(x)
    for c, (dx, dy, dz) in enumerate(_CUBE_CORNERS):
        f += vals[c] * (x if dx else 1 - x) * (y if dy else 1 - y) * (z if dz else 1 - z)
    inside = f > level
    lab_in, _ = ndimage.label(inside)
    lab_out, _ = ndimage.label(~inside)
    at = lambda lab, c: int(lab[_CUBE_CORNERS[c][0] * n, _CUBE_CORNERS[c][1] * n, _CUBE_CORNERS[c][2] * n])
    ins = [c for c in range(8) if vals[c] > level]
    outs = [c for c in range(8) if not vals[c] > level]
    return _partition({c: at(lab_in, c) for c in ins}, ins), _partition({c: at(lab_out, c) for c in outs}, outs)


def _union_find(n):
    parent = list(range(n))

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x

    def union(a, b):
        ra, rb = find(a), find(b)
        if ra != rb:
            parent[max(ra, rb)] = min(ra, rb)
    return find, union


def surface_corner_groups(vals, level=0.0):
    """Corner groups ON THE SURFACE of the cube: same-side corners joined along cube edges, and across an ambiguous face the diagonal pair the
    asymptotic decider joins (inside pair when the saddle value is inside, else the outside pair).  -> (inside partition, outside partition)."""
    s = [v > level for v in vals]
    find, union = _union_find(8)
    for a in range(8):
        for ax in range(3):
            b = a | (1 << ax)
            if b != a and s[a] == s[b]:
                union(a, b)
    for ax in range(3):
        u, v = (ax + 1) % 3, (ax + 2) % 3
        for side in (0, 1):
            c = lambda a, b: (side << ax) | (a << u) | (b << v)
            c00, c10, c01, c11 = c(0, 0), c(1, 0), c(0, 1), c(1, 1)
            if s[c00] == s[c11] and s[c10] == s[c01] and s[c00] != s[c10]:
                det = (vals[c00] - level) * (vals[c11] - level) - (vals[c10] - level) * (vals[c01] - level)
                saddle_inside = det > 0 if s[c00] else det < 0
                pair = (c00, c11) if s[c00] == saddle_inside else (c10, c01)
                union(*pair)
    lab = {c: find(c) for c in range(8)}
    return _partition(lab, [c for c in range(8) if s[c]]), _partition(lab, [c for c in range(8) if not s[c]])


def interior_corner_groups(vals, level=0.0):
    """surface_corner_groups + the interior test: for each side (inside: w = v - level, outside: w = level - v), each sweep axis and each diagonal
    pair (X, Y) of the four columns of the sweep (B, D the other two), all linear in the height t:  g(t) = X(t) Y(t) - B(t) D(t).  If g has a maximum
    at t* = -g1 / (2 g2) strictly inside (0, 1) with X(t*) > 0, Y(t*) > 0 and g(t*) > 0, the columns X and Y are joined inside the plane at height
    t* (asymptotic decider of that plane), and -- w being linear and positive at t* along each column -- so are the positive end corners of X and Y.
    -> (inside partition, outside partition)."""
    s = [v > level for v in vals]
    gin, gout = surface_corner_groups(vals, level)
    find, union = _union_find(8)
    for g in list(gin) + list(gout):
        g = sorted(g)
        for c in g[1:]:
            union(g[0], c)
    for side in (True, False):
        w = [(v - level) if side else (level - v) for v in vals]
        for ax in range(3):
            u, v_ = (ax + 1) % 3, (ax + 2) % 3
            col = lambda a, b: ((a << u) | (b << v_), (a << u) | (b << v_) | (1 << ax))
            for (xa, xb), (ya, yb), (ba, bb), (da, db) in (((0, 0), (1, 1), (1, 0), (0, 1)), ((1, 0), (0, 1), (0, 0), (1, 1))):
                X, Y, B, D = col(xa, xb), col(ya, yb), col(ba, bb), col(da, db)
                x0, x1, y0, y1, b0, b1, d0, d1 = w[X[0]], w[X[1]], w[Y[0]], w[Y[1]], w[B[0]], w[B[1]], w[D[0]], w[D[1]]
                g2 = (x1 - x0) * (y1 - y0) - (b1 - b0) * (d1 - d0)
                g1 = (y0 * (x1 - x0) + x0 * (y1 - y0)) - (d0 * (b1 - b0) + b0 * (d1 - d0))
                if not g2 < 0:
                    continue
                t = -g1 / (2 * g2)
                if not 0 < t < 1:
                    continue
                xt, yt, bt, dt = x0 + (x1 - x0) * t, y0 + (y1 - y0) * t, b0 + (b1 - b0) * t, d0 + (d1 - d0) * t
                if xt > 0 and yt > 0 and xt * yt - bt * dt > 0:
                    ex = [c for c in X if s[c] == side]
                    ey = [c for c in Y if s[c] == side]
                    if ex and ey:
                        union(ex[0], ey[0])
    lab = {c: find(c) for c in range(8)}
    return _partition(lab, [c for c in range(8) if s[c]]), _partition(lab, [c for c in range(8) if not s[c]])


def _patch_components(verts, f):
    """(euler characteristic, boundary loops) of every edge-connected component of the faces f (a 2-complex)."""
    lab = components_union_find(f)
    out = []
    for l in np.unique(lab):
        ff = f[lab == l]
        e = np.sort(np.concatenate([ff[:, [0, 1]], ff[:, [1, 2]], ff[:, [2, 0]]]), axis=1)
        ue, cnt = np.unique(e, axis=0, return_counts=True)
        assert cnt.max() <= 2, 'a non-manifold edge inside a cube'
        chi = np.unique(ff).shape[0] - ue.shape[0] + ff.shape[0]
        bnd = ue[cnt == 1]
        ids = {int(v): k for k, v in enumerate(np.unique(bnd))}
        find, union = _union_find(len(ids))
        for a, b in bnd:
            union(ids[int(a)], ids[int(b)])
        loops = len({find(k) for k in range(len(ids))})
        out.append((int(chi), int(loops)))
    return sorted(out)


def cube_patch_topology(verts, faces, origin):
    """The faces of the mesh that lie in the closed unit cube at integer `origin`, as a 2-complex: -> list of (euler characteristic, boundary loops)
    per edge-connected component.  A disc is (1, 1), an annulus (tube) (0, 2).  A face belongs to the cube its centroid lies in (marching-cubes
    faces never lie IN a cube face)."""
    verts = np.asarray(verts, dtype=np.float64)
    faces = np.asarray(faces, dtype=np.int64)
    cube = np.floor(verts[faces].mean(axis=1)).astype(np.int64)
    f = faces[(cube == np.asarray(origin, dtype=np.int64)).all(axis=1)]
    return _patch_components(verts, f) if f.shape[0] else []


def check_cube_topology(verts, faces, volume, level=0.0, cubes=None):
    """For every cube with eight finite corners that the surface crosses (or the listed `cubes`): with R = number of surface corner groups (both
    sides) and G = number of groups after the interior test, the mesh patch in the cube has R - 1 boundary loops in total, exactly R - G annuli
    (one tube per pair of groups joined through the cube), and discs otherwise.  A mesher that closes every loop with a disc FAILS in a cube whose
    interior test joins two groups.  Returns {'cubes': checked, 'multi_loop': cubes with more than one loop, 'tunnels': cubes with a tube}."""
    vol = np.asarray(volume, dtype=np.float64)
    verts = np.asarray(verts, dtype=np.float64)
    faces = np.asarray(faces, dtype=np.int64)
    nx, ny, nz = vol.shape
    checked = multi = tunnels = 0
    by_cube = {}
    if faces.shape[0]:
        cube = np.floor(verts[faces].mean(axis=1)).astype(np.int64)
        cid = (cube[:, 0] * ny + cube[:, 1]) * nz + cube[:, 2]
        order = np.argsort(cid, kind='stable')
        cs = cid[order]
        start = np.nonzero(np.concatenate([[True], cs[1:] != cs[:-1]]))[0]
        for a, b in zip(start, list(start[1:]) + [cs.shape[0]]):
            by_cube[int(cs[a])] = order[a:b]
    if cubes is None:
        # cubes the surface crosses: vectorised pre-selection, the per-cube work below is scalar
        corner = [vol[dx:nx - 1 + dx, dy:ny - 1 + dy, dz:nz - 1 + dz] for dx, dy, dz in _CUBE_CORNERS]
        fin = np.ones(corner[0].shape, dtype=bool)
        anyin = np.zeros(corner[0].shape, dtype=bool)
        allin = np.ones(corner[0].shape, dtype=bool)
        onlevel = np.zeros(corner[0].shape, dtype=bool)
        with np.errstate(invalid='ignore'):
            for v in corner:
                fin &= np.isfinite(v)
                anyin |= v > level
                allin &= v > level
                onlevel |= v == level
        cubes = [tuple(int(t) for t in c) for c in np.stack(np.nonzero(fin & anyin & ~allin & ~onlevel), axis=1)]
    for (x, y, z) in cubes:
        vals = [float(vol[x + dx, y + dy, z + dz]) for dx, dy, dz in _CUBE_CORNERS]
        if not np.isfinite(vals).all():
            continue
        s = [v > level for v in vals]
        if all(s) or not any(s) or any(v == level for v in vals):
            continue                                           # (a crossing ON a corner: the loop structure is degenerate there)
        sin, sout = surface_corner_groups(vals, level)
        r = len(sin) + len(sout)
        idx = by_cube.get((x * ny + y) * nz + z)
        assert idx is not None, 'cube {} is crossed by the surface but has no faces'.format((x, y, z))
        checked += 1
        if r == 2 and idx.shape[0] <= 4:
            continue                                           # one loop of <= 6 crossings closed by <= 4 faces: a disc (the manifold test covers it)
        iin, iout = interior_corner_groups(vals, level)
        g = len(iin) + len(iout)
        patch = _patch_components(verts, faces[idx])
        loops = sum(l for _, l in patch)
        assert loops == r - 1, 'cube {}: {} boundary loops for {} surface regions'.format((x, y, z), loops, r)
        want = sorted([(0, 2)] * (r - g) + [(1, 1)] * (r - 1 - 2 * (r - g)))
        assert patch == want, ('cube {}: the interior test joins {} pair(s) of corner groups (tube expected), the mesh patch has components '
                               '(euler characteristic, boundary loops) = {}, expected {}'.format((x, y, z), r - g, patch, want))
        multi += 1 if r > 2 else 0
        tunnels += 1 if r > g else 0
    return {'cubes': checked, 'multi_loop': multi, 'tunnels': tunnels}
