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
  check_marching_cubes      vertex multiset == edge_crossings; every face inside ONE grid cube; the faces form a closed, consistently
                            oriented 2-manifold (every directed edge once, its reverse once) wherever the surface does not run into unseen
                            voxels; normals point from values > level towards values <= level
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
    # 1. vertex multiset
    assert verts.shape[0] == want.shape[0], 'vertex count {} != {} level crossings on edges of finite cubes'.format(verts.shape[0], want.shape[0])
    got = verts[np.lexsort((verts[:, 2], verts[:, 1], verts[:, 0]))]
    if want.shape[0]:
        # lexicographic order is not stable under rounding noise in x: compare through nearest partners of equal rank in a coarser sort
        err = np.abs(got - want).max()
        if err > atol:
            key = lambda a: np.lexsort((np.round(a[:, 2], 6), np.round(a[:, 1], 6), np.round(a[:, 0], 6)))
            err = np.abs(verts[key(verts)] - want[key(want)]).max()
        assert err <= atol, 'vertex positions differ from the linear-interpolation crossings by {}'.format(err)
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
    return {'vertices': int(verts.shape[0]), 'faces': int(faces.shape[0]), 'boundary_edges': nb}


def components_union_find(faces):
    """Label of the connected component of every face, components joined across shared UNDIRECTED edges (trimesh face_adjacency,
    source/base/mesh.py:27) -- sequential union-find with path halving, no sparse-graph library."""
    faces = np.asarray(faces, dtype=np.int64)
    nf = faces.shape[0]
    parent = list(range(nf))

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x

    owner = {}
    for f in range(nf):
        a, b, c = faces[f]
        for u, v in ((a, b), (b, c), (c, a)):
            k = (u, v) if u < v else (v, u)
            g = owner.get(k)
            if g is None:
                owner[k] = f
            else:
                ra, rb = find(f), find(g)
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
