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
order, so they agree and the surface has no cracks.
INTERIOR ambiguities [round 5] (Chernyaev's / Lewiner's cases 4, 6, 7, 10, 12, 13: two same-sign groups of corners that no face joins may still be
connected THROUGH the cube by the trilinear interpolant) are resolved by the interior test of the same papers in its general form: sweep a plane
along a cube axis; in the plane at height t the function is bilinear with corner values A(t), B(t), C(t), D(t) linear in t, and two diagonal
corners A, C of one sign are joined in that plane iff A C - B D > 0 (the asymptotic decider again); g(t) = A C - B D is quadratic, so the groups
are connected iff for some axis g has an interior maximum t* = -g1 / (2 g2) in (0, 1) with g(t*) > 0 and A(t*), C(t*) of the groups' sign
(Lewiner et al. 2003, test_interior, cases 4 / 10: exactly these formulas).  A connected pair of loops is closed by a TUBE (annulus between the two
loops) instead of two discs.  Which pairs of loops can be joined, which (axis, diagonal) sweeps decide it, and the tube's triangles are derived
below like the rest of the table; the decision itself is data (`interior_connected`).  The specification and the numerical ground truth
(connected components of the densely sampled trilinear interpolant) are in oracle/mesh_oracle.py.
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


def _slice_columns(ax):
    """The four columns (cube edges along `ax`) of a sweep along axis ax, at (u, v) = (0,0), (1,0), (0,1), (1,1) with u = ax+1, v = ax+2 (mod 3):
    [(corner at t = 0, corner at t = 1)] * 4.  Diagonal pair 0 = columns 0, 3; pair 1 = columns 1, 2."""
    u, v = (ax + 1) % 3, (ax + 2) % 3
    return [((a << u) | (b << v), (a << u) | (b << v) | (1 << ax)) for b in (0, 1) for a in (0, 1)]


_SLICE_COLS = [_slice_columns(ax) for ax in range(3)]
_DIAG = ((0, 3, 1, 2), (1, 2, 0, 3))             # (X, Y, B, D) column numbers of the two diagonal pairs of a slice


def _surface_regions(case, dec):
    """Connectivity of the cube corners ON THE SURFACE of the cube: same-sign corners are joined along cube edges, and across an ambiguous face the
    inside diagonal pair is joined when the face's decision bit is set, the outside pair when it is clear.  -> region label per corner."""
    lab = list(range(8))

    def find(x):
        while lab[x] != x:
            lab[x] = lab[lab[x]]
            x = lab[x]
        return x

    ins = [(case >> c) & 1 for c in range(8)]
    for a, b in _EDGES:
        if ins[a] == ins[b]:
            lab[find(a)] = find(b)
    for f, (c00, c10, c01, c11) in enumerate(_FACE_UV):
        if ins[c00] == ins[c11] and ins[c10] == ins[c01] and ins[c00] != ins[c10]:
            joined_inside = (dec >> f) & 1
            pair = (c00, c11) if bool(ins[c00]) == bool(joined_inside) else (c10, c01)
            lab[find(pair[0])] = find(pair[1])
    return [find(c) for c in range(8)]


def _edge_mid(e):
    a, b = _EDGES[e]
    return (_CORNERS[a] + _CORNERS[b]) * 0.5


_TUBE_CACHE = {}


def _tube(loop_a, loop_b, edge_faces):
    key = (tuple(loop_a), tuple(loop_b))
    if key not in _TUBE_CACHE:
        _TUBE_CACHE[key] = _tube_search(loop_a, loop_b, edge_faces)
    return _TUBE_CACHE[key]


_PATHS = {}


def _lattice_paths(na, nb):
    """All orders of na A-steps and nb B-steps: bool [paths, na + nb] (True = A-step), cached."""
    import itertools
    key = (na, nb)
    if key not in _PATHS:
        n = na + nb
        combos = np.array(list(itertools.combinations(range(n), na)), dtype=np.int64)
        m = np.zeros((combos.shape[0], n), dtype=bool)
        m[np.arange(combos.shape[0])[:, None], combos] = True
        _PATHS[key] = m
    return _PATHS[key]


def _tube_search(loop_a, loop_b, edge_faces):
    """Triangles of an annulus between two directed loops (both as the table stores loops: inside on the left seen from outside the cube), with the
    winding of the disc triangles (loop edges are traversed backwards: normals towards lower values).
    The annulus is a cyclic sequence of BRIDGES a_i - b_j: from bridge (i, j) a step along A emits the triangle (a_i+1, a_i, b_j) and leads to
    (i + 1, j), a step along B emits (b_j, b_j-1, a_i) and leads to (i, j - 1): the loops run in opposite senses around the tube.  All start
    bridges (a_0, b_j0) and all orders of the na + nb steps are enumerated (vectorised); a tube may use every bridge once.
    A bridge whose two cube edges lie in one cube face would be a chord flush with that face (the neighbouring cube sees the same two vertices, and
    the face decider's segments would be contradicted): tubes without such bridges are preferred, shortest summed squared bridge length (between
    edge midpoints) first.  For most loop pairs every tube has flush bridges; then the run of consecutive bridges that contains all of them is cut
    out, together with the triangles on both sides of it, and the hole -- a polygon bounded by two clean bridges and the loop edges between them --
    is closed by a fan around ONE extra vertex inside the cube (id CENTER, at the mean of the polygon's corners, like the centre vertex of the
    long disc loops): the shortest such run is taken.  -> (triangles, with the fan first, or None if no tube exists)."""
    na, nb = len(loop_a), len(loop_b)
    n = na + nb
    mid_a = np.array([_edge_mid(e) for e in loop_a])
    mid_b = np.array([_edge_mid(e) for e in loop_b])
    flush_ab = np.array([[1 if (edge_faces[ea] & edge_faces[eb]) else 0 for eb in loop_b] for ea in loop_a], dtype=np.int64)
    len_ab = ((mid_a[:, None, :] - mid_b[None, :, :]) ** 2).sum(-1)
    paths = _lattice_paths(na, nb)                                       # [P, n]
    ia = np.concatenate([np.zeros((paths.shape[0], 1), dtype=np.int64), np.cumsum(paths, axis=1)[:, :-1]], axis=1)           # A-steps before step t
    kb = np.concatenate([np.zeros((paths.shape[0], 1), dtype=np.int64), np.cumsum(~paths, axis=1)[:, :-1]], axis=1)
    best = None
    for j0 in range(nb):
        ai, bj = ia % na, (j0 - kb) % nb                                 # bridge before step t
        bid = np.sort(ai * nb + bj, axis=1)
        ok = (bid[:, 1:] != bid[:, :-1]).all(axis=1)                     # every bridge once (a path that spends all its A-steps in one run is a cone)
        fl = flush_ab[ai, bj].astype(bool)                               # [P, n]
        length = len_ab[ai, bj]
        # shortest cyclic window of bridges that contains every flush one = n - (longest cyclic run of clean bridges); the run must hold >= 2 bridges
        clean = ~fl
        dbl = np.concatenate([clean, clean], axis=1)
        run = np.zeros(dbl.shape[0], dtype=np.int64)
        best_run = np.zeros(dbl.shape[0], dtype=np.int64)
        best_end = np.zeros(dbl.shape[0], dtype=np.int64)
        for t in range(2 * n):
            run = np.where(dbl[:, t], run + 1, 0)
            better = run > best_run
            best_run = np.where(better, run, best_run)
            best_end = np.where(better, t, best_end)
        full = best_run >= n                                             # no flush bridge at all
        best_run = np.minimum(best_run, n)
        window = n - best_run                                            # bridges cut out (0: a tube without flush bridges)
        # the triangles that stay (between consecutive bridges of the clean run) must include a step along A AND a step along B: otherwise the
        # piece that is cut out contains a whole loop and the polygon around it touches itself
        ca = np.cumsum(np.concatenate([paths, paths], axis=1), axis=1)
        ca = np.concatenate([np.zeros((ca.shape[0], 1), dtype=np.int64), ca], axis=1)            # ca[:, t] = A-steps among steps 0 .. t - 1 (doubled)
        first_d = best_end - best_run + 1
        rows_ = np.arange(paths.shape[0])
        kept_a = ca[rows_, np.maximum(first_d + best_run - 1, 0)] - ca[rows_, np.maximum(first_d, 0)]
        kept = best_run - 1
        ok &= full | ((kept_a >= 1) & (kept - kept_a >= 1))
        if not ok.any():
            continue
        reg_len = np.where(fl.any(axis=1), 0.0, length.sum(axis=1))      # ranking inside the clean tubes: summed bridge length
        cand = np.nonzero(ok)[0]
        keyed = sorted((int(window[p]), round(float(reg_len[p] if window[p] == 0 else length[p].sum()), 9), j0, int(p)) for p in cand)
        if best is None or keyed[0] < best[0]:
            p = keyed[0][3]
            best = (keyed[0], j0, paths[p].copy(), int(best_end[p]) % n, int(best_run[p]), int(window[p]))
    if best is None:
        # No tube whose flush bridges fit into one simple polygon (a triangle loop against a hexagon: the six clean bridges come in three isolated
        # pairs).  Lewiner's table for the same configuration (case 7.4.2) has chords in cube faces too: take the tube with the FEWEST flush bridges.
        # Such a chord is an interior edge of this cube's patch (both its triangles lie in this cube), not a face segment.
        for j0 in range(nb):
            ai, bj = ia % na, (j0 - kb) % nb
            bid = np.sort(ai * nb + bj, axis=1)
            ok = (bid[:, 1:] != bid[:, :-1]).all(axis=1)
            nfl = flush_ab[ai, bj].sum(axis=1)
            length = len_ab[ai, bj].sum(axis=1)
            for p in np.nonzero(ok)[0]:
                key = (int(nfl[p]), round(float(length[p]), 9), j0, int(p))
                if best is None or key < best[0]:
                    best = (key, j0, paths[p].copy(), 0, n, 0)
        if best is None:
            return None, 1
    _, j0, steps, run_end, run_len, window = best
    # triangles in step order; triangle t lies between bridge t and bridge t + 1 (cyclic)
    tris, i, j = [], 0, j0
    for t in range(n):
        ea, eb = loop_a[i % na], loop_b[j % nb]
        if steps[t]:
            tris.append((loop_a[(i + 1) % na], ea, eb))
            i += 1
        else:
            tris.append((eb, loop_b[(j - 1) % nb], ea))
            j -= 1
    if window == 0:
        return tris, 0
    # clean run = bridges run_end - run_len + 1 .. run_end (cyclic); the triangles strictly between two clean bridges of the run stay, the rest
    # (window + 1 triangles, from bridge run_end to bridge run_end - run_len + 1 going forward) is replaced by the fan
    first = (run_end - run_len + 1) % n
    keep = [(first + d) % n for d in range(run_len - 1)]                 # triangle t sits between bridges t and t + 1
    cut = [t for t in range(n) if t not in keep]
    edges = {}
    for t in cut:
        x, y, z = tris[t]
        for u, v in ((x, y), (y, z), (z, x)):
            if (v, u) in edges:
                del edges[(v, u)]
            else:
                edges[(u, v)] = True
    nxt = {u: v for u, v in edges}
    assert len(nxt) == len(edges), 'the polygon around the flush bridges is not simple'
    start = min(nxt)
    cyc, e = [], start
    while True:
        cyc.append(e)
        e = nxt[e]
        if e == start:
            break
    assert len(cyc) == len(edges)
    fan = [(CENTER, cyc[q], cyc[(q + 1) % len(cyc)]) for q in range(len(cyc))]
    return fan + [tris[t] for t in keep], 0


def _build_table():
    """-> (tri int8 [R, W, 3] cube-edge ids (-1 = none), ntri uint8 [R], amb uint8 [256], tun_index int32 [256 * 64, 2], tun_cand int32 [C, 3]).
    Rows 0 .. 16383: row case * 64 + dec, dec = one bit per face: 1 = the inside corners of that (ambiguous) face are JOINED through the face; every
    loop closed by a disc.  Rows whose dec has bits outside amb[case] are copies of the row with those bits cleared, so a lookup may mask or not.
    tun_index[row] = (first, count) into tun_cand: the pairs of loops of that row that the interior test may join; tun_cand[c] = (sign: 1 inside /
    0 outside groups, mask: bit 2 * axis + diagonal = this sweep decides the pair, alternative row >= 16384: the same cube with a tube between the
    two loops).  The first candidate (in list order) whose test succeeds replaces the row."""
    faces = _face_cycles()
    edge_faces = {i: frozenset(fi for fi, cyc in enumerate(faces) if a in cyc and b in cyc) for i, (a, b) in enumerate(_EDGES)}
    rows, cands = {}, {}
    amb = np.zeros(256, dtype=np.uint8)
    extra = []                                          # alternative rows (tubes), appended behind the 16384 regular ones
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
            loops, seen = [], set()
            for start in sorted(nxt):
                if start in seen:
                    continue
                loop, e = [], start
                while e not in seen:
                    seen.add(e)
                    loop.append(e)
                    e = nxt[e]
                loops.append(loop)
            discs = [[(a, c, b) for a, b, c in _triangulate(loop, edge_faces)] for loop in loops]      # winding: normals point towards LOWER values

            def assemble(pieces):
                # a fan around the centre vertex (at most one per cube: such loops have >= 8 of the 12 edges) is listed FIRST in its row: the
                # kernels read "this cube has a centre vertex" off the row's first entry
                pieces = sorted(pieces, key=lambda t: 0 if t[0][0] == CENTER else 1)
                assert sum(1 for t in pieces if t[0][0] == CENTER) <= 1
                return [t for lp in pieces for t in lp]

            rows[(case, dec)] = assemble(discs)
            # ---- interior ambiguity: pairs of loops the interior test may join by a tube --------------------------------------------------------
            if len(loops) < 2:
                continue
            region = _surface_regions(case, dec)
            sides = []                                  # per loop: (inside region, outside region) it separates
            for loop in loops:
                rin = {region[c] for e in loop for c in _EDGES[e] if inside[c]}
                rout = {region[c] for e in loop for c in _EDGES[e] if not inside[c]}
                assert len(rin) == 1 and len(rout) == 1, (case, dec)
                sides.append((rin.pop(), rout.pop()))
            lst = []
            for sign in (1, 0):                         # inside groups first
                mine, other = (0, 1) if sign else (1, 0)
                for la in range(len(loops)):
                    for lb in range(la + 1, len(loops)):
                        # two DIFFERENT groups of this sign with one common neighbour region between the loops
                        if sides[la][mine] == sides[lb][mine] or sides[la][other] != sides[lb][other]:
                            continue
                        grp_a = {c for c in range(8) if region[c] == sides[la][mine]}
                        grp_b = {c for c in range(8) if region[c] == sides[lb][mine]}
                        mask = 0
                        for ax in range(3):
                            cols = _SLICE_COLS[ax]
                            for dg, (x, y, _b, _d) in enumerate(_DIAG):
                                cx, cy = set(cols[x]), set(cols[y])
                                if (cx & grp_a and cy & grp_b) or (cx & grp_b and cy & grp_a):
                                    mask |= 1 << (2 * ax + dg)
                        if mask == 0:
                            continue
                        tube, flush = _tube(loops[la], loops[lb], edge_faces)
                        if flush:
                            continue                    # (does not occur: asserted by tests/test_mesh_oracle.py on the whole table)
                        pieces = [tube] + [d for k, d in enumerate(discs) if k not in (la, lb)]
                        extra.append(assemble(pieces))
                        lst.append((sign, mask, 256 * 64 + len(extra) - 1))
            if lst:
                cands[(case, dec)] = lst
    width = max(len(t) for t in list(rows.values()) + extra)
    nrow = 256 * 64 + len(extra)
    tri = np.full((nrow, width, 3), -1, dtype=np.int8)
    ntri = np.zeros(nrow, dtype=np.uint8)
    tun_index = np.zeros((256 * 64, 2), dtype=np.int32)
    tun_cand = []
    for case in range(256):
        first = {}
        for dec in range(64):
            key = (case, dec & int(amb[case]))
            t = rows[key]
            ntri[case * 64 + dec] = len(t)
            if t:
                tri[case * 64 + dec, :len(t)] = np.array(t, dtype=np.int8)
            if key in cands:
                if key not in first:
                    first[key] = len(tun_cand)
                    tun_cand += cands[key]
                tun_index[case * 64 + dec] = (first[key], len(cands[key]))
    for k, t in enumerate(extra):
        ntri[256 * 64 + k] = len(t)
        tri[256 * 64 + k, :len(t)] = np.array(t, dtype=np.int8)
    return tri, ntri, amb, tun_index, np.array(tun_cand, dtype=np.int32).reshape(-1, 3)


_TRI_TABLE, _NTRI, _AMB, _TUN_INDEX, _TUN_CAND = _build_table()
N_BASE_ROWS = 256 * 64
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


def interior_sweep_connected(corner_vals, level, sign, ax, dg):
    """One sweep of the interior test, vectorised: corner_vals = 8 arrays (cube corners), sign 1: groups of inside corners (v - level), 0: of outside
    corners (level - v); plane orthogonal to axis `ax`, diagonal pair `dg` of the plane's four columns (_SLICE_COLS, _DIAG).  With X, Y the diagonal
    columns and B, D the other two, each linear in the height t:  g(t) = X Y - B D = g2 t^2 + g1 t + g0.  True where g has an interior maximum
    t* = -g1 / (2 g2) in (0, 1) (g2 < 0) with X(t*) > 0, Y(t*) > 0 and g(t*) > 0: the two columns are joined inside the plane at height t* (the
    asymptotic decider of that plane), hence their groups through the cube.  Same operations in the same order as the kernel
    (csrc/pps_mc.hip::sweep_connected; no fused multiply-add on either side)."""
    cols = _SLICE_COLS[ax]
    x, y, b, d = _DIAG[dg]
    sg = 1.0 if sign else -1.0
    val = lambda c: sg * (corner_vals[c] - level)
    x0, x1, y0, y1 = val(cols[x][0]), val(cols[x][1]), val(cols[y][0]), val(cols[y][1])
    b0, b1, d0, d1 = val(cols[b][0]), val(cols[b][1]), val(cols[d][0]), val(cols[d][1])
    dx, dy, db, dd = x1 - x0, y1 - y0, b1 - b0, d1 - d0
    g2 = dx * dy - db * dd
    g1 = (y0 * dx + x0 * dy) - (d0 * db + b0 * dd)
    with np.errstate(divide='ignore', invalid='ignore'):
        t = -g1 / (2.0 * g2)
        xt, yt, bt, dt = x0 + dx * t, y0 + dy * t, b0 + db * t, d0 + dd * t
        return (g2 < 0.0) & (t > 0.0) & (t < 1.0) & (xt > 0.0) & (yt > 0.0) & (xt * yt - bt * dt > 0.0)


def interior_rows(rows, corner_vals, level):
    """rows: int64 table rows (case * 64 + dec) of some cubes, corner_vals: their 8 corner value arrays -> the rows after the interior test: the first
    candidate pair of loops (tun_cand order) that one of its sweeps finds connected replaces the row by its tube row."""
    rows = rows.copy()
    first, count = _TUN_INDEX[rows, 0], _TUN_INDEX[rows, 1]
    todo = np.nonzero(count > 0)[0]
    if todo.size == 0:
        return rows
    sub = [v[todo] for v in corner_vals]
    done = np.zeros(todo.size, dtype=bool)
    for k in range(int(count.max())):
        has = (count[todo] > k) & ~done
        if not has.any():
            break
        cand = _TUN_CAND[np.where(has, first[todo] + k, 0)]
        hit = np.zeros(todo.size, dtype=bool)
        for sign in (1, 0):
            for ax in range(3):
                for dg in range(2):
                    sel = has & (cand[:, 0] == sign) & (((cand[:, 1] >> (2 * ax + dg)) & 1) == 1)
                    if sel.any():
                        hit |= sel & interior_sweep_connected(sub, level, sign, ax, dg)
        rows[todo[hit]] = cand[hit, 2]
        done |= hit
    return rows


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
    cvals = [v[cx, cy, cz] for v in corner_vals]
    rows = interior_rows(case[cx, cy, cz] * 64 + dec, cvals, level)  # interior ambiguity: a connected pair of loops gets a tube (its own table row)
    tris = _TRI_TABLE[rows].astype(np.int64)                         # [n, W, 3] local edge ids
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
        a, b = _two_owner_pairs((es[1:] == es[:-1]).all(axis=1), ow, np)
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
    """(tri int8 [rows, W, 3], ntri uint8 [rows], amb uint8 [256], tun_index int32 [16384, 2], tun_cand int32 [C, 3]) on `dev` (uploaded once)."""
    import torch
    t = _DEV_TABLES.get(str(dev))
    if t is None:
        t = (torch.from_numpy(_TRI_TABLE).to(dev).contiguous(), torch.from_numpy(_NTRI).to(dev), torch.from_numpy(_AMB).to(dev),
             torch.from_numpy(_TUN_INDEX).to(dev).contiguous(), torch.from_numpy(_TUN_CAND).to(dev).contiguous())
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


def _two_owner_pairs(same, ow, xp):
    """(a, b): the two faces of every edge that belongs to EXACTLY two faces.  `same[i]` = sorted edge i + 1 equals sorted edge i, `ow` = the owning
    face of every sorted edge; xp = numpy or torch.  An edge with three or more owners (non-manifold) pairs nothing: trimesh's face_adjacency,
    which source/base/mesh.py:27 builds its components from, keeps the edges that occur twice (`require_count=2`)."""
    n = ow.shape[0]
    if n < 2:
        return ow[:0], ow[:0]
    if xp is np:
        f = np.zeros(1, dtype=bool)
        prev, nxt = np.concatenate([f, same]), np.concatenate([same, f])             # edge i equals edge i - 1 / edge i + 1
        after = np.concatenate([nxt[1:], f])                                          # edge i + 1 equals edge i + 2
    else:
        f = xp.zeros(1, dtype=xp.bool, device=same.device)
        prev, nxt = xp.cat([f, same]), xp.cat([same, f])
        after = xp.cat([nxt[1:], f])
    first = ~prev & nxt & ~after                                                      # first edge of a run of exactly two
    idx = xp.nonzero(first)[0] if xp is np else xp.nonzero(first)[:, 0]
    return ow[idx], ow[idx + 1]


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
    if welded and grid_coords and verts.is_cuda and verts.dtype == torch.float64 and faces.dtype == torch.int64:
        # device meshes: the corner weld and the face filter are HIP kernels (csrc/pps_mesh.hip); same result as the torch form below, which host
        # tensors (the CPU suite) still take
        from . import ops
        skip = True
        remap, hot, merged = ops.mesh_corner_weld(verts, digits)
        if merged:
            faces = remap[faces]
            faces = faces[ops.mesh_face_filter(faces, hot)]
    elif welded and grid_coords:
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
    if faces.shape[0] and min_component_faces is not None and faces.is_cuda and faces.dtype == torch.int64 and int(min_component_faces) <= 32:
        from . import ops
        faces = faces[~ops.mesh_small_components(faces, nv, int(min_component_faces))]       # HIP kernels (csrc/pps_mesh.hip)
    elif faces.shape[0] and min_component_faces is not None:
        nf = faces.shape[0]
        e = torch.cat([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]])
        e = torch.sort(e, dim=1)[0]
        ekey = e[:, 0] * nv + e[:, 1]
        owner = torch.arange(nf, device=dev).repeat(3)
        skey, order = torch.sort(ekey, stable=True)
        ow = owner[order]
        a, b = _two_owner_pairs(skey[1:] == skey[:-1], ow, torch)
        faces = faces[~_small_component_faces(a, b, nf, int(min_component_faces))]
    used = torch.zeros(verts.shape[0], dtype=torch.bool, device=dev)           # compaction of the referenced vertices, order kept
    used[faces.reshape(-1)] = True
    remap = torch.cumsum(used, 0) - 1
    return verts[used], remap[faces]
