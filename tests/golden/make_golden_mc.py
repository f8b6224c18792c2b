"""Fixture for the interior ambiguity of Marching Cubes (tests/test_mesh_oracle.py): cubes in which the trilinear interpolant joins two same-side
groups of corners THROUGH the cube although no cube face joins them, found by random search and DENSE SAMPLING (96^3 lattice, connected components,
oracle/mesh_oracle.py::trilinear_corner_groups) -- no analytic test and no triangle table is involved in producing the expected partitions.
Three cubes per number of inside corners (2 .. 6: Chernyaev's cases 4, 6 / 7, 10 / 12 / 13 and their complements).

    python tests/golden/make_golden_mc.py        ->  tests/golden/mc_tunnel_cubes.npz  (vals [n,8], inside_label [n,8], outside_label [n,8]; -1 = other side)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from oracle import mesh_oracle as M          # noqa: E402


def main():
    rng = np.random.default_rng(2025)
    per_count = {k: [] for k in (2, 3, 4, 5, 6)}
    while any(len(v) < 3 for v in per_count.values()):
        vals = np.round(rng.standard_normal(8), 3)
        if np.abs(vals).min() < 0.2:
            continue
        k = int((vals > 0).sum())
        if k not in per_count or len(per_count[k]) >= 3:
            continue
        sin, sout = M.surface_corner_groups(vals)
        if len(sin) + len(sout) < 3:
            continue
        coarse = M.trilinear_corner_groups(vals, n=32)
        if coarse == (sin, sout):
            continue
        fine = M.trilinear_corner_groups(vals, n=96)
        if fine != coarse:
            continue                                           # not resolved robustly: skip
        per_count[k].append((vals, fine))
    vals_all, lin, lout = [], [], []
    for k in sorted(per_count):
        for vals, (pin, pout) in per_count[k]:
            a, b = -np.ones(8, dtype=np.int64), -np.ones(8, dtype=np.int64)
            for g, grp in enumerate(sorted(pin, key=min)):
                a[list(grp)] = g
            for g, grp in enumerate(sorted(pout, key=min)):
                b[list(grp)] = g
            vals_all.append(vals); lin.append(a); lout.append(b)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'mc_tunnel_cubes.npz')
    np.savez(out, vals=np.array(vals_all), inside_label=np.array(lin), outside_label=np.array(lout))
    print(out, len(vals_all))


if __name__ == '__main__':
    main()
