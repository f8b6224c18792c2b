"""Small REAL-data fixture: four shapes of the reference's datasets/abc_minimal (MIT-licensed data shipped with the reference),
clouds sub-sampled to at most 15000 points (seeded) and re-written with this package's PLY writer, query points / signed distances
copied as they are (float32 .npy), set lists.   python tests/golden/make_abc_fixture.py   (build container only)"""
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from ppsurf_amd import data, meshio  # noqa: E402

SRC = '/root/reference/datasets/abc_minimal'
DST = os.path.join(HERE, 'abc_mini4')


def main():
    train = data.read_shape_list(os.path.join(SRC, 'trainset.txt'))[:3]
    test = data.read_shape_list(os.path.join(SRC, 'testset.txt'))[:1]
    rng = np.random.default_rng(0)
    for name in train + test:
        pts = meshio.read_ply_vertices(os.path.join(SRC, '04_pts_vis', name + '.xyz.ply'))[:, :3]
        sel = np.sort(rng.choice(pts.shape[0], min(15000, pts.shape[0]), replace=False))
        meshio.write_ply_points(os.path.join(DST, '04_pts_vis', name + '.xyz.ply'), pts[sel])
        for sub in ('05_query_pts', '05_query_dist'):
            os.makedirs(os.path.join(DST, sub), exist_ok=True)
            shutil.copyfile(os.path.join(SRC, sub, name + '.ply.npy'), os.path.join(DST, sub, name + '.ply.npy'))
    for fname, names in (('trainset.txt', train), ('valset.txt', test), ('testset.txt', test + train[:1])):
        with open(os.path.join(DST, fname), 'w') as f:
            f.write('\n'.join(names) + '\n')
    print(sum(os.path.getsize(os.path.join(r, f)) for r, _, fs in os.walk(DST) for f in fs) / 1e6, 'MB')


if __name__ == '__main__':
    main()
