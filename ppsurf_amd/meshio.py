"""Minimal PLY / XYZ / NPY IO for the predict path (the reference uses trimesh, absent here).

Point clouds of datasets/abc_minimal are binary little-endian PLY written by trimesh
(`element vertex n`, `property float x/y/z`, optional normals, `element face 0`); meshes are written the same way
(source/poco_model.py:269 `mesh.export`).  Replaces the IO part of source/occupancy_data_module.py:174-225.
"""
import os

import numpy as np

_PLY_TYPES = {'char': 'i1', 'uchar': 'u1', 'short': 'i2', 'ushort': 'u2', 'int': 'i4', 'uint': 'u4', 'float': 'f4', 'double': 'f8',
              'int8': 'i1', 'uint8': 'u1', 'int16': 'i2', 'uint16': 'u2', 'int32': 'i4', 'uint32': 'u4', 'float32': 'f4', 'float64': 'f8'}


def read_ply_vertices(path):
    """Vertex x,y,z (+ nx,ny,nz if present) of an ascii or binary PLY -> float array [n, 3|6]."""
    with open(path, 'rb') as f:
        if f.readline().strip() != b'ply':
            raise ValueError('not a PLY file: {}'.format(path))
        fmt, nvert, props, in_vertex = None, 0, [], False
        while True:
            line = f.readline()
            if not line:
                raise ValueError('unterminated PLY header: {}'.format(path))
            tok = line.decode('ascii', 'replace').split()
            if not tok:
                continue
            if tok[0] == 'format':
                fmt = tok[1]
            elif tok[0] == 'element':
                in_vertex = tok[1] == 'vertex'
                if in_vertex:
                    nvert = int(tok[2])
            elif tok[0] == 'property' and in_vertex:
                if tok[1] == 'list':
                    raise ValueError('list property on vertices is not supported')
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == 'end_header':
                break
        if fmt == 'ascii':
            data = np.loadtxt(f, max_rows=nvert, ndmin=2)
            cols = {name: data[:, i] for i, (name, _) in enumerate(props)}
        else:
            end = '<' if fmt == 'binary_little_endian' else '>'
            rec = np.frombuffer(f.read(nvert * sum(np.dtype(t).itemsize for _, t in props)),
                                dtype=np.dtype([(n, end + t) for n, t in props]), count=nvert)
            cols = {name: rec[name] for name, _ in props}
    names = ['x', 'y', 'z'] + (['nx', 'ny', 'nz'] if all(k in cols for k in ('nx', 'ny', 'nz')) else [])
    return np.stack([np.asarray(cols[k], dtype=np.float64) for k in names], axis=1)


def load_pts(pts_file: str) -> np.ndarray:
    """source/occupancy_data_module.py:174-225 for the formats that need no third-party package."""
    ext = os.path.splitext(pts_file)[1].lower()
    if ext == '.npy':
        return np.load(pts_file)
    if ext == '.npz':
        return np.load(pts_file)['arr_0']
    if ext == '.xyz':
        return np.loadtxt(pts_file, ndmin=2)
    if ext == '.ply':
        return read_ply_vertices(pts_file)
    raise ValueError('Unknown point cloud type: {}'.format(pts_file))


def write_ply_mesh(path, verts: np.ndarray, faces: np.ndarray):
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    verts = np.asarray(verts, dtype='<f4')
    faces = np.asarray(faces, dtype='<i4')
    header = ('ply\nformat binary_little_endian 1.0\ncomment ppsurf_amd\nelement vertex {}\nproperty float x\nproperty float y\n'
              'property float z\nelement face {}\nproperty list uchar int vertex_indices\nend_header\n').format(verts.shape[0], faces.shape[0])
    rec = np.empty(faces.shape[0], dtype=[('n', 'u1'), ('v', '<i4', (3,))])
    rec['n'] = 3
    rec['v'] = faces
    with open(path, 'wb') as f:
        f.write(header.encode('ascii'))
        f.write(verts.tobytes())
        f.write(rec.tobytes())


def write_ply_points(path, pts: np.ndarray):
    """Binary little-endian PLY with float x/y/z vertices and zero faces (layout of datasets/*/04_pts_vis/*.xyz.ply)."""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    pts = np.asarray(pts, dtype='<f4')
    header = ('ply\nformat binary_little_endian 1.0\ncomment ppsurf_amd\nelement vertex {}\nproperty float x\nproperty float y\n'
              'property float z\nelement face 0\nproperty list uchar int vertex_indices\nend_header\n').format(pts.shape[0])
    with open(path, 'wb') as f:
        f.write(header.encode('ascii'))
        f.write(pts.tobytes())
