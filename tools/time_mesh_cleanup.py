"""Marching Cubes + the two clean-ups of a reconstruction at R = 257 (259^3 volume, +-4 band of a bumpy sphere, ~4 x 10^5 faces), device kernels.
    python tools/time_mesh_cleanup.py  -> ms per call (HIP kernels of csrc/pps_mc.hip / pps_mesh.hip; host tensors would take the torch form)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppsurf_amd import mcubes  # noqa: E402


def main():
    n = 259
    g = torch.arange(n, dtype=torch.float64, device='cuda:0')
    x, y, z = torch.meshgrid(g, g, g, indexing='ij')
    vol = 100.0 - torch.sqrt((x - 129.3) ** 2 + (y - 128.1) ** 2 + (z - 130.7) ** 2) + 3.0 * torch.sin(x * 0.21) * torch.cos(y * 0.17)
    vol[vol.abs() > 4.0] = float('nan')

    def timed(fn, reps=5):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3, out

    t_mc, (v, f) = timed(lambda: mcubes.marching_cubes_torch(vol, 0.0))
    v = v.to(torch.float32).to(torch.float64)
    t_c1, (v1, f1) = timed(lambda: mcubes.clean_mesh_torch(v, f, min_component_faces=6, welded=True, grid_coords=True))
    vm = v1 * 0.0039 - 0.5
    t_c2, _ = timed(lambda: mcubes.clean_mesh_torch(vm, f1, min_component_faces=6, welded=True, grid_coords=False))
    print('{} vertices, {} faces: marching cubes {:.2f} ms, clean-up in index space {:.2f} ms, clean-up after refinement {:.2f} ms'.format(
        v.shape[0], f.shape[0], t_mc, t_c1, t_c2))


if __name__ == '__main__':
    main()
