"""Stage timing of one full reconstruction (latent loop, region growing, MC, refinement) on one GPU (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
from golden_util import filled_sd
from source.ppsurf_model import PPSurfModel
from ppsurf_amd import reconstruct, mcubes
from ppsurf_amd.synthetic import make_cloud

N = int(os.environ.get('N', 100000)); R = int(os.environ.get('R', 129)); ITERS = int(os.environ.get('ITERS', 10))
DEV = 'cuda:0'
model = PPSurfModel(pointnet_latent_size=256, output_names=['imp_surf_sign'], in_channels=3, out_channels=2, k=64, lambda_l1=0.0,
                    debug=False, in_file='x.npy', results_dir='/tmp/res', padding_factor=0.05, name='t', network_latent_size=256,
                    gen_subsample_manifold_iter=ITERS, gen_subsample_manifold=10000, gen_resolution_global=R, num_pts_local=50,
                    rec_batch_size=50000, gen_refine_iter=10, workers=1)
model.network.load_state_dict(filled_sd('', key='ppsurf'))
model = model.to(DEV).eval()
cloud = make_cloud(N, seed=42)
pts_cf = torch.from_numpy(cloud.T.copy()).to(DEV)
def sync(): torch.cuda.synchronize(); return time.time()
t0 = sync(); lat = model.encode_latents(pts_cf); t1 = sync()
print('latent loop: {:.3f} s for N={} ({} x {} passes)'.format(t1 - t0, N, ITERS, -(-N // 10000)), flush=True)
# one encoder pass split
from ppsurf_amd import spatial
ids = torch.randperm(N, device=DEV)[:10000]
d = {'pts': pts_cf[:, ids].unsqueeze(0)}
t0 = sync(); sp = spatial.get_fkaconv_ids(d); t1 = sync(); d.update(sp); model.network.encoder.forward_point_major(d, 0); t2 = sync()
print('  one pass: ids {:.2f} ms, encoder {:.2f} ms'.format((t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
shape = {'pts': pts_cf.unsqueeze(0), 'latents': lat.t().unsqueeze(0)}
field = reconstruct.OccupancyField(model.network, shape, torch.from_numpy(cloud).unsqueeze(0), 50000, 50)
bmin, bmax = cloud.min(), cloud.max(); step = (bmax - bmin) / (R - 1)
pts_ids = torch.from_numpy(((cloud - bmin) / step + 1).astype(np.int32).astype(np.int64)).to(DEV)
t0 = sync(); vol = reconstruct.create_volume(field, pts_ids, R, step, bmin - step); t1 = sync()
print('region growing: {:.3f} s, {} queries -> {:.3e} q/s'.format(t1 - t0, field.n_queries, field.n_queries / (t1 - t0)), flush=True)
v = vol.cpu().numpy(); t2 = time.time()
frac_pos = float((v[~np.isnan(v)] > 0).mean())
verts, faces = mcubes.marching_cubes(v, 0.0); t3 = time.time()
verts, faces = mcubes.clean_mesh(verts, faces); t4 = time.time()
print('volume D2H {:.3f} s, MC {:.3f} s ({} verts, {} faces), clean {:.3f} s; positive fraction {:.3f}'.format(t2 - t1, t3 - t2, verts.shape[0], faces.shape[0], t4 - t3, frac_pos), flush=True)
nq0 = field.n_queries
q = torch.from_numpy((verts * step + (bmin - step)).astype(np.float32)).to(DEV)
t0 = sync()
for _ in range(10): field(q)
t1 = sync()
print('refinement decode: {:.3f} s for 10 x {} queries'.format(t1 - t0, q.shape[0]))
