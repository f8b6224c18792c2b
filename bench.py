"""Headline benchmark: occupancy query-points/sec of the PPSurf 50NN decoder path at gen_resolution_global=257.

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank per GPU)

One STEP = one pass of the hot path over one chunk of Q = rec_batch_size = 50000 grid-band queries of a synthetic
100k-point cloud (configs/poco.yaml:51-52, configs/ppsurf_50nn.yaml): brute-force 64-NN -> 50-NN patch gather +
normalisation -> interpolation-attention + PointNet + MLP -> occupancy.  Inputs (cloud, queries, per-point table,
weights) are resident in HBM before the timed region.  Multi-GPU: the query blocks of the Marching-Cubes band are
sharded over the ranks, no collective on the data path (weak scaling: every rank decodes its own 50000-query block).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))

N_POINTS = 100_000
Q_CHUNK = 50_000
K_PROJ = 64
P_LOCAL = 50
RES = 257
ALG_MFLOP_PER_QUERY = 53.21                    # conv+matmul FLOPs of from_latent at P=50
# dominant kernel (pps_interp_pool_f32): reference work it replaces per query, poco_model.py:400-414:
# 64 neighbours x (fc1 66304 + fc2 65536 + fc3 65536 + fc_query 16384 + fc_value 65536) MAC + 16384 MAC pooling
INTERP_ALG_FLOP_PER_QUERY = 2.0 * (64 * 279_296 + 16_384)
INTERP_EXEC_FLOP_PER_QUERY = 2320 * 4 * 2048.0   # MFMAs issued per query x flop per v_mfma_f32_16x16x4_f32
PEAK_F32_MFMA_TFLOPS = 157.3                   # /opt/skills/guides/MI355X_MICROARCH.md


def cpu_baseline(sd, cloud, qry, lat, n_sample=8192, chunk=1024, budget_s=20.0):
    """The oracle (CPU restatement of the reference, kind 'port') on a bounded sample of the same workload:
    chunks of 1024 queries until ~budget_s seconds of CPU work are spent (at most n_sample queries)."""
    from oracle import ppsurf_oracle as O
    # torch CPU ops on these small per-chunk tensors stop scaling (and collapse) beyond a few dozen threads
    threads = max(1, min(os.cpu_count() or 1, 32))
    torch.set_num_threads(threads)
    os.environ['OMP_NUM_THREADS'] = str(threads)
    sel = np.linspace(0, qry.shape[0] - 1, n_sample).astype(np.int64)
    pts_cf = torch.from_numpy(cloud.T.copy()).unsqueeze(0)
    latt = torch.from_numpy(lat)
    done = 0
    t0 = time.time()
    for s in range(0, n_sample, chunk):
        if done > 0 and time.time() - t0 > budget_s:
            break
        q = qry[sel[s:s + chunk]]
        patches = O.get_pts_local_ps(cloud, q, P_LOCAL)
        data = {'latents': latt, 'pts': pts_cf, 'pts_query': torch.from_numpy(q).unsqueeze(0),
                'pts_local_ps': torch.from_numpy(patches).unsqueeze(0)}
        with torch.no_grad():
            O.predict_from_latent(O.ppsurf_from_latent(sd, data, k=K_PROJ))
        done += q.shape[0]
    dt = time.time() - t0
    n_sample = done
    return {'value': n_sample / dt, 'unit': 'queries/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': '{} of the step\'s {} queries (N={} cloud, k=64, P=50), torch fp32 + OpenMP C kNN, {:.1f} s'.format(
                n_sample, qry.shape[0], cloud.shape[0], dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--backend', default='nccl', help="torch.distributed backend for N>1 ('nccl' = RCCL; 'gloo' only for single-GPU rehearsals)")
    ap.add_argument('--same-gpu', action='store_true', help='rehearsal: every rank uses cuda:0 (needs --backend gloo)')
    ap.add_argument('--overlap', action='store_true', help='A/B only: spatial queries of chunk i+1 on a side stream -- SLOWER (16.8 vs 12.7 ms/step): a resident kNN block '
                         'keeps one of the two persistent decoder workgroups of its CU from starting')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit('--gpus {} needs torch.distributed.run with {} ranks (WORLD_SIZE={})'.format(args.gpus, args.gpus, world))
    if args.same_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    from golden_util import filled_sd
    from ppsurf_amd import ops
    from ppsurf_amd.decoder import DecoderPlan
    from ppsurf_amd.synthetic import make_cloud, make_band_queries, make_latents

    sd = filled_sd('', key='ppsurf')
    plan = DecoderPlan(sd, dev)
    cloud = make_cloud(N_POINTS, seed=42)
    # every rank owns a different block of the band (query-block sharding); rank 0's block is the N=1 workload
    qry = make_band_queries(cloud, Q_CHUNK, resolution=RES, seed=1 + rank)
    lat = make_latents(256, N_POINTS, seed=77)
    pts = torch.from_numpy(cloud).to(dev)
    qd = torch.from_numpy(qry).to(dev)
    table = plan.point_table(torch.from_numpy(lat[0]).to(dev))       # per-shape, outside the per-chunk step

    from ppsurf_amd.decoder import ChunkPipeline
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    # the product's chunk loop (ppsurf_amd.reconstruct.OccupancyField uses the same class): 64-NN search, patch gather (the
    # 50-NN are a prefix of the 64-NN), decoder kernels, all on one stream; --overlap moves the spatial queries of chunk i+1 to
    # a side stream (measured: slower, the persistent decoder kernels are sized to own every CU).
    pipe = ChunkPipeline(plan, table, pts, pts, K_PROJ, P_LOCAL, same_cloud=True, max_chunk=Q_CHUNK, overlap=args.overlap)

    pipe.run([qd] * args.warmup)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    results = pipe.run([qd] * args.steps, want_occ=True, interp_events=ev)        # EXACTLY `steps` chunks of Q_CHUNK queries
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    logits, occ = results[-1]
    from ppsurf_amd import sharding
    dt = sharding.max_over_ranks(dt, dev if args.backend == 'nccl' else 'cpu')
    assert bool(torch.isfinite(occ).all())

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        value = world * Q_CHUNK * args.steps / dt
        k_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
        achieved = INTERP_ALG_FLOP_PER_QUERY * Q_CHUNK / (k_ms * 1e-3) / 1e12
        traffic = None
        pmc = os.path.join(REPO, 'profiles', 'round1_pmc.json')
        if os.path.isfile(pmc):
            traffic = json.load(open(pmc)).get('interp_pool_hbm_bytes_per_launch')
        out = {
            'metric': 'occupancy query-points/sec @ res=257, 50NN', 'value': value, 'unit': 'queries/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_step,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'ppsurf_50nn predict, R=257 band queries, 100k-point synthetic cloud, '
                                   '{} queries per step per GPU (rec_batch_size), k=64, P=50'.format(Q_CHUNK),
                       'parallelism': 'query-block sharding x{}'.format(world), 'weights': 'formula-filled (no checkpoint offline)'},
            'roofline': {'kernel': 'interp_pool_kernel (pps_interp_pool_f32)', 'bound': 'mfma', 'achieved': achieved,
                         'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': achieved / PEAK_F32_MFMA_TFLOPS,
                         'traffic': traffic, 'avg_kernel_ms': k_ms,
                         'executed_tflops': INTERP_EXEC_FLOP_PER_QUERY * Q_CHUNK / (k_ms * 1e-3) / 1e12,
                         'frac_executed': INTERP_EXEC_FLOP_PER_QUERY * Q_CHUNK / (k_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                         'note': 'achieved / frac count the reference\'s ALGORITHMIC flops the kernel replaces (frac > 1 because two exact '
                                 'identities remove work); executed_tflops / frac_executed count the MFMA flops actually issued = '
                                 'utilisation of the fp32 matrix pipe (DESIGN.md 4.1)'},
            'whole_path_algorithmic_tflops': ALG_MFLOP_PER_QUERY * 1e6 * value / world / 1e12,
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(sd, cloud, qry, lat)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
