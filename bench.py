"""Headline benchmark: occupancy query-points/sec of the PPSurf 50NN decoder path at gen_resolution_global=257.

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank per GPU)

One STEP = one pass of the hot path over one chunk of Q = rec_batch_size = 50000 queries of the Marching-Cubes band of a
synthetic 100k-point cloud (configs/poco.yaml:51-52, configs/ppsurf_50nn.yaml): exact 64-NN -> 50-NN patch gather +
normalisation -> interpolation-attention + PointNet + MLP -> occupancy, through the PRODUCT's chunk loop
(ppsurf_amd.decoder.ChunkPipeline, the class reconstruct.OccupancyField drives) and the product's C entry point
pps_decode_fwd(_events)_f32.  The K chunks are DISTINCT: the first region-growing round of create_volume over as many synthetic
shapes as it takes (47 full chunks per shape at R = 257).  Inputs (clouds, query lists, per-point tables, weights) are resident
in HBM before the timed region.

--scaling weak (default): every rank decodes its own chunks of its own shapes, no collective on the data path.
--scaling strong: ONE shape is reconstructed by all ranks together (PPS_SHARD=queries: growth and refinement rounds split
into contiguous query ranges + one all-gather of 4 B/query per round, encoder passes dealt round-robin + one all-reduce of the
latent sums per wave); value = the shape's decoder queries / wall time, plus shapes_per_hour and the collective time share.

Extra keys on the same JSON line (N=1 weak only, skipped with --quick): `shapes_per_hour` (one whole R=257 reconstruction, first
shape and steady state), `fit_ms_per_step` (BASELINE config 3, bf16-mixed, B=10), `cpu_baseline`.
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

N_POINTS = 100_000
Q_CHUNK = 50_000
K_PROJ = 64
P_LOCAL = 50
RES = 257
ALG_MFLOP_PER_QUERY = 53.21                    # conv+matmul FLOPs of the reference's from_latent at P=50
# dominant kernel (interp_pool_kernel).  Reference work it replaces per query, poco_model.py:400-414:
# 64 neighbours x (fc1 66304 + fc2 65536 + fc3 65536 + fc_query 16384 + fc_value 65536) MAC + 16384 MAC pooling
INTERP_ALG_FLOP_PER_QUERY = 2.0 * (64 * 279_296 + 16_384)
# work it EXECUTES per query: 2320 k-steps x 4 v_mfma_f32_16x16x4_f32 x 2048 flop (fc1 hoisted to the per-point table, fc_value
# moved behind the pooling: DESIGN.md section 2)
INTERP_EXEC_FLOP_PER_QUERY = 2320 * 4 * 2048.0
STAGE_EXEC_MFMA_PER_QUERY = {'interp_pool': 9280, 'pointnet_stn_rows': 2413, 'pointnet_stn_fc': 296, 'pointnet_feat_rows': 2670, 'decode_tail': 200}
PEAK_F32_MFMA_TFLOPS = 157.3                   # /opt/skills/guides/MI355X_MICROARCH.md
PEAK_F16_MFMA_TFLOPS = 2500.0                  # dense f16 / bf16 matrix peak, same guide
STAGES = ('interp_pool', 'pointnet_stn_rows', 'pointnet_stn_fc', 'pointnet_feat_rows', 'decode_tail')


def cpu_model():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(sd, cloud, qry, lat, q_call=20_000, reps=3):
    """The oracle (CPU restatement of the reference, kind 'port') on a bounded sample of the same workload: `reps` calls of
    Q = 20 000 queries (SURVEY.md 8d / BASELINE.md 4), N = 100k cloud, kNN + patches + from_latent + occupancy."""
    from oracle import ppsurf_oracle as O
    # torch CPU ops on these tensors stop scaling (and collapse) beyond a few dozen threads; measured on the 256-core host:
    # 50 queries/s at 256 threads, ~2100 at 32
    threads = max(1, min(os.cpu_count() or 1, 32))
    torch.set_num_threads(threads)
    os.environ['OMP_NUM_THREADS'] = str(threads)
    pts_cf = torch.from_numpy(cloud.T.copy()).unsqueeze(0)
    latt = torch.from_numpy(lat)
    times = []
    for r in range(reps):
        q = qry[r * q_call:(r + 1) * q_call]
        t0 = time.time()
        patches = O.get_pts_local_ps(cloud, q, P_LOCAL)
        data = {'latents': latt, 'pts': pts_cf, 'pts_query': torch.from_numpy(q).unsqueeze(0), 'pts_local_ps': torch.from_numpy(patches).unsqueeze(0)}
        with torch.no_grad():
            O.predict_from_latent(O.ppsurf_from_latent(sd, data, k=K_PROJ))
        times.append(time.time() - t0)
    return {'value': q_call / float(np.median(times)), 'unit': 'queries/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'cpu_model': cpu_model(), 'host_cores': os.cpu_count(),
            'sample': '{} calls of {} band queries (N={} cloud, k=64, P=50): torch fp32 from_latent + OpenMP C kNN/patches; '
                      'median of {} s per call'.format(reps, q_call, cloud.shape[0], ', '.join('{:.1f}'.format(t) for t in times))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None, help='default 200 (weak: chunks) / 3 (strong: whole reconstructions)')
    ap.add_argument('--warmup', type=int, default=None, help='default 10 (weak) / 1 (strong)')
    ap.add_argument('--scaling', choices=['weak', 'strong'], default='weak')
    ap.add_argument('--quick', action='store_true', help='query throughput only: no shapes/hour, fit step or CPU baseline legs')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--dtype', choices=['f32', 'f16x3'], default='f32', help="decoder dtype of the timed loop; 'f16x3' (opt-in split precision) is for profiling "
                    'runs -- the headline value is the f32 run')
    ap.add_argument('--backend', default='nccl', help="torch.distributed backend for N>1 ('nccl' = RCCL; 'gloo' only for single-GPU rehearsals)")
    ap.add_argument('--same-gpu', action='store_true', help='rehearsal: every rank uses cuda:0 (needs --backend gloo)')
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 200 if args.scaling == 'weak' else 3
    if args.warmup is None:
        args.warmup = 10 if args.scaling == 'weak' else 1

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

    from ppsurf_amd import sharding, workloads
    from ppsurf_amd.decoder import DecoderPlan, ChunkPipeline
    from ppsurf_amd.synthetic import make_cloud, make_latents, network_state_dict

    red_dev = dev if args.backend == 'nccl' else 'cpu'
    if args.scaling == 'strong':
        return strong(args, rank, world, dev, dist, red_dev)

    sd = network_state_dict('ppsurf')
    plan = DecoderPlan(sd, dev, dtype=args.dtype)
    # ---- resident inputs: as many shapes as it takes to give every step its own chunk of a real first-round band ----------
    total_chunks = args.warmup + args.steps
    shapes, work = [], []                                        # work: (pipeline, chunk) per step
    s = 0
    while len(work) < total_chunks:
        cloud = make_cloud(N_POINTS, seed=42 + 1000 * rank + s)
        pts = torch.from_numpy(cloud).to(dev)
        lat = make_latents(256, N_POINTS, seed=77 + s)
        table = plan.point_table(torch.from_numpy(lat[0]).to(dev))       # per-shape, outside the per-chunk step
        chunks, n_band = workloads.band_chunks(cloud, RES, Q_CHUNK, dev)
        pipe = ChunkPipeline(plan, table, pts, pts, K_PROJ, P_LOCAL, same_cloud=True, max_chunk=Q_CHUNK)
        shapes.append({'cloud': cloud, 'lat': lat, 'band': n_band, 'chunks': len(chunks)})
        work += [(pipe, c) for c in chunks]
        s += 1
    work = work[:total_chunks]
    ev = [workloads.HipEvents(6) for _ in range(args.steps)]

    for pipe, c in work[:args.warmup]:
        pipe.run([c])
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i, (pipe, c) in enumerate(work[args.warmup:]):            # EXACTLY `steps` distinct chunks of Q_CHUNK queries
        res = pipe.run([c], want_occ=True, stage_events=[ev[i].arr])
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    occ = res[-1][1]
    dt = sharding.max_over_ranks(dt, red_dev)
    assert bool(torch.isfinite(occ).all())

    out = None
    if rank == 0:
        ms_step = dt / args.steps * 1e3
        value = world * Q_CHUNK * args.steps / dt
        stage_ms = {name: float(np.mean([e.elapsed_ms(j, j + 1) for e in ev])) for j, name in enumerate(STAGES)}
        k_ms = stage_ms['interp_pool']
        executed = INTERP_EXEC_FLOP_PER_QUERY * Q_CHUNK / (k_ms * 1e-3) / 1e12
        traffic, traffic_src = None, None
        pmc = os.path.join(REPO, 'profiles', 'round2_f32_pmc.json')
        if os.path.isfile(pmc):
            traffic = json.load(open(pmc)).get('interp_pool_hbm_bytes_per_launch')
            traffic_src = 'profiles/round2_f32_pmc.json: 2 x FETCH_SIZE + WRITE_SIZE of separate rocprofv3 --pmc passes of this bench (a previous run, NOT measured in this run)'
        # --dtype f16x3: the dense layers run as three f16 MFMA products per fp32 product -> executed flops x 3, priced against the dense f16 peak
        f16 = args.dtype == 'f16x3'
        peak, mult = (PEAK_F16_MFMA_TFLOPS, 3.0) if f16 else (PEAK_F32_MFMA_TFLOPS, 1.0)
        executed *= mult
        stage_frac = {n: mult * STAGE_EXEC_MFMA_PER_QUERY[n] * 2048.0 * Q_CHUNK / (stage_ms[n] * 1e-3) / 1e12 / peak for n in STAGES}
        out = {
            'metric': 'occupancy query-points/sec @ res=257, 50NN', 'value': value, 'unit': 'queries/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_step,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': 'ppsurf_50nn predict, R=257: {} distinct first-growth-round band chunks of {} queries '
                                   '(rec_batch_size) over {} synthetic 100k-point clouds per GPU, k=64, P=50'.format(args.steps, Q_CHUNK, len(shapes)),
                       'parallelism': 'query-block sharding x{}'.format(world), 'weights': 'formula-filled (no checkpoint offline)',
                       'entry': 'ChunkPipeline.run -> pps_knn_blocked_f32, pps_patch_normalize_f32, pps_decode_fwd_events_f32'},
            'roofline': {'kernel': ('interp_pool_f16x3_kernel' if f16 else 'interp_pool_kernel') + ' (inside pps_decode_fwd_events_f32)', 'bound': 'mfma',
                         'achieved': executed, 'peak': peak, 'unit': 'TFLOP/s', 'frac': executed / peak,
                         'traffic': traffic, 'traffic_source': traffic_src, 'avg_kernel_ms': k_ms,
                         'algorithmic_tflops': INTERP_ALG_FLOP_PER_QUERY * Q_CHUNK / (k_ms * 1e-3) / 1e12,
                         'note': ('achieved / frac = f16 MFMA flops the split-precision kernel executes (3 f16 products per fp32 product of the 9280 x 2048 '
                                  'flop per query) / its HIP-event duration / the dense f16 matrix peak; the kernel is bound by its 576 KB weight stream '
                                  'per pass (LDS-DMA), not by the f16 pipe (DESIGN.md section 4.1b)' if f16 else
                                  'achieved / frac = MFMA flops the kernel EXECUTES (9280 v_mfma_f32_16x16x4_f32 per query x 2048) / its HIP-event '
                                  'duration / the fp32 matrix peak; algorithmic_tflops prices the reference work it replaces (two exact identities '
                                  'remove 47 % of it, DESIGN.md section 2) and is not a hardware fraction')},
            'stage_ms': stage_ms, 'stage_mfma_frac': stage_frac,
            'spatial_ms': ms_step - sum(stage_ms.values()),
            'whole_path_algorithmic_tflops': ALG_MFLOP_PER_QUERY * 1e6 * value / world / 1e12,
            'whole_path_executed_mfma_frac': mult * sum(STAGE_EXEC_MFMA_PER_QUERY.values()) * 2048.0 * value / world / 1e12 / peak,
        }
    extra = world == 1 and not args.quick and args.dtype == 'f32'
    if extra:
        # ---- opt-in split-precision decoder (dtype "f16x3": fc2 / fc3 / fc_query of the interpolation branch as 3 f16 MFMA products
        # per fp32 product; logits within ~1e-5 of the fp32 path, tests/test_gpu_decoder.py).  NOT the headline value. ----------
        plan16 = DecoderPlan(sd, dev, dtype='f16x3')
        n16 = min(args.steps, 100)
        w16 = [(ChunkPipeline(plan16, pipe.table, pipe.pts, pipe.pts, K_PROJ, P_LOCAL, same_cloud=True, max_chunk=Q_CHUNK), c)
               for pipe, c in work[args.warmup:args.warmup + 1]]
        pipes16 = {}
        ev16 = [workloads.HipEvents(6) for _ in range(n16)]
        seq = work[args.warmup:args.warmup + n16]
        for pipe, c in seq:
            if id(pipe) not in pipes16:
                pipes16[id(pipe)] = ChunkPipeline(plan16, pipe.table, pipe.pts, pipe.pts, K_PROJ, P_LOCAL, same_cloud=True, max_chunk=Q_CHUNK)
        for pipe, c in seq[:5]:
            pipes16[id(pipe)].run([c])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i, (pipe, c) in enumerate(seq):
            r16 = pipes16[id(pipe)].run([c], want_occ=True, stage_events=[ev16[i].arr])
        torch.cuda.synchronize()
        dt16 = time.perf_counter() - t0
        ref_occ = pipe.run([c], want_occ=True)[0][1]
        f16_stats = {'value': Q_CHUNK * len(seq) / dt16, 'unit': 'queries/s', 'ms_per_step': dt16 / len(seq) * 1e3, 'steps': len(seq),
                        'stage_ms': {name: float(np.mean([e.elapsed_ms(j, j + 1) for e in ev16])) for j, name in enumerate(STAGES)},
                        'max_abs_occ_diff_vs_f32_last_chunk': float((r16[0][1] - ref_occ).abs().max()),
                        'note': 'opt-in decoder dtype (DecoderPlan(dtype="f16x3") / network.decoder_dtype / PPS_DECODER_DTYPE): the dense layers of '
                                'the interpolation and PointNet branches on the f16 matrix pipe in split precision (3 f16 products per fp32 product, '
                                'fp32 accumulation); per-point table, xyz layers, softmax / pooling and tail fp32; same chunks as the fp32 run'}
        out['f16x3'] = f16_stats
        del plan16, pipes16, w16, ev16
        del work, ev
        torch.cuda.empty_cache()
        model = workloads.make_model(RES, P_LOCAL, Q_CHUNK, dev)
        runs = [workloads.reconstruct_steered(model, N_POINTS, seed=42 + i, device=dev) for i in range(3)]
        steady = min(r['total_s'] for r in runs[1:])
        out['shapes_per_hour'] = 3600.0 / steady
        out['reconstruction'] = {'first_shape_s': runs[0]['total_s'], 'steady_s': steady, 'latent_loop_s': runs[-1]['latent_s'],
                                 'surface_s': runs[-1]['surface_s'], 'decoder_queries': runs[-1]['decoder_queries'],
                                 'vertices': runs[-1]['vertices'], 'first_shape_per_hour': 3600.0 / runs[0]['total_s'],
                                 'encoder_passes_per_s': 10.0 * (N_POINTS // 10000) / runs[-1]['latent_s'],
                                 'note': 'whole R=257 reconstruction of a 100k-point cloud by the product driver: latent loop (100 encoder '
                                         'passes), region growing, Marching Cubes + clean-up, 10 refinement rounds; every query decoded by the real '
                                         'kernels, growth steered by the analytic shape (formula-filled weights describe no surface)'}
        model.network.decoder_dtype = 'f16x3'
        runs16 = [workloads.reconstruct_steered(model, N_POINTS, seed=42 + i, device=dev) for i in range(2)]
        out['f16x3']['shapes_per_hour'] = 3600.0 / runs16[-1]['total_s']
        out['f16x3']['reconstruction_steady_s'] = runs16[-1]['total_s']
        del model
        torch.cuda.empty_cache()
        fit = workloads.FitStep(batch=10, precision='bf16-mixed', device=dev, graph=True)      # as `pps.py fit` runs it: replayed HIP graph, loader thread
        for _ in range(6):
            fit()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n_fit = 20
        for _ in range(n_fit):
            loss = fit()
        torch.cuda.synchronize()
        out['fit_ms_per_step'] = (time.perf_counter() - t0) / n_fit * 1e3
        out['fit'] = {'config': 'ppsurf_50nn fit step: B=10 shapes x 10000 points, 2000 queries/shape, P=50, bf16-mixed, AdamW; id tables + '
                                'patches built on the device by the loader thread on a second stream, step replayed as a HIP graph (the defaults of pps.py fit)',
                      'steps_timed': n_fit, 'loss': float(loss),
                      'shapes_per_s': 10.0 / (out['fit_ms_per_step'] * 1e-3)}
        fit.close()
        del fit
        if not args.no_cpu_baseline:
            qry = torch.cat(workloads.band_chunks(shapes[0]['cloud'], RES, Q_CHUNK, dev)[0][:2]).cpu().numpy()
            out['cpu_baseline'] = cpu_baseline(sd, shapes[0]['cloud'], qry, shapes[0]['lat'])
    elif rank == 0 and world > 1:
        out['shapes_per_hour'] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def strong(args, rank, world, dev, dist, red_dev):
    """One shape, all ranks: PPS_SHARD=queries.  `steps` = reconstructions timed, `warmup` = untimed ones."""
    from ppsurf_amd import sharding, workloads
    sharding.set_query_sharding(world > 1)
    model = workloads.make_model(RES, P_LOCAL, Q_CHUNK, dev)
    model.shard_queries = world > 1
    steps, warm = args.steps, args.warmup
    for i in range(warm):
        workloads.reconstruct_steered(model, N_POINTS, seed=42, device=dev)
    sharding.profile_collectives(True)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    runs = [workloads.reconstruct_steered(model, N_POINTS, seed=43 + i, device=dev) for i in range(steps)]
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = sharding.max_over_ranks(time.perf_counter() - t0, red_dev)
    coll = sharding.collective_seconds()
    mine = sum(r['decoder_queries'] for r in runs)
    total_q = mine
    if dist is not None:
        t = torch.tensor([float(mine)], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t)
        total_q = int(t.item())
    if rank == 0:
        print(json.dumps({
            'metric': 'occupancy query-points/sec @ res=257, 50NN', 'value': total_q / dt, 'unit': 'queries/s', 'n_gpus': world,
            'steps': steps, 'warmup': warm, 'ms_per_step': dt / steps * 1e3, 'higher_is_better': True, 'scaling': 'strong',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'ppsurf_50nn predict, ONE 100k-point synthetic shape at R=257 reconstructed by all ranks together '
                                   '(step = one whole reconstruction: latent loop, region growing, MC, 10 refinement rounds)',
                       'parallelism': 'PPS_SHARD=queries x{}: per growth/refinement round contiguous query ranges + all-gather of 4 B/query; '
                                      'encoder passes round-robin + all-reduce of latent sums'.format(world)},
            'shapes_per_hour': 3600.0 * steps / dt, 'decoder_queries_per_shape': total_q / steps,
            'collective_s_per_shape_rank0': coll / steps, 'collective_share_rank0': coll / dt,
            'collectives_per_shape': sharding.STATS['calls'] / steps}), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
