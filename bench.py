"""Headline benchmark: occupancy query-points/sec of the PPSurf 50NN decoder path at gen_resolution_global=257.

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU over RCCL.  Either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`
(RANK / LOCAL_RANK / WORLD_SIZE in the environment), or plainly as `python bench.py --gpus N`: the script then re-executes itself under
torch.distributed.run (127.0.0.1, a free port) and passes the ranks' output and exit code through.

One STEP = one pass of the hot path over one chunk of Q = rec_batch_size = 50000 queries of the Marching-Cubes band of a
synthetic 100k-point cloud (configs/poco.yaml:51-52, configs/ppsurf_50nn.yaml): exact 64-NN -> 50-NN patch gather +
normalisation -> interpolation-attention + PointNet + MLP -> occupancy, through the PRODUCT's chunk loop
(ppsurf_amd.decoder.ChunkPipeline, the class reconstruct.OccupancyField drives) and the product's C entry point
pps_decode_fwd(_events)_f32.  The K chunks are DISTINCT: the first region-growing round of create_volume over as many synthetic
shapes as it takes (47 full chunks per shape at R = 257).  Inputs (clouds, query lists, per-point tables, weights) are resident
in HBM before the timed region.

N > 1 (the driver's SCALE run is this one command per N) puts THREE measurements on the one JSON line:
  * `value` -- shape-level replicas: every rank decodes its own chunks of its own shapes, no collective on the data path (what a test-set
    reconstruction with PPS_SHARD=shapes does; `scaling: weak`);
  * `strong` -- SURVEY 8(e)'s query-block sharding: ONE R = 257 shape reconstructed by all ranks together (PPS_SHARD=queries: every growth /
    refinement round split into contiguous query ranges + one RCCL all-gather of 4 B/query, encoder passes dealt round-robin + one all-reduce
    of the latent sums per wave): queries/s, shapes/hour, collectives per shape and their share of the wall time;
  * `fit` -- BASELINE config 3 under data parallelism: B = 50 // N shapes per rank (source/base/mp.py:91), backward in three replayed stages with
    the gradient bucket of stage k on the wire (RCCL all-reduce) while stage k + 1 runs: ms/step, the all-reduces alone, the same step with every
    collective behind backward, and the share of the all-reduce time the overlap hides.
--scaling strong prints the `strong` measurement as a line of its own (value = the shape's decoder queries / wall time).
--single-rank-collectives (with --gpus 1): a ONE-rank process group runs the N > 1 code path -- every collective call site executes on the real
backend (RCCL accepts a one-rank communicator) on a 1-GPU box.

Extra keys on the same JSON line (skipped with --quick): `shapes_per_hour` (whole R=257 reconstructions by the product driver; N=1: first
shape and steady state, N>1: every rank reconstructs its own shapes, the figure is all shapes of all ranks / the slowest rank's time),
and at N=1 only `fit_ms_per_step` (BASELINE config 3, bf16-mixed, B=10) and `cpu_baseline`.

The timed region is at least MIN_TIMED_S long: when `steps` distinct chunks take less, the same chunk list is decoded `repeats` times
(`steps` is reported as given, `repeats` and `timed_s` next to it; ms_per_step and value are per chunk over all repeats).
"""
import argparse
import json
import math
import os
import socket
import subprocess
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
MIN_TIMED_S = 2.0                              # the timed region is stretched to this by repeating the chunk list (see `repeats`)
ALG_MFLOP_PER_QUERY = 53.21                    # conv+matmul FLOPs of the reference's from_latent at P=50
# dominant kernel (interp_pool_kernel).  Reference work it replaces per query, poco_model.py:400-414:
# 64 neighbours x (fc1 66304 + fc2 65536 + fc3 65536 + fc_query 16384 + fc_value 65536) MAC + 16384 MAC pooling
INTERP_ALG_FLOP_PER_QUERY = 2.0 * (64 * 279_296 + 16_384)
# work it EXECUTES per query: 2320 k-steps x 4 v_mfma_f32_16x16x4_f32 x 2048 flop (fc1 hoisted to the per-point table, fc_value
# moved behind the pooling: DESIGN.md section 2)
INTERP_EXEC_FLOP_PER_QUERY = 2320 * 4 * 2048.0
# executed v_mfma_f32_16x16x4_f32-equivalents per query at P = 50 (f16x3: 3/8 as many v_mfma_f32_16x16x32_f16 per dense layer).  Round 4: conv1 lives in
# the per-query matrix and conv3 behind the attention pooling (ppsurf_amd/decoder.py): the feature kernel runs 4 tiles x (4 + 64 + 64 + 128), the
# tail (256 + 128) -> 256 -> 256 -> 2 on 16-query tiles
STAGE_EXEC_MFMA_PER_QUERY = {'interp_pool': 9280, 'pointnet_stn_rows': 2413, 'pointnet_stn_fc': 296, 'pointnet_feat_rows': 1040, 'decode_tail': 168}
# ALGORITHMIC flops per query of the reference work each stage replaces at P = 50 (2 x MACs of the conv / matmul ops;
# the stages sum to 53.2 MFLOP with fc8's 65 536 MAC, which the tail carries): STN rows = 50 x (conv0a 192 + conv0b 4096 + stn.conv1 4096 +
# conv2 8192 + conv3 32768), STN head 256x128 + 128x64 + 64x4096, feature rows = 50 x (trans2 4096 + conv1 4096 + conv2 8192 + conv3 32768 +
# att.fc_query 256 + att.fc_value 65536 + pooling 256), tail = fc8 65536 + MLP 131584 (nn.py:162-190,305-373,376-417)
STAGE_ALG_FLOP_PER_QUERY = {'interp_pool': INTERP_ALG_FLOP_PER_QUERY, 'pointnet_stn_rows': 2.0 * 50 * 49_344, 'pointnet_stn_fc': 2.0 * 303_104,
                            'pointnet_feat_rows': 2.0 * 50 * 115_200, 'decode_tail': 2.0 * (65_536 + 131_584)}
PEAK_F32_MFMA_TFLOPS = 157.3                   # /opt/skills/guides/MI355X_MICROARCH.md
PEAK_F16_MFMA_TFLOPS = 2500.0                  # dense f16 / bf16 matrix peak, same guide
MEASURED_F16_MFMA_TFLOPS = 1800.0              # bare v_mfma_f32_16x16x32_f16 loop, random operands, measured (1797-1917; 2274 on zeros)
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
    Q = 20 000 queries (SURVEY.md 8d / BASELINE.md 4), N = 100k cloud, kNN + patches + from_latent + occupancy.  SURVEY 8(d) says
    os.cpu_count() threads; torch CPU ops on these tensors stop scaling (and collapse) beyond a few dozen threads, so a short probe (1000 queries
    per thread count) is run first, its rates travel in the JSON (`threads_tried`) and the fastest count is the one the baseline is quoted on."""
    from oracle import ppsurf_oracle as O
    pts_cf = torch.from_numpy(cloud.T.copy()).unsqueeze(0)
    latt = torch.from_numpy(lat)

    def call(q):
        t0 = time.time()
        patches = O.get_pts_local_ps(cloud, q, P_LOCAL)
        data = {'latents': latt, 'pts': pts_cf, 'pts_query': torch.from_numpy(q).unsqueeze(0), 'pts_local_ps': torch.from_numpy(patches).unsqueeze(0)}
        with torch.no_grad():
            O.predict_from_latent(O.ppsurf_from_latent(sd, data, k=K_PROJ))
        return time.time() - t0

    host = os.cpu_count() or 1
    tried = {}
    for th in sorted({t for t in (8, 32, 64, 128, host) if t <= host}):
        torch.set_num_threads(th)
        os.environ['OMP_NUM_THREADS'] = str(th)
        call(qry[:200])                                          # thread pool warm-up
        tried[th] = 1000 / call(qry[:1000])
        if tried[th] < 0.25 * max(tried.values()):               # collapsing: larger counts only get slower, and each probe would take minutes
            break
    threads = max(tried, key=tried.get)
    torch.set_num_threads(threads)
    os.environ['OMP_NUM_THREADS'] = str(threads)
    times = [call(qry[r * q_call:(r + 1) * q_call]) for r in range(reps)]
    return {'value': q_call / float(np.median(times)), 'unit': 'queries/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'cpu_model': cpu_model(), 'host_cores': host,
            'threads_tried': {str(k): round(v, 1) for k, v in tried.items()},
            'threads_note': 'queries/s of a 1000-query call per torch/OpenMP thread count; `cores` = the fastest of them (SURVEY 8d asks for all host '
                            'cores: torch CPU ops on [1,256,Q,64] tensors slow down beyond a few dozen threads on this host)',
            'sample': '{} calls of {} band queries (N={} cloud, k=64, P=50): torch fp32 from_latent + OpenMP C kNN/patches; '
                      'median of {} s per call'.format(reps, q_call, cloud.shape[0], ', '.join('{:.1f}'.format(t) for t in times))}


def gpu_state():
    """Clocks / power / temperature of GPU 0 as rocm-smi reports them right now (None if the tool is missing): the fit step's kernels are HBM- and
    fabric-bound, and the same step runs 19.9 ms on a chip that has just started and 21-22 ms after a minute of sustained load (power management);
    the bench line carries the state next to `fit_ms_per_step` so that a slow box can be told from a slow program."""
    try:
        p = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--showtemp', '--json'], capture_output=True, text=True, timeout=20)
        d = json.loads(p.stdout)
        card = d.get('card0') or next(iter(d.values()))
        keep = {}
        for k, v in card.items():
            kl = k.lower()
            if any(t in kl for t in ('sclk', 'mclk', 'fclk', 'socclk', 'power', 'temperature (sensor junction)', 'temperature (sensor memory)')):
                keep[k] = v
        return keep
    except Exception as exc:                                # noqa: BLE001 (a diagnostic only)
        return {'unavailable': type(exc).__name__}


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def launch_cmd(n, argv, port):
    """The command line `python bench.py --gpus n ...` turns itself into when no launcher set WORLD_SIZE: one rank per GPU of this node."""
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
            '--master-port', str(port), os.path.abspath(__file__)] + [a for a in argv if a != '--spawn']


def self_spawn(n, argv):
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')          # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault('OMP_NUM_THREADS', '8')
    return subprocess.call(launch_cmd(n, argv, free_port()), env=env)


HEADLINE_DTYPE = 'f16x3'     # decoder dtype of `value` = the product default (ppsurf_amd.decoder.DecoderPlan); the fp32 run travels as an extra key of the N=1 line


def rank_values(x, rank, world, dist, red_dev):
    """[x of rank 0, ..., x of rank world-1] on every rank."""
    if dist is None:
        return [float(x)]
    t = torch.zeros(world, dtype=torch.float64, device=red_dev)
    t[rank] = float(x)
    dist.all_reduce(t)
    return [float(v) for v in t.cpu()]


def build_work(plan, n_chunks, rank, dev, queries='band'):
    """Resident inputs: as many synthetic shapes as it takes to give every step its own chunk of a real first-round band."""
    import bench_workloads as workloads
    from ppsurf_amd.decoder import ChunkPipeline
    from ppsurf_amd.synthetic import make_cloud, make_latents
    shapes, work = [], []                                        # work: (pipeline, chunk) per step
    s = 0
    while len(work) < n_chunks:
        cloud = make_cloud(N_POINTS, seed=42 + 1000 * rank + s)
        pts = torch.from_numpy(cloud).to(dev)
        lat = make_latents(256, N_POINTS, seed=77 + s)
        table = plan.point_table(torch.from_numpy(lat[0]).to(dev))       # per-shape, outside the per-chunk step
        if queries == 'dense':
            chunks, n_band = workloads.dense_chunks(cloud, RES, Q_CHUNK, dev, n_chunks=47)      # as many per shape as the band gives: same shape count
            n_band *= Q_CHUNK
        else:
            chunks, n_band = workloads.band_chunks(cloud, RES, Q_CHUNK, dev)
        pipe = ChunkPipeline(plan, table, pts, pts, K_PROJ, P_LOCAL, same_cloud=True, max_chunk=Q_CHUNK)
        shapes.append({'cloud': cloud, 'lat': lat, 'band': n_band, 'chunks': len(chunks)})
        work += [(pipe, c) for c in chunks]
        s += 1
    return shapes, work[:n_chunks]


def chunk_loop(work, steps, warmup, rank, world, dist, red_dev, min_timed_s=MIN_TIMED_S):
    """`warmup` untimed chunks, then `repeats` passes over the next `steps` DISTINCT chunks between barrier + synchronize on both sides, handed to
    the product's chunk loop the way reconstruct.OccupancyField hands it a growth round: ChunkPipeline.run(list of the shape's chunks), which deals
    long lists to two HIP streams (ChunkPipeline lanes).  Per-kernel HIP events cannot be read under two lanes (a kernel's interval then contains
    the moments it shares the chip), so the per-stage times and the roofline come from a SECOND, single-lane pass over the same chunks
    (`single_lane`), timed the same way.
    Returns the max-over-ranks wall time, this rank's own time, repeats, per-stage HIP-event means (ms) and the last occupancy."""
    import bench_workloads as workloads
    from ppsurf_amd import sharding
    from ppsurf_amd.fit import HostGcPacer
    torch.cuda.synchronize()
    tw = time.perf_counter()
    for pipe, c in work[:warmup]:
        pipe.run([c])
    torch.cuda.synchronize()
    est = (time.perf_counter() - tw) / max(warmup, 1)              # s per chunk, first-call overheads included: an over-estimate
    timed = work[warmup:warmup + steps]
    groups = []                                                   # consecutive chunks of one shape: one run() call each
    for pipe, c in timed:
        if groups and groups[-1][0] is pipe:
            groups[-1][1].append(c)
        else:
            groups.append((pipe, [c]))
    repeats = 1
    if warmup > 0 and min_timed_s > 0:
        # one more untimed pass over the timed chunks' first group gives a steady-state estimate (and allocates the second lane's buffers)
        probe = groups[0]
        torch.cuda.synchronize()
        tw = time.perf_counter()
        probe[0].run(probe[1])
        torch.cuda.synchronize()
        est = (time.perf_counter() - tw) / len(probe[1])
        for pipe, cs in groups[1:]:
            pipe.run(cs[:min(len(cs), 4)])
        repeats = max(1, int(math.ceil(min_timed_s / (est * steps))))
    repeats = int(round(sharding.max_over_ranks(float(repeats), red_dev)))
    pacer = HostGcPacer().__enter__()                             # as the predict loop of ppsurf_amd.runner decodes the chunks of a shape
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(repeats):
        for pipe, cs in groups:                                   # `steps` distinct chunks of Q_CHUNK queries, `repeats` times
            res = pipe.run(cs, want_occ=True)
            pacer.tick()
    torch.cuda.synchronize()
    mine = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
    dt = sharding.max_over_ranks(time.perf_counter() - t0, red_dev)
    occ = res[-1][1]
    assert bool(torch.isfinite(occ).all())
    # ---- the SAME loop with the chunk lanes switched off: what the second HIP stream is worth (the difference between `value`
    #      and the per-kernel pass below is NOT the lanes' gain -- that pass issues one chunk per call with HIP events between the kernels)
    saved = [(pipe, pipe.lanes) for pipe, _ in groups]
    for pipe, _ in groups:
        pipe.lanes = 1
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    for _ in range(repeats):
        for pipe, cs in groups:
            pipe.run(cs, want_occ=True)
            pacer.tick()
    torch.cuda.synchronize()
    one_lane = (time.perf_counter() - t2) / (repeats * steps)
    for pipe, ln in saved:
        pipe.lanes = ln
    lanes = max(min(pipe.lanes if pipe.lanes is not None else (2 if len(cs) >= pipe.LANE_MIN_CHUNKS else 1), len(cs)) for pipe, cs in groups)
    # ---- single-lane pass with HIP events around every kernel inside the product call: stage times, roofline -----------------------------------
    n_ev = min(len(timed), 100)
    ev = [workloads.HipEvents(6) for _ in range(n_ev)]
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for i, (pipe, c) in enumerate(timed[:n_ev]):
        pipe.run([c], want_occ=True, stage_events=[ev[i].arr])
        pacer.tick()
    torch.cuda.synchronize()
    single = (time.perf_counter() - t1) / n_ev
    pacer.close()
    stage_ms = {name: float(np.mean([e.elapsed_ms(j, j + 1) for e in ev])) for j, name in enumerate(STAGES)}
    return {'dt': dt, 'mine': mine, 'repeats': repeats, 'stage_ms': stage_ms, 'occ': occ, 'chunks': steps * repeats, 'lanes': lanes,
            'single_lane_ms_per_step': single * 1e3, 'single_lane_steps': n_ev, 'one_lane_loop_ms_per_step': one_lane * 1e3}


def pmc_traffic(dtype):
    """HBM bytes per launch of the dominant kernel from the committed counter passes of this bench -- a previous run of the same command, never
    measured inside this run.  The file records a digest of the kernel sources it was measured on (bench_workloads.csrc_digest); if the sources
    have changed since, the figure is NOT reported (traffic: null) instead of silently going stale."""
    import bench_workloads as workloads
    here = workloads.csrc_digest()
    for tag in ('round6', 'round5', 'round4', 'round3', 'round2'):
        path = os.path.join(REPO, 'profiles', '{}_{}_pmc.json'.format(tag, dtype))
        if os.path.isfile(path):
            d = json.load(open(path))
            if d.get('csrc_digest') != here:
                return None, 'profiles/{} was measured on other kernel sources (digest {} at commit {}, now {}): not reported'.format(
                    os.path.basename(path), d.get('csrc_digest', 'unrecorded'), d.get('git_head', 'unrecorded'), here)
            src = 'profiles/{}: 2 x FETCH_SIZE + WRITE_SIZE of separate rocprofv3 --pmc passes of this bench at commit {}, same kernel sources (digest {}); a previous run, NOT measured in this run'.format(
                os.path.basename(path), d.get('git_head', 'unrecorded'), here)
            return d.get('interp_pool_hbm_bytes_per_launch'), src
    return None, None


def roofline_block(dtype, stage_ms):
    f16 = dtype == 'f16x3'
    peak, mult = (PEAK_F16_MFMA_TFLOPS, 3.0) if f16 else (PEAK_F32_MFMA_TFLOPS, 1.0)
    k_ms = stage_ms['interp_pool']
    executed = mult * INTERP_EXEC_FLOP_PER_QUERY * Q_CHUNK / (k_ms * 1e-3) / 1e12
    traffic, traffic_src = pmc_traffic(dtype)
    algorithmic = INTERP_ALG_FLOP_PER_QUERY * Q_CHUNK / (k_ms * 1e-3) / 1e12
    return {'kernel': ('interp_pool_f16x3_kernel' if f16 else 'interp_pool_kernel') + ' (inside pps_decode_fwd_events_f32)', 'bound': 'mfma',
            # SURVEY 8(d)'s quantity: ALGORITHMIC flops -- the reference work this kernel replaces, 35.78 MFLOP per query (poco_model.py:400-414) x the
            # 50 000 queries of a launch -- / the kernel's HIP-event duration / the guide's dense peak for the MFMA type the kernel issues.
            # fp32: two exact identities remove 47 % of the reference's flops before the kernel runs, so this can exceed 1 there
            'achieved': algorithmic, 'peak': peak, 'unit': 'TFLOP/s', 'frac': algorithmic / peak,
            # what the kernel EXECUTES on the matrix pipe (f16x3: the 3 f16 products per fp32 product counted as work) -- a utilisation figure, not
            # the contract's fraction
            'executed_tflops': executed, 'frac_executed': executed / peak,
            # what a bare loop of the same MFMA instruction sustains on random operands on this chip (it clocks down under matrix load:
            # tools/ubench/mfma_power_probe.hip, profiles/round4_mfma_power_probe.txt); NOT a roofline fraction
            'peak_sustained_random_operands': MEASURED_F16_MFMA_TFLOPS if f16 else None,
            'executed_over_sustained': executed / MEASURED_F16_MFMA_TFLOPS if f16 else None,
            'traffic': traffic, 'traffic_source': traffic_src, 'avg_kernel_ms': k_ms,
            'note': 'achieved / frac = algorithmic flops per launch (SURVEY 8d: 35.78 MFLOP x 50 000 queries) / HIP-event duration of the kernel / ' +
                    ('the dense f16 matrix peak; executed_tflops counts the 3 f16 MFMAs per fp32 product of the 9280 x 2048 flop per query the '
                     'kernel runs (DESIGN.md section 4.1)' if f16 else
                     'the fp32 matrix peak; executed_tflops = the 9280 v_mfma_f32_16x16x4_f32 per query x 2048 flop the kernel runs (two exact '
                     'identities remove 47 % of the reference work, DESIGN.md section 2: frac > 1 is possible, frac_executed is the hardware fraction)')}


def dtype_stats(dtype, r, world):
    """Per-dtype figures of one chunk_loop result `r`."""
    f16 = dtype == 'f16x3'
    peak, mult = (PEAK_F16_MFMA_TFLOPS, 3.0) if f16 else (PEAK_F32_MFMA_TFLOPS, 1.0)
    value = world * Q_CHUNK * r['chunks'] / r['dt']
    ms_step = r['dt'] / r['chunks'] * 1e3
    sm = r['stage_ms']
    return {'value': value, 'unit': 'queries/s', 'ms_per_step': ms_step, 'repeats': r['repeats'], 'timed_s': r['dt'], 'stage_ms': sm,
            'stage_mfma_frac': {n: STAGE_ALG_FLOP_PER_QUERY[n] * Q_CHUNK / (sm[n] * 1e-3) / 1e12 / peak for n in STAGES},      # algorithmic (8d)
            'stage_mfma_frac_executed': {n: mult * STAGE_EXEC_MFMA_PER_QUERY[n] * 2048.0 * Q_CHUNK / (sm[n] * 1e-3) / 1e12 / peak for n in STAGES},
            'lanes': r.get('lanes', 1),
            # the timed loop again with ONE chunk lane (same calls, no events): value / this = what the second stream gains
            'one_lane_loop': {'ms_per_step': r['one_lane_loop_ms_per_step'], 'value': world * Q_CHUNK / (r['one_lane_loop_ms_per_step'] * 1e-3),
                              'lanes_gain': r['one_lane_loop_ms_per_step'] / ms_step - 1.0},
            # the same chunks one at a time on ONE stream with HIP events around every kernel: what stage_ms / roofline / spatial_ms describe
            'single_lane': {'ms_per_step': r['single_lane_ms_per_step'], 'value': world * Q_CHUNK / (r['single_lane_ms_per_step'] * 1e-3),
                            'steps': r['single_lane_steps']},
            'spatial_ms': r['single_lane_ms_per_step'] - sum(sm.values()),
            'whole_path_algorithmic_tflops': ALG_MFLOP_PER_QUERY * 1e6 * value / world / 1e12,
            'whole_path_algorithmic_frac': ALG_MFLOP_PER_QUERY * 1e6 * value / world / 1e12 / peak,
            'whole_path_executed_mfma_frac': mult * sum(STAGE_EXEC_MFMA_PER_QUERY.values()) * 2048.0 * value / world / 1e12 / peak}


DTYPE_NOTE = {
    'f32': 'every product an fp32 MFMA (v_mfma_f32_16x16x4_f32, bit-for-bit an fp32 fmaf chain)',
    'f16x3': 'split precision: every fp32 product of the dense layers is carried as 3 f16 MFMAs (hi.hi + hi.lo + lo.hi of x = hi + lo, f16 parts, '
             '~21 significand bits) with fp32 accumulation -- wider than the 16-mixed autocast arithmetic the reference runs its GPU predict in '
             '(configs/poco.yaml:10); per-point table, xyz layers, softmax / pooling and tail stay fp32; parity held at the same 1e-4 bar as fp32 '
             '(tests/test_gpu_decoder.py, test_gpu_api.py, test_gpu_configs.py)'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None, help='default 200 (weak: chunks) / 3 (strong: whole reconstructions)')
    ap.add_argument('--warmup', type=int, default=None, help='default 10 (weak) / 1 (strong)')
    ap.add_argument('--scaling', choices=['weak', 'strong'], default='weak')
    ap.add_argument('--quick', action='store_true', help='query throughput only: no shapes/hour, fit step or CPU baseline legs')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--dtype', choices=['f32', 'f16x3'], default=HEADLINE_DTYPE, help='decoder dtype of the timed loop and of `value`')
    ap.add_argument('--backend', default='nccl', help="torch.distributed backend for N>1 ('nccl' = RCCL; 'gloo' only for single-GPU rehearsals)")
    ap.add_argument('--same-gpu', action='store_true', help='rehearsal: every rank uses cuda:0 (needs --backend gloo)')
    ap.add_argument('--spawn', action='store_true', help='re-execute under torch.distributed.run even for --gpus 1 (what --gpus N>1 does by itself)')
    ap.add_argument('--queries', choices=['band', 'dense'], default='band',
                    help="band: the first region-growing round (what a reconstruction evaluates); dense: z-slab blocks of the whole (R+2)^3 grid "
                         "(SURVEY 8d(i): 'every voxel of the Marching-Cubes grid')")
    ap.add_argument('--shapes', type=int, default=2, help='timed whole reconstructions per rank of the shapes/hour leg')
    ap.add_argument('--single-rank-collectives', action='store_true',
                    help='--gpus 1 only: initialise a ONE-rank process group and run the N > 1 code path (strong + fit legs with every collective issued)')
    ap.add_argument('--only', choices=['config2', 'config5', 'fit'], default=None, help='run ONE extra leg and print its object (profiling: tools/profile_config5.sh; the default line runs `fit` this way)')
    ap.add_argument('--fit-in-process', action='store_true', help='N = 1: the fit leg at the end of this process instead of in a process of its own')
    ap.add_argument('--legs', default='replicas,strong,fit', help='N > 1: which measurements go on the line (comma list of replicas, strong, fit)')
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 200 if args.scaling == 'weak' else 3
    if args.warmup is None:
        args.warmup = 10 if args.scaling == 'weak' else 1

    launched = 'RANK' in os.environ and 'WORLD_SIZE' in os.environ
    if not launched and (args.gpus > 1 or args.spawn):
        # plain `python bench.py --gpus N`: become N ranks (one per GPU) under torch.distributed.run and pass their result through
        sys.exit(self_spawn(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        raise SystemExit('--gpus {} but the launcher started {} ranks (WORLD_SIZE)'.format(args.gpus, world))
    if args.same_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    dist = None
    from ppsurf_amd import sharding
    if args.single_rank_collectives:
        if world != 1:
            raise SystemExit('--single-rank-collectives is the one-rank rehearsal of the N > 1 path: use it with --gpus 1')
        os.environ['PPS_SINGLE_RANK_COLLECTIVES'] = '1'
    multi = world > 1 or args.single_rank_collectives
    if multi:
        import torch.distributed as dist
        sharding.init_process_group(dev, backend=args.backend)     # nccl: communicator bound to this rank's GPU (device_id)
    import bench_workloads as workloads
    from ppsurf_amd.decoder import DecoderPlan
    from ppsurf_amd.synthetic import network_state_dict

    red_dev = dev if args.backend == 'nccl' else 'cpu'
    if args.only is not None:
        if args.only == 'fit':
            print(json.dumps(fit_leg(dev)), flush=True)
        else:
            print(json.dumps({'config2': config2_leg, 'config5': config5_leg}[args.only](dev, args.dtype)), flush=True)
        return
    if args.scaling == 'strong':
        st = strong_leg(args, rank, world, dev, dist, red_dev, args.steps, args.warmup)
        if rank == 0:
            print(json.dumps(strong_line(args, world, st)), flush=True)
        if dist is not None:
            dist.destroy_process_group()
        return
    legs = set(args.legs.split(',')) if multi else {'replicas'}

    sd = network_state_dict('ppsurf')
    plan = DecoderPlan(sd, dev, dtype=args.dtype)
    shapes, work = build_work(plan, args.warmup + args.steps, rank, dev, queries=args.queries)
    r = chunk_loop(work, args.steps, args.warmup, rank, world, dist, red_dev)
    per_rank = [Q_CHUNK * r['chunks'] / t for t in rank_values(r['mine'], rank, world, dist, red_dev)]

    out = None
    if rank == 0:
        st = dtype_stats(args.dtype, r, world)
        out = {
            'metric': 'occupancy query-points/sec @ res=257, 50NN', 'value': st['value'], 'unit': 'queries/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': st['ms_per_step'],
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'repeats': r['repeats'], 'timed_s': r['dt'], 'world_size_seen': world,
            'backend': ('RCCL (torch.distributed nccl)' if args.backend == 'nccl' else args.backend) if multi else None,
            'per_rank_queries_per_s': {'min': min(per_rank), 'max': max(per_rank), 'all': per_rank},
            'dtype_note': DTYPE_NOTE[args.dtype],
            'config': {'workload': 'ppsurf_50nn predict, R=257: {} distinct {} chunks of {} queries '
                                   '(rec_batch_size) over {} synthetic 100k-point clouds per GPU, k=64, P=50'.format(
                                       args.steps, 'first-growth-round band' if args.queries == 'band' else 'dense z-slab (every voxel of the (R+2)^3 grid)',
                                       Q_CHUNK, len(shapes)),
                       'queries': args.queries,
                       'parallelism': 'shape-level replicas x{} (every rank its own shapes, no collective on the data path; the query-block sharding '
                                      'of ONE shape over all ranks is the `strong` key of this line)'.format(world) if multi else 'one GPU',
                       'weights': 'formula-filled (no checkpoint offline)',
                       'entry': 'ChunkPipeline.run -> pps_knn_blocked_f32, pps_patch_normalize_f32, pps_decode_fwd_events_f32'},
            'roofline': roofline_block(args.dtype, r['stage_ms']),
            'lanes': st['lanes'], 'single_lane': st['single_lane'], 'one_lane_loop': st['one_lane_loop'],
            'lanes_note': 'value / ms_per_step: the product chunk loop (ChunkPipeline.run on the chunk list of a shape; lists of >= 4 chunks are dealt to 2 HIP '
                          'streams).  one_lane_loop: the same calls with one lane -- the gain of the second stream is value / one_lane_loop.value - 1.  '
                          'stage_ms, roofline, spatial_ms: a separate pass, ONE chunk per call with a HIP event between the kernels (single_lane): its '
                          'step is longer than one_lane_loop by the per-call overhead, not by lanes',
            'stage_ms': st['stage_ms'], 'stage_mfma_frac': st['stage_mfma_frac'], 'stage_mfma_frac_executed': st['stage_mfma_frac_executed'], 'spatial_ms': st['spatial_ms'],
            'whole_path_algorithmic_tflops': st['whole_path_algorithmic_tflops'], 'whole_path_algorithmic_frac': st['whole_path_algorithmic_frac'],
            'whole_path_executed_mfma_frac': st['whole_path_executed_mfma_frac'],
        }
    if not args.quick:
        other = 'f16x3' if args.dtype == 'f32' else 'f32'
        if not multi:
            # ---- the other decoder dtype on the same chunks: an extra key, never `value` ----------------------------------------------
            plan2 = DecoderPlan(sd, dev, dtype=other)
            from ppsurf_amd.decoder import ChunkPipeline
            pipes2, work2 = {}, []
            for pipe, c in work:
                if id(pipe) not in pipes2:
                    pipes2[id(pipe)] = ChunkPipeline(plan2, pipe.table, pipe.pts, pipe.pts, K_PROJ, P_LOCAL, same_cloud=True, max_chunk=Q_CHUNK)
                work2.append((pipes2[id(pipe)], c))
            n2, w2 = min(args.steps, 100), min(args.warmup, 5)
            r2 = chunk_loop(work2, n2, w2, rank, world, dist, red_dev, min_timed_s=1.0)
            last_pipe, last_c = work[w2 + n2 - 1]                       # the chunk r2['occ'] belongs to, decoded by the headline dtype
            ref_occ = last_pipe.run([last_c], want_occ=True)[0][1]
            out[other] = dict(dtype_stats(other, r2, world), steps=n2, note=DTYPE_NOTE[other], roofline=roofline_block(other, r2['stage_ms']))
            out[other]['max_abs_occ_diff_vs_{}_last_chunk'.format(args.dtype)] = float((r2['occ'] - ref_occ).abs().max())
            del plan2, pipes2, work2, r2
        if not multi and args.queries == 'band':
            # ---- the north star's other reading of the workload: dense z-slab blocks of the whole grid, same step, same dtype ---------------
            _, workd = build_work(plan, 5 + 40, rank, dev, queries='dense')
            rd = chunk_loop(workd, 40, 5, rank, world, dist, red_dev, min_timed_s=0.5)
            out['dense'] = dict(dtype_stats(args.dtype, rd, world), steps=40,
                                note='z-slab blocks of the (R+2)^3 Marching-Cubes grid in index order, 50 000 voxels each, evenly spaced over the volume '
                                     '(poco_utils.py:52-58,212-213): what a reconstruction WITHOUT region growing would evaluate')
            del workd, rd
        del work, r
        torch.cuda.empty_cache()
        # ---- shapes/hour: whole R=257 reconstructions by the product driver, every rank its own shapes ----------------------------------
        model = workloads.make_model(RES, P_LOCAL, Q_CHUNK, dev)
        model.network.decoder_dtype = args.dtype
        first = workloads.reconstruct_steered(model, N_POINTS, seed=42 + 1000 * rank, device=dev)
        from ppsurf_amd.fit import HostGcPacer
        runs = []
        pacer = HostGcPacer(every=1).__enter__()            # the predict loop of ppsurf_amd.runner: a young-generation collection between shapes,
        if dist is not None:                                # the full one when the loop is over
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.shapes):
            runs.append(workloads.reconstruct_steered(model, N_POINTS, seed=43 + 1000 * rank + i, device=dev))
            pacer.tick()
        torch.cuda.synchronize()
        mine = time.perf_counter() - t0
        if dist is not None:
            dist.barrier()
        dt_shapes = sharding.max_over_ranks(time.perf_counter() - t0, red_dev)
        pacer.close()
        rank_sph = [3600.0 * args.shapes / t for t in rank_values(mine, rank, world, dist, red_dev)]
        if rank == 0:
            steady = min(x['total_s'] for x in runs)
            # all timed shapes of all ranks / the slowest rank's wall time (N = 1: the mean over the timed shapes); the best single shape travels as
            # reconstruction.steady_best_shapes_per_hour
            out['shapes_per_hour'] = 3600.0 * world * args.shapes / dt_shapes
            out['reconstruction'] = {'first_shape_s': first['total_s'], 'steady_s': steady, 'steady_best_shapes_per_hour': 3600.0 / steady,
                                     'cloud_noise_sigma': 0.005, 'latent_loop_s': runs[-1]['latent_s'],
                                     'surface_s': runs[-1]['surface_s'], 'decoder_queries': runs[-1]['decoder_queries'],
                                     'vertices': runs[-1]['vertices'], 'first_shape_per_hour': 3600.0 / first['total_s'],
                                     'encoder_passes_per_s': 10.0 * (N_POINTS // 10000) / runs[-1]['latent_s'],
                                     'shapes_timed_per_rank': args.shapes, 'timed_s': dt_shapes, 'decoder_dtype': args.dtype,
                                     'per_rank_shapes_per_hour': {'min': min(rank_sph), 'max': max(rank_sph), 'all': rank_sph},
                                     'note': 'whole R=257 reconstruction of a 100k-point cloud by the product driver: latent loop (100 encoder '
                                             'passes), region growing, Marching Cubes + clean-up, 10 refinement rounds; every query decoded by the real '
                                             'kernels, growth steered by the analytic shape (formula-filled weights describe no surface); N>1: shape-level '
                                             'sharding (PPS_SHARD=shapes), no collective on the data path'}
        if not multi:
            model.network.decoder_dtype = other
            runs2 = [workloads.reconstruct_steered(model, N_POINTS, seed=42 + i, device=dev) for i in range(3)]      # the first warms the other dtype's plan
            out[other]['shapes_per_hour'] = 3600.0 * len(runs2[1:]) / sum(x['total_s'] for x in runs2[1:])
            out[other]['reconstruction_steady_s'] = runs2[-1]['total_s']
        del model
        torch.cuda.empty_cache()
    if multi and not args.quick:
        # ---- the two measurements that need every rank at once -------------------------------------------------------------------
        # (an exception every rank raises alike -- a bug, an unsupported configuration -- is reported in the line; the replica figure is not lost)
        if 'strong' in legs:
            try:
                st = strong_leg(args, rank, world, dev, dist, red_dev, steps=2, warm=1)
                if rank == 0:
                    out['strong'] = strong_block(world, st)
            except Exception as exc:                        # noqa: BLE001
                if rank == 0:
                    out['strong'] = {'error': '{}: {}'.format(type(exc).__name__, str(exc)[-1500:])}
            torch.cuda.empty_cache()
        if 'fit' in legs:
            try:
                fb = ddp_fit_leg(rank, world, dev, dist, red_dev)
                if rank == 0:
                    out['fit'] = fb
                    out['fit_ms_per_step'] = fb['ms_per_step']
            except Exception as exc:                        # noqa: BLE001
                if rank == 0:
                    out['fit'] = {'error': '{}: {}'.format(type(exc).__name__, str(exc)[-1500:])}
                    out['fit_ms_per_step'] = None
            torch.cuda.empty_cache()
    if not multi and not args.quick:
        # ---- BASELINE.md section 3's other per-config report lines ---------------------------------------------------------
        out['config2'] = config2_leg(dev, args.dtype)
        torch.cuda.empty_cache()
        out['config5'] = config5_leg(dev, args.dtype)
        torch.cuda.empty_cache()
    if not multi and not args.quick:
        try:
            out.update(fit_leg(dev) if args.fit_in_process else fit_leg_fresh_process(args))
            if args.fit_in_process:
                out['fit']['process'] = 'at the end of the bench process (--fit-in-process)'
        except Exception as exc:                            # noqa: BLE001 -- the headline line must not be lost to a failed extra leg
            out['fit_ms_per_step'] = None
            out['fit'] = {'error': '{}: {}'.format(type(exc).__name__, str(exc)[-1500:])}
        if not args.no_cpu_baseline:
            qry = torch.cat(workloads.band_chunks(shapes[0]['cloud'], RES, Q_CHUNK, dev)[0][:2]).cpu().numpy()
            out['cpu_baseline'] = cpu_baseline(sd, shapes[0]['cloud'], qry, shapes[0]['lat'])
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def fit_roofline(ms_per_step):
    """Executed matrix flops and HBM bytes of one fit step from the committed counter passes of the REPLAYED step (tools/profile_train_pmc.sh ->
    profiles/round5_train_pmc.json) against the step time measured in THIS run -- only if the file was measured on the same kernel sources and
    training code (bench_workloads.fit_digest); otherwise no roofline is reported (the round-3 object mixed an eager step of an older
    commit with the replayed step's time)."""
    import bench_workloads as workloads
    paths = [os.path.join(REPO, 'profiles', '{}_train_pmc.json'.format(tag)) for tag in ('round6', 'round5', 'round4')]
    paths = [q for q in paths if os.path.isfile(q)]
    if not paths:
        return {'roofline': None, 'roofline_note': 'no counter passes of the replayed fit step committed'}
    path = paths[0]
    name = 'profiles/' + os.path.basename(path)
    d = json.load(open(path))
    if d.get('fit_digest') != workloads.fit_digest():
        return {'roofline': None, 'roofline_note': '{} was measured on other sources (digest {} at commit {}, now {}): not reported'.format(
            name, d.get('fit_digest', 'unrecorded'), d.get('git_head', 'unrecorded'), workloads.fit_digest())}
    s = ms_per_step * 1e-3
    return {'roofline': {'source': '{} (tools/profile_train_pmc.sh: rocprofv3 --pmc passes of {} at commit {}, same sources; per-step sums over all '
                                   'kernels incl. the data preparation on the second stream); step time from this run'.format(name, d.get('command', '?'), d.get('git_head', 'unrecorded')),
                         'mfma_flops_per_step': d['mfma_flops_per_step'], 'hbm_bytes_per_step': d['hbm_bytes_per_step'],
                         'achieved_tflops': d['mfma_flops_per_step'] / s / 1e12, 'mfma_frac_of_bf16_peak': d['mfma_flops_per_step'] / s / 1e12 / PEAK_F16_MFMA_TFLOPS,
                         'achieved_hbm_tb_s': d['hbm_bytes_per_step'] / s / 1e12, 'hbm_frac': d['hbm_bytes_per_step'] / s / 8e12,
                         'bound': d.get('bound', 'hbm')}}


def config2_leg(dev, dtype, shapes=2):
    """BASELINE config 2 (ppsurf_50nn predict, ONE shape, gen_resolution_global = 129, 1 GPU; BASELINE.md section 3: q/s, s/shape): whole reconstructions
    by the product driver at R = 129 -- latent loop, region growing, Marching Cubes + clean-up, 10 refinement rounds -- on the synthetic 100k-point
    cloud (growth steered by the analytic shape like the R = 257 leg: formula-filled weights describe no surface; the real ABC shape of this config
    runs in tests/test_gpu_configs.py::test_config2_* with learned toy weights)."""
    import bench_workloads as workloads
    model = workloads.make_model(129, P_LOCAL, Q_CHUNK, dev)
    model.network.decoder_dtype = dtype
    first = workloads.reconstruct_steered(model, N_POINTS, seed=42, device=dev)
    runs = [workloads.reconstruct_steered(model, N_POINTS, seed=43 + i, device=dev) for i in range(shapes)]
    best = min(runs, key=lambda r: r['total_s'])
    tot_q, tot_s = sum(r['decoder_queries'] for r in runs), sum(r['total_s'] for r in runs)
    del model
    return {'config': 'ppsurf_50nn predict, one 100k-point synthetic shape, gen_resolution_global=129, rec_batch_size=50000, 1 GPU (BASELINE config 2)',
            'queries_per_s': tot_q / tot_s, 's_per_shape': tot_s / shapes, 'best_s_per_shape': best['total_s'], 'first_shape_s': first['total_s'],
            'decoder_queries_per_shape': tot_q / shapes, 'latent_loop_s': best['latent_s'], 'surface_s': best['surface_s'],
            'vertices': best['vertices'], 'shapes_timed': shapes, 'decoder_dtype': dtype,
            'surface_queries_per_s': best['decoder_queries'] / best['surface_s'],
            'note': 'queries_per_s = decoder queries of the timed shapes / their whole wall time (latent loop, driver, Marching Cubes included); '
                    'surface_queries_per_s = the same without the latent loop'}


C5_N, C5_P, C5_Q, C5_RES = 250_000, 200, 25_000, 513          # configs/ppsurf_200nn.yaml:8 + BASELINE config 5
# algorithmic bytes per query of the two gather kernels of config 5: the 200-NN search reads the query (12 B) and writes 200 int64 ids (the cloud,
# 3 MB, is shared by all queries); the patch kernel reads those ids, gathers 200 points of 12 B and writes 200 normalised points
C5_KNN_BYTES, C5_PATCH_BYTES = 12 + 200 * 8, 12 + 200 * 8 + 200 * 12 + 200 * 12


def config5_pmc():
    """HBM bytes per launch of the k = 200 search and the patch kernel from the committed counter passes of `python bench.py --only config5`
    (profiles/round6_config5_pmc.json), only if measured on these kernel sources."""
    import bench_workloads as workloads
    path = os.path.join(REPO, 'profiles', 'round6_config5_pmc.json')
    if not os.path.isfile(path):
        return None, 'no counter passes of the config-5 leg committed'
    d = json.load(open(path))
    if d.get('csrc_digest') != workloads.csrc_digest():
        return None, 'profiles/round6_config5_pmc.json was measured on other kernel sources (digest {} at commit {}, now {}): not reported'.format(
            d.get('csrc_digest'), d.get('git_head'), workloads.csrc_digest())
    return d, 'profiles/round6_config5_pmc.json: 2 x FETCH_SIZE + WRITE_SIZE per launch, separate rocprofv3 --pmc passes of `python bench.py --only config5` at commit {}'.format(d.get('git_head'))


def config5_leg(dev, dtype, steps=20, warmup=3):
    """BASELINE config 5's chunk (ppsurf_200nn: P = 200 patch points, rec_batch_size = 25 000, N = 250 000-point synthetic cloud, R = 513 band): the same
    step as the headline through the product's chunk loop -- ONE exact 200-NN search serves the patches and (its first 64 columns) the
    interpolation ids -- with per-kernel times and the HBM rates BASELINE.md section 3 asks for of the k = 200 search and the patch gather."""
    import bench_workloads as workloads
    from ppsurf_amd import ops
    from ppsurf_amd.decoder import DecoderPlan, ChunkPipeline
    from ppsurf_amd.synthetic import make_cloud, make_latents, network_state_dict
    plan = DecoderPlan(network_state_dict('ppsurf', num_pts_local=C5_P), dev, dtype=dtype)
    cloud = make_cloud(C5_N, seed=5)
    pts = torch.from_numpy(cloud).to(dev)
    table = plan.point_table(torch.from_numpy(make_latents(256, C5_N, seed=6)[0]).to(dev))
    chunks, n_band = workloads.band_chunks(cloud, C5_RES, C5_Q, dev)
    chunks = chunks[::max(1, len(chunks) // (steps + warmup))][:steps + warmup]
    pipe = ChunkPipeline(plan, table, pts, pts, K_PROJ, C5_P, same_cloud=True, max_chunk=C5_Q)
    pipe.run(chunks[:warmup])
    timed = chunks[warmup:]
    pipe.run(timed[:4])                                            # allocates the second lane's buffers
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        res = pipe.run(timed, want_occ=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (reps * len(timed))
    assert bool(torch.isfinite(res[-1][1]).all())
    ev = [workloads.HipEvents(6) for _ in timed]
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for i, c in enumerate(timed):
        pipe.run([c], want_occ=True, stage_events=[ev[i].arr])
    torch.cuda.synchronize()
    single = (time.perf_counter() - t1) / len(timed)
    stage_ms = {name: float(np.mean([e.elapsed_ms(j, j + 1) for e in ev])) for j, name in enumerate(STAGES)}

    def timed_op(fn):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn(timed[0])
        torch.cuda.synchronize()
        a.record()
        for c in timed:
            fn(c)
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / len(timed)
    ids = torch.empty((C5_Q, C5_P), dtype=torch.int64, device=dev)
    patches = torch.empty((C5_Q, C5_P, 3), dtype=torch.float32, device=dev)
    knn_ms = timed_op(lambda c: pipe.raw_blocks.query(c, C5_P, out=ids))
    patch_ms = timed_op(lambda c: ops.patch_normalize(pts, c, ids, C5_P, out=patches))
    pmc, src = config5_pmc()
    gbs = lambda nbytes, ms: nbytes / (ms * 1e-3) / 1e9
    out = {'config': 'ppsurf_200nn predict chunk: N=250000-point synthetic cloud, P=200, k=64, rec_batch_size=25000, {} band chunks of the R=513 grid '
                     '(BASELINE config 5; configs/ppsurf_200nn.yaml:8)'.format(len(timed)),
           'queries_per_s': C5_Q / dt, 'ms_per_step': dt * 1e3, 'single_lane_ms_per_step': single * 1e3, 'steps': len(timed), 'repeats': reps,
           'decoder_dtype': dtype, 'band_queries_of_the_shape': n_band, 'stage_ms': stage_ms,
           'spatial_ms': single * 1e3 - sum(stage_ms.values()),
           'kernel_ms': {'knn_blocked_k200': knn_ms, 'patch_normalize_p200': patch_ms},
           'gather_kernels': {
               'knn_blocked_k200': {'ms': knn_ms, 'algorithmic_bytes_per_launch': C5_KNN_BYTES * C5_Q, 'algorithmic_GB_s': gbs(C5_KNN_BYTES * C5_Q, knn_ms),
                                    'hbm_bytes_per_launch': pmc and pmc.get('knn_hbm_bytes_per_launch'),
                                    'hbm_GB_s': pmc and pmc.get('knn_hbm_bytes_per_launch') and gbs(pmc['knn_hbm_bytes_per_launch'], knn_ms)},
               'patch_normalize_p200': {'ms': patch_ms, 'algorithmic_bytes_per_launch': C5_PATCH_BYTES * C5_Q, 'algorithmic_GB_s': gbs(C5_PATCH_BYTES * C5_Q, patch_ms),
                                        'hbm_bytes_per_launch': pmc and pmc.get('patch_hbm_bytes_per_launch'),
                                        'hbm_GB_s': pmc and pmc.get('patch_hbm_bytes_per_launch') and gbs(pmc['patch_hbm_bytes_per_launch'], patch_ms)},
               'hbm_source': src,
               'note': 'ms: the kernel alone on the stream (HIP events over the chunks); algorithmic bytes: query + 200 int64 ids (search), ids + 200 gathered '
                       'points + 200 written points (patches), the 3 MB cloud is cache-resident; hbm_*: counted L2 <-> fabric traffic of a separate '
                       'profiled run, reported only for matching kernel sources'}}
    del pipe, plan, table
    return out


def fit_leg(dev):
    """BASELINE config 3 on one GPU as `pps.py fit` runs it -> {'fit_ms_per_step': ..., 'fit': {...}}."""
    import bench_workloads as workloads
    res = {}
    fit = workloads.FitStep(batch=10, precision='bf16-mixed', device=dev, graph=True)      # as `pps.py fit` runs it: replayed HIP graph, loader thread
    for _ in range(6):
        fit()
    torch.cuda.synchronize()
    from ppsurf_amd.fit import HostGcPacer
    state0 = gpu_state()
    n_fit = 120
    with HostGcPacer() as pacer:                      # as the epoch loop of ppsurf_amd.fit runs its steps
        t0 = time.perf_counter()
        for _ in range(n_fit):
            loss = fit()
            pacer.tick()
        torch.cuda.synchronize()
        res['fit_ms_per_step'] = (time.perf_counter() - t0) / n_fit * 1e3
    res['fit'] = {'config': 'ppsurf_50nn fit step: B=10 shapes x 10000 points, 2000 queries/shape, P=50, bf16-mixed, AdamW; id tables + '
                            'patches built on the device by the loader thread on a second stream, step replayed as a HIP graph (the defaults of pps.py fit)',
                  'steps_timed': n_fit, 'loss': float(loss),
                  'shapes_per_s': 10.0 / (res['fit_ms_per_step'] * 1e-3)}
    res['fit']['gpu_state'] = {'before': state0, 'after': gpu_state(), 'note': 'rocm-smi before / right after the timed steps'}
    res['fit'].update(fit_roofline(res['fit_ms_per_step']))
    # where a slow box loses its time: a second, instrumented pass of 60 steps -- HIP events on the step's stream and on the
    # loader's, host time blocked on the loader thread.  Not part of fit_ms_per_step (the events cost a few microseconds per step).
    fit.start_trace()
    with HostGcPacer() as pacer:
        for _ in range(60):
            fit()
            pacer.tick()
    tr = fit.read_trace()
    res['fit'].update({'queue_busy_ms': tr['queue_busy_ms'], 'loader_wait_ms': tr['loader_wait_ms'], 'batch_wait_ms': tr['batch_wait_ms'],
                       'loader_host_ms': tr['loader_host_ms'], 'traced_call_ms': tr['call_ms'], 'trace_note': tr['note'], 'host_cores': os.cpu_count()})
    fit.close()
    del fit
    return res


def fit_leg_fresh_process(args):
    """The fit leg in a process of its own (`python bench.py --only fit`): `pps.py fit` IS a process of its own, and the step's time depends on which
    stream of torch's pool the loader thread gets -- the same box, the same clocks (fit.gpu_state): 19.8 ms in a fresh process (the loader's stream
    is the first one made), 21.6 ms when the legs in front have made 10, 14 or 18 streams before it (a period of four: in that mapping the
    loader's kernels take an equal share of the chip instead of running in the step's shadow; profiles/NOTES_r6.md section 2).  --fit-in-process
    measures it at the end of this process like rounds 1-5 did."""
    cmd = [sys.executable, os.path.abspath(__file__), '--only', 'fit']
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=1800)
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    if p.returncode != 0 or not lines:
        raise RuntimeError('the fit leg failed in its own process (exit {}): {}'.format(p.returncode, p.stderr[-2000:]))
    res = json.loads(lines[-1])
    res['fit']['process'] = 'a process of its own (python bench.py --only fit), like pps.py fit; --fit-in-process runs it at the end of the bench process'
    return res


def strong_leg(args, rank, world, dev, dist, red_dev, steps, warm):
    """One shape, all ranks: PPS_SHARD=queries.  `steps` = reconstructions timed, `warm` = untimed ones.  Every rank calls this;
    returns the measured figures (the same on every rank up to rank 0's own collective time)."""
    from ppsurf_amd import sharding
    import bench_workloads as workloads
    on = sharding.multi()
    sharding.set_query_sharding(on)
    model = workloads.make_model(RES, P_LOCAL, Q_CHUNK, dev)
    model.network.decoder_dtype = args.dtype
    model.shard_queries = on
    try:
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
        calls, items = sharding.STATS['calls'], sharding.STATS['items']
    finally:
        sharding.profile_collectives(False)
        sharding.set_query_sharding(False)
    mine = sum(r['decoder_queries'] for r in runs)
    total_q = mine
    if dist is not None:
        t = torch.tensor([float(mine)], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t)
        total_q = int(t.item())
    per_rank_q = rank_values(mine / steps, rank, world, dist, red_dev)
    del model
    return {'dt': dt, 'steps': steps, 'warm': warm, 'total_q': total_q, 'coll': coll, 'calls': calls, 'items': items, 'per_rank_q': per_rank_q,
            'latent_s': runs[-1]['latent_s'], 'surface_s': runs[-1]['surface_s'], 'vertices': runs[-1]['vertices']}


STRONG_PARALLELISM = ('PPS_SHARD=queries x{}: per growth / refinement round contiguous query ranges + one all-gather of 4 B/query '
                      '(sharding.sharded_map); encoder passes round-robin + all-reduce of latent sums and counts per wave (sharding.allreduce_latents)')


def strong_block(world, st):
    """The `strong` key of the N > 1 line: SURVEY 8(e)'s query-block sharding of ONE shape over all ranks."""
    dt, steps = st['dt'], st['steps']
    return {'value': st['total_q'] / dt, 'unit': 'queries/s', 'scaling': 'strong', 'shapes_per_hour': 3600.0 * steps / dt,
            's_per_shape': dt / steps, 'shapes_timed': steps, 'warmup_shapes': st['warm'],
            'decoder_queries_per_shape': st['total_q'] / steps,
            'per_rank_decoder_queries_per_shape': st['per_rank_q'],
            'collectives_per_shape': st['calls'] / steps, 'gathered_queries_per_shape': st['items'] / steps,
            'collective_s_per_shape_rank0': st['coll'] / steps, 'collective_share_rank0': st['coll'] / dt,
            'latent_loop_s': st['latent_s'], 'surface_s': st['surface_s'], 'vertices': st['vertices'],
            'parallelism': STRONG_PARALLELISM.format(world),
            'note': 'ONE 100k-point synthetic shape at R=257 reconstructed by all ranks together (latent loop, region growing, Marching Cubes, 10 '
                    'refinement rounds); collective_* = HIP-event time around the data-path collectives on rank 0 (includes waiting for the slowest '
                    'rank of a round); collectives_per_shape counts the per-round all-gathers (the latent all-reduces are in the time, not the count)'}


def strong_line(args, world, st):
    """`--scaling strong`: the same measurement as a bench line of its own."""
    b = strong_block(world, st)
    return {'metric': 'occupancy query-points/sec @ res=257, 50NN', 'value': b['value'], 'unit': 'queries/s', 'n_gpus': world,
            'steps': st['steps'], 'warmup': st['warm'], 'ms_per_step': st['dt'] / st['steps'] * 1e3, 'higher_is_better': True, 'scaling': 'strong',
            'vs_baseline': None, 'dtype': args.dtype, 'dtype_note': DTYPE_NOTE[args.dtype], 'data': 'synthetic',
            'config': {'workload': 'ppsurf_50nn predict, ONE 100k-point synthetic shape at R=257 reconstructed by all ranks together '
                                   '(step = one whole reconstruction: latent loop, region growing, MC, 10 refinement rounds)',
                       'parallelism': STRONG_PARALLELISM.format(world)},
            'shapes_per_hour': b['shapes_per_hour'], 'decoder_queries_per_shape': b['decoder_queries_per_shape'],
            'collective_s_per_shape_rank0': b['collective_s_per_shape_rank0'], 'collective_share_rank0': b['collective_share_rank0'],
            'collectives_per_shape': b['collectives_per_shape']}


DDP_GLOBAL_BATCH = int(os.environ.get('PPS_BENCH_DDP_BATCH', '50'))          # source/base/mp.py:91: `--data.init_args.batch_size 50 // num_gpus` (the env override: tests)


def ddp_fit_leg(rank, world, dev, dist, red_dev, n_steps=40):
    """BASELINE config 3 data-parallel: B = 50 // N shapes per rank (mp.py:91; configs/device_server.yaml:13 writes 12 for its 4 GPUs), the step of
    ppsurf_amd.fit for several ranks (fit.StagedStep: backward in three replayed HIP graphs, bucket k's RCCL all-reduce issued between replays k
    and k + 1; buffer broadcast, mask collective and AdamW behind).  Timed three ways, each between barrier + synchronize with the max over
    ranks: (a) as the product runs it, (b) every bucket all-reduce held back until backward is over (no overlap), (c) the three bucket
    all-reduces alone.  overlap_share = (b - a) / c: the part of the all-reduce time the staged issue hides behind backward."""
    import bench_workloads as workloads
    from ppsurf_amd import sharding
    from ppsurf_amd.fit import HostGcPacer
    b = max(1, DDP_GLOBAL_BATCH // world)
    fit = workloads.FitStepDDP(batch=b, precision='bf16-mixed', device=dev, rank=rank)

    def timed(n):
        with HostGcPacer() as pacer:
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                loss = fit()
                pacer.tick()
            torch.cuda.synchronize()
            mine = time.perf_counter() - t0
            if dist is not None:
                dist.barrier()
        return sharding.max_over_ranks(mine, red_dev) / n * 1e3, float(loss)

    for _ in range(fit.WARMUP_STEPS):                     # eager steps, the capture of the three stage graphs, first replays
        fit()
    t_overlap, loss = timed(n_steps)
    fit.buckets.hold = True                               # reduce(k) between the replays does nothing: finish() sends every bucket behind backward
    for _ in range(3):
        fit()
    t_serial, _ = timed(n_steps)
    fit.buckets.hold = False
    t_overlap2, _ = timed(n_steps)                        # (a) again after (b): the two orders bracket any drift of the box
    t_ar = fit.allreduce_alone_ms(reps=10, dist=dist)
    t_ar = sharding.max_over_ranks(t_ar, red_dev)
    ta = min(t_overlap, t_overlap2)
    out = {'config': 'ppsurf_50nn fit step, data-parallel: B = {} // {} = {} shapes per rank x 10000 points, 2000 queries/shape, P=50, bf16-mixed, AdamW; '
                     'id tables + patches built on the device by the loader thread on a second stream; forward + backward replayed as {} HIP graphs per '
                     'step with bucket k all-reduced between replays k and k+1 (fit.StagedStep), buffer broadcast / mask collective / optimizer eager '
                     'behind (the defaults of pps.py fit under torch.distributed.run)'.format(DDP_GLOBAL_BATCH, world, b, fit.n_stage_graphs()),
           'ranks': world, 'batch_per_rank': b, 'global_batch': b * world,
           'ms_per_step': ta, 'ms_per_step_runs': [t_overlap, t_overlap2], 'shapes_per_s': b * world / (ta * 1e-3),
           'ms_per_step_collectives_behind_backward': t_serial, 'allreduce_ms': t_ar,
           'overlap_share': max(0.0, min(1.0, (t_serial - ta) / t_ar)) if t_ar > 0 else None,
           'gradient_bytes_per_step': fit.bucket_bytes(), 'bucket_dtype': str(fit.buckets.comm_dtype or torch.float32),
           'graphs_captured': fit.n_stage_graphs(), 'capture_failed': bool(fit.core.failed), 'steps_timed': n_steps, 'loss': loss,
           'allreduce_note': 'allreduce_ms = the step\'s three bucket all-reduces issued back to back with nothing else on the GPU (max over ranks, mean '
                             'of 10); overlap_share = (ms_per_step_collectives_behind_backward - ms_per_step) / allreduce_ms, clipped to [0, 1]'}
    fit.close()
    return out


if __name__ == '__main__':
    main()
