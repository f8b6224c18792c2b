"""Where the cycles of one wave of interp_pool_f16x3_kernel go, per pipeline step (a 32 KiB weight chunk): needs the trace build
    python -m ppsurf_amd.build --variant trace -DPPS_TRACE         (then)        PPS_LIB_VARIANT=trace python tools/trace_interp16.py [f16x3|f32]
The build stamps the cycle counter around the four parts of stream_step (copy issue | compute | wait for the own copies | barrier) in wave 0 of
workgroup 0 and sums them; this script runs ONLY the interpolation kernel on a band chunk and prints the averages per step."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench_workloads as workloads
from ppsurf_amd import _lib, ops
from ppsurf_amd.decoder import DecoderPlan
from ppsurf_amd.synthetic import make_cloud, make_latents, network_state_dict

DEV = 'cuda:0'
dtype = sys.argv[1] if len(sys.argv) > 1 else 'f16x3'
L = _lib.lib()
raw = ctypes.CDLL(_lib.LIB_PATH)
plan = DecoderPlan(network_state_dict('ppsurf'), DEV, dtype=dtype)
cloud = make_cloud(100000, seed=42)
pts = torch.from_numpy(cloud).to(DEV)
table = plan.point_table(torch.from_numpy(make_latents(256, 100000, seed=77)[0]).to(DEV))
q = workloads.band_chunks(cloud, 257, 50000, DEV)[0][5]
idx = ops.KnnBlocks(pts).query(q, 64)
pooled = torch.empty((50000, 256), device=DEV)
st = torch.cuda.current_stream().cuda_stream
buf = (ctypes.c_ulonglong * 8)()


def run():
    if dtype == 'f16x3':
        _lib.check(L.pps_interp_pool_f16x3(table.data_ptr(), pts.data_ptr(), q.data_ptr(), idx.data_ptr(), 50000, 64, plan.w['ip_w'].data_ptr(),
                                           plan._w16_t[0].data_ptr(), plan.w['ip_b'].data_ptr(), pooled.data_ptr(), st), 'interp')
    else:
        _lib.check(L.pps_interp_pool_f32(table.data_ptr(), pts.data_ptr(), q.data_ptr(), idx.data_ptr(), 50000, 64, plan.w['ip_w'].data_ptr(),
                                         plan.w['ip_b'].data_ptr(), pooled.data_ptr(), st), 'interp')


run(); torch.cuda.synchronize()
raw.pps_debug_trace_read(buf)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    run()
e1.record(); torch.cuda.synchronize()
raw.pps_debug_trace_read(buf)
v = np.array(list(buf), dtype=np.float64)
n = v[4]
print('{}: kernel {:.3f} ms; wave 0 of workgroup 0, {} pipeline steps: cycles per step (s_memtime ticks)  issue {:.0f} | compute {:.0f} | own-copy wait {:.0f} | barrier {:.0f} '
      '| sum {:.0f}'.format(dtype, e0.elapsed_time(e1) / 5, int(n), v[0] / n, v[1] / n, v[2] / n, v[3] / n, v[:4].sum() / n))
