"""Run-to-run identity of the chain kernel over many launches: python tools/dbg/head_chain_repeat.py [launches=100] [nq=2003]
(PPS_LIB_VARIANT picks the library build).  Prints the number of launches whose h1/y2/y3/qy differ from the first launch's and the rows affected."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
from ppsurf_amd import _lib
import test_gpu_head_chain as T
L = _lib.lib()
dt = torch.bfloat16
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 2003
table, ids, pts, query, wx, w2, b2, w3, b3, wq, bq = T._case(nq, 64, 4000, 7 + nq, dt)
rows = nq * 64
pad = (rows + 255) // 256 * 256
ws = torch.empty((L.pps_head_chain_ws_bytes(),), device='cuda', dtype=torch.uint8)
first, bad_launches, bad_rows = None, 0, 0
for rep in range(reps):
    h1, y2, y3 = (torch.zeros((pad, 256), device='cuda', dtype=dt) for _ in range(3))
    qy = torch.zeros((pad, 64), device='cuda', dtype=dt)
    _lib.check(L.pps_head_chain_fwd(table.data_ptr(), ids.data_ptr(), pts.data_ptr(), query.data_ptr(), nq, 64, 1, wx.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                                    w3.data_ptr(), b3.data_ptr(), wq.data_ptr(), bq.data_ptr(), h1.data_ptr(), y2.data_ptr(), y3.data_ptr(), qy.data_ptr(),
                                    ws.data_ptr(), torch.cuda.current_stream().cuda_stream), 'x')
    torch.cuda.synchronize()
    out = (h1[:rows], y2[:rows], y3[:rows], qy[:rows])
    if first is None:
        first = [o.clone() for o in out]
        continue
    bad = torch.zeros(rows, dtype=torch.bool, device='cuda')
    for a, b in zip(out, first):
        bad |= (a != b).any(dim=1)
    n = int(bad.sum())
    bad_launches += n > 0
    bad_rows += n
print('variant {!r}: {} launches of {} units each: {} launches differ from the first ({} rows)'.format(
    os.environ.get('PPS_LIB_VARIANT', ''), reps, pad // 256, bad_launches, bad_rows))
