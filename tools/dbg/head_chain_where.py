"""Where do two runs of the chain kernel differ?  (rows -> unit, position of the unit in its workgroup's sequence)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
from ppsurf_amd import train_ops, _lib
import test_gpu_head_chain as T
L = _lib.lib()
dt = torch.bfloat16
for nq in (300, 2003):
    table, ids, pts, query, wx, w2, b2, w3, b3, wq, bq = T._case(nq, 64, 4000, 7 + nq, dt)
    rows = nq * 64
    pad = (rows + 255) // 256 * 256
    units = pad // 256
    grid = min(units, 256)
    outs = []
    for rep in range(4):
        h1, y2, y3 = (torch.zeros((pad, 256), device='cuda', dtype=dt) for _ in range(3))
        qy = torch.zeros((pad, 64), device='cuda', dtype=dt)
        ws = torch.empty((L.pps_head_chain_ws_bytes(),), device='cuda', dtype=torch.uint8)
        _lib.check(L.pps_head_chain_fwd(table.data_ptr(), ids.data_ptr(), pts.data_ptr(), query.data_ptr(), nq, 64, 1, wx.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                                        w3.data_ptr(), b3.data_ptr(), wq.data_ptr(), bq.data_ptr(), h1.data_ptr(), y2.data_ptr(), y3.data_ptr(), qy.data_ptr(),
                                        ws.data_ptr(), torch.cuda.current_stream().cuda_stream), 'x')
        torch.cuda.synchronize()
        outs.append((h1[:rows].clone(), y2[:rows].clone(), y3[:rows].clone(), qy[:rows].clone()))
    for name, i in (('h1', 0), ('y2', 1), ('y3', 2), ('qy', 3)):
        bad = torch.zeros(rows, dtype=torch.bool, device='cuda')
        for o in outs[1:]:
            bad |= (o[i] != outs[0][i]).any(dim=1)
        r = torch.nonzero(bad)[:, 0].cpu()
        u = (r // 256)
        info = sorted(set((int(x), int(x) // grid, (units - 1 - int(x)) // grid) for x in u.tolist()))[:12]
        cols = torch.nonzero(((outs[1][i] != outs[0][i]) | (outs[2][i] != outs[0][i])).any(dim=0))[:, 0].tolist()[:16]
        print(nq, name, 'rows differing', int(bad.sum()), 'of', rows, '(unit, pass of its workgroup, passes after it):', info, 'cols', cols)
