"""Outputs of N launches of the chain kernel into a file, and the comparison of two such files.
    python tools/dbg/head_chain_dump.py run OUT.pt [launches=4]      (PPS_LIB_VARIANT picks the library build)
    python tools/dbg/head_chain_dump.py cmp A.pt B.pt"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
NQ = 2003
if sys.argv[1] == 'run':
    from ppsurf_amd import _lib
    import test_gpu_head_chain as T
    L = _lib.lib()
    dt = torch.bfloat16
    table, ids, pts, query, wx, w2, b2, w3, b3, wq, bq = T._case(NQ, 64, 4000, 7 + NQ, dt)
    rows = NQ * 64
    pad = (rows + 255) // 256 * 256
    ws = torch.empty((L.pps_head_chain_ws_bytes(),), device='cuda', dtype=torch.uint8)
    outs = []
    for rep in range(int(sys.argv[3]) if len(sys.argv) > 3 else 4):
        h1, y2, y3 = (torch.zeros((pad, 256), device='cuda', dtype=dt) for _ in range(3))
        qy = torch.zeros((pad, 64), device='cuda', dtype=dt)
        _lib.check(L.pps_head_chain_fwd(table.data_ptr(), ids.data_ptr(), pts.data_ptr(), query.data_ptr(), NQ, 64, 1, wx.data_ptr(), w2.data_ptr(),
                                        b2.data_ptr(), w3.data_ptr(), b3.data_ptr(), wq.data_ptr(), bq.data_ptr(), h1.data_ptr(), y2.data_ptr(),
                                        y3.data_ptr(), qy.data_ptr(), ws.data_ptr(), torch.cuda.current_stream().cuda_stream), 'x')
        torch.cuda.synchronize()
        outs.append([t[:rows].cpu() for t in (h1, y2, y3, qy)])
    torch.save(outs, sys.argv[2])
else:
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    ref = a[0]
    for which, runs in (('A', a), ('B', b)):
        for r, out in enumerate(runs):
            line = []
            for name, x, y in zip(('h1', 'y2', 'y3', 'qy'), out, ref):
                d = (x != y)
                rows_bad = d.any(dim=1)
                cols = torch.nonzero(d.any(dim=0))[:, 0].tolist()
                units = sorted(set((torch.nonzero(rows_bad)[:, 0] // 256).tolist()))
                worst = float((x.float() - y.float()).abs().max())
                line.append('{} rows {} units {} cols {} worst {:.3g}'.format(name, int(rows_bad.sum()), len(units), len(cols), worst))
                if name == 'h1' and units:
                    line.append('[units >= 256: {} of {}]'.format(sum(u >= 256 for u in units), len(units)))
                if name == 'h1' and 0 < len(units) <= 8:
                    line.append('[units {} cols {}]'.format(units, cols[:8]))
            print(which, r, ' | '.join(line))
