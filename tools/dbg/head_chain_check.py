"""Bitwise checks of the chain kernel: run-to-run determinism and h1 / y2 / y3 / qy against the separate launches."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
from ppsurf_amd import train_ops, _lib
import test_gpu_head_chain as T

L = _lib.lib()
for dt in (torch.bfloat16, torch.float16):
    for nq, k, n in ((300, 64, 4000), (2003, 64, 10000), (20000, 64, 100000)):
        table, ids, pts, query, wx, w2, b2, w3, b3, wq, bq = T._case(nq, k, n, 7 + nq, dt)
        rows = nq * k
        pad = (rows + 255) // 256 * 256
        outs = []
        for rep in range(3):
            h1, y2, y3 = (torch.zeros((pad, 256), device='cuda', dtype=dt) for _ in range(3))
            qy = torch.zeros((pad, 64), device='cuda', dtype=dt)
            ws = torch.empty((L.pps_head_chain_ws_bytes(),), device='cuda', dtype=torch.uint8)
            _lib.check(L.pps_head_chain_fwd(table.data_ptr(), ids.data_ptr(), pts.data_ptr(), query.data_ptr(), nq, k, train_ops._code(dt), wx.data_ptr(), w2.data_ptr(),
                                            b2.data_ptr(), w3.data_ptr(), b3.data_ptr(), wq.data_ptr(), bq.data_ptr(), h1.data_ptr(), y2.data_ptr(), y3.data_ptr(),
                                            qy.data_ptr(), ws.data_ptr(), torch.cuda.current_stream().cuda_stream), 'x')
            torch.cuda.synchronize()
            outs.append((h1[:rows].clone(), y2[:rows].clone(), y3[:rows].clone(), qy[:rows].clone()))
        det = [all(torch.equal(a, b) for a, b in zip(outs[0], o)) for o in outs[1:]]
        with torch.no_grad(), torch.autocast('cuda', dtype=dt):
            s1 = train_ops.head_input(table, ids, pts, query, k, wx)
            s2 = train_ops.rows_layer(train_ops.Act(s1, None, True), w2, b2, None, True)
            s3 = train_ops.rows_layer(s2, w3, b3, None, True)
            sq = train_ops.rows_layer(s3, wq, bq, None, False)
        diffs = [float((a.float() - b.float()).abs().max()) for a, b in zip(outs[0], (s1, s2.raw, s3.raw, sq.raw))]
        frac = [float((a != b).float().mean()) for a, b in zip(outs[0], (s1, s2.raw, s3.raw, sq.raw))]
        print(dt, nq, 'deterministic', det, 'max |fused - separate| h1 y2 y3 qy', diffs, 'fraction of differing entries', ['%.2e' % f for f in frac])
        if nq == 300:
            print('  per tensor run-to-run equal:', [bool(torch.equal(a, b)) for a, b in zip(outs[0], outs[1])])
            bad = torch.nonzero(outs[0][0] != s1)
            print('  h1 mismatches vs head_input:', bad.shape[0], 'rows', sorted(set(bad[:, 0].tolist()))[:20], 'cols', sorted(set(bad[:, 1].tolist()))[:40])
            bad2 = torch.nonzero(outs[0][0] != outs[1][0])
            print('  h1 run0 vs run1:', bad2.shape[0], bad2[:8].tolist())
            r = int(bad[0, 0]); c = int(bad[0, 1])
            print('  row', r, 'unit', r // 256, 'wave', (r % 256) // 32, 'tile', (r % 32) // 16, 'n', r % 16, 'col', c, 'fused', float(outs[0][0][r, c]), 'separate', float(s1[r, c]),
                  'table', float(table[ids[r], c]))
