"""What did a wrong h1 entry of the chain kernel use?  python tools/dbg/head_chain_diag.py  (PPS_LIB_VARIANT = the build under test)
For every h1 entry that differs from the separate head_input kernel: column, wave, tile, and which (table column c', Wx row c'') reproduces it."""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
from ppsurf_amd import _lib, train_ops
import test_gpu_head_chain as T
L = _lib.lib()
dt = torch.bfloat16
NQ = 2003
table, ids, pts, query, wx, w2, b2, w3, b3, wq, bq = T._case(NQ, 64, 4000, 7 + NQ, dt)
rows = NQ * 64
pad = (rows + 255) // 256 * 256
ws = torch.empty((L.pps_head_chain_ws_bytes(),), device='cuda', dtype=torch.uint8)
with torch.no_grad():
    ref = train_ops.head_input(table, ids, pts, query, 64, wx)
h1, y2, y3 = (torch.zeros((pad, 256), device='cuda', dtype=dt) for _ in range(3))
qy = torch.zeros((pad, 64), device='cuda', dtype=dt)
_lib.check(L.pps_head_chain_fwd(table.data_ptr(), ids.data_ptr(), pts.data_ptr(), query.data_ptr(), NQ, 64, 1, wx.data_ptr(), w2.data_ptr(),
                                b2.data_ptr(), w3.data_ptr(), b3.data_ptr(), wq.data_ptr(), bq.data_ptr(), h1.data_ptr(), y2.data_ptr(),
                                y3.data_ptr(), qy.data_ptr(), ws.data_ptr(), torch.cuda.current_stream().cuda_stream), 'x')
torch.cuda.synchronize()
bad = torch.nonzero(h1[:rows] != ref.view(rows, 256)).cpu()
print('bad entries', len(bad))
rel = (query.repeat_interleave(64, dim=0) - pts[ids]).float()                 # [rows, 3]
cols, waves, tiles, gs, js, ss = (collections.Counter() for _ in range(6))
expl = collections.Counter()
tab = table.float()
for r, c in bad[:4000].tolist():
    cols[c] += 1; waves[(r % 256) // 32] += 1; tiles[(r % 32) // 16] += 1
    gs[(c % 64) // 16] += 1; js[c % 8] += 1; ss[2 * (c // 64) + ((c % 16) // 8)] += 1
    got = float(h1[r, c])
    t = tab[ids[r]]                                                          # [256]
    dots = (wx * rel[r]).sum(dim=1)                                          # [256]: Wx row c'' . rel
    # candidates: table column c + Wx row c'' ; table column c' + Wx row c
    cand1 = (t[c] + dots).to(dt).float()
    cand2 = (t + dots[c]).to(dt).float()
    m1 = torch.nonzero(cand1 == got)[:, 0].tolist()
    m2 = torch.nonzero(cand2 == got)[:, 0].tolist()
    if m1 and len(m1) <= 3:
        expl['Wx row of channel c%+d' % (m1[0] - c)] += 1
    elif m2 and len(m2) <= 3:
        expl['table column c%+d' % (m2[0] - c)] += 1
    else:
        expl['other'] += 1
for name, cnt in (('column', cols), ('wave', waves), ('tile t', tiles), ('g', gs), ('j', js), ('s', ss)):
    print(name, sorted(cnt.items()))
print(expl.most_common(12))
