"""Weight-gradient kernel of the fused row layers alone (pps_rows_layer_bwd with dx = NULL), per layer shape; PPS_LIB_VARIANT selects an ablation
build (python -m ppsurf_amd.build --variant NAME -DPPS_ABL_DW_NOMATH / -DPPS_ABL_DW_NOPROD).  Usage: python tools/time_rows_dw.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppsurf_amd import _lib          # noqa: E402

DEV = 'cuda:0'
L = _lib.lib()
for rows, cin, cout, bn in [(1000000, 64, 64, True), (1000000, 128, 256, True), (1280000, 256, 256, False), (1280000, 256, 64, False)]:
    x = torch.randn(rows, cin, device=DEV).to(torch.bfloat16)
    y = torch.randn(rows, cout, device=DEV).to(torch.bfloat16)
    gy = torch.randn(rows, cout, device=DEV).to(torch.bfloat16)
    w = torch.randn(cout, cin, device=DEV)
    aff = torch.stack([torch.ones(cin, device=DEV), torch.zeros(cin, device=DEV)])
    gamma, save, daff = torch.ones(cout, device=DEV), torch.stack([torch.zeros(cout, device=DEV), torch.ones(cout, device=DEV)]), torch.ones(2, cout, device=DEV)
    dw, db, dg, dbt = torch.empty(cout, cin, device=DEV), torch.empty(cout, device=DEV), torch.empty(cout, device=DEV), torch.empty(cout, device=DEV)
    ws = torch.empty(L.pps_rows_layer_ws_bytes(cin, cout), device=DEV, dtype=torch.uint8)
    st = torch.cuda.current_stream().cuda_stream

    def run():
        rc = L.pps_rows_layer_bwd(x.data_ptr(), y.data_ptr(), gy.data_ptr(), rows, cin, cout, 1, aff.data_ptr(), aff.data_ptr() + 4 * cin, 1, w.data_ptr(),
                                  gamma.data_ptr() if bn else None, save.data_ptr() if bn else None, daff.data_ptr() if bn else None, None, None, None,
                                  dw.data_ptr(), db.data_ptr(), dg.data_ptr() if bn else None, dbt.data_ptr() if bn else None, ws.data_ptr(), st)
        assert rc == 0
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    gb = rows * ((2 if bn else 1) * cout + cin) * 2 / 1e9
    print('{} {}->{} bn={} : {:.3f} ms  {:.2f} TB/s'.format(rows, cin, cout, int(bn), ms, gb / ms))
