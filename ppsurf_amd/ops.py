"""Thin torch-facing wrappers of the spatial-query entry points of the C ABI (device tensors in, device tensors out)."""
import torch

from . import _lib


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def knn_point_major(pts: torch.Tensor, query: torch.Tensor, k: int, return_d2: bool = False):
    """pts [n,3], query [m,3] float32 contiguous on the GPU -> int64 [m,k] (sorted by (d2, index)); k <= min(n, 64)."""
    if not (pts.is_cuda and query.is_cuda):
        raise _lib.PpsError('pps_knn_f32 needs device tensors; there is no CPU fallback')
    pts = pts.contiguous().float()
    query = query.contiguous().float()
    m = query.shape[0]
    idx = torch.empty((m, k), dtype=torch.int64, device=pts.device)
    d2 = torch.empty((m, k), dtype=torch.float32, device=pts.device) if return_d2 else None
    _lib.check(_lib.lib().pps_knn_f32(pts.data_ptr(), pts.shape[0], query.data_ptr(), m, int(k), idx.data_ptr(),
                                      d2.data_ptr() if return_d2 else None, _stream(pts)), 'pps_knn_f32')
    return (idx, d2) if return_d2 else idx


def patch_normalize(raw: torch.Tensor, query: torch.Tensor, idx: torch.Tensor, p: int) -> torch.Tensor:
    """raw [n,3], query [q,3], idx int64 [q,>=p] -> patches [q,p,3] in patch space (ppsurf_data_loader.py:91-123)."""
    raw = raw.contiguous().float()
    query = query.contiguous().float()
    assert idx.dtype == torch.int64 and idx.stride(1) == 1
    q = query.shape[0]
    out = torch.empty((q, p, 3), dtype=torch.float32, device=raw.device)
    _lib.check(_lib.lib().pps_patch_normalize_f32(raw.data_ptr(), query.data_ptr(), idx.data_ptr(), idx.stride(0), q, int(p),
                                                  out.data_ptr(), _stream(raw)), 'pps_patch_normalize_f32')
    return out
