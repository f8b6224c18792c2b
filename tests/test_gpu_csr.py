"""pps_csr_build (csrc/pps_csr.hip): the CSR of an id table by a counting sort on the device must be EXACTLY what the stable library sort +
binary search returned until round 5 (so every scatter of the backward pass adds its contributions in the same order: gradients bit-equal)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _reference(flat, rows):
    keys, order = torch.sort(flat.to(torch.int32), stable=True)
    offsets = torch.searchsorted(keys, torch.arange(rows + 1, dtype=keys.dtype, device=flat.device))
    return order, offsets


@pytest.mark.parametrize('b,m,k,n', [(1, 1, 1, 1), (3, 39, 16, 39), (10, 2000, 64, 10000), (10, 10000, 16, 10000), (2, 2500, 1, 625), (50, 2000, 64, 10000),
                                     (1, 5000, 16, 2049), (4, 4097, 3, 2048)])
def test_csr_of_a_batch_table_equals_stable_sort_and_searchsorted(b, m, k, n):
    from ppsurf_amd import train_ops
    g = torch.Generator(device='cuda').manual_seed(b * 1000 + m + k)
    ids = torch.randint(0, n, (b, m, k), device='cuda', generator=g)
    if n > 8:                                                      # a heavy row (a point that is everybody's neighbour) and an empty one
        ids[:, ::3, 0] = 5
        ids[ids == 7] = 6
    flat, order, offsets = train_ops.csr_build_table(ids, m * k, n, b * n, False)
    ref_flat = (ids + torch.arange(b, device='cuda').view(b, 1, 1) * n).reshape(-1)
    ro, rf = _reference(ref_flat, b * n)
    assert torch.equal(flat, ref_flat) and torch.equal(order, ro) and torch.equal(offsets, rf)
    assert order.dtype == torch.int64 and offsets.dtype == torch.int64 and int(offsets[-1]) == b * m * k
    o2, f2 = train_ops.csr_build(ref_flat, b * n)                  # the flat form (per_item = 0) the backward pass falls back to
    assert torch.equal(o2, ro) and torch.equal(f2, rf)


def test_csr_of_an_upsampling_table_counts_minus_one_as_row_zero_and_is_run_to_run_identical():
    from ppsurf_amd import train_ops
    b, m, n = 10, 10000, 2500
    g = torch.Generator(device='cuda').manual_seed(3)
    ids = torch.randint(-1, n, (b, m, 1), device='cuda', generator=g)
    flat, order, offsets = train_ops.csr_build_table(ids, m, n, b * n, True)
    t = torch.where(ids > -1, ids, torch.zeros_like(ids))
    ref_flat = (t + torch.arange(b, device='cuda').view(b, 1, 1) * n).reshape(-1)
    ro, rf = _reference(ref_flat, b * n)
    assert torch.equal(flat, ref_flat) and torch.equal(order, ro) and torch.equal(offsets, rf)
    for _ in range(20):                                            # the fill phase uses atomics; the rank phase makes the result independent of them
        _, o, f = train_ops.csr_build_table(ids, m, n, b * n, True)
        assert torch.equal(o, order) and torch.equal(f, offsets)


def test_segment_sum_through_the_new_csr_equals_index_add():
    """End to end: the backward of the row gather (segmented sum in CSR order) against a float64 index_add."""
    from ppsurf_amd import train_ops
    x = torch.randn(5000, 64, device='cuda', requires_grad=True)
    idx = torch.randint(0, 5000, (40000,), device='cuda')
    y = train_ops._GatherRows.apply(x, idx)
    gy = torch.randn_like(y)
    y.backward(gy)
    ref = torch.zeros(5000, 64, device='cuda', dtype=torch.float64).index_add_(0, idx, gy.double())
    assert torch.allclose(x.grad.double(), ref, atol=1e-4, rtol=0)
