"""Region-growing kernels on byte masks (csrc/pps_grow.hip) against the torch expressions of the driver (source/poco_utils.py:181-196,245-246):
bit-exact, for volume sizes that are and are not multiples of the vector width."""
import numpy as np
import pytest
import torch

from ppsurf_amd import ops

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _torch_dilate(mask, r):
    m = torch.nn.functional.max_pool3d(mask[None, None].float(), kernel_size=2 * r + 1, stride=1, padding=r)
    return m[0, 0] > 0


@pytest.mark.parametrize('shape', [(259, 259, 259), (35, 35, 35), (5, 7, 3), (1, 1, 9), (64, 32, 16), (13, 4, 66), (7, 5, 2), (6, 6, 1), (9, 3, 3)])
@pytest.mark.parametrize('r', [0, 1, 2, 3])
def test_box_dilation_equals_max_pool(shape, r):
    g = torch.Generator().manual_seed(shape[0] * 7 + r)
    mask = (torch.rand(shape, generator=g) < 0.002).to(DEV)
    mask[0, 0, 0] = True
    mask[-1, -1, -1] = True                     # clipping at every border
    got = ops.dilate_box(mask, r)
    assert got.dtype == torch.bool and torch.equal(got, _torch_dilate(mask, r))
    dense = (torch.rand(shape, generator=g) < 0.4).to(DEV)
    assert torch.equal(ops.dilate_box(dense, r), _torch_dilate(dense, r))
    assert not ops.dilate_box(torch.zeros(shape, dtype=torch.bool, device=DEV), r).any()


def test_dilation_of_a_mask_with_a_storage_offset():
    """A contiguous bool tensor that starts 1 / 3 bytes into its storage is passed through unchanged by contiguous(): the kernel's 32-bit fast path
    must not be taken for it."""
    g = torch.Generator().manual_seed(3)
    for off in (1, 3):
        flat = (torch.rand(off + 20 * 12 * 8, generator=g) < 0.05).to(DEV)
        mask = flat[off:].view(20, 12, 8)
        assert mask.is_contiguous() and mask.data_ptr() % 4 == off % 4
        assert torch.equal(ops.dilate_box(mask, 2), _torch_dilate(mask.clone(), 2))


def test_point_list_semantics_of_the_reference():
    """`_dilate_binary` marks arr[p - r : p + r + 1] per point, clipped (poco_utils.py:181-196): the same set as dilating the seed mask."""
    n, r = 40, 2
    rng = np.random.default_rng(0)
    pts = rng.integers(0, n, (300, 3))
    want = np.zeros((n, n, n), dtype=bool)
    for p in pts:
        lo, hi = np.maximum(p - r, 0), np.minimum(p + r + 1, n)
        want[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]] = True
    seeds = torch.zeros((n, n, n), dtype=torch.bool, device=DEV)
    t = torch.from_numpy(pts).to(DEV)
    seeds[t[:, 0], t[:, 1], t[:, 2]] = True
    assert np.array_equal(ops.dilate_box(seeds, r).cpu().numpy(), want)


@pytest.mark.parametrize('n', [259, 33, 6])
def test_frontier_and_band_masks_equal_the_torch_expressions(n):
    g = torch.Generator().manual_seed(n)
    vol = torch.randn((n, n, n), generator=g, dtype=torch.float64)
    vol[torch.rand((n, n, n), generator=g) < 0.5] = float('nan')
    vol[torch.rand((n, n, n), generator=g) < 0.05] = 0.0                           # exact zeros count for BOTH signs (>= 0 and <= 0)
    vol = vol.to(DEV)
    neg, pos, see, band = [(torch.rand((n, n, n), generator=g) < 0.5).to(DEV) for _ in range(4)]
    want = (neg & (vol >= 0) & see) | (pos & (vol <= 0) & see)
    assert torch.equal(ops.grow_frontier(vol, neg, pos, see), want)
    assert torch.equal(ops.grow_band_todo(vol, band), band & torch.isnan(vol))
