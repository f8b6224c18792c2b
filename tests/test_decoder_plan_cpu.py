"""CPU checks of the host side of the decoder: C ABI loads and exports every declared symbol, the packing is a
bijection, and the folded/composed/packed weight images reproduce the reference's from_latent (golden fixture)."""
import os
import re

import numpy as np
import torch

from golden_util import REPO, load_golden, filled_sd
from ppsurf_amd import _lib
from ppsurf_amd.decoder import DecoderPlan, pack_dense, pack_xyz
from ppsurf_amd.synthetic import make_latents
import emulate


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(REPO, 'include', 'ppsurf_amd.h')).read()
    declared = set(re.findall(r'\b(pps_[a-z0-9_]+)\s*\(', header))
    assert declared, 'no entry points parsed from the header'
    lib = _lib.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), 'libppsurf_amd.so does not export ' + name
    assert declared == set(_lib.SIGNATURES.keys())
    assert lib.pps_abi_version() == 2


def test_pack_roundtrip_and_padding():
    rng = np.random.default_rng(0)
    for out, inp in ((256, 256), (64, 256), (128, 64), (2, 256), (4096, 64)):
        w = rng.standard_normal((out, inp)).astype(np.float32)
        p = pack_dense(w)
        assert p.shape[0] == _lib.lib().pps_packed_dense_floats(out, inp) == ((out + 31) // 32 * 32) * ((inp + 15) // 16 * 16)
        assert np.array_equal(emulate.unpack_dense(p, out, inp).astype(np.float32), w)
    w = rng.standard_normal((64, 3)).astype(np.float32)
    assert np.array_equal(emulate.unpack_xyz(pack_xyz(w), 64).astype(np.float32), w)
    # documented A-operand order: packed[ob][kb][l][s] = W[16 ob + (l & 15)][16 kb + 4 (l >> 4) + s]
    w = np.arange(32 * 32, dtype=np.float32).reshape(32, 32)
    p = pack_dense(w).reshape(2, 2, 64, 4)
    assert p[1, 0, 17, 2] == w[16 + 1, 4 * 1 + 2] and p[0, 1, 63, 3] == w[15, 16 + 12 + 3]


def test_folded_weights_reproduce_reference_from_latent():
    g = load_golden('ppsurf_from_latent')
    plan = DecoderPlan(filled_sd('', key='ppsurf'), 'cpu', dtype='f32')
    w = {k: v.numpy() for k, v in plan.w.items()}
    cloud = g['cloud']
    logits, _ = emulate.decode(w, make_latents(256, cloud.shape[0], 77)[0], cloud, g['query'], g['proj_ids'][0], g['patches'])
    np.testing.assert_allclose(logits, g['logits'][0].T, rtol=1e-4, atol=1e-4)
