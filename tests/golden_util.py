"""Helpers shared by the test modules: golden fixtures and formula-filled state dicts."""
import json
import os

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, 'tests', 'golden')


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False))


_MANIFEST = None


def manifest(prefix):
    """[(name, shape)] of the reference module recorded under `prefix` by make_golden.py."""
    global _MANIFEST
    if _MANIFEST is None:
        with open(os.path.join(GOLDEN, 'manifest.json')) as f:
            _MANIFEST = json.load(f)
    return [(k, tuple(s)) for k, s in _MANIFEST[prefix]]


def filled_sd(prefix, key=None, as_torch=True):
    """Regenerate the formula-filled state dict of golden case `prefix` ({prefix+name: tensor})."""
    from ppsurf_amd.synthetic import fill_param
    names = manifest(key if key is not None else prefix)
    sd = {prefix + k: fill_param(prefix + k, s) for k, s in names}
    return {k: torch.from_numpy(v) for k, v in sd.items()} if as_torch else sd


def sd_digest(sd):
    from ppsurf_amd.synthetic import state_dict_digest
    return state_dict_digest({k: (v.numpy() if torch.is_tensor(v) else v) for k, v in sd.items()})
