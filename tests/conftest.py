import json
import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'timeout: per-test time limit (pytest-timeout)')


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        # a test that stops making progress fails after 10 minutes instead of holding the GPU box for the whole tier (the longest test takes ~100 s);
        # pytest-timeout is in the image, the marker is inert without it
        if config.pluginmanager.hasplugin('timeout'):
            for item in items:
                if 'gpu' in item.keywords and item.get_closest_marker('timeout') is None:
                    item.add_marker(pytest.mark.timeout(600, method='thread'))       # (thread: also fires inside a blocked HIP call)
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)
