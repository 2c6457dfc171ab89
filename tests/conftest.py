"""Shared pytest configuration.

Markers: ``gpu`` = needs a real MI355X (run by ``pytest -m gpu`` on the GPU box);
everything else must pass on a CPU-only container.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: test needs a real MI355X GPU (HIP extension + cuda:0)')


@pytest.fixture(scope='session')
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name + '.npz'))
    return load


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no GPU is visible, so that an unmarked
    ``pytest tests`` still works in the build container."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for it in items:
        if 'gpu' in it.keywords:
            it.add_marker(skip)
