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
    ``pytest tests`` still works in the build container.  Every test gets a time limit (pytest-timeout, if installed): a hung kernel
    or a stuck rendezvous fails ONE test instead of eating the whole run."""
    import torch
    if config.pluginmanager.hasplugin('timeout'):
        for it in items:
            if it.get_closest_marker('timeout') is None:
                it.add_marker(pytest.mark.timeout(900))
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for it in items:
        if 'gpu' in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope='session')
def spread_gate():
    """Gates of the trajectory-level comparisons (weights / scores after truncated GN-CG fits), DERIVED from the reference's own
    run-to-run spread instead of asserted constants: tests/golden/g_spread.npz holds, per output, the largest deviation of the
    reference from itself over 7 re-runs (features scaled by a few ulp; 1, 2, 4 instead of 8 threads), relative to max|value|
    (oracle/make_golden_r2.py: spread).  gate(key) = 2 x that spread for the CPU restatement (same formulation as the reference;
    measured 0.02-0.9 x spread on G3/G4); the HIP tests pass mult=3: the HIP path is an 8th draw from the same noise -- another
    summation order AND the low-resolution form of the normal equations -- and measured 0.8-2.2 x the 7-run spread on G4
    (1.61e-4 vs a 7.5e-5 spread on the projection weights).  ``at_most`` keeps a previously asserted constant as an upper
    bound so that a derived gate never loosens a test."""
    s = np.load(os.path.join(GOLDEN, 'g_spread.npz'))

    def gate(key, mult=2.0, at_most=None):
        g = mult * float(s[key + '_spread'])
        return g if at_most is None else min(g, at_most)
    return gate
