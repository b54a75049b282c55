import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu on the GPU box")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden():
    return load_golden


_SD_CACHE = {}


def state_dict_for(seed, style):
    """Seeded oracle weights (cached per session: 34 M parameters take ~1 s to draw)."""
    from oracle import weights
    key = (int(seed), str(style))
    if key not in _SD_CACHE:
        _SD_CACHE[key] = weights.make_state_dict(int(seed), str(style))
    return _SD_CACHE[key]


@pytest.fixture(scope="session")
def make_sd():
    return state_dict_for
