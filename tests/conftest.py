import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def bunny_octree():
    from raytracer_amd import scenes
    return scenes.default_octree()


@pytest.fixture(scope="session")
def test_sky():
    import numpy as np
    return np.random.default_rng(1).integers(0, 256, size=(6, 64, 64, 4), dtype=np.uint8)
