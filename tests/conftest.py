"""pytest configuration: registers the `gpu` marker and puts the product tree on sys.path.

`rq-vae-recommender_amd/` mirrors the reference repo root (same module paths: modules.quantize,
modules.rqvae, init.kmeans, ...), so it is added to sys.path exactly as the reference root would be.
The oracle (oracle/) is imported by tests only.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "rq-vae-recommender_amd")
GOLDEN = os.path.join(ROOT, "tests", "golden")

for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def load_golden(name):
    import numpy as np
    return dict(np.load(os.path.join(GOLDEN, name)))
