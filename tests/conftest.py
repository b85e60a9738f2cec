"""pytest configuration: registers the `gpu` marker and puts the repo root on sys.path."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class Fixture:
    """A tests/golden/*.npz file. int8-grid arrays are dequantised on access (value = q / scale)."""

    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN, name + ".npz"))

    def __getitem__(self, key):
        a = self.z[key]
        if key + "#scale" in self.z.files:
            a = a.astype(np.float32) / np.float32(self.z[key + "#scale"])
        return a

    def weights(self, prefix="w/"):
        return {k[len(prefix):]: self[k] for k in self.z.files if k.startswith(prefix) and not k.endswith("#scale")}

    def keys(self):
        return [k for k in self.z.files if not k.endswith("#scale")]


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = Fixture(name)
        return cache[name]

    return load
