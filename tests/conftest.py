import glob
import importlib
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "oracle", "py")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_pkg():
    """the package directory is `mental-poker_amd` (not a valid identifier): import it by name"""
    try:
        # torch ships a HIP runtime of its own: it has to be in the process BEFORE libmpshuffle.so pulls in /opt/rocm's, or
        # torch.cuda finds no GPU afterwards (bench.py imports torch first for the same reason)
        import torch  # noqa: F401
    except ImportError:
        pass
    return importlib.import_module("mental-poker_amd")


def golden_cases():
    return sorted(glob.glob(os.path.join(GOLDEN, "shuffle_*.json")))


def load_json(path):
    with open(path) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def coracle():
    import coracle as co
    co.build()
    return co


@pytest.fixture(scope="session")
def mp():
    return load_pkg()
