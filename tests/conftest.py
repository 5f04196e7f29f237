import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """Serial, bit-reproducible oracle build (test infrastructure)."""
    import oracle as O
    return O.Oracle(fast=False)


@pytest.fixture(scope="session")
def oracle_fast():
    """OpenMP/vectorised oracle build: same source, same bits (checked in test_oracle_kat.py)."""
    import oracle as O
    return O.Oracle(fast=True)


@pytest.fixture(scope="session")
def nb():
    """The product package (directory name has a hyphen, hence importlib)."""
    return importlib.import_module("mini-nbody_amd")


def has_gpu():
    try:
        return os.path.exists("/dev/kfd") and len(os.listdir("/sys/class/kfd/kfd/topology/nodes")) > 1
    except OSError:
        return False
