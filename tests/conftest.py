"""pytest configuration: the `gpu` marker and import paths.

`-m "not gpu"` runs in the build container (no GPU): oracle vs goldens, host logic,
C-ABI load/exports, gloo sharding.  `-m gpu` runs on the MI355X box: HIP path vs oracle.
"""
import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PKG_NAME = "handposeestimation-with-3d-cnns_amd"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    """The product package (its directory name has hyphens, so import it by string)."""
    return importlib.import_module(PKG_NAME)


@pytest.fixture(scope="session")
def synth():
    return importlib.import_module(PKG_NAME + ".synth")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def golden_names():
    with open(os.path.join(ROOT, "tests", "golden", "MANIFEST.txt")) as f:
        return [ln.strip() for ln in f if ln.strip() and not ln.startswith("#")]
