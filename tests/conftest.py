"""pytest configuration: markers + shared fixtures.

`-m "not gpu"`: oracle self-checks, host parsers, C-ABI export check and the host-emulated kernels vs the oracle.
`-m gpu`      : parity tests proper — the HIP path through the C ABI of libqmhip.so vs the oracle / golden fixtures.
"""
import os
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

REFERENCE = "/root/reference"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def blobs():
    from qm_control_amd import scenarios
    return scenarios.load_blobs()


@pytest.fixture(scope="session")
def oblobs():
    """the oracle's blobs (numpy front-end); `blobs` are the product's (its own C++ ingestion): no common-mode model input"""
    import pyoracle
    return pyoracle.load_blobs()


@pytest.fixture(scope="session")
def oracle(oblobs):
    import pyoracle
    return pyoracle.Oracle(*oblobs)


def rel_err(a, b):
    a = np.asarray(a, float); b = np.asarray(b, float)
    return float(np.abs(a - b).max() / max(1e-300, np.abs(b).max()))


from blocks import BLOCKS, block_errs, assert_blocks      # noqa: E402,F401  (per-block parity metric: tests/blocks.py)
