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


# ---- per-block parity assertions (north_star: 1e-6 relative on optimal state / input trajectories and WBC torques).
# One relative error over a whole array would let the largest block hide the others (contact forces ~134 N next to joint velocities
# ~0.1 rad/s), so every physical block is held to the tolerance on its own scale: max|a − b| <= tol · max(max|b|, floor).
BLOCKS = {
    "x":   [("momentum", slice(0, 6), 1e-2), ("base pose", slice(6, 12), 1e-2), ("joints", slice(12, 30), 1e-2)],          # centroidal state (30)
    "u":   [("contact forces", slice(0, 12), 1.0), ("joint velocities", slice(12, 30), 1e-2)],                             # input (30)
    "wbc": [("vdot", slice(0, 24), 1e-2), ("contact forces", slice(24, 36), 1.0), ("torques", slice(36, 54), 1e-1)],       # WBC output (54)
}


def block_errs(a, b, kind):
    """{block name: max|a − b| / max(max|b|, floor)} over the last axis' blocks"""
    a = np.asarray(a, float); b = np.asarray(b, float)
    assert a.shape == b.shape, (a.shape, b.shape)
    out = {}
    for name, sl, floor in BLOCKS[kind]:
        d = np.abs(a[..., sl] - b[..., sl])
        out[name] = float(d.max() / max(float(np.abs(b[..., sl]).max()), floor)) if d.size else 0.0
    return out


def assert_blocks(a, b, kind, tol, what=""):
    errs = block_errs(a, b, kind)
    bad = {k: v for k, v in errs.items() if not v <= tol}
    assert not bad, "%s: block errors above %.1e: %s (all: %s)" % (what, tol, bad, errs)
