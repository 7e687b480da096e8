"""tests/blocks.py — TEST INFRASTRUCTURE: the per-block parity metric shared by the tests, __graft_entry__.smoke() and the sweep tools."""
import numpy as np

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
