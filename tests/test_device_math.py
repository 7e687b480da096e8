"""The hand-written replacements of library functions in the kernels (qm_dev_common.h), evaluated on the host through the emulator build: qm_sincos (Cody-Waite +
minimax kernels instead of the device library's full-range sin / cos), qm_frcp (reciprocal estimate + one third-order step instead of an IEEE division), qm_log
(frexp + atanh series).  Checked against numpy's long double."""
import ctypes as C
import numpy as np
import emu_harness

_dp = C.POINTER(C.c_double)


def _call(name, x):
    lib = C.CDLL(emu_harness.build()); x = np.ascontiguousarray(x, float); outs = [np.zeros_like(x) for _ in range(2 if name == "emu_sincos" else 1)]
    getattr(lib, name)(C.c_int(x.size), x.ctypes.data_as(_dp), *[o.ctypes.data_as(_dp) for o in outs])
    return outs


def test_sincos_matches_long_double_over_the_range_angles_can_take():
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-s, s, 200000) for s in (1.0, 7.0, 100.0, 9.0e4)] + [np.arange(-40, 41) * (np.pi / 2), np.arange(-40, 41) * (np.pi / 4), [0.0, -0.0, 1e-300, 5e-324]])
    s, c = _call("emu_sincos", x); xl = x.astype(np.longdouble)
    assert np.abs(s - np.sin(xl)).max() < 2.5e-16 and np.abs(c - np.cos(xl)).max() < 2.5e-16
    assert np.abs(s * s + c * c - 1.0).max() < 5e-16


def test_sincos_over_its_whole_range_and_nan_beyond():
    """round 5: no library path — the two-constant reduction holds to |x| < 2^31 (absolute error of the reduced argument < 8e-17, qm_dev_common.h); beyond that, and for
    infinities / NaN, the pair is NaN (the solve then reports a failure) — the rounds 1-4 fallback `sin(x); cos(x)` was inlined at every call site: most of the kinematics kernels' code"""
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.uniform(-s, s, 100000) for s in (1.0e5, 1.0e7, 2.0e9)] + [[1.0e5, -3.0e7, 2147483647.0, -2147483647.5], np.arange(1, 1300) * 1.0e6 * (np.pi / 2)])
    s, c = _call("emu_sincos", x); xl = x.astype(np.longdouble)
    assert np.abs(s - np.sin(xl)).max() < 3e-16 and np.abs(c - np.cos(xl)).max() < 3e-16
    bad = np.array([2147483648.0, -2147483648.0, 1.0e10, -1.0e300, np.inf, -np.inf, np.nan])
    s, c = _call("emu_sincos", bad)
    assert np.isnan(s).all() and np.isnan(c).all()


def test_fast_reciprocal_and_log_are_within_an_ulp_or_two():
    rng = np.random.default_rng(1)
    x = np.concatenate([10.0 ** rng.uniform(-12, 12, 200000), rng.uniform(0.5, 2.0, 100000)])
    (r,) = _call("emu_frcp", x)
    assert (np.abs(r * x.astype(np.longdouble) - 1.0)).max() < 4.5e-16
    (lg,) = _call("emu_log", x); ref = np.log(x.astype(np.longdouble))
    assert (np.abs(lg - ref) <= 1e-15 * np.maximum(1.0, np.abs(ref))).all()
