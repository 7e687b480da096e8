"""The GENERAL HoQp cascade of the product (csrc/kernels/k_hoqp.h behind qmhip_hoqp_solve: arbitrary task hierarchies, own inequality rows below the first level,
the reference's row / slack pairing of HoQp.cpp:46,152-158) against
  * the oracle's general level solve (oracle/src/wbc.h) and the literal 80-bit cascade (tests/hoqp_literal.py)          [-m "not gpu": kernel on the host emulator]
  * the same plus the specialised whole-body-control kernel on the controller's own task matrices                       [-m gpu: through the C ABI]"""
import numpy as np
import pytest
from test_hoqp_literal import _random_cascade
from wbc_cases import MODES, hard_wbc_inputs, random_wbc_inputs

SHAPES = [[(3, 5), (9, 4)], [(1, 6), (11, 5)], [(3, 5), (2, 4), (7, 0)], [(2, 4), (2, 6), (2, 3), (6, 0)], [(4, 0), (3, 0), (5, 0)]]


def _wbc_tasks(oracle, c, variant=0):
    oracle.wbc_reset(); oracle.wbc(c["xd"], c["il"], c["rbd"], c["mode"], 0.002, c["time"], mpc_variant=bool(variant))
    ref, st, d = oracle.wbc(c["xd"], c["ud"], c["rbd"], c["mode"], 0.002, c["time"], mpc_variant=bool(variant), debug=True)
    return oracle.wbc_tasks(), ref


@pytest.mark.parametrize("shapes", SHAPES[:4])
def test_emulated_general_kernel_vs_oracle_and_literal(shapes):
    import emu_harness, pyoracle
    from hoqp_literal import hoqp_literal
    rng = np.random.default_rng(5 + len(shapes)); compared = 0
    for rep in range(6):
        tasks = _random_cascade(rng, 12, shapes)
        x, st, it = pyoracle.hoqp(tasks); xe, ste = emu_harness.hoqp(tasks)
        assert np.array_equal(st, ste), (rep, st, ste)                      # incl. the ill-posed cascades of the pairing quirk (status 2 / 3)
        if (st == 0).all():
            assert np.abs(x - xe).max() <= 1e-9 * max(1.0, np.abs(x).max())
            xl, _ = hoqp_literal(tasks); assert np.abs(xe - xl).max() <= 1e-8 * max(1.0, np.abs(xl).max()); compared += 1
    assert compared >= 2


def test_emulated_general_kernel_on_the_controllers_tasks(blobs, oracle):
    """the two shipped hierarchies are special cases: on the task matrices WbcBase builds, the general kernel returns what the specialised cascade returns"""
    import emu_harness
    for c in random_wbc_inputs(oracle, blobs, 2, 21, 0.05, MODES) + hard_wbc_inputs(oracle, blobs, 2, 31):
        tasks, ref = _wbc_tasks(oracle, c)
        x, st = emu_harness.hoqp(tasks)
        assert (st == 0).all() and np.abs(x - ref[:36]).max() <= 1e-8 * np.abs(ref[:36]).max()


def _equality_cascade(rng, n, ma):
    """one equality-only level with ma rows on n variables (feasible by construction)"""
    A = rng.normal(size=(ma, n)); return [dict(A=A, b=A @ rng.normal(size=n) if ma <= n else rng.normal(size=ma), D=np.zeros((0, n)), f=np.zeros(0))]


def test_emulated_pipeline_reuse_with_fewer_variables_and_more_rows():
    """advisor finding (round 3): the pipeline sized b / f by rows * n; a later solve with fewer variables but more rows (n = 36, ma = 10 -> n = 10, ma = 36) overran them"""
    import emu_harness, pyoracle
    rng = np.random.default_rng(3)
    for n, ma in ((36, 10), (10, 36), (36, 10), (12, 30)):
        tasks = _equality_cascade(rng, n, ma)
        x, st, _ = pyoracle.hoqp(tasks); xe, ste = emu_harness.hoqp(tasks)
        assert np.array_equal(st, ste) and np.abs(x - xe).max() <= 1e-9 * max(1.0, np.abs(x).max()), (n, ma)


@pytest.mark.gpu
def test_context_reuse_with_fewer_variables_and_more_rows(blobs):
    import pyoracle
    from qm_control_amd import api
    itf = api.QMInterface(blobs=blobs, max_batch=1, max_nodes=8, max_ref_knots=2, max_events=2); hq = api.HoQp(itf); rng = np.random.default_rng(3)
    for n, ma in ((36, 10), (10, 36), (36, 10), (12, 30)):
        tasks = _equality_cascade(rng, n, ma)
        xo, sto, _ = pyoracle.hoqp(tasks); x, st = hq.solve(tasks)
        assert np.array_equal(np.ravel(st), sto) and np.abs(np.ravel(x) - xo).max() <= 1e-9 * max(1.0, np.abs(xo).max()), (n, ma)
    itf.close()


@pytest.mark.gpu
def test_general_kernel_batches_vs_oracle(blobs):
    import pyoracle
    from qm_control_amd import api
    itf = api.QMInterface(blobs=blobs, max_batch=1, max_nodes=8, max_ref_knots=2, max_events=2); hq = api.HoQp(itf)
    for shapes in SHAPES:
        rng = np.random.default_rng(11 + len(shapes)); B = 24
        cascades = [_random_cascade(rng, 12, shapes) for _ in range(B)]
        tasks = [dict((k, np.stack([cs[lev][k] for cs in cascades])) for k in ("A", "b", "D", "f")) for lev in range(len(shapes))]
        x, st = hq.solve(tasks); ok = 0
        for i, cs in enumerate(cascades):
            xo, sto, _ = pyoracle.hoqp(cs)
            assert np.array_equal(st[i], sto), (shapes, i, st[i], sto)
            if (sto == 0).all():
                assert np.abs(x[i] - xo).max() <= 1e-9 * max(1.0, np.abs(xo).max()), (shapes, i); ok += 1
        assert ok >= B // 4
    # shape limits are refused, not truncated
    with pytest.raises(api.QmhipError, match="bad argument"):
        hq.solve([dict(A=np.zeros((1, 37)), b=np.zeros(1), D=np.zeros((0, 37)), f=np.zeros(0))])
    itf.close()


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [0, 1])
def test_general_kernel_equals_the_wbc_kernel_and_the_literal_cascade(blobs, oracle, variant):
    from qm_control_amd import api
    from hoqp_literal import hoqp_literal
    from conftest import assert_blocks
    itf = api.QMInterface(blobs=blobs, max_batch=16, max_nodes=8, max_ref_knots=2, max_events=2); hq = api.HoQp(itf); wbc = api.HierarchicalWbc(itf, mpc_variant=bool(variant))
    cases = random_wbc_inputs(oracle, blobs, 8, 21 + variant, 0.05, MODES) + hard_wbc_inputs(oracle, blobs, 8, 31 + variant)
    arr = lambda k: np.array([c[k] for c in cases])
    wbc.reset(); wbc.update(arr("xd"), arr("il"), arr("rbd"), arr("mode"), 0.002, arr("time"))
    out, stw = wbc.update(arr("xd"), arr("ud"), arr("rbd"), arr("mode"), 0.002, arr("time"))
    for i, c in enumerate(cases):                      # task shapes differ with the contact mode: one cascade per call
        tasks, ref = _wbc_tasks(oracle, c, variant)
        x, st = hq.solve(tasks)
        assert (st == 0).all() and (stw[i] == 0).all()
        xl, _ = hoqp_literal(tasks)
        assert_blocks(np.concatenate([x, out[i, 36:]]), np.concatenate([xl, out[i, 36:]]), "wbc", 1e-8, "literal, case %d" % i)
        assert np.abs(x - out[i, :36]).max() <= 1e-8 * np.abs(out[i, :36]).max(), i      # the specialised kernel's [v̇; F]
    itf.close()
