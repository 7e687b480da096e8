"""GPU parity tests of policy evaluation + the hierarchical WBC (K5..K7) and of the whole control step,
through the C ABI, against the CPU oracle.  Tolerance 1e-6 relative on torques / decision vector
(BASELINE.json north_star); qp status must be 0 on both sides."""
import numpy as np
import pytest
from conftest import assert_blocks, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-6


from wbc_cases import random_wbc_inputs as _random_wbc_inputs


@pytest.mark.parametrize("variant", [0, 1])
def test_wbc_random_states(blobs, oracle, variant):
    from qm_control_amd import api
    cases = _random_wbc_inputs(oracle, blobs, 24, 11 + variant, 0.05)
    B = len(cases)
    itf = api.QMInterface(blobs=blobs, max_batch=B, max_nodes=8, max_ref_knots=2, max_events=2)
    wbc = api.HierarchicalWbc(itf, mpc_variant=bool(variant))
    arr = lambda k: np.array([c[k] for c in cases])
    wbc.reset()
    wbc.update(arr("xd"), arr("il"), arr("rbd"), arr("mode"), 0.002, arr("time"))       # primes inputLast_
    out, st = wbc.update(arr("xd"), arr("ud"), arr("rbd"), arr("mode"), 0.002, arr("time"))
    for b, c in enumerate(cases):
        oracle.wbc_reset(); oracle.wbc(c["xd"], c["il"], c["rbd"], c["mode"], 0.002, c["time"], mpc_variant=bool(variant))
        ref, sto = oracle.wbc(c["xd"], c["ud"], c["rbd"], c["mode"], 0.002, c["time"], mpc_variant=bool(variant))
        assert list(sto) == [0, 0, 0], (b, sto)
        assert list(st[b]) == [0, 0, 0], (b, st[b])
        assert_blocks(out[b], ref, "wbc", TOL, b)
    itf.close()


@pytest.mark.parametrize("name,B,N", [("C2", 1, 100), ("C3", 32, 40), ("C5", 8, 150)])
def test_control_step_vs_oracle(blobs, oracle, name, B, N):
    """whole benchmark step: SQP iteration + policy at t0 + WBC on the synthetic measured state"""
    import pyoracle
    from qm_control_amd import api, scenarios
    cfg = scenarios.make_config(name, batch=B, n_intervals=N)
    bad, xf, uf, w = pyoracle.batch_step(*pyoracle.load_blobs(), 8, cfg["t0"], cfg["horizon"], cfg["x0"], cfg["ref_t"], cfg["ref_x"], cfg["ev"], cfg["modes"], cfg["period"], cfg["time"])
    assert bad == 0
    itf = api.QMInterface(blobs=blobs, max_batch=B, max_nodes=N + 40, max_ref_knots=2, max_events=cfg["ev"].shape[1])
    mpc = api.SqpMpc(itf); wbc = api.HierarchicalWbc(itf)
    mpc.set_problem(cfg["t0"], cfg["x0"], cfg["ref_t"], cfg["ref_x"], cfg["ev"], cfg["modes"])
    wbc.reset()
    mpc.control_step_resident(cfg["horizon"], cfg["period"], cfg["time"])
    out, st = wbc.download(B)
    xd, ud, mode = mpc.evaluatePolicy(cfg["t0"])
    assert_blocks(xd, xf, "x", TOL, "policy x"); assert_blocks(ud, uf, "u", TOL, "policy u")
    assert (st == 0).all()
    for b in range(B):
        assert_blocks(out[b], w[b], "wbc", TOL, b)
    itf.close()


@pytest.mark.parametrize("gait", ["flying_trot", "pace", "dynamic_walk", "static_walk", "amble", "skipping", "pawup"])
def test_control_step_other_gaits(blobs, oracle, gait):
    """whole control step on the gaits the benchmark does not walk: 0-, 1-, 2- (lateral) and 3-leg support in the WBC's contact tasks"""
    import pyoracle
    from qm_control_amd import api, scenarios
    B, N = 8, 30
    cfg = scenarios.gait_config(gait, batch=B, n_intervals=N, seed=11)
    bad, xf, uf, w = pyoracle.batch_step(*pyoracle.load_blobs(), 8, cfg["t0"], cfg["horizon"], cfg["x0"], cfg["ref_t"], cfg["ref_x"], cfg["ev"], cfg["modes"], cfg["period"], cfg["time"])
    assert bad == 0
    itf = api.QMInterface(blobs=blobs, max_batch=B, max_nodes=96, max_ref_knots=2, max_events=cfg["ev"].shape[1])
    mpc = api.SqpMpc(itf); wbc = api.HierarchicalWbc(itf)
    mpc.set_problem(cfg["t0"], cfg["x0"], cfg["ref_t"], cfg["ref_x"], cfg["ev"], cfg["modes"])
    wbc.reset()
    mpc.control_step_resident(cfg["horizon"], cfg["period"], cfg["time"])
    out, st = wbc.download(B)
    assert (st == 0).all()
    for b in range(B):
        assert_blocks(out[b], w[b], "wbc", TOL, b)
    itf.close()


@pytest.mark.parametrize("gait,inst", [("standing_pace", 244), ("lindyhop", 137)])
def test_degenerate_vertices_are_resolved(blobs, oracle, gait, inst):
    """regression: instances whose level-2 working set fills the (small) null space — a further blocking row used to be reported as a working-set
    overflow (status 2) although the torques were right; every instance of the batch must finish with status 0 and the named one match the oracle"""
    import pyoracle
    from qm_control_amd import api, scenarios
    B, N = 256, 60
    cfg = scenarios.gait_config(gait, batch=B, n_intervals=N, seed=3)
    itf = api.QMInterface(blobs=blobs, max_batch=B, max_nodes=N + 80, max_ref_knots=2, max_events=cfg["ev"].shape[1])
    mpc = api.SqpMpc(itf); wbc = api.HierarchicalWbc(itf)
    mpc.set_problem(cfg["t0"], cfg["x0"], cfg["ref_t"], cfg["ref_x"], cfg["ev"], cfg["modes"])
    wbc.reset(); mpc.control_step_resident(cfg["horizon"], cfg["period"], cfg["time"])
    res = mpc.download(); out, st = wbc.download(B)
    assert (res["status"] == 0).all() and (st == 0).all()
    idx = np.array([inst])
    bad, xf, uf, w = pyoracle.batch_step(*pyoracle.load_blobs(), 1, cfg["t0"][idx], cfg["horizon"], cfg["x0"][idx], cfg["ref_t"][idx], cfg["ref_x"][idx], cfg["ev"][idx], cfg["modes"][idx], cfg["period"], cfg["time"])
    assert bad == 0
    assert_blocks(out[inst], w[0], "wbc", TOL, inst)
    itf.close()


@pytest.mark.parametrize("variant", [0, 1])
def test_reconfigured_gains_reach_the_device(blobs, oblobs, variant):
    """the rqt_reconfigure path (WbcBase::dynamicCallback, WbcBase.cpp:69-116): gains changed BY NAME mid-run through the C ABI (qmhip_wbc_gain_index ->
    qmhip_set_setting, what adaptors/QmhipWbc.h does with the server's parameter_updates message) take effect on the next tick and match the oracle run with the same gains;
    a name the reference's callback does not read changes nothing."""
    import pyoracle
    from qm_control_amd import api, layout as L
    oracle = pyoracle.Oracle(*oblobs)
    cases = _random_wbc_inputs(oracle, blobs, 12, 41 + variant, 0.05)
    B = len(cases); arr = lambda k: np.array([c[k] for c in cases])
    itf = api.QMInterface(blobs=blobs, max_batch=B, max_nodes=8, max_ref_knots=2, max_events=2); witf = itf.wbc_context()      # the control thread's context, as the adaptor uses it
    wbc = api.HierarchicalWbc(witf, mpc_variant=bool(variant))
    def tick():
        wbc.reset(); wbc.update(arr("xd"), arr("il"), arr("rbd"), arr("mode"), 0.002, arr("time"))
        return wbc.update(arr("xd"), arr("ud"), arr("rbd"), arr("mode"), 0.002, arr("time"))
    def oracle_tick(o):
        outs = []
        for c in cases:
            o.wbc_reset(); o.wbc(c["xd"], c["il"], c["rbd"], c["mode"], 0.002, c["time"], mpc_variant=bool(variant))
            outs.append(o.wbc(c["xd"], c["ud"], c["rbd"], c["mode"], 0.002, c["time"], mpc_variant=bool(variant))[0])
        return np.array(outs)
    out0, st0 = tick()
    assert (st0 == 0).all(); assert_blocks(out0, oracle_tick(oracle), "wbc", TOL, "shipped gains")
    new = {"kp_swing": 200.0, "kd_swing": 20.0, "baseHeightKp": 250.0, "baseHeightKd": 60.0, "kp_base_linear": 300.0, "kd_base_linear": 50.0, "kp_base_angular": 150.0, "kd_base_angular": 90.0,
           "kp_arm_joint_2": 2500.0, "kd_arm_joint_5": 40.0, "kp_ee_linear_x": 1500.0, "kd_ee_linear_z": 50.0, "kp_ee_angular_y": 900.0, "kd_ee_angular_x": 30.0}
    for name, v in new.items():
        assert witf.set_gain(name, v)
        oracle.set_setting(witf.lib.qmhip_wbc_gain_index(name.encode()), v)
    assert not witf.set_gain("d_ee_x", 0.3) and not witf.set_gain("da_ee_z", 1.0)          # not read by the reference's callback: no slot
    out1, st1 = tick()
    assert (st1 == 0).all(); assert_blocks(out1, oracle_tick(oracle), "wbc", TOL, "reconfigured gains")
    assert np.abs(out1[:, 36:] - out0[:, 36:]).max() > 1e-3                                  # the gains did change the torques
    witf.close(); itf.close()
