"""GPU parity tests of the MPC path (K0..K4) through the C ABI, against the CPU oracle.

Tolerance: BASELINE.json north_star — optimal state/input trajectories within 1e-6 relative; integer
contact-mode schedules / node event tags bit-exact.
"""
import numpy as np
from qm_control_amd import layout as L
import pytest
from conftest import assert_blocks, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-6


def _oracle_solve(oracle, cfg, b):
    oracle.set_schedule(cfg["ev"][b], cfg["modes"][b])
    oracle.set_target(cfg["ref_t"][b], cfg["ref_x"][b])
    return oracle.mpc_step(cfg["t0"][b], cfg["t0"][b] + cfg["horizon"], cfg["x0"][b])


def _gpu_solve(blobs, cfg, B, max_nodes):
    from qm_control_amd import api
    itf = api.QMInterface(blobs=blobs, max_batch=B, max_nodes=max_nodes, max_ref_knots=cfg["ref_t"].shape[1], max_events=cfg["ev"].shape[1])
    mpc = api.SqpMpc(itf)
    res = mpc.run(cfg["t0"][:B], cfg["x0"][:B], cfg["ref_t"][:B], cfg["ref_x"][:B], cfg["ev"][:B], cfg["modes"][:B], cfg["horizon"])
    return itf, mpc, res


def _compare(res, b, r):
    n = len(r["t"])
    assert res["status"][b] == 0
    assert res["num_nodes"][b] == n
    assert np.array_equal(res["t"][b, :n], r["t"])                    # same f64 operations -> identical times
    assert np.array_equal(res["event"][b, :n], r["ev"])                # integer: bit-exact
    assert np.array_equal(res["mode"][b, :n], r["mode"])               # integer: bit-exact
    assert_blocks(res["x"][b, :n], r["x"], "x", TOL, "x* instance %d" % b)
    assert_blocks(res["u"][b, :n], r["u"], "u", TOL, "u* instance %d" % b)
    assert res["perf"][b, 8] == r["alpha"]
    assert rel_err(res["perf"][b, :8], r["perf"][:8]) <= 1e-6


@pytest.mark.parametrize("name", ["C1", "C2"])
def test_single_instance(blobs, oracle, name):
    from qm_control_amd import scenarios
    cfg = scenarios.make_config(name)
    r = _oracle_solve(oracle, cfg, 0)
    itf, mpc, res = _gpu_solve(blobs, cfg, 1, len(r["t"]) + 4)
    _compare(res, 0, r)
    itf.close()


def test_batch_random_trot(blobs, oracle):
    from qm_control_amd import scenarios
    cfg = scenarios.make_config("C3", batch=64, n_intervals=40)
    itf, mpc, res = _gpu_solve(blobs, cfg, 64, 64)
    for b in (0, 1, 7, 31, 63):
        _compare(res, b, _oracle_solve(oracle, cfg, b))
    itf.close()


ALL_GAITS = ["stance", "trot", "standing_trot", "flying_trot", "pace", "standing_pace", "dynamic_walk", "static_walk", "amble", "lindyhop", "skipping", "pawup"]


@pytest.mark.parametrize("gait", ALL_GAITS)
def test_every_gait_template(blobs, oracle, gait):
    """All 12 templates of gait.info: 1-, 2-, 3-leg support and flight phases (nc = 12..16, m = 14..18)."""
    from qm_control_amd import scenarios
    cfg = scenarios.gait_config(gait, batch=4, n_intervals=30)
    itf, mpc, res = _gpu_solve(blobs, cfg, 4, 96)
    for b in range(4):
        _compare(res, b, _oracle_solve(oracle, cfg, b))
    itf.close()


def test_ee_tracking_schedule_switch(blobs, oracle):
    from qm_control_amd import scenarios
    cfg = scenarios.make_config("C5", batch=16, n_intervals=150)
    itf, mpc, res = _gpu_solve(blobs, cfg, 16, 180)
    for b in (0, 5, 15):
        _compare(res, b, _oracle_solve(oracle, cfg, b))
    itf.close()


def test_node_buffer_too_small_is_reported(blobs):
    from qm_control_amd import scenarios
    cfg = scenarios.make_config("C2")
    itf, mpc, res = _gpu_solve(blobs, cfg, 1, 32)
    assert res["status"][0] == -1
    itf.close()


def test_receding_horizon_closed_loop(blobs, oracle):
    """SURVEY.md §8(f) rank 1 through the C ABI: warm-started MPC calls chained on the device (advance along the policy), against the
    oracle doing the same thing; and qmhip_closed_loop_resident == the same calls issued one by one."""
    from qm_control_amd import api, scenarios
    B, steps, dt_mpc = 4, 4, 0.02
    cfg = scenarios.make_config("C3", batch=B, n_intervals=40)
    itf = api.QMInterface(blobs=blobs, max_batch=B, max_nodes=80, max_ref_knots=cfg["ref_t"].shape[1], max_events=cfg["ev"].shape[1])
    mpc = api.SqpMpc(itf); wbc = api.HierarchicalWbc(itf)
    mpc.set_problem(cfg["t0"], cfg["x0"], cfg["ref_t"], cfg["ref_x"], cfg["ev"], cfg["modes"])
    mpc.solve_resident(cfg["horizon"]); got = [mpc.download()]
    for k in range(1, steps):
        mpc.advance(dt_mpc); mpc.solve_resident(cfg["horizon"], warm=True); got.append(mpc.download())
    for b in range(B):
        oracle.set_schedule(cfg["ev"][b], cfg["modes"][b]); oracle.set_target(cfg["ref_t"][b], cfg["ref_x"][b])
        t0 = float(cfg["t0"][b]); x0 = cfg["x0"][b]
        for k in range(steps):
            if k > 0:
                t0 += dt_mpc; x0, _, _ = oracle.eval_policy(t0)
            r = oracle.mpc_step(t0, t0 + cfg["horizon"], x0, warm=(k > 0)); n = len(r["t"]); g = got[k]
            assert g["status"][b] == 0 and g["num_nodes"][b] == n
            assert np.array_equal(g["t"][b, :n], r["t"]) and np.array_equal(g["event"][b, :n], r["ev"]) and np.array_equal(g["mode"][b, :n], r["mode"])
            assert_blocks(g["x"][b, :n], r["x"], "x", TOL, (b, k)); assert_blocks(g["u"][b, :n], r["u"], "u", TOL, (b, k))
    # the fused closed loop (MPC + policy + WBC per step) reproduces the step-by-step sequence bit for bit
    wbc.reset(); mpc.set_problem(cfg["t0"], cfg["x0"], cfg["ref_t"], cfg["ref_x"], cfg["ev"], cfg["modes"])
    mpc.closed_loop_resident(steps, dt_mpc, cfg["horizon"], cfg["period"], cfg["time"])
    last = mpc.download(); out, qps = wbc.download(B)
    assert np.array_equal(last["x"], got[-1]["x"]) and np.array_equal(last["u"], got[-1]["u"]) and (qps == 0).all() and np.isfinite(out).all()
    itf.close()


def test_structured_input_weight_paths_equal_the_dense_ones_on_the_device(blobs):
    """Round 6: with the shipped block-diagonal input weight K1b forms r = R0 (u - u_nom) as a three-term row product in the order the dense path's DPP tree adds the same
    terms (k_lq.h) and the trial evaluation multiplies 54 instead of 900 entries (k_ls.h).  Forced onto the dense instances (`r_dense` 1) the device must give the SAME BITS:
    the whole primal solution, the merit terms and the step lengths over warm-started receding-horizon solves (a cold start has u = u_nom exactly and would not see a wrong
    association — the first build of the structured product passed a cold-start test and failed this one: profiles/r06_ab_lq_regions.log)."""
    from qm_control_amd import api, scenarios
    B, steps, dt_mpc = 128, 3, 0.01
    cfg = scenarios.make_config("C3", batch=B)

    def run(dense):
        itf = api.QMInterface(blobs=blobs, max_batch=B, max_nodes=128, max_ref_knots=2, max_events=cfg["ev"].shape[1]); itf.debug_set("r_dense", dense)
        assert itf.debug_get("r_blocks") == (0 if dense else 1)
        mpc = api.SqpMpc(itf); mpc.set_problem(cfg["t0"], cfg["x0"], cfg["ref_t"], cfg["ref_x"], cfg["ev"], cfg["modes"])
        outs = []
        for k in range(steps):
            if k > 0: mpc.advance(dt_mpc)
            mpc.solve_resident(cfg["horizon"], warm=(k > 0)); outs.append(mpc.download())
        itf.close(); return outs

    st, de = run(0), run(1)
    assert np.abs(st[1]["u"][:, :, 12:24]).max() > 1e-3          # the warm solves do start from non-zero joint velocities
    for k in range(steps):
        for key in ("x", "u", "status", "t", "perf"):
            assert np.array_equal(st[k][key], de[k][key]), (k, key)


def test_device_line_search_tail_equals_the_host_driven_loop(blobs, oracle):
    """Round 6: after the first trial the line search finishes in ONE launch on the device (qm_ls_tail_kernel, k_ls.h) instead of one host round trip per trial.  On
    warm-started receding-horizon solves of the benchmark workload (256 instances, N = 100: ~ 15 % of the instances reject the full step, a few take three trials —
    tools/warm_ls_histogram.py) the tail must give BIT-IDENTICAL results to the host-driven loop of rounds 1-5 (`ls_device_tail` 0): step lengths, status, the whole
    primal solution, the number of trials (the merit sums to 1e-13); and the accepted step lengths are the oracle's on a sample of instances that backtracked."""
    from qm_control_amd import api, scenarios
    B, steps, dt_mpc = 256, 5, 0.01
    cfg = scenarios.make_config("C3", batch=B)

    def run(tail):
        itf = api.QMInterface(blobs=blobs, max_batch=B, max_nodes=128, max_ref_knots=2, max_events=cfg["ev"].shape[1]); itf.debug_set("ls_device_tail", tail)
        mpc = api.SqpMpc(itf); mpc.set_problem(cfg["t0"], cfg["x0"], cfg["ref_t"], cfg["ref_x"], cfg["ev"], cfg["modes"])
        outs = []
        for k in range(steps):
            if k > 0: mpc.advance(dt_mpc)
            mpc.solve_resident(cfg["horizon"], warm=(k > 0)); outs.append(mpc.download())
        itf.close(); return outs

    dev, host = run(1), run(0)
    for k in range(steps):
        a, b = dev[k], host[k]
        assert a["ls_trials"] == b["ls_trials"], (k, a["ls_trials"], b["ls_trials"])
        for key in ("x", "u", "status", "t"):
            assert np.array_equal(a[key], b[key]), (k, key)
        # step lengths and Armijo metric bit for bit; the merit sums of a later trial to rounding (the node terms are the same function inlined into two kernels: the compiler's
        # multiply-add contraction may differ — observed: ONE sum of 256 x 4 off by one unit in the last place over five steps, 1e-22 on 5e-7)
        assert np.array_equal(a["perf"][:, 8:], b["perf"][:, 8:]) and np.array_equal(a["perf"][:, :4], b["perf"][:, :4]), k
        assert np.allclose(a["perf"][:, 4:8], b["perf"][:, 4:8], rtol=1e-13, atol=0.0), k
    alphas = np.stack([d["perf"][:, 8] for d in dev])                      # [steps][B] accepted step lengths
    assert (alphas[0] == 1.0).all() and (alphas[2:] < 1.0).any() and max(d["ls_trials"] for d in dev) >= 2, alphas.min(axis=1)      # the warm solves DO backtrack
    # the oracle's loop on a sample: the instances with the smallest accepted step lengths + a few that never backtracked
    order = np.argsort(alphas[1:].min(axis=0)); sample = list(order[:5]) + list(order[-2:])
    for b in sample:
        oracle.set_schedule(cfg["ev"][b], cfg["modes"][b]); oracle.set_target(cfg["ref_t"][b], cfg["ref_x"][b])
        t0 = float(cfg["t0"][b]); x0 = cfg["x0"][b]
        for k in range(steps):
            if k > 0:
                t0 += dt_mpc; x0, _, _ = oracle.eval_policy(t0)
            r = oracle.mpc_step(t0, t0 + cfg["horizon"], x0, warm=(k > 0)); n = len(r["t"])
            assert dev[k]["perf"][b, 8] == r["alpha"], (b, k, dev[k]["perf"][b, 8], r["alpha"])
            assert_blocks(dev[k]["x"][b, :n], r["x"], "x", TOL, (b, k)); assert_blocks(dev[k]["u"][b, :n], r["u"], "u", TOL, (b, k))


def test_fused_policy_at_t0_equals_the_policy_kernel_on_the_device(blobs):
    """Round 6: in a control step the kernels that decide the step length write the policy at t0, the WBC starts behind them on its own stream and the batch's apply runs
    beside it (qmhip.hip: control_step).  Against the rounds-1-5 order (apply -> qm_policy_kernel -> WBC; `fused_policy` 0) on the benchmark workload — a cold step and five
    warm-started closed-loop steps, where instances backtrack — torques, QP statuses, the policy at t0 and the primal solution must be bit-identical."""
    from qm_control_amd import api, scenarios
    B = 256
    cfg = scenarios.make_config("C3", batch=B)

    def run(fused):
        itf = api.QMInterface(blobs=blobs, max_batch=B, max_nodes=128, max_ref_knots=2, max_events=cfg["ev"].shape[1]); itf.debug_set("fused_policy", fused)
        mpc = api.SqpMpc(itf); wbc = api.HierarchicalWbc(itf)
        mpc.set_problem(cfg["t0"], cfg["x0"], cfg["ref_t"], cfg["ref_x"], cfg["ev"], cfg["modes"]); wbc.reset()
        mpc.control_step_resident(cfg["horizon"], cfg["period"], cfg["time"]); cold = (wbc.download(B), mpc.download(), mpc.evaluatePolicy(cfg["t0"]))
        wbc.reset(); mpc.closed_loop_resident(5, 0.01, cfg["horizon"], cfg["period"], cfg["time"]); itf.synchronize()
        warm = (wbc.download(B), mpc.download()); itf.close()
        return cold, warm

    (c1, w1), (c0, w0) = run(1), run(0)
    for a, b in ((c1, c0), (w1, w0)):
        assert np.array_equal(a[0][0], b[0][0]) and np.array_equal(a[0][1], b[0][1]) and (a[0][1] == 0).all()      # WBC output, QP statuses
        for key in ("x", "u", "perf", "status", "t"): assert np.array_equal(a[1][key], b[1][key]), key
    for k in range(3): assert np.array_equal(c1[2][k], c0[2][k])
    assert (w1[1]["perf"][:, 8] < 1.0).any() and w1[1]["ls_trials"] >= 2      # the warm steps did backtrack


@pytest.mark.parametrize("B,max_nodes", [(97, 192), (5, 512)])
def test_line_search_tail_ragged_batch_and_largest_node_capacity(blobs, oracle, B, max_nodes):
    """The device-side line-search tail and the fused policy on the shapes the benchmark does not exercise: a batch that is no multiple of the 64-row blocks the
    thread-per-(node, instance) kernels move together (97), and the largest node capacity (512: the tail's workgroup then carves 88 KB of LDS), on C5 (N = 150, schedule
    switching, arm near its limits) with a tightened filter and arms started off their references (some far outside the joint limits) so that part of the batch backtracks.  Bit-identical to the host-driven loop + policy kernel, every
    status valid, and the accepted step lengths and the control step's torques equal the oracle's on a sample."""
    from qm_control_amd import api, scenarios
    import pyoracle
    st = blobs[1].copy(); st[L.ST_G_MAX] = 1e-9; st[L.ST_DELTA_TOL] = 1e-12
    cfg = scenarios.make_config("C5", batch=B)
    cfg["x0"][::3, 24:30] += 0.3; cfg["x0"][1::7, 24:30] += 3.3; cfg["x0"][2::5, 12:24] += 0.5      # arms / legs off their references (some arms far outside the joint limits): part of the batch refuses the full step

    def run(tail, fused):
        itf = api.QMInterface(blobs=(blobs[0], st), max_batch=B, max_nodes=max_nodes, max_ref_knots=2, max_events=cfg["ev"].shape[1])
        itf.debug_set("ls_device_tail", tail); itf.debug_set("fused_policy", fused)
        mpc = api.SqpMpc(itf); wbc = api.HierarchicalWbc(itf)
        mpc.set_problem(cfg["t0"], cfg["x0"], cfg["ref_t"], cfg["ref_x"], cfg["ev"], cfg["modes"]); wbc.reset()
        mpc.control_step_resident(cfg["horizon"], cfg["period"], cfg["time"])
        res = mpc.download(); out, qps = wbc.download(B); itf.close()
        return res, out, qps

    (r1, o1, q1), (r0, o0, q0) = run(1, 1), run(0, 0)
    assert r1["ls_trials"] == r0["ls_trials"] and r1["ls_trials"] >= 2, (r1["ls_trials"], r0["ls_trials"])
    for key in ("x", "u", "status", "t", "num_nodes"): assert np.array_equal(r1[key], r0[key]), key
    assert np.array_equal(r1["perf"][:, 8], r0["perf"][:, 8]) and np.array_equal(o1, o0) and np.array_equal(q1, q0)
    assert (r1["status"] >= 0).all() and (r1["perf"][:, 8] < 1.0).any() and (r1["perf"][:, 8] == 1.0).any()      # a mixed batch (the WBC of a robot whose arm is 3.3 rad off may report an iteration limit: not asserted)
    o = pyoracle.Oracle(pyoracle.load_blobs()[0], st)
    for b in (0, B // 2, B - 1):
        o.set_schedule(cfg["ev"][b], cfg["modes"][b]); o.set_target(cfg["ref_t"][b], cfg["ref_x"][b])
        r = o.mpc_step(float(cfg["t0"][b]), float(cfg["t0"][b]) + cfg["horizon"], cfg["x0"][b]); n = len(r["t"])
        assert r1["perf"][b, 8] == r["alpha"] and r1["num_nodes"][b] == n, (b, r1["perf"][b, 8], r["alpha"])
        assert_blocks(r1["x"][b, :n], r["x"], "x", TOL, b); assert_blocks(r1["u"][b, :n], r["u"], "u", TOL, b)


def test_update_references_keeps_the_warm_start(blobs, oracle):
    """what the MPC_BASE adaptor does on every call after the first (adaptors/QmhipMpc.h): new targets / schedule from preSolverRun, new observation,
    warm-started iteration from the PREVIOUS primal solution — qmhip_mpc_update_references must not drop it (qmhip_mpc_upload would)"""
    from qm_control_amd import api, scenarios
    B = 3
    cfg = scenarios.make_config("C3", batch=B, n_intervals=30)
    itf = api.QMInterface(blobs=blobs, max_batch=B, max_nodes=64, max_ref_knots=2, max_events=cfg["ev"].shape[1])
    mpc = api.SqpMpc(itf)
    mpc.set_problem(cfg["t0"], cfg["x0"], cfg["ref_t"], cfg["ref_x"], cfg["ev"], cfg["modes"]); mpc.solve_resident(cfg["horizon"])
    first = mpc.download()
    ref_x2 = cfg["ref_x"].copy(); ref_x2[:, 1, 6] += 0.2; ref_x2[:, 1, 30] += 0.05                 # the operator moved the base and the EE goal
    t1 = cfg["t0"] + 0.03; x1, _, _ = mpc.evaluatePolicy(t1)
    mpc.update_references(ref_t=cfg["ref_t"], ref_x=ref_x2)                                        # schedule left as is
    mpc.set_initial(t1, x1); mpc.solve_resident(cfg["horizon"], warm=True)
    got = mpc.download()
    for b in range(B):
        oracle.set_schedule(cfg["ev"][b], cfg["modes"][b]); oracle.set_target(cfg["ref_t"][b], cfg["ref_x"][b])
        r0 = oracle.mpc_step(cfg["t0"][b], cfg["t0"][b] + cfg["horizon"], cfg["x0"][b]); n0 = len(r0["t"])
        assert_blocks(first["x"][b, :n0], r0["x"], "x", TOL, b)
        xo, _, _ = oracle.eval_policy(t1[b])
        oracle.set_target(cfg["ref_t"][b], ref_x2[b])
        r = oracle.mpc_step(t1[b], t1[b] + cfg["horizon"], xo, warm=True); n = len(r["t"])
        assert got["status"][b] == 0 and got["num_nodes"][b] == n and np.array_equal(got["mode"][b, :n], r["mode"])
        assert_blocks(got["x"][b, :n], r["x"], "x", TOL, b); assert_blocks(got["u"][b, :n], r["u"], "u", TOL, b)
        # a cold start from the same observation is a different iterate: the warm start really was used
        rc = oracle.mpc_step(t1[b], t1[b] + cfg["horizon"], xo)
        assert np.abs(rc["u"] - r["u"]).max() > 1e-3
    itf.close()


def test_multiple_sqp_iterations(blobs, oracle):
    """sqp.sqpIteration = 3 through qmhip_set_setting: three SQP iterations per MPC call, against the oracle iterating on its own iterate"""
    from qm_control_amd import api, scenarios
    B = 2
    cfg = scenarios.make_config("C3", batch=B, n_intervals=40)
    itf = api.QMInterface(blobs=blobs, max_batch=B, max_nodes=80, max_ref_knots=cfg["ref_t"].shape[1], max_events=cfg["ev"].shape[1])
    itf.set_setting(L.ST_SQP_ITER, 3.0)
    mpc = api.SqpMpc(itf)
    got = mpc.run(cfg["t0"], cfg["x0"], cfg["ref_t"], cfg["ref_x"], cfg["ev"], cfg["modes"], cfg["horizon"])
    for b in range(B):
        oracle.set_schedule(cfg["ev"][b], cfg["modes"][b]); oracle.set_target(cfg["ref_t"][b], cfg["ref_x"][b])
        t0 = float(cfg["t0"][b]); r = oracle.mpc_step(t0, t0 + cfg["horizon"], cfg["x0"][b])
        for _ in range(2):
            r = oracle.mpc_step(t0, t0 + cfg["horizon"], cfg["x0"][b], warm="iterate")
        n = len(r["t"])
        assert got["status"][b] == 0 and got["num_nodes"][b] == n
        assert_blocks(got["x"][b, :n], r["x"], "x", TOL, b); assert_blocks(got["u"][b, :n], r["u"], "u", TOL, b)
        assert rel_err(got["perf"][b, :8], r["perf"][:8]) <= 1e-5
    itf.close()


def test_discrete_ilqr_solver(blobs, oracle):
    """qmhip_set_setting(ST_SOLVER, 1): the discrete iLQR behind the same MPC entry points on a batch whose instances stop their line
    searches at different step lengths, cold and warm, against oracle/src/ilqr.h; switching back restores the SQP"""
    from qm_control_amd import api, scenarios
    B, N = 8, 24
    cfg = scenarios.make_config("C3", batch=B, n_intervals=N)
    itf = api.QMInterface(blobs=blobs, max_batch=B, max_nodes=N + 24, max_ref_knots=2, max_events=cfg["ev"].shape[1])
    mpc = api.SqpMpc(itf)
    itf.set_setting(L.ST_SOLVER, 1.0)
    res = mpc.run(cfg["t0"], cfg["x0"], cfg["ref_t"], cfg["ref_x"], cfg["ev"], cfg["modes"], cfg["horizon"])
    t1 = cfg["t0"] + 0.02; x1, _, _ = mpc.evaluatePolicy(t1)
    mpc.set_initial(t1, x1); mpc.solve_resident(cfg["horizon"], warm=True); res2 = mpc.download()
    alphas = set()
    for b in range(B):
        oracle.set_schedule(cfg["ev"][b], cfg["modes"][b]); oracle.set_target(cfg["ref_t"][b], cfg["ref_x"][b])
        r = oracle.ilqr_step(cfg["t0"][b], cfg["t0"][b] + cfg["horizon"], cfg["x0"][b]); n = len(r["t"]); alphas.add(r["alpha"])
        assert res["status"][b] == 0 and res["num_nodes"][b] == n and np.array_equal(res["mode"][b, :n], r["mode"]) and res["perf"][b, 8] == r["alpha"]
        assert_blocks(res["x"][b, :n], r["x"], "x", TOL, b); assert_blocks(res["u"][b, :n], r["u"], "u", TOL, b)
        xo, _, _ = oracle.eval_policy(t1[b])
        r2 = oracle.ilqr_step(t1[b], t1[b] + cfg["horizon"], xo, warm=True); n2 = len(r2["t"])
        assert res2["status"][b] == 0 and res2["num_nodes"][b] == n2 and res2["perf"][b, 8] == r2["alpha"]
        assert_blocks(res2["x"][b, :n2], r2["x"], "x", TOL, b); assert_blocks(res2["u"][b, :n2], r2["u"], "u", TOL, b)
    assert len(alphas) >= 2 and min(alphas) > 0.0
    itf.set_setting(L.ST_SOLVER, 0.0)
    res3 = mpc.run(cfg["t0"], cfg["x0"], cfg["ref_t"], cfg["ref_x"], cfg["ev"], cfg["modes"], cfg["horizon"])
    _compare(res3, 0, _oracle_solve(oracle, cfg, 0))
    itf.close()


def test_ipm_solver_slot(blobs, oracle):
    """qmhip_set_setting(ST_SOLVER, 2): the `ipm` block (task.info:94-125; loaded at QMInterface.cpp:72, never instantiated).  No hard inequality rows in this OCP, so
    the interior-point iteration is the multiple-shooting step on the ipm block's parameters: changed through the C ABI here (dt 0.02, two iterations) and compared
    with the oracle on the same settings; switching back restores the SQP block"""
    from qm_control_amd import api, scenarios
    B, N = 4, 30
    cfg = scenarios.make_config("C3", batch=B, n_intervals=N)
    itf = api.QMInterface(blobs=blobs, max_batch=B, max_nodes=N + 24, max_ref_knots=2, max_events=cfg["ev"].shape[1])
    mpc = api.SqpMpc(itf)
    changes = ((L.ST_IPM_DT, 0.02), (L.ST_IPM_ITER, 2.0), (L.ST_SOLVER, 2.0))
    for idx, v in changes: itf.set_setting(idx, v); oracle.set_setting(idx, v)
    try:
        got = mpc.run(cfg["t0"], cfg["x0"], cfg["ref_t"], cfg["ref_x"], cfg["ev"], cfg["modes"], cfg["horizon"])
        for b in range(B):
            oracle.set_schedule(cfg["ev"][b], cfg["modes"][b]); oracle.set_target(cfg["ref_t"][b], cfg["ref_x"][b])
            t0 = float(cfg["t0"][b]); r = oracle.mpc_step(t0, t0 + cfg["horizon"], cfg["x0"][b]); r = oracle.mpc_step(t0, t0 + cfg["horizon"], cfg["x0"][b], warm="iterate")
            n = len(r["t"])
            assert got["status"][b] == 0 and got["num_nodes"][b] == n and n < N + 1 and np.array_equal(got["t"][b, :n], r["t"])
            assert_blocks(got["x"][b, :n], r["x"], "x", TOL, b); assert_blocks(got["u"][b, :n], r["u"], "u", TOL, b)
    finally:
        for idx in (L.ST_IPM_DT, L.ST_IPM_ITER, L.ST_SOLVER): oracle.set_setting(idx, float(blobs[1][idx]))
    itf.set_setting(L.ST_SOLVER, 0.0)
    res = mpc.run(cfg["t0"], cfg["x0"], cfg["ref_t"], cfg["ref_x"], cfg["ev"], cfg["modes"], cfg["horizon"])
    _compare(res, 0, _oracle_solve(oracle, cfg, 0))
    with pytest.raises(api.QmhipError): itf.set_setting(L.ST_SOLVER, 4.0)
    itf.close()


@pytest.mark.parametrize("name,N,fric,iters", [("C2", 100, 0.3, 1), ("C2", 40, 0.12, 3), ("C5", 30, 0.3, 3)])
def test_ipm_hard_cones(blobs, oblobs, name, N, fric, iters):
    """qmhip_set_setting(ST_SOLVER, 3): the interior-point method with HARD friction cones and arm position / velocity boxes (the `ipm` block of
    task.info:94-125) through the C ABI against oracle/src/ipm.h — itself pinned by a dense solve of the horizon's primal-dual Newton system (tests/test_ipm.py): x*, u* 1e-6 per
    block, integers bit-exact, step lengths / barrier parameter / slack / dual 1e-6, over `iters` iterations of one solve (ipm.ipmIteration); friction coefficient 0.12: a cone
    that binds; C5: arm within 0.1 rad of its joint limits.  Every slack stays positive, i.e. every iterate is strictly inside the linearised cones and boxes."""
    import pyoracle
    from qm_control_amd import api, scenarios
    cfg = scenarios.make_config(name, batch=2, n_intervals=N)
    itf = api.QMInterface(blobs=blobs, max_batch=2, max_nodes=N + 16, max_ref_knots=cfg["ref_t"].shape[1], max_events=cfg["ev"].shape[1])
    ost = oblobs[1].copy()
    for idx, v in ((L.ST_IPM_MU, 1e-2), (L.ST_FRIC_COEF, fric), (L.ST_IPM_ITER, float(iters)), (L.ST_SOLVER, 3.0)):
        itf.set_setting(idx, v); ost[idx] = v
    mpc = api.SqpMpc(itf)
    res = mpc.run(cfg["t0"], cfg["x0"], cfg["ref_t"], cfg["ref_x"], cfg["ev"], cfg["modes"], cfg["horizon"]); assert (res["status"] == 0).all(), res["status"]
    info = itf.debug_read("ipm_info", (2, 8)); nm = itf.max_nodes
    s_dev = itf.debug_read("ipm_s", (nm, 2, 28)); l_dev = itf.debug_read("ipm_l", (nm, 2, 28))
    for b in range(2):
        o = pyoracle.Oracle(oblobs[0], ost); o.set_schedule(cfg["ev"][b], cfg["modes"][b]); o.set_target(cfg["ref_t"][b], cfg["ref_x"][b])
        for it in range(iters):
            r = o.ipm_step(cfg["t0"][b], cfg["t0"][b] + cfg["horizon"], cfg["x0"][b], mode="cold" if it == 0 else "iterate")
        n = len(r["t"]); what = "%s instance %d" % (name, b)
        assert res["num_nodes"][b] == n and np.array_equal(res["t"][b, :n], r["t"]) and np.array_equal(res["event"][b, :n], r["ev"]) and np.array_equal(res["mode"][b, :n], r["mode"]), what
        assert_blocks(res["x"][b, :n], r["x"], "x", TOL, what + " x*"); assert_blocks(res["u"][b, :n], r["u"], "u", TOL, what + " u*")
        assert res["perf"][b, 8] == pytest.approx(r["alpha"], rel=1e-9) and info[b, 1] == pytest.approx(r["alpha_primal_max"], rel=1e-6) and info[b, 2] == pytest.approx(r["alpha_dual_max"], rel=1e-6)
        assert info[b, 4] == pytest.approx(r["barrier"], rel=1e-12) and np.abs(res["perf"][b, :8] - r["perf"][:8]).max() <= 1e-6 * max(1.0, np.abs(r["perf"][:8]).max()), what
        for i in range(n - 1):
            if r["ev"][i] == 1:
                continue
            p = o.ipm_node(i); on = p["on"] == 1
            assert (s_dev[i, b][on] > 0.0).all() and (l_dev[i, b][on] > 0.0).all(), (what, i)
            assert np.abs(s_dev[i, b][on] - p["slack"][on]).max() <= 1e-6 * max(1.0, np.abs(p["slack"][on]).max()) and np.abs(l_dev[i, b][on] - p["dual"][on]).max() <= 1e-6 * max(1e-3, np.abs(p["dual"][on]).max()), (what, i)
    itf.close()


def test_warned_solve_policy_at_t0_matches_the_robust_grid(blobs):
    """What the WBC consumes of a solve that carries QM_MPC_WARN_PIVOT — the policy at the observation time (and one control period later) — against the same solve on the
    robust grid (the degenerate node merged into the event node): <= 5e-6 per block — the bound of the whole-trajectory comparison; measured 2.8e-6 on the joint velocities, 7e-7 on the contact forces.  Every gait event inside C2's horizon x offsets -9e-7 ... -1e-12."""
    from qm_control_amd import api, scenarios
    from test_grid_fuzz import degenerate_cases
    cfg, cases = degenerate_cases(scenarios.make_config("C2", batch=1, n_intervals=100))
    B = cfg["B"]; pol = {}
    for name, dt_min in (("up", L.QM_GRID_DT_MIN_UPSTREAM), ("rob", L.QM_GRID_DT_MIN_ROBUST)):
        itf = api.QMInterface(blobs=blobs, max_batch=B, max_nodes=128, max_ref_knots=cfg["ref_t"].shape[1], max_events=cfg["ev"].shape[1])
        itf.set_setting(L.ST_GRID_DT_MIN, dt_min); mpc = api.SqpMpc(itf)
        res = mpc.run(cfg["t0"], cfg["x0"], cfg["ref_t"], cfg["ref_x"], cfg["ev"], cfg["modes"], cfg["horizon"])
        assert (res["status"] == (L.QM_MPC_WARN_PIVOT if name == "up" else 0)).all(), (name, res["status"])
        pol[name] = [mpc.evaluatePolicy(cfg["t0"] + d) for d in (0.0, 0.002)]; itf.close()
    for (xu, uu, mu), (xr, ur, mr) in zip(pol["up"], pol["rob"]):
        assert np.array_equal(mu, mr)
        assert_blocks(xu, xr, "x", 5e-6, "policy state at t0: warned solve vs robust grid"); assert_blocks(uu, ur, "u", 5e-6, "policy input at t0: warned solve vs robust grid")


def test_nan_observation_is_a_failed_solve_on_the_device(blobs):
    """-m gpu twin of tests/test_grid_fuzz.py::test_nan_observation_and_indefinite_stage_are_failures_not_warnings through the C ABI: a NaN in one instance's observation
    gives THAT instance status -4 (a failure: the adaptor throws, as [upstream] SqpSolver does on HPIPM's NaN status) and leaves its neighbour's solve untouched; a
    negated input weight (Huu indefinite on stages of positive duration) fails every instance."""
    from qm_control_amd import api, scenarios
    B = 2
    cfg = scenarios.make_config("C3", batch=B, n_intervals=20)
    kw = dict(max_batch=B, max_nodes=48, max_ref_knots=cfg["ref_t"].shape[1], max_events=cfg["ev"].shape[1])
    itf = api.QMInterface(blobs=blobs, **kw); mpc = api.SqpMpc(itf)
    clean = mpc.run(cfg["t0"], cfg["x0"], cfg["ref_t"], cfg["ref_x"], cfg["ev"], cfg["modes"], cfg["horizon"]); assert (clean["status"] == 0).all()
    x0 = cfg["x0"].copy(); x0[0, 7] = np.nan
    bad = mpc.run(cfg["t0"], x0, cfg["ref_t"], cfg["ref_x"], cfg["ev"], cfg["modes"], cfg["horizon"])
    assert bad["status"][0] == -4 and bad["status"][1] == 0, bad["status"]
    n = clean["num_nodes"][1]; assert np.array_equal(bad["x"][1, :n], clean["x"][1, :n]) and np.array_equal(bad["u"][1, :n], clean["u"][1, :n])
    itf.close()
    st = blobs[1].copy(); st[L.ST_R:L.ST_R + 900] *= -1.0
    itf = api.QMInterface(blobs=(blobs[0], st), **kw)
    r = api.SqpMpc(itf).run(cfg["t0"], cfg["x0"], cfg["ref_t"], cfg["ref_x"], cfg["ev"], cfg["modes"], cfg["horizon"]); itf.close()
    assert (r["status"] == -4).all(), r["status"]


def test_degenerate_interval_survives_on_the_device(blobs, oblobs):
    """-m gpu twin of tests/test_grid_fuzz.py::test_degenerate_interval_survives, FULL matrix: every gait event inside C2's horizon x offsets {-9e-7 ... -1e-12} of a
    shooting node in front of it.  Through the C ABI on [upstream]'s grid: status == QM_MPC_WARN_PIVOT (a warning, >= 0), integers and node times identical to the oracle's,
    x* / u* within 1e-6 per block of the oracle on the WHOLE trajectories, and within 5e-6 of the robust-grid solve everywhere but the degenerate interval's input.
    ST_RICCATI_STRICT = 1 turns the same solve into the hard failure -4 of rounds 1-3."""
    import pyoracle
    from qm_control_amd import api, scenarios
    from test_grid_fuzz import degenerate_cases, check_degenerate_against_robust
    cfg, cases = degenerate_cases(scenarios.make_config("C2", batch=1, n_intervals=100))
    B = cfg["B"]; assert B >= 15
    runs = {}
    for name, dt_min, strict in (("up", L.QM_GRID_DT_MIN_UPSTREAM, 0.0), ("rob", L.QM_GRID_DT_MIN_ROBUST, 0.0), ("strict", L.QM_GRID_DT_MIN_UPSTREAM, 1.0)):
        itf = api.QMInterface(blobs=blobs, max_batch=B, max_nodes=128, max_ref_knots=cfg["ref_t"].shape[1], max_events=cfg["ev"].shape[1])
        itf.set_setting(L.ST_GRID_DT_MIN, dt_min); itf.set_setting(L.ST_RICCATI_STRICT, strict)
        runs[name] = api.SqpMpc(itf).run(cfg["t0"], cfg["x0"], cfg["ref_t"], cfg["ref_x"], cfg["ev"], cfg["modes"], cfg["horizon"]); itf.close()
    assert (runs["up"]["status"] == L.QM_MPC_WARN_PIVOT).all() and (runs["rob"]["status"] == 0).all() and (runs["strict"]["status"] == -4).all()
    ost = oblobs[1].copy(); ostrob = ost.copy(); ostrob[L.ST_GRID_DT_MIN] = L.QM_GRID_DT_MIN_ROBUST
    for b, (k_ev, off) in enumerate(cases):
        what = "event %d offset %g" % (k_ev, off); dev = {}
        for name, s in (("up", ost), ("rob", ostrob)):
            o = pyoracle.Oracle(oblobs[0], s); r = _oracle_solve(o, cfg, b); res = runs[name]; n = len(r["t"])
            assert r["warn"] == (L.QM_MPC_WARN_PIVOT if name == "up" else 0), what
            assert res["num_nodes"][b] == n and np.array_equal(res["t"][b, :n], r["t"]) and np.array_equal(res["event"][b, :n], r["ev"]) and np.array_equal(res["mode"][b, :n], r["mode"]), what
            assert_blocks(res["x"][b, :n], r["x"], "x", TOL, what + " x* vs oracle (%s)" % name); assert_blocks(res["u"][b, :n], r["u"], "u", TOL, what + " u* vs oracle (%s)" % name)
            dev[name] = dict(t=res["t"][b, :n], ev=res["event"][b, :n], mode=res["mode"][b, :n], x=res["x"][b, :n], u=res["u"][b, :n])
        check_degenerate_against_robust(dev["up"], dev["rob"], 5e-6, what + " (device)")
