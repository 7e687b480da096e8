"""GPU parity tests of the reference / gait front-end through the C ABI, against oracle/frontend.py.

Bar: mode schedules (integers AND event times: same f64 additions in the same order) bit-exact; targets 1e-12; the MPC fed by the
device-resident front-end within 1e-6 of the oracle fed by the oracle's front-end.
"""
import os
import sys
import numpy as np
import pytest
from conftest import assert_blocks, rel_err

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
pytestmark = pytest.mark.gpu
TOL = 1e-6
PTS = 0.1


def _oracle_gait(fe, gaits, default="stance"):
    g = gaits[default]
    return fe.GaitSchedule([0.5], [15, 15], g["switchingTimes"], g["modeSequence"], PTS)


def test_gait_schedule_command_streams(blobs):
    """random gait commands over all 12 templates, 40 MPC calls, 64 instances: the device-resident GaitSchedule state and the exported
    solver schedule equal the oracle's std::vector restatement bit for bit"""
    import frontend as fe
    from qm_control_amd import api, scenarios
    gaits = scenarios.load_gaits(); names = list(gaits.keys())
    B, horizon = 64, 1.5
    itf = api.QMInterface(blobs=blobs, max_batch=B, max_nodes=16, max_ref_knots=2, max_events=128)
    assert itf.settings_blob[986] == PTS                     # ST_PHASE_TRANS_STANCE, task.info:11
    mpc = api.SqpMpc(itf)
    gs = api.GaitSchedule(itf, gaits, B)
    orc = [_oracle_gait(fe, gaits) for _ in range(B)]
    rng = np.random.default_rng(17)
    t = rng.uniform(0.0, 0.5, B); x0 = np.tile(itf.getInitialState(), (B, 1))
    mpc.B = B
    for step in range(40):
        mpc.set_initial(t, x0)
        req = [names[rng.integers(0, 12)] if rng.uniform() < 0.3 else None for _ in range(B)]
        gs.preSolverRun(req, t, horizon)
        for b in range(B):
            if req[b] is not None:
                g = gaits[req[b]]; orc[b].pre_solver_run_insert(g["switchingTimes"], g["modeSequence"], t[b], t[b] + horizon)
        gs.updateSolverSchedule(horizon)
        sched = [o.modify_references(t[b], horizon) for b, o in enumerate(orc)]
        dev = gs.download(); ev, mo = gs.solver_schedule()
        assert (dev["status"] == 0).all()
        for b in range(B):
            n = dev["n"][b]; o = orc[b]
            assert n == len(o.event_times) and np.array_equal(dev["event_times"][b, :n], np.array(o.event_times)) and np.array_equal(dev["mode_sequence"][b, :n + 1], np.array(o.mode_sequence))
            assert dev["template"][b] == (names.index(req[b]) if req[b] is not None else dev["template"][b])
            assert np.array_equal(ev[b, :n], np.array(sched[b][0])) and np.array_equal(mo[b, :n + 1], np.array(sched[b][1])) and (mo[b, n + 1:] == 15).all()
        t = t + rng.uniform(0.01, 0.5, B)
    itf.close()


def test_schedule_capacity_is_reported(blobs):
    from qm_control_amd import api, scenarios
    gaits = scenarios.load_gaits()
    itf = api.QMInterface(blobs=blobs, max_batch=2, max_nodes=16, max_ref_knots=2, max_events=8)       # too few solver slots for 3 x 1.5 s of pace
    mpc = api.SqpMpc(itf); mpc.B = 2
    gs = api.GaitSchedule(itf, gaits, 2, default_gait="pace")
    mpc.set_initial(np.array([1.0, 1.0]), np.tile(itf.getInitialState(), (2, 1)))
    gs.updateSolverSchedule(1.5)
    assert (gs.download()["status"] == -3).all()
    itf.close()


def test_targets_from_commands(blobs, oracle):
    import frontend as fe
    from qm_control_amd import api, scenarios
    B = 32
    itf = api.QMInterface(blobs=blobs, max_batch=B, max_nodes=16, max_ref_knots=3, max_events=8)
    mpc = api.SqpMpc(itf); mpc.B = B
    mb, st = blobs
    rng = np.random.default_rng(23)
    x0 = np.tile(itf.getInitialState(), (B, 1)) + rng.uniform(-0.05, 0.05, (B, 30)); t0 = rng.uniform(0.5, 5.0, B)
    mpc.set_initial(t0, x0)
    # EE state of the observation: the oracle's forward kinematics (what QMController publishes, QMController.cpp:173-174)
    ee = np.array([oracle.rbd_from_q(np.concatenate([x0[b, 6:9], x0[b, 9:12], x0[b, 12:30]]))[48:55] for b in range(B)])
    kind = rng.integers(1, 4, B).astype(np.int32); kind[0] = 0
    cmd = np.zeros((B, 7))
    for b in range(B):
        if kind[b] == 1: cmd[b, :4] = rng.uniform(-0.5, 0.5, 4)
        elif kind[b] == 2: cmd[b, :3] = rng.uniform(-0.2, 0.2, 3)
        else:
            cmd[b, :3] = ee[b, :3] + rng.uniform(-0.3, 0.3, 3); q = rng.normal(size=4); cmd[b, 3:] = q / np.linalg.norm(q)
    qnom = mb[scenarios.MB_QNOM:scenarios.MB_QNOM + 18]
    for thru in (False, True):
        pub = api.TargetTrajectoriesPublisher(itf, B)             # time_to_target = mpc.timeHorizon of the settings
        T = itf.settings_blob[996]
        pub.publish(kind, cmd, ee_state=ee, ee_through_float=thru)
        rt, rx, last = pub.download()
        for b in range(1, B):
            o = fe.TargetPublisher(qnom, 0.4, 0.3, 0.1, T)
            eb = fe.ee_state_through_float(ee[b]) if thru else ee[b]
            ort, orx = {1: o.cmd_vel, 2: o.ee_cmd_vel, 3: o.ee_goal}[int(kind[b])](cmd[b], t0[b], x0[b], eb)
            assert np.allclose(rt[b, :2], ort, rtol=1e-13, atol=0) and np.allclose(rx[b, :2], orx, rtol=1e-12, atol=1e-13), b
            assert np.allclose(last[b], o.last_ee, rtol=1e-13, atol=0) and np.array_equal(rx[b, 2], rx[b, 1]) and rt[b, 2] > rt[b, 1]
    # forward kinematics on the device (ee_state = None) == the oracle's EE pose
    pub = api.TargetTrajectoriesPublisher(itf, B)
    pub.publish(np.full(B, api.CMD_EE_VEL, np.int32), np.zeros((B, 7)))
    rt, rx, last = pub.download()
    assert rel_err(rx[:, 0, 30:37], ee) <= 1e-10
    itf.close()


def test_closed_loop_with_gait_switch_and_velocity_command(blobs, oracle):
    """stance -> trot switch requested during the second MPC call, walking under a cmd_vel target: front-end, warm-started MPC and the
    plant all on the device, against the oracle driven by the oracle's front-end"""
    import frontend as fe
    from qm_control_amd import api, scenarios
    gaits = scenarios.load_gaits()
    B, N, steps, dt_mpc = 3, 40, 6, 0.2
    mb, st = blobs
    horizon = N * st[scenarios.ST_SQP_DT]
    itf = api.QMInterface(blobs=blobs, max_batch=B, max_nodes=96, max_ref_knots=2, max_events=48)
    mpc = api.SqpMpc(itf); mpc.B = B
    gs = api.GaitSchedule(itf, gaits, B)
    pub = api.TargetTrajectoriesPublisher(itf, B, time_to_target=1.0)
    rng = np.random.default_rng(2)
    x0 = np.tile(itf.getInitialState(), (B, 1)); x0[:, 6:8] += rng.uniform(-0.02, 0.02, (B, 2)); x0[:, 12:24] += rng.uniform(-0.03, 0.03, (B, 12))
    t0 = np.array([0.1, 0.23, 0.4])
    cmd = np.zeros((B, 7)); cmd[:, 0] = [0.3, 0.2, -0.2]; cmd[:, 3] = [0.0, 0.2, -0.1]
    qnom = mb[scenarios.MB_QNOM:scenarios.MB_QNOM + 18]
    ee_of = lambda x: oracle.rbd_from_q(np.concatenate([x[6:9], x[9:12], x[12:30]]))[48:55]
    got = []
    mpc.set_initial(t0, x0)
    for k in range(steps):
        if k > 0:
            mpc.advance(dt_mpc)
        if k == 1:
            gs.preSolverRun(["trot", "trot", "standing_trot"], t0 + k * dt_mpc, horizon)
        if k in (0, 3):
            pub.publish(np.full(B, api.CMD_VEL, np.int32), cmd)                  # EE state by forward kinematics of the resident x0
        gs.updateSolverSchedule(horizon)
        mpc.solve_resident(horizon, warm=(k > 0))
        got.append((mpc.download(), gs.solver_schedule(), pub.download()))
    for b in range(B):
        og = _oracle_gait(fe, gaits); op = fe.TargetPublisher(qnom, 0.4, 0.3, 0.1, 1.0)
        t = float(t0[b]); x = x0[b]
        for k in range(steps):
            if k > 0:
                t += dt_mpc; x, _, _ = oracle.eval_policy(t)
            if k == 1:
                g = gaits[["trot", "trot", "standing_trot"][b]]; og.pre_solver_run_insert(g["switchingTimes"], g["modeSequence"], float(t0[b]) + k * dt_mpc, float(t0[b]) + k * dt_mpc + horizon)
            if k in (0, 3):
                rt, rx = op.cmd_vel(cmd[b], t, x, ee_of(x))
            ev, mo = og.modify_references(t, horizon)
            res, (dev_ev, dev_mo), (dev_rt, dev_rx, _) = got[k]
            assert np.array_equal(dev_ev[b, :len(ev)], np.array(ev)) and np.array_equal(dev_mo[b, :len(mo)], np.array(mo)), (b, k)
            assert rel_err(dev_rt[b], rt) <= 1e-12 and rel_err(dev_rx[b], rx) <= 1e-9, (b, k)
            oracle.set_schedule(np.array(ev), np.array(mo, dtype=np.int32)); oracle.set_target(rt, rx)
            r = oracle.mpc_step(t, t + horizon, x, warm=(k > 0)); n = len(r["t"])
            assert res["status"][b] == 0 and res["num_nodes"][b] == n, (b, k)
            assert np.array_equal(res["event"][b, :n], r["ev"]) and np.array_equal(res["mode"][b, :n], r["mode"]), (b, k)
            assert_blocks(res["x"][b, :n], r["x"], "x", TOL, (b, k)); assert_blocks(res["u"][b, :n], r["u"], "u", TOL, (b, k))
        assert set(res["mode"][b, :n]) - {15} != set(), b       # the new gait is inside the horizon by the last call
    itf.close()


def test_fused_closed_loop_refreshes_the_schedule(blobs):
    """qmhip_closed_loop_resident with an active gait front-end == the same steps issued one by one (bit for bit), trotting from a stance start"""
    from qm_control_amd import api, scenarios
    gaits = scenarios.load_gaits()
    B, N, steps, dt_mpc = 8, 40, 8, 0.15
    horizon = N * blobs[1][scenarios.ST_SQP_DT]
    rng = np.random.default_rng(4)
    x0 = np.tile(blobs[1][scenarios.ST_XINIT:scenarios.ST_XINIT + 30], (B, 1)); x0[:, 12:24] += rng.uniform(-0.03, 0.03, (B, 12)); t0 = rng.uniform(0.0, 0.3, B)
    cmd = np.zeros((B, 7)); cmd[:, 0] = 0.25
    outs = []
    for fused in (False, True):
        itf = api.QMInterface(blobs=blobs, max_batch=B, max_nodes=96, max_ref_knots=2, max_events=48)
        mpc = api.SqpMpc(itf); mpc.B = B; wbc = api.HierarchicalWbc(itf); wbc.reset()
        gs = api.GaitSchedule(itf, gaits, B); pub = api.TargetTrajectoriesPublisher(itf, B, time_to_target=2.0)
        mpc.set_initial(t0, x0)
        gs.insertModeSequenceTemplate("trot", 0.6, 1.0)                                  # trot from t = 0.6 on
        pub.publish(np.full(B, api.CMD_VEL, np.int32), cmd)
        if fused:
            mpc.closed_loop_resident(steps, dt_mpc, horizon, 0.002, 20.0)
        else:
            for k in range(steps):
                if k > 0:
                    mpc.advance(dt_mpc)
                gs.updateSolverSchedule(horizon); mpc.solve_resident(horizon, warm=(k > 0))
                xd, ud, mode = mpc.evaluatePolicy(t0 + k * dt_mpc)
        res = mpc.download(); st = gs.download()
        assert (res["status"] == 0).all() and (st["status"] == 0).all()
        outs.append((res, gs.solver_schedule()))
        if fused:
            out, qps = wbc.download(B); assert (qps == 0).all() and np.isfinite(out).all()
        itf.close()
    (r0, s0), (r1, s1) = outs
    assert np.array_equal(s0[0], s1[0]) and np.array_equal(s0[1], s1[1])
    assert np.array_equal(r0["x"], r1["x"]) and np.array_equal(r0["u"], r1["u"]) and np.array_equal(r0["mode"], r1["mode"])
    assert (r1["mode"][:, :r1["num_nodes"].min()] != 15).any()                          # the robots are trotting by the end


def test_closed_loop_reports_a_failed_schedule_update(blobs):
    """with too few solver event slots the schedule refresh inside the fused closed loop fails (-3); K0 used to reset the status to 0 and the
    MPC silently solved on the stale schedule.  Now: sticky front-end status, non-zero MPC status for the failed instances, the others unaffected."""
    from qm_control_amd import api, scenarios
    gaits = scenarios.load_gaits()
    B, N = 4, 20
    horizon = N * blobs[1][scenarios.ST_SQP_DT]
    itf = api.QMInterface(blobs=blobs, max_batch=B, max_nodes=64, max_ref_knots=2, max_events=8)      # stance fits, [t − T, t + 2T] of trot does not
    mpc = api.SqpMpc(itf); mpc.B = B; wbc = api.HierarchicalWbc(itf); wbc.reset()
    gs = api.GaitSchedule(itf, gaits, B); pub = api.TargetTrajectoriesPublisher(itf, B, time_to_target=2.0)
    x0 = np.tile(itf.getInitialState(), (B, 1)); t0 = np.full(B, 0.1)
    mpc.set_initial(t0, x0)
    pub.publish(np.full(B, api.CMD_VEL, np.int32), np.zeros((B, 7)))
    mpc.closed_loop_resident(2, 0.05, horizon, 0.002, 20.0)
    assert (mpc.download()["status"] == 0).all() and (gs.download()["status"] == 0).all()            # standing: fine
    gs.insertModeSequenceTemplate(["trot", None, "trot", None], 0.3, 5.0)                            # instances 0 and 2 switch to a long trot
    mpc.closed_loop_resident(3, 0.05, horizon, 0.002, 20.0)
    st = mpc.download()["status"]; fs = gs.download()["status"]
    assert list(fs) == [-3, 0, -3, 0] and list(st) == [-3, 0, -3, 0], (fs, st)
    mpc.closed_loop_resident(1, 0.05, horizon, 0.002, 20.0)                                          # ... and it stays reported
    assert list(mpc.download()["status"]) == [-3, 0, -3, 0]
    itf.close()


def test_golden_front_end_stream(blobs):
    """the committed fixture tests/golden/frontend_stream.npz (tools/gen_golden_frontend.py) through the C ABI"""
    from qm_control_amd import api, scenarios
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import gen_golden_frontend as gg
    G = np.load(os.path.join(root, "tests", "golden", "frontend_stream.npz"))
    gaits = scenarios.load_gaits(); names, ts, reqs = gg.stream()
    itf = api.QMInterface(blobs=blobs, max_batch=gg.B, max_nodes=16, max_ref_knots=2, max_events=gg.CAP)
    mpc = api.SqpMpc(itf); mpc.B = gg.B
    gs = api.GaitSchedule(itf, gaits, gg.B)
    x0 = np.tile(itf.getInitialState(), (gg.B, 1))
    for s in range(gg.STEPS):
        mpc.set_initial(ts[s], x0)
        gs.preSolverRun([None if r < 0 else names[r] for r in reqs[s]], ts[s], gg.HORIZON)
        gs.updateSolverSchedule(gg.HORIZON)
        ev, mo = gs.solver_schedule()
        for b in range(gg.B):
            n = G["n"][s, b]
            assert np.array_equal(ev[b, :n], G["ev"][s, b, :n]) and np.array_equal(mo[b, :n + 1], G["mo"][s, b, :n + 1])
    itf.close()
    itf = api.QMInterface(blobs=blobs, max_batch=6, max_nodes=16, max_ref_knots=2, max_events=8)
    mpc = api.SqpMpc(itf); mpc.B = 6; mpc.set_initial(G["tgt_t0"], G["tgt_x0"])
    pub = api.TargetTrajectoriesPublisher(itf, 6, time_to_target=1.0)
    pub.publish(G["tgt_kind"], G["tgt_cmd"], ee_state=G["tgt_ee"])
    rt, rx, last = pub.download()
    assert np.allclose(rt, G["tgt_rt"], rtol=1e-13, atol=0) and np.allclose(rx, G["tgt_rx"], rtol=1e-12, atol=1e-13) and np.allclose(last, G["tgt_last"], rtol=1e-13, atol=0)
    itf.close()
