"""Batched rigid-body plant: the emulated kernel against the CPU oracle's restatement (oracle/src/sim.h) — command law with the
delay buffer of QMHWSim::writeSim, forward dynamics, penalty contact — plus physical sanity checks of the oracle itself."""
import numpy as np
from qm_control_amd import layout as L
import pytest
from conftest import rel_err


def nominal_q(st, z=None):
    q = np.array(st[936:960], float)      # initialState: base pose + joints
    if z is not None:
        q[2] = z
    return q


def stand_height(oracle, st):
    """base height at which the nominal feet just touch the ground (foot radius 0.02)"""
    q = nominal_q(st); rbd = oracle.rbd_from_q(q)
    import ctypes as C
    pos = np.zeros(3); R = np.zeros(9)
    zs = []
    for f in range(4):
        oracle.lib.qmo_frame_pose(oracle.h, q.ctypes.data_as(C.POINTER(C.c_double)), C.c_int(f), pos.ctypes.data_as(C.POINTER(C.c_double)), R.ctypes.data_as(C.POINTER(C.c_double)))
        zs.append(pos[2])
    return q[2] - min(zs) + 0.02


def random_cases(oracle, st, n, seed):
    rng = np.random.default_rng(seed); z0 = stand_height(oracle, st); cases = []
    for k in range(n):
        q = nominal_q(st, z0 - 0.004 + 0.01 * rng.random()); q[3:6] += 0.05 * rng.normal(size=3); q[6:] += 0.05 * rng.normal(size=18)
        v = 0.2 * rng.normal(size=24)
        kp = np.concatenate([np.full(12, 0.0 if k % 2 else 60.0), np.full(6, 20.0)]); kd = np.concatenate([np.full(12, 3.0), np.full(6, 0.5)])   # kd as the reference commands them (QMController.cpp:183,188; cfg/weight.cfg:8)
        cases.append(dict(q=q, v=v, pos=nominal_q(st)[6:] + 0.05 * rng.normal(size=18), vel=0.1 * rng.normal(size=18), kp=kp, kd=kd, ff=5.0 * rng.normal(size=18)))
    return cases


def test_oracle_free_fall_and_static_stance(blobs, oracle):
    mb, st = blobs
    # free fall far above the ground with no command: base accelerates with g, total momentum in x/y stays zero
    oracle.sim_params(); q = nominal_q(st, 2.0); oracle.sim_reset(q, np.zeros(24), 0.0); oracle.sim_command(0, 0, 0, 0, 0)
    for _ in range(10):
        r = oracle.sim_step(0.001, 1)
    assert r["status"] == 0 and not r["contact"].any()
    assert abs(r["v"][2] + 9.81 * 0.01) < 5e-3 and np.abs(r["force"]).max() == 0.0
    # standing on compliant legs with the contact active: the conjugate momentum of the base z translation, (M v)[2] = total vertical momentum, changes by
    # the impulse of (sum of normal forces − weight) — Newton's law for the whole tree, whatever the joints do
    z0 = stand_height(oracle, st); q = nominal_q(st, z0 - 0.002)
    kp = np.concatenate([np.full(12, 300.0), np.full(6, 20.0)]); kd = np.concatenate([np.full(12, 3.0), np.full(6, 0.5)])   # explicit joint law: kd h must stay below 2 x the joint inertia (wrist: 5.8e-4)
    oracle.sim_reset(q, np.zeros(24), 0.0); oracle.sim_command(q[6:], 0.0, kp, kd, 0.0)
    weight = mb[L.MB_ROBOTMASS] * 9.81; imp = 0.0; h = 0.001
    for k in range(200):
        r = oracle.sim_step(h, 1); imp += (r["force"][2::3].sum() - weight) * h
        if k % 50 == 49:
            M = oracle.wbc(st[L.ST_XINIT:L.ST_XINIT + 30], np.zeros(30), oracle.rbd_from_q(r["q"], np.zeros(24)), 15, 0.002, 20.0, debug=True)[2]["M"]
            assert abs((M @ r["v"])[2] - imp) < 0.02 * max(1.0, abs(imp)), (k, (M @ r["v"])[2], imp)
    assert r["status"] == 0 and r["contact"].all()
    assert abs(r["q"][2] - z0) < 0.01 and np.abs(r["q"][3:6]).max() < 0.05 and np.isfinite(r["v"]).all()


def test_oracle_command_delay(blobs, oracle):
    """a feed-forward torque step reaches the joints `delay` after it was commanded (QMHWSim.cpp:100-113)"""
    mb, st = blobs
    oracle.sim_params(delay=0.009, saturate_effort=0.0); q = nominal_q(st, 2.0); oracle.sim_reset(q, np.zeros(24), 0.0); oracle.sim_command(0, 0, 0, 0, 0)
    for _ in range(3):
        oracle.sim_step(0.001, 1)
    ff = np.zeros(18); ff[17] = 1.0                       # last arm joint
    oracle.sim_command(0, 0, 0, 0, ff)
    acc = []
    for k in range(14):
        v0 = oracle.sim_step(0.001, 1)["v"][23]; acc.append(v0)
    dv = np.diff(np.array([0.0] + acc))                   # joint-velocity increments per step
    first = int(np.argmax(np.abs(dv) > 1e-3 * np.abs(dv).max()))
    assert first in (8, 9), (first, dv)          # the applied command is the oldest one not older than the delay: its age lies in (delay - period, delay]


@pytest.mark.parametrize("nsub", [1, 2])
def test_emulated_plant_vs_oracle(blobs, oracle, nsub):
    import emu_harness
    mb, st = blobs
    cases = random_cases(oracle, st, 4, 5)
    B = len(cases); arr = lambda k: np.array([c[k] for c in cases])
    e = emu_harness.Emu(mb, st, 8, 8, 2, 2)
    e.sim_params(); e.sim_reset(arr("q"), arr("v"), 1.0); e.sim_command(arr("pos"), arr("vel"), arr("kp"), arr("kd"), arr("ff"))
    steps = 12
    out = [e.sim_step(0.001, nsub) for _ in range(steps)]
    for b, c in enumerate(cases):
        oracle.sim_params(); oracle.sim_reset(c["q"], c["v"], 1.0); oracle.sim_command(c["pos"], c["vel"], c["kp"], c["kd"], c["ff"])
        for k in range(steps):
            r = oracle.sim_step(0.001, nsub)
            assert r["status"] == 0 and out[k]["status"][b] == 0
            assert rel_err(out[k]["q"][b], r["q"]) < 1e-9 and rel_err(out[k]["v"][b], r["v"]) < 1e-7, (b, k)
            assert rel_err(out[k]["force"][b], r["force"]) < 1e-6 and list(out[k]["contact"][b]) == list(r["contact"]), (b, k)
            assert rel_err(out[k]["rbd"][b], r["rbd"]) < 1e-7, (b, k)


def _rot_zyx(z, y, x):
    cz, sz, cy, sy, cx, sx = np.cos(z), np.sin(z), np.cos(y), np.sin(y), np.cos(x), np.sin(x)
    return np.array([[cz * cy, cz * sy * sx - sz * cx, cz * sy * cx + sz * sx], [sz * cy, sz * sy * sx + cz * cx, sz * sy * cx - cz * sx], [-sy, cy * sx, cy * cx]])


def centroidal_from_rbd(mb, rbd):
    """computeCentroidalStateFromRbdModel [upstream], SRBD: normalized momentum = A_b(q) v_base / m"""
    from qm_control_amd import scenarios as sc
    R = _rot_zyx(*rbd[0:3]); Inom = mb[sc.MB_INOM:sc.MB_INOM + 9].reshape(3, 3); rnom = mb[sc.MB_RNOM:sc.MB_RNOM + 3]; m = mb[sc.MB_ROBOTMASS]
    w = rbd[24:27]; x = np.zeros(30)
    x[0:3] = rbd[27:30] + np.cross(R @ rnom, w); x[3:6] = (R @ Inom @ R.T @ w) / m; x[6:9] = rbd[3:6]; x[9:12] = rbd[0:3]; x[12:30] = rbd[6:24]
    return x


def robust_grid_settings(st):
    """settings blob of a fixed-rate loop: the time grid's robust minimum step (what api.QMHWSim(robust_grid=True) sets on a device context)"""
    from qm_control_amd import layout as L
    st = np.array(st, float); st[L.ST_GRID_DT_MIN] = L.QM_GRID_DT_MIN_ROBUST
    return st


def oracle_closed_loop(oracle, mb, cfg, q0, n_ticks, period, nsub, mpc_every, horizon, arm_kp, arm_kd, time0, pipelined=False, controller=0, robust_grid=True):
    """robust_grid: the loop's observation times (time0 + k ms) share a raster with the gait events, so a grid node can land within weakEpsilon of an event; like the
    device loops (api.QMHWSim(robust_grid=True)) the oracle then runs with ST_GRID_DT_MIN = QM_GRID_DT_MIN_ROBUST for the duration of the loop"""
    from qm_control_amd import layout as L
    old = oracle.set_setting(L.ST_GRID_DT_MIN, L.QM_GRID_DT_MIN_ROBUST if robust_grid else L.QM_GRID_DT_MIN_UPSTREAM)
    try:
        return _oracle_closed_loop(oracle, mb, cfg, q0, n_ticks, period, nsub, mpc_every, horizon, arm_kp, arm_kd, time0, pipelined, controller)
    finally:
        oracle.set_setting(L.ST_GRID_DT_MIN, old)


def _oracle_closed_loop(oracle, mb, cfg, q0, n_ticks, period, nsub, mpc_every, horizon, arm_kp, arm_kd, time0, pipelined=False, controller=0):
    """QMController::update around the oracle's plant, same order of operations as qmhip_closed_loop_sim; pipelined: as qmhip_closed_loop_sim_pipelined — the MPC
    triggered at a tick observes the plant there, its solution is used from the next MPC period on (the first one is synchronous)"""
    oracle.set_schedule(cfg["ev"][0], cfg["modes"][0]); oracle.set_target(cfg["ref_t"][0], cfg["ref_x"][0])
    oracle.wbc_reset(); oracle.sim_params(); oracle.sim_reset(q0, np.zeros(24), time0); oracle.sim_command(0, 0, 0, 0, 0)
    st = dict(rbd=oracle.rbd_from_q(q0, np.zeros(24)), time=time0, k=0); log = []
    pos = np.zeros(18); vel = np.zeros(18); kp = np.zeros(18); kd = np.zeros(18); ff = np.zeros(18)
    arm_hold = np.array(q0[18:24], float); arm_last = np.full(6, float(time0))      # QMMpcController: held position commands, last_time_ (QMController.cpp:127)

    def tick():
        time, rbd = st["time"], st["rbd"]
        xd, ud, mode = oracle.eval_policy(time)
        if st["k"] == 0:
            oracle.wbc_set_input_last(ud)          # inputLast_ primed with the planned input at the first tick (qmhip_closed_loop_sim)
        out, wst = oracle.wbc(xd, ud, rbd, mode, period, time, mpc_variant=(controller == 1))
        if controller == 1:                        # QMMpcController::updateControlLaw (QMController.cpp:431-445)
            pos[:12] = xd[12:24]; vel[:12] = ud[12:24]; kp[:12] = 0.0; kd[:12] = 3.0; ff[:12] = out[36:48]
            for j in range(6):
                if time - arm_last[j] > 1.0 / 100.0:
                    arm_hold[j] = rbd[18 + j] + ud[24 + j] * 1.0 / 100.0; arm_last[j] = time
            pos[12:] = arm_hold; vel[12:] = 0.0; kp[12:] = arm_kp; kd[12:] = arm_kd; ff[12:] = 0.0
        else:
            if time > 10.0:
                pos[:12] = xd[12:24]; vel[:12] = ud[12:24]; kp[:12] = 0.0; kd[:12] = 3.0; ff[:12] = out[36:48]
            pos[12:] = xd[24:30]; vel[12:] = 0.0; kp[12:] = arm_kp; kd[12:] = arm_kd; ff[12:] = out[48:54]
        oracle.sim_command(pos, vel, kp, kd, ff)
        r = oracle.sim_step(period, nsub); st["rbd"] = r["rbd"]; st["time"] = r["time"]; st["k"] += 1
        log.append(dict(q=r["q"].copy(), v=r["v"].copy(), tau=out[36:].copy(), wbc_status=list(wst), mode=mode))

    if not pipelined:
        for k in range(n_ticks):
            if k % mpc_every == 0:
                oracle.mpc_step(st["time"], st["time"] + horizon, centroidal_from_rbd(mb, st["rbd"]), warm=(k > 0))
            tick()
        return log
    assert n_ticks % mpc_every == 0
    for p in range(n_ticks // mpc_every):
        t_obs, x_obs = st["time"], centroidal_from_rbd(mb, st["rbd"])
        if p == 0:
            oracle.mpc_step(t_obs, t_obs + horizon, x_obs, warm=False)
        for _ in range(mpc_every):
            tick()
        if p > 0:
            oracle.mpc_step(t_obs, t_obs + horizon, x_obs, warm=True)
    return log


def test_emulated_closed_loop_around_the_plant_vs_oracle(blobs, oracle):
    """the product's tick sequence (qm_closed_loop_sim_ticks: state estimate -> MPC -> policy -> WBC -> updateControlLaw -> simulation step) on the host emulator
    against the same loop built from the oracle's pieces: 10 ticks of a stance -> trot schedule with MPC calls at ticks 0 and 6"""
    import os, sys, emu_harness
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from sim_closed_loop_demo import setup
    mb, st = blobs
    horizon = 0.45; c = setup("trot", 1, horizon, t_start=20.2)
    q0 = c["xbar"][6:30].copy(); q0[2] = 0.385
    e = emu_harness.Emu(mb, robust_grid_settings(st), 1, 64, 2, c["ev"].shape[1])
    c["horizon"] = horizon; c["B"] = 1; e.grid_only(c, batch=1)   # uploads reference and schedule
    e.lib.emu_wbc_reset(e.h); e.sim_params(); e.sim_reset(q0[None], np.zeros((1, 24)), 20.2); e.sim_command(0, 0, 0, 0, 0)
    n_ticks = 10; dev = []
    for k in range(n_ticks):
        e.closed_loop_sim(1, 0.001, horizon, nsub=2, mpc_every=6, restart=(k == 0)); dev.append(e.sim_state())
    log = oracle_closed_loop(oracle, mb, c, q0, n_ticks, 0.001, 2, 6, horizon, 0.0, 0.5, 20.2)
    for k in range(n_ticks):
        assert dev[k]["mpc_status"][0] == 0 and list(dev[k]["wbc_status"][0]) == [0, 0, 0], k
        assert rel_err(dev[k]["tau"][0], log[k]["tau"]) < 1e-6 and rel_err(dev[k]["q"][0], log[k]["q"]) < 1e-9 and rel_err(dev[k]["v"][0], log[k]["v"]) < 1e-7, k


def test_emulated_mpc_controller_loop_vs_oracle(blobs, oracle):
    """the QMMpcController variant of the loop (HierarchicalMpcWbc; legs commanded on every tick, the arm as position commands re-published at 100 Hz:
    QMController.cpp:410-414, 431-445) on the host emulator against the same loop built from the oracle's pieces; time < 10, so the QMController law
    would NOT command the legs here — 24 ticks cover two arm publications"""
    import os, sys, emu_harness
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from sim_closed_loop_demo import setup
    mb, st = blobs
    horizon = 0.45; t_start = 5.2; c = setup("trot", 1, horizon, t_start=t_start)
    q0 = c["xbar"][6:30].copy(); q0[2] = 0.385
    e = emu_harness.Emu(mb, robust_grid_settings(st), 1, 64, 2, c["ev"].shape[1])
    c["horizon"] = horizon; c["B"] = 1; e.grid_only(c, batch=1)
    e.sim_set_controller(1)
    e.lib.emu_wbc_reset(e.h); e.sim_params(); e.sim_reset(q0[None], np.zeros((1, 24)), t_start); e.sim_command(0, 0, 0, 0, 0)
    n_ticks = 24; dev = []; arm_kp, arm_kd = 60.0, 2.0
    for k in range(n_ticks):
        e.closed_loop_sim(1, 0.001, horizon, nsub=2, mpc_every=8, arm_kp=arm_kp, arm_kd=arm_kd, restart=(k == 0)); dev.append(e.sim_state())
    log = oracle_closed_loop(oracle, mb, c, q0, n_ticks, 0.001, 2, 8, horizon, arm_kp, arm_kd, t_start, controller=1)
    for k in range(n_ticks):
        assert dev[k]["mpc_status"][0] == 0 and list(dev[k]["wbc_status"][0]) == [0, 0, 0], k
        assert rel_err(dev[k]["tau"][0], log[k]["tau"]) < 1e-6 and rel_err(dev[k]["q"][0], log[k]["q"]) < 1e-9 and rel_err(dev[k]["v"][0], log[k]["v"]) < 1e-7, k
    # the legs really were driven (time < 10): the plant did not simply collapse as it would under the QMController law before the legs are switched on
    assert dev[-1]["q"][0][2] > 0.36
    e.sim_set_controller(0)


def test_oracle_reproduces_the_plant_goldens(blobs, oracle):
    """the frozen plant / closed-loop trajectories (tools/gen_golden_sim.py) against a fresh run of the oracle"""
    import os, sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from sim_closed_loop_demo import setup
    mb, st = blobs
    g = np.load(os.path.join(ROOT, "tests", "golden", "sim_plant_B4_T12.npz"))
    for b in range(4):
        oracle.sim_params(); oracle.sim_reset(g["in_q_%d" % b], g["in_v_%d" % b], 1.0); oracle.sim_command(*[g["in_%s_%d" % (k, b)] for k in ("pos", "vel", "kp", "kd", "ff")])
        for k in range(12):
            r = oracle.sim_step(0.001, 2)
            assert rel_err(r["q"], g["q_%d" % b][k]) < 1e-12 and rel_err(r["v"], g["v_%d" % b][k]) < 1e-10 and list(r["contact"]) == list(g["contact_%d" % b][k])
    g = np.load(os.path.join(ROOT, "tests", "golden", "sim_closed_loop_trot_T20.npz"))
    c = setup("trot", 1, float(g["horizon"]), t_start=float(g["t_start"]))
    log = oracle_closed_loop(oracle, mb, c, g["q0"], 20, 0.001, 2, int(g["mpc_every"]), float(g["horizon"]), 0.0, 0.5, float(g["t_start"]))
    for k in range(20):
        assert rel_err(log[k]["q"], g["q"][k]) < 1e-10 and rel_err(log[k]["tau"], g["tau"][k]) < 1e-7 and log[k]["mode"] == int(g["mode"][k]), k



def test_emulated_pipelined_loop_vs_oracle(blobs, oracle):
    """qm_closed_loop_sim_pipelined (the MPC beside the ticks, its solution used one period after its observation) on the host emulator against the oracle's loop
    with the same latency: 3 periods of 4 ticks"""
    import os, sys, emu_harness
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from sim_closed_loop_demo import setup
    mb, st = blobs
    horizon = 0.45; c = setup("trot", 1, horizon, t_start=20.2); c["horizon"] = horizon; c["B"] = 1
    q0 = c["xbar"][6:30].copy(); q0[2] = 0.385
    e = emu_harness.Emu(mb, robust_grid_settings(st), 1, 64, 2, c["ev"].shape[1]); e.grid_only(c, batch=1)
    e.lib.emu_wbc_reset(e.h); e.sim_params(); e.sim_reset(q0[None], np.zeros((1, 24)), 20.2); e.sim_command(0, 0, 0, 0, 0)
    dev = []
    for p in range(3):
        e.closed_loop_sim(4, 0.001, horizon, nsub=2, mpc_every=4, restart=(p == 0), pipelined=True); dev.append(e.sim_state())
    log = oracle_closed_loop(oracle, mb, c, q0, 12, 0.001, 2, 4, horizon, 0.0, 0.5, 20.2, pipelined=True)
    for p in range(3):
        k = 4 * p + 3
        assert dev[p]["mpc_status"][0] == 0 and list(dev[p]["wbc_status"][0]) == [0, 0, 0], p
        assert rel_err(dev[p]["tau"][0], log[k]["tau"]) < 1e-6 and rel_err(dev[p]["q"][0], log[k]["q"]) < 1e-9 and rel_err(dev[p]["v"][0], log[k]["v"]) < 1e-7, p
