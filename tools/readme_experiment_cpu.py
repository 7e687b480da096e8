"""tools/readme_experiment_cpu.py — the README's end-effector stability experiment (/root/reference/README.md:109-116: the EE is commanded to hold its pose while the base
travels 30 cm; EE deviation at most 3.5 mm / 2.6 deg in Gazebo) on the CPU ORACLE's closed loop (tests/test_sim.py: QMController::update around oracle/src/sim.h), with a log
that separates the MPC's PLAN (end-effector pose of the planned state) from what the plant does.  Test infrastructure / investigation aid: the product's loop is
qmhip_closed_loop_sim (tools/sim_closed_loop_demo.py, tools/readme_experiment.py).
usage: python tools/readme_experiment_cpu.py [key=value ...]   keys: ticks nsub mpc_every horizon arm_kp arm_kd pipelined stiffness damping delay plant(=sim|perfect) dist T drive(=pose|cmdvel) vx walk_s pub_every"""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pyoracle
import frontend
from qm_control_amd import scenarios, layout as L
from test_sim import centroidal_from_rbd


def quat_angle_deg(q, qref):
    return float(np.degrees(2.0 * np.arccos(np.clip(abs(float(np.dot(q, qref))), 0.0, 1.0))))


def run(ticks=6000, nsub=2, mpc_every=10, horizon=1.0, arm_kp=0.0, arm_kd=0.5, pipelined=0, stiffness=4.0e4, damping=200.0, delay=0.009, plant="sim", dist=0.3, T=3.0, z0=0.385, threads=3, log_every=100, gait="trot", drive="pose", vx=0.3, walk_s=1.0, pub_every=20, ee_dx=0.0, ee_dy=0.0, ee_dz=0.0, settle=0.0, **kw):
    mb, st = pyoracle.load_blobs(); o = pyoracle.Oracle(mb, st); o.set_threads(int(threads)); o.set_setting(L.ST_GRID_DT_MIN, L.QM_GRID_DT_MIN_ROBUST)
    for k_, v_ in kw.items():
        if k_.startswith('st_'): o.set_setting(int(k_[3:]), float(v_)); st[int(k_[3:])] = float(v_)      # ablations: st_<slot>=value
    t_start = 20.0; period = 0.001
    xbar = st[L.ST_XINIT:L.ST_XINIT + 30].copy(); qnom = mb[L.MB_QNOM:L.MB_QNOM + 18].copy()
    if gait == "stance": e, m = scenarios.stance_schedule(t_start, 100.0)
    else:
        g = scenarios.load_gaits()[gait]; e, m = scenarios.tile_gait(g["switchingTimes"], g["modeSequence"], t_start + 0.5, t_start + 30.0)
    q0 = xbar[6:30].copy(); q0[2] = z0
    rbd0 = o.rbd_from_q(q0, np.zeros(24)); ee = rbd0[48:55].copy()                 # EE target = EE pose of the start posture (what the publisher's lastEeTarget_ starts from)
    base = xbar[6:12].copy(); base[2] = scenarios.COM_HEIGHT; goal = base.copy(); goal[0] += dist
    rt, rx = scenarios.make_target(t_start, T, base, goal, qnom, ee, ee)
    if drive == "cmdvel":      # the README experiment's drive: a cmd_vel stream into QmTargetTrajectoriesPublisher's cmdVelToTargetTrajectories (_node.cpp:71-116): the base target rides `vx * TIME_TO_TARGET`
        # ahead of the measured base, the commanded velocity is the momentum reference, the end-effector target is lastEeTarget_ (the pose at the start)
        pub = frontend.TargetPublisher(qnom, scenarios.COM_HEIGHT, 0.3, 0.1, horizon); pub.last_ee = ee.copy()
        rt, rx = scenarios.make_target(t_start, horizon, base, base, qnom, ee, ee)
    o.set_schedule(np.asarray(e, float), np.asarray(m, np.int32)); o.set_target(rt, rx)
    o.wbc_reset(); o.sim_params(contact_stiffness=stiffness, contact_damping=damping, delay=delay); o.sim_reset(q0, np.zeros(24), t_start); o.sim_command(0, 0, 0, 0, 0)
    s = dict(rbd=rbd0, time=t_start, k=0); pos = np.zeros(18); vel = np.zeros(18); kp = np.zeros(18); kd = np.zeros(18); ff = np.zeros(18)
    log = []; dev = dict(p=0.0, a=0.0, plan_p=0.0, plan_a=0.0, track_p=0.0, init_p=0.0, init_a=0.0); pend = None
    # round 6: the drive may start from ANOTHER end-effector pose than the start posture's — the target is ramped by (ee_dx, ee_dy, ee_dz) during `settle` seconds of trotting on
    # the spot (slowly: the publisher re-anchors a target further than 0.1 m from the measured pose, _node.cpp:95-96), and deviations are recorded from the start of the drive on,
    # against the target AND against the end-effector pose measured at that moment (the README's "deviation from its initial position")
    ee0 = ee.copy(); ee_off = np.array([ee_dx, ee_dy, ee_dz], float); walk0 = 0.5 + 0.35 + float(settle); ee_init = [None]

    def ee_of_state(x):
        return o.rbd_from_q(x[6:30], np.zeros(24))[48:55]

    def tick():
        time_, rbd = s["time"], s["rbd"]
        xd, ud, mode = o.eval_policy(time_)
        if s["k"] == 0: o.wbc_set_input_last(ud)
        out, wst = o.wbc(xd, ud, rbd, mode, period, time_)
        if plant == "sim":
            pos[:12] = xd[12:24]; vel[:12] = ud[12:24]; kp[:12] = 0.0; kd[:12] = 3.0; ff[:12] = out[36:48]
            pos[12:] = xd[24:30]; vel[12:] = 0.0; kp[12:] = arm_kp; kd[12:] = arm_kd; ff[12:] = out[48:54]
            o.sim_command(pos, vel, kp, kd, ff); r = o.sim_step(period, int(nsub)); s["rbd"] = r["rbd"]; s["time"] = r["time"]; force = r["force"]
        else:                                                                         # perfect-tracking plant: the next observation is the policy's state one period on
            xn, un, _ = o.eval_policy(time_ + period); s["rbd"] = o.rbd_from_q(xn[6:30], np.zeros(24)); s["time"] = time_ + period; force = ud[:12]
            s["xn"] = xn
        s["k"] += 1
        eep = ee_of_state(xd); eem = s["rbd"][48:55]
        dp = float(np.linalg.norm(eem[:3] - ee[:3])); da = quat_angle_deg(eem[3:], ee[3:]); pp = float(np.linalg.norm(eep[:3] - ee[:3])); pa = quat_angle_deg(eep[3:], ee[3:])
        if s["time"] - t_start < walk0: dp = da = pp = pa = 0.0      # (settling: not part of the experiment)
        elif ee_init[0] is None: ee_init[0] = eem.copy()
        if ee_init[0] is not None: dev["init_p"] = max(dev["init_p"], float(np.linalg.norm(eem[:3] - ee_init[0][:3]))); dev["init_a"] = max(dev["init_a"], quat_angle_deg(eem[3:], ee_init[0][3:]))
        dev["p"] = max(dev["p"], dp); dev["a"] = max(dev["a"], da); dev["plan_p"] = max(dev["plan_p"], pp); dev["plan_a"] = max(dev["plan_a"], pa); dev["track_p"] = max(dev["track_p"], float(np.linalg.norm(eem[:3] - eep[:3])))
        if s["k"] % int(log_every) == 0:
            q = s["rbd"]; log.append(dict(t=round(s["time"] - t_start, 3), base_x=float(q[3]), base_z=float(q[5]), zyx=[float(v) for v in q[0:3]], plan_base_x=float(xd[6]), ref_base_x=(float(np.interp(s["time"], rt, rx[:, 6])) if drive == "pose" else float("nan")),
                                       ee_dev_mm=1e3 * dp, ee_dev_deg=da, plan_ee_dev_mm=1e3 * pp, plan_ee_dev_deg=pa, fz=[float(v) for v in force[2::3]], mode=int(mode), wbc=[int(v) for v in wst]))

    def observe():
        if plant == "sim": return centroidal_from_rbd(mb, s["rbd"])
        return s.get("xn", xbar.copy() if s["k"] else np.concatenate([np.zeros(6), q0]))

    t0 = time.time()
    for k in range(int(ticks)):
        if drive == "cmdvel" and k % int(pub_every) == 0:
            tr = s["time"] - t_start; cmd = np.zeros(6); cmd[0] = vx if (walk0 <= tr < walk0 + walk_s) else 0.0
            if settle > 0.0:
                f = min(1.0, max(0.0, (tr - 0.85) / max(1e-9, settle - 1.0))); ee[:3] = ee0[:3] + f * ee_off; pub.last_ee = ee.copy()
            a, b = pub.cmd_vel(cmd, s["time"], observe(), s["rbd"][48:55]); o.set_target(a, b)
        if k % int(mpc_every) == 0:
            if int(pipelined) and pend is not None:
                o.mpc_step(pend[0], pend[0] + horizon, pend[1], warm=True)           # the solve triggered one MPC period ago becomes available now
            if not int(pipelined) or pend is None:
                o.mpc_step(s["time"], s["time"] + horizon, observe(), warm=(k > 0))
            pend = (s["time"], observe())
        tick()
    q = s["rbd"]; print('final arm q', np.round(q[18+6:18+12] if False else s['rbd'][6+12+6:6+12+12], 3), file=sys.stderr)
    return dict(config=dict(ticks=int(ticks), nsub=int(nsub), mpc_every=int(mpc_every), horizon=horizon, arm_kp=arm_kp, arm_kd=arm_kd, pipelined=int(pipelined), stiffness=stiffness, damping=damping, delay=delay, plant=plant, dist=dist, T=T, gait=gait, drive=drive, vx=vx, walk_s=walk_s, pub_every=int(pub_every)),
                base_travel_m=float(q[3] - 0.0), ee_dev_max_mm=1e3 * dev["p"], ee_dev_max_deg=dev["a"], plan_ee_dev_max_mm=1e3 * dev["plan_p"], plan_ee_dev_max_deg=dev["plan_a"], ee_vs_plan_max_mm=1e3 * dev["track_p"], ee_dev_from_initial_mm=1e3 * dev["init_p"], ee_dev_from_initial_deg=dev["init_a"], arm_q_final=[float(v) for v in s["rbd"][18:24]],
                wall_s=time.time() - t0, log=log)


if __name__ == "__main__":
    kw = {}
    for a in sys.argv[1:]:
        k, v = a.split("=", 1)
        try: kw[k] = float(v) if ("." in v or "e" in v) else int(v)
        except ValueError: kw[k] = v
    out = kw.pop("out", None)
    r = run(**kw)
    print(json.dumps({k: v for k, v in r.items() if k != "log"}))
    for e in r["log"]: print(e)
    if out: json.dump(r, open(out, "w"), indent=1)
