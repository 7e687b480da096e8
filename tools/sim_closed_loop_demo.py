"""tools/sim_closed_loop_demo.py — the whole controller around the batched plant, device resident (qmhip_closed_loop_sim): stance or trot, prints the base pose,
contact forces and joint-speed bound every 50 ticks.  Usage: python tools/sim_closed_loop_demo.py [stance|trot] [ticks] [batch]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from qm_control_amd import api, scenarios


def setup(gait, B, horizon, t_start=20.0, ee_pose=None):
    mb, st = scenarios.load_blobs()
    xbar = st[scenarios.ST_XINIT:scenarios.ST_XINIT + 30].copy(); qnom = mb[scenarios.MB_QNOM:scenarios.MB_QNOM + 18].copy()
    if gait == "stance":
        e, m = scenarios.stance_schedule(t_start, 100.0)
    else:
        g = scenarios.load_gaits()["trot"]                      # stance until t_start + 0.5, then the trot template tiled from there (what a gait command does)
        e, m = scenarios.tile_gait(g["switchingTimes"], g["modeSequence"], t_start + 0.5, t_start + 30.0)
    ev, modes = scenarios._pad_schedules([e] * B, [m] * B)
    base = xbar[6:12].copy(); base[2] = scenarios.COM_HEIGHT
    goal = base.copy(); goal[0] += 0.0 if gait == "stance" else 0.3
    ee = np.concatenate([scenarios.EE_NOMINAL_POS, scenarios.EE_NOMINAL_QUAT]) if ee_pose is None else np.asarray(ee_pose, float)   # the publisher's lastEeTarget_ starts at the measured EE pose
    rt, rx = scenarios.make_target(t_start, 3.0, base, goal, qnom, ee, ee)
    return dict(mb=mb, st=st, xbar=xbar, t0=np.full(B, t_start), x0=np.tile(xbar, (B, 1)), ref_t=np.tile(rt, (B, 1)), ref_x=np.tile(rx, (B, 1, 1)), ev=ev, modes=modes)


if __name__ == "__main__":
    gait = sys.argv[1] if len(sys.argv) > 1 else "stance"; ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 300; B = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    horizon = 1.0
    c = setup(gait, B, horizon)
    z0 = float(os.environ.get("Z0", "0.385"))
    if os.environ.get("EE_NOMINAL", "0") != "1":               # EE target = EE pose of the start posture (FK through the plant's hand-over of the reset state)
        itf0 = api.QMInterface(blobs=(c["mb"], c["st"]), max_batch=1, max_nodes=8, max_ref_knots=2, max_events=2); s0 = api.QMHWSim(itf0)
        q0 = c["xbar"][6:30].copy(); q0[2] = z0; s0.reset(q0[None], np.zeros((1, 24)), 20.0); rbd0, _ = s0.step(1e-9, 1); itf0.close()
        c = setup(gait, B, horizon, ee_pose=rbd0[0, 48:55])
    itf = api.QMInterface(blobs=(c["mb"], c["st"]), max_batch=B, max_nodes=128, max_ref_knots=2, max_events=c["ev"].shape[1])
    mpc = api.SqpMpc(itf); wbc = api.HierarchicalWbc(itf); sim = api.QMHWSim(itf, robust_grid=True)
    extra = {k.lower()[4:]: float(v) for k, v in os.environ.items() if k.startswith("SIM_")}     # e.g. SIM_DELAY=0 SIM_CONTACT_DAMPING=400
    if extra: sim.set_params(**extra)
    mpc.set_problem(c["t0"], c["x0"], c["ref_t"], c["ref_x"], c["ev"], c["modes"]); wbc.reset()
    q = np.tile(c["xbar"][6:30], (B, 1)); q[:, 2] = z0
    sim.reset(q, np.zeros((B, 24)), 20.0)
    t = time.time(); ee0 = None; dev_p = 0.0; dev_a = 0.0
    for k in range(0, ticks, 50):
        sim.closed_loop(50, 0.001, horizon, n_substeps=int(os.environ.get("NSUB", "2")), mpc_every=int(os.environ.get("MPC_EVERY", "10")))
        s = sim.state(); res = mpc.download(); out, st3 = wbc.download(B)
        rbd, _ = sim.step(1e-12, 1, download=True) if False else (itf.debug_read("sim_rbd", (B, 55)), None)
        if ee0 is None: ee0 = c["ref_x"][0, 0, 30:37].copy()
        dp = np.linalg.norm(rbd[:, 48:51] - ee0[:3], axis=1); dq = np.abs((rbd[:, 51:55] * ee0[3:]).sum(1)); da = np.degrees(2.0 * np.arccos(np.clip(dq, 0.0, 1.0)))
        dev_p = max(dev_p, dp.max()); dev_a = max(dev_a, da.max())
        print("tick %4d  z %.4f  x %.4f  zyx %s  max|qd| %.2f  fz %s  mpc status %s wbc %s" % (k + 50, s["q"][0, 2], s["q"][0, 0], s["q"][0, 3:6].round(3), np.abs(s["v"][0, 6:]).max(), s["force"][0, 2::3].round(1), res["status"][:2], st3[0]))
    itf.synchronize(); print("wall %.2f s for %d ticks x %d instances" % (time.time() - t, ticks, B))
    print("base travel %.3f m; end-effector deviation from its commanded (initial) pose: max %.1f mm, %.2f deg (README: 3.5 mm / 2.6 deg over 30 cm in Gazebo)" % (s["q"][:, 0].mean(), 1e3 * dev_p, dev_a))
