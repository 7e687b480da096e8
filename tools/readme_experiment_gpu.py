"""tools/readme_experiment_gpu.py — the README's end-effector stability experiment on the PRODUCT's device-resident loop (qmhip_closed_loop_sim + the device target publisher's
cmdVelToTargetTrajectories), the drive the figure shows: the base backs away in -x under a cmd_vel stream for 10 s while the end-effector target stays where it was
(/root/reference/README.md:109-116, docs/position_err.png; ablation on the CPU oracle's loop: tools/readme_experiment.py -> profiles/r05_readme_experiment.json).
usage: python tools/readme_experiment_gpu.py [vx] [arm_kd] [batch] [seconds]"""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from qm_control_amd import api, scenarios


def run(blobs, vx=-0.1, arm_kd=0.0, B=4, walk_s=10.0, pub_every=20, seed=None):
    mb, st = blobs; horizon = 1.0; t_start = 20.0
    xbar = st[scenarios.ST_XINIT:scenarios.ST_XINIT + 30].copy(); qnom = mb[scenarios.MB_QNOM:scenarios.MB_QNOM + 18].copy()
    g = scenarios.load_gaits()["trot"]; e, m = scenarios.tile_gait(g["switchingTimes"], g["modeSequence"], t_start + 0.5, t_start + 40.0)
    ev, modes = scenarios._pad_schedules([e] * B, [m] * B)
    itf = api.QMInterface(blobs=blobs, max_batch=B, max_nodes=128, max_ref_knots=2, max_events=ev.shape[1])
    mpc = api.SqpMpc(itf); wbc = api.HierarchicalWbc(itf); sim = api.QMHWSim(itf, robust_grid=True)
    q = np.tile(xbar[6:30], (B, 1)); q[:, 2] = 0.385
    if seed is not None: q[:, 6:18] += 0.02 * np.random.default_rng(seed).normal(size=(B, 12))
    sim.reset(q, np.zeros((B, 24)), t_start); rbd0, _ = sim.step(1e-9, 1); ee0 = rbd0[:, 48:55].copy()                     # the end-effector pose of the start posture
    base = xbar[6:12].copy(); base[2] = scenarios.COM_HEIGHT
    ref_t = np.zeros((B, 2)); ref_x = np.zeros((B, 2, 37))
    for b in range(B): ref_t[b], ref_x[b] = scenarios.make_target(t_start, horizon, base, base, qnom, ee0[b], ee0[b])
    mpc.set_problem(np.full(B, t_start), np.tile(xbar, (B, 1)), ref_t, ref_x, ev, modes); wbc.reset(); sim.reset(q, np.zeros((B, 24)), t_start)
    pub = api.TargetTrajectoriesPublisher(itf, B, time_to_target=horizon, last_ee_target=ee0[0])
    walk0 = 0.85; n_chunks = int(round((walk0 + walk_s + 0.15) * 1000 / pub_every)); dev_p = np.zeros(B); dev_a = np.zeros(B); ok = True; t = time.time()
    for k in range(n_chunks):
        tr = k * pub_every * 1e-3; cmd = np.zeros((B, 7)); cmd[:, 0] = vx if walk0 <= tr < walk0 + walk_s else 0.0
        pub.publish(np.full(B, api.CMD_VEL, np.int32), cmd)
        sim.closed_loop(pub_every, 0.001, horizon, n_substeps=2, mpc_every=10, arm_kp=0.0, arm_kd=arm_kd)
        rbd = itf.debug_read("sim_rbd", (B, 55))
        dev_p = np.maximum(dev_p, np.linalg.norm(rbd[:, 48:51] - ee0[:, :3], axis=1))
        dev_a = np.maximum(dev_a, np.degrees(2.0 * np.arccos(np.clip(np.abs((rbd[:, 51:55] * ee0[:, 3:]).sum(1)), 0.0, 1.0))))
        if k % 50 == 49:
            res = mpc.download(); _, st3 = wbc.download(B); ok = ok and bool((res["status"] >= 0).all() and (st3 == 0).all())
    s = sim.state(); itf.close()
    return dict(vx=vx, arm_kd=arm_kd, batch=B, seconds=n_chunks * pub_every * 1e-3, base_travel_m=[float(v) for v in s["q"][:, 0]], ee_dev_max_mm=[float(1e3 * v) for v in dev_p], ee_dev_max_deg=[float(v) for v in dev_a],
                base_z=[float(v) for v in s["q"][:, 2]], all_status_ok=ok, wall_s=time.time() - t)


if __name__ == "__main__":
    a = sys.argv[1:]; vx = float(a[0]) if a else -0.1; kd = float(a[1]) if len(a) > 1 else 0.0; B = int(a[2]) if len(a) > 2 else 4; secs = float(a[3]) if len(a) > 3 else 10.0
    print(json.dumps(run(scenarios.load_blobs(), vx, kd, B, secs)))
