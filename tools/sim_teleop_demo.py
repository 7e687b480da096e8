"""tools/sim_teleop_demo.py — the reference's operator interface on the batched plant: the robots stand, receive a gait command (GaitJoyPublisher -> GaitReceiver,
device-resident GaitSchedule), walk under a cmd_vel stream (QmTargetTrajectoriesPublisher's cmdVelToTargetTrajectories on the resident observation, refreshed every
0.1 s), and are sent back to stance — all through qmhip_gait_* / qmhip_target_* / qmhip_closed_loop_sim.  Usage: python tools/sim_teleop_demo.py [gait] [batch] [vx]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from qm_control_amd import api, scenarios


def run(blobs, gait="trot", B=8, vx=0.3, walk_s=3.0, verbose=True, seed=2):
    mb, st = blobs; gaits = scenarios.load_gaits(); horizon = 1.0; t_start = 20.0; rng = np.random.default_rng(seed)
    xbar = st[scenarios.ST_XINIT:scenarios.ST_XINIT + 30].copy(); qnom = mb[scenarios.MB_QNOM:scenarios.MB_QNOM + 18].copy()
    itf = api.QMInterface(blobs=blobs, max_batch=B, max_nodes=160, max_ref_knots=2, max_events=48)
    mpc = api.SqpMpc(itf); wbc = api.HierarchicalWbc(itf); sim = api.QMHWSim(itf, robust_grid=True)
    q = np.tile(xbar[6:30], (B, 1)); q[:, 2] = 0.385; q[:, 6:18] += 0.02 * rng.normal(size=(B, 12))
    sim.reset(q, np.zeros((B, 24)), t_start); rbd0, _ = sim.step(1e-9, 1)
    base = xbar[6:12].copy(); base[2] = scenarios.COM_HEIGHT
    ref_t = np.zeros((B, 2)); ref_x = np.zeros((B, 2, 37))
    for b in range(B):
        ref_t[b], ref_x[b] = scenarios.make_target(t_start, horizon, base, base, qnom, rbd0[b, 48:55], rbd0[b, 48:55])
    e, m = scenarios.stance_schedule(t_start, 100.0); ev, modes = scenarios._pad_schedules([e] * B, [m] * B)
    evp = np.full((B, 48), ev[0, -1] if ev.shape[1] else 0.0); evp[:, :ev.shape[1]] = ev; mop = np.full((B, 49), 15, np.int32)
    mpc.set_problem(np.full(B, t_start), np.tile(xbar, (B, 1)), ref_t, ref_x, np.sort(evp, axis=1), mop); wbc.reset(); sim.reset(q, np.zeros((B, 24)), t_start)
    gs = api.GaitSchedule(itf, gaits, B); pub = api.TargetTrajectoriesPublisher(itf, B, time_to_target=horizon, last_ee_target=rbd0[0, 48:55])
    log = []
    def chunk(n, label):
        sim.closed_loop(n, 0.001, horizon, n_substeps=2, mpc_every=10); s = sim.state(); res = mpc.download(); _, st3 = wbc.download(B)
        ok = bool((res["status"] == 0).all() and (st3 == 0).all() and (gs.download()["status"] == 0).all())
        log.append(dict(label=label, t=float(s["time"][0]), x=s["q"][:, 0].copy(), z=s["q"][:, 2].copy(), tilt=np.abs(s["q"][:, 3:5]).max(1), ok=ok, mode=int(res["mode"][0, 0])))
        if verbose: print("%-8s t %.2f  x %.3f  z %.3f  tilt %.3f  mode %2d  ok %s" % (label, log[-1]["t"], s["q"][:, 0].mean(), s["q"][:, 2].mean(), log[-1]["tilt"].max(), log[-1]["mode"], ok))
    chunk(500, "stand")
    gs.preSolverRun(gait, sim.state()["time"], horizon)                      # the gait command arrives: the template starts at the end of the current horizon
    cmd = np.zeros((B, 7)); cmd[:, 0] = vx
    for k in range(int(round(walk_s / 0.1)) + 10):                            # 1 s for the gait to come in, then walk_s of cmd_vel
        pub.publish(np.full(B, api.CMD_VEL, np.int32), cmd); chunk(100, "walk")
    gs.preSolverRun("stance", sim.state()["time"], horizon); pub.publish(np.full(B, api.CMD_VEL, np.int32), np.zeros((B, 7)))
    for k in range(20):
        chunk(100, "stop")
    itf.close()
    return log


if __name__ == "__main__":
    gait = sys.argv[1] if len(sys.argv) > 1 else "trot"; B = int(sys.argv[2]) if len(sys.argv) > 2 else 8; vx = float(sys.argv[3]) if len(sys.argv) > 3 else 0.3
    log = run(scenarios.load_blobs(), gait, B, vx)
    print("travelled %.3f m (commanded %.2f m/s for 3 s); all statuses ok: %s; final mode %d" % (log[-1]["x"].mean(), vx, all(l["ok"] for l in log), log[-1]["mode"]))
