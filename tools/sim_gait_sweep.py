"""tools/sim_gait_sweep.py — the whole controller around the plant through every gait template of gait.info: stance for 0.5 s, then the template tiled, base commanded
0.3 m ahead; reports per gait whether the instances stay upright and the worst MPC / WBC status.  Usage: python tools/sim_gait_sweep.py [ticks] [batch]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
from qm_control_amd import api, scenarios
from sim_closed_loop_demo import setup

ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 2000; B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
gaits = scenarios.load_gaits(); horizon = 1.0; rng = np.random.default_rng(11)
for name, g in gaits.items():
    c = setup("stance", B, horizon)
    e, m = scenarios.tile_gait(g["switchingTimes"], g["modeSequence"], 20.5, 20.0 + 1e-3 * ticks + 3.0)
    c["ev"], c["modes"] = scenarios._pad_schedules([e] * B, [m] * B)
    c["ref_x"][:, 1, 6] += 0.3
    q = np.tile(c["xbar"][6:30], (B, 1)); q[:, 2] = 0.385; q[:, 6:18] += 0.02 * rng.normal(size=(B, 12))
    itf = api.QMInterface(blobs=(c["mb"], c["st"]), max_batch=B, max_nodes=160, max_ref_knots=2, max_events=c["ev"].shape[1])
    mpc = api.SqpMpc(itf); wbc = api.HierarchicalWbc(itf); sim = api.QMHWSim(itf, robust_grid=True)
    sim.reset(q, np.zeros((B, 24)), 20.0); rbd0, _ = sim.step(1e-9, 1)
    for b in range(B): c["ref_x"][b, :, 30:37] = rbd0[b, 48:55]
    mpc.set_problem(c["t0"], c["x0"], c["ref_t"], c["ref_x"], c["ev"], c["modes"]); wbc.reset(); sim.reset(q, np.zeros((B, 24)), 20.0)
    bad_mpc = np.zeros(B, bool); bad_wbc = np.zeros(B, bool)
    for k in range(0, ticks, 100):
        sim.closed_loop(100, 0.001, horizon, n_substeps=2, mpc_every=10); res = mpc.download(); _, st3 = wbc.download(B)
        bad_mpc |= res["status"] != 0; bad_wbc |= (st3 != 0).any(1)
    s = sim.state(); up = np.isfinite(s["q"]).all(1) & (np.abs(s["q"][:, 3:5]).max(1) < 0.4) & (s["q"][:, 2] > 0.25)
    print("%-22s upright %2d/%d  travel %.3f m  z %.3f  MPC status != 0 on %d, WBC on %d" % (name, up.sum(), B, np.nanmean(s["q"][up, 0]) if up.any() else float("nan"), np.nanmean(s["q"][up, 2]) if up.any() else float("nan"), bad_mpc.sum(), bad_wbc.sum()), flush=True)
    itf.close()
