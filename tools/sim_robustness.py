"""tools/sim_robustness.py — 1024 perturbed instances of the whole controller around the plant (stance -> trot, base commanded 0.3 m ahead): status counts, how many
stay upright, base travel and end-effector deviation statistics.  Usage: python tools/sim_robustness.py [ticks] [batch]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
from qm_control_amd import api
from sim_closed_loop_demo import setup

ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 3000; B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
horizon = 1.0; rng = np.random.default_rng(7)
c = setup("trot", B, horizon)
q = np.tile(c["xbar"][6:30], (B, 1)); q[:, 2] = 0.385; q[:, 6:18] += 0.03 * rng.normal(size=(B, 12)); q[:, 18:] += 0.1 * rng.normal(size=(B, 6)); q[:, 5] += 0.1 * rng.normal(size=B)
itf = api.QMInterface(blobs=(c["mb"], c["st"]), max_batch=B, max_nodes=128, max_ref_knots=2, max_events=c["ev"].shape[1])
mpc = api.SqpMpc(itf); wbc = api.HierarchicalWbc(itf); sim = api.QMHWSim(itf, robust_grid=True)
sim.reset(q, np.zeros((B, 24)), 20.0); rbd0, _ = sim.step(1e-9, 1)          # EE target of every instance = its own start pose
for b in range(B):
    c["ref_x"][b, :, 30:37] = rbd0[b, 48:55]; c["ref_x"][b, :, 11] = q[b, 5]; c["ref_x"][b, :, 9] = 0.0
mpc.set_problem(c["t0"], c["x0"], c["ref_t"], c["ref_x"], c["ev"], c["modes"]); wbc.reset(); sim.reset(q, np.zeros((B, 24)), 20.0)
t = time.time(); bad_mpc = np.zeros(B, bool); bad_wbc = np.zeros(B, bool); dev = np.zeros(B)
for k in range(0, ticks, 100):
    sim.closed_loop(100, 0.001, horizon, n_substeps=2, mpc_every=10, pipelined=bool(int(os.environ.get("PIPELINED", "0"))))
    res = mpc.download(); _, st3 = wbc.download(B); rbd = itf.debug_read("sim_rbd", (B, 55))
    bad_mpc |= res["status"] != 0; bad_wbc |= (st3 != 0).any(1); dev = np.maximum(dev, np.linalg.norm(rbd[:, 48:51] - rbd0[:, 48:51], axis=1))
s = sim.state(); up = np.isfinite(s["q"]).all(1) & (np.abs(s["q"][:, 3:5]).max(1) < 0.3) & (s["q"][:, 2] > 0.3)
print("%d instances x %d ticks in %.2f s; upright %d; MPC status != 0 seen on %d, WBC status != 0 on %d (sampled every 100 ticks)" % (B, ticks, time.time() - t, up.sum(), bad_mpc.sum(), bad_wbc.sum()))
print("base travel: mean %.3f m (min %.3f max %.3f); EE deviation from the commanded pose: median %.1f mm, 95 %% %.1f mm, max %.1f mm" % (s["q"][up, 0].mean(), s["q"][up, 0].min(), s["q"][up, 0].max(), 1e3 * np.median(dev[up]), 1e3 * np.percentile(dev[up], 95), 1e3 * dev[up].max()))
