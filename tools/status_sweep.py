"""tools/status_sweep.py — robustness sweep: many random instances of every scenario family, all solver / QP statuses must be 0 and a random sample
must agree with the oracle (one-off validation tool; the regular tests cover fixed seeds)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pyoracle
from blocks import block_errs
from qm_control_amd import api, scenarios
blobs = scenarios.load_blobs(); worst = 0.0
cases = [("C4", 4096, 100, s) for s in (1235, 77, 78)] + [("C5", 2048, 150, 1236), ("C5", 1024, 150, 5)] + [("gait:" + g, 256, 60, 3) for g in scenarios.load_gaits()]
for name, B, N, seed in cases:
    cfg = scenarios.gait_config(name[5:], batch=B, n_intervals=N, seed=seed) if name.startswith("gait:") else scenarios.make_config(name, batch=B, n_intervals=N, seed=seed)
    itf = api.QMInterface(blobs=blobs, max_batch=B, max_nodes=N + 80, max_ref_knots=2, max_events=cfg["ev"].shape[1])
    mpc = api.SqpMpc(itf); wbc = api.HierarchicalWbc(itf)
    mpc.set_problem(cfg["t0"], cfg["x0"], cfg["ref_t"], cfg["ref_x"], cfg["ev"], cfg["modes"])
    wbc.reset(); mpc.control_step_resident(cfg["horizon"], cfg["period"], cfg["time"])
    res = mpc.download(); out, qps = wbc.download(B)
    bad_m = np.nonzero(res["status"] < 0)[0]; warn_m = np.nonzero(res["status"] > 0)[0]; bad_w = np.nonzero((qps != 0).any(1))[0]      # status > 0: warning bits on a valid solution
    idx = np.random.default_rng(seed).choice(B, 6, replace=False)
    bad, xf, uf, w = pyoracle.batch_step(*pyoracle.load_blobs(), 6, cfg["t0"][idx], cfg["horizon"], cfg["x0"][idx], cfg["ref_t"][idx], cfg["ref_x"][idx], cfg["ev"][idx], cfg["modes"][idx], cfg["period"], cfg["time"])
    err = max(block_errs(out[idx], w, "wbc").values()); worst = max(worst, err)      # worst BLOCK (v̇ / forces / torques against their own scales)
    print("%-20s B %5d N %3d seed %5d  mpc failed %d  mpc warnings %d  wbc bad %d %s  finite %s  sample worst-block rel err %.1e (oracle bad %d)" % (name, B, N, seed, len(bad_m), len(warn_m), len(bad_w), qps[bad_w[:3]].tolist(), bool(np.isfinite(out).all()), err, bad))
    itf.close()
print("worst sample block rel err %.2e" % worst)
