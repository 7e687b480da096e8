"""tools/wbc_algo_ab.py — primal vs dual active set of the WBC hard rows: same torques?, kernel time, status."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from qm_control_amd import api, scenarios
res = {}
for name, B, N in (("C4", 1024, 100), ("C5", 512, 150)):
    cfg = scenarios.make_config(name, batch=B, n_intervals=N)
    itf = api.QMInterface(blobs=scenarios.load_blobs(), max_batch=B, max_nodes=N + 60, max_ref_knots=2, max_events=cfg["ev"].shape[1])
    mpc = api.SqpMpc(itf); wbc = api.HierarchicalWbc(itf)
    mpc.set_problem(cfg["t0"], cfg["x0"], cfg["ref_t"], cfg["ref_x"], cfg["ev"], cfg["modes"])
    outs = []
    for algo in (0, 1):
        itf.debug_set("wbc_algo", algo)
        wbc.reset(); mpc.control_step_resident(cfg["horizon"], cfg["period"], cfg["time"]); itf.synchronize()
        itf.set_profiling(True); itf.reset_kernel_ms()
        for _ in range(5): wbc.reset(); mpc.control_step_resident(cfg["horizon"], cfg["period"], cfg["time"])
        itf.synchronize(); ms, n = itf.kernel_ms("wbc"); itf.set_profiling(False)
        out, qps = wbc.download(B); outs.append(out)
        itf.debug_set("wbc_stop", -4); wbc.reset(); mpc.control_step_resident(cfg["horizon"], cfg["period"], cfg["time"]); itf.synchronize(); itf.debug_set("wbc_stop", 0)
        its = itf.debug_read("wbc_scratch", (B, 432))[:, 13:15]
        print("   iterations level1: mean %.1f max %d   level2: mean %.1f max %d   total max %d" % (its[:, 0].mean(), its[:, 0].max(), its[:, 1].mean(), its[:, 1].max(), its.sum(1).max()))
        print(name, "algo", algo, "wbc ms %.3f" % (ms / n), "status counts", np.bincount(qps.ravel(), minlength=3).tolist(), "finite", bool(np.isfinite(out).all()))
    d = np.abs(outs[0] - outs[1]).max(axis=1) / np.abs(outs[0]).max(axis=1)
    print(name, "max rel diff dual vs primal %.3e" % d.max(), "instances > 1e-9:", int((d > 1e-9).sum()))
    itf.close()
