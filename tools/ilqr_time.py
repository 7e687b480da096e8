import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from qm_control_amd import api, scenarios, layout as L
for B in (64, 1024):
    cfg = scenarios.make_config("C4", batch=B)
    itf = api.QMInterface(blobs=scenarios.load_blobs(), max_batch=B, max_nodes=128, max_ref_knots=2, max_events=cfg["ev"].shape[1])
    mpc = api.SqpMpc(itf); mpc.set_problem(cfg["t0"], cfg["x0"], cfg["ref_t"], cfg["ref_x"], cfg["ev"], cfg["modes"])
    for s in (0, 1):
        itf.set_setting(L.ST_SOLVER, float(s))
        mpc.solve_resident(cfg["horizon"]); itf.synchronize(); t = time.perf_counter()
        for _ in range(3): mpc.solve_resident(cfg["horizon"])
        itf.synchronize(); dt = (time.perf_counter() - t) / 3
        r = mpc.download()
        print("B %4d solver %d: %.2f ms per MPC iteration, trials %d, status ok %s" % (B, s, dt * 1e3, r["ls_trials"], bool((r["status"] == 0).all())))
    itf.close()
