import os, sys, json
sys.path.insert(0, "/root/repo")
import numpy as np
from qm_control_amd import api, scenarios
B = 1024
cfg = scenarios.make_config("C4", batch=B)
itf = api.QMInterface(blobs=scenarios.load_blobs(), max_batch=B, max_nodes=128, max_ref_knots=2, max_events=cfg["ev"].shape[1])
mpc = api.SqpMpc(itf); mpc.set_problem(cfg["t0"], cfg["x0"], cfg["ref_t"], cfg["ref_x"], cfg["ev"], cfg["modes"])
mpc.solve_resident(cfg["horizon"]); itf.synchronize()
itf.debug_set("lq_prof", 1)
mpc.solve_resident(cfg["horizon"]); itf.synchronize()
nm = 128; SR = 7360
stage = itf.debug_read("stage", (B * nm, SR))
rows = stage[np.arange(B)[:, None] * nm + np.arange(5, 95)[None, :]].reshape(-1, SR)[:, 4752:4761]
ln = ["P0 stage inputs", "I jacobian columns", "I RK2 composition", "II constraint rows", "II projector", "II projected dynamics", "III cost model (input / state terms, barriers)", "III EE term, [Q | q], R assembly", "III projected cost + stores"]
print(json.dumps({n: float(rows[:, i].mean()) for i, n in enumerate(ln)}, indent=1)); print("LQ cycles/node between the first and the last stamp", rows[:, :9].sum(1).mean())
