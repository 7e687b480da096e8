"""tools/wbc_prof.py — in-kernel cycle counters of the WBC kernel (profiling only)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from qm_control_amd import api, scenarios
B = 1024
cfg = scenarios.make_config("C4", batch=B)
itf = api.QMInterface(blobs=scenarios.load_blobs(), max_batch=B, max_nodes=128, max_ref_knots=2, max_events=cfg["ev"].shape[1])
mpc = api.SqpMpc(itf); mpc.set_problem(cfg["t0"], cfg["x0"], cfg["ref_t"], cfg["ref_x"], cfg["ev"], cfg["modes"])
mpc.control_step_resident(cfg["horizon"], cfg["period"], cfg["time"]); itf.synchronize()
for stop, names in ((-1, ["init+rigid_body", "task build + AZ/g0", "L0 G build + active rows", "L0 QR solve", "L0 Z_times/d0_apply/c0c1", "L0 line search", "L0 iteration tail", "null space", "L>=1 factor + DZ", "L>=1 solves / appends / drops", "L>=1 iteration rest / level tail", "output"]),
                    (-2, ["tq_solve", "tq_mult", "tq_drop", "tq_append", "  append 1-2 t, coefficients", "  append 3 column sweep", "  append 4 row rotations", "  append 5 row + y", "(other)"]),
                    (-3, ["L0 load", "L0 pivoted QR", "L0 backward accumulation", "L0 write Zp", "(start -> rigid-body passes)", "L1 load", "L1 pivoted QR", "L1 backward accumulation", "(start -> end of rigid-body passes)", "L1 Zp <- Zp Q2"])):
    itf.debug_set("wbc_stop", stop)
    mpc.control_step_resident(cfg["horizon"], cfg["period"], cfg["time"]); itf.synchronize()
    cyc = itf.debug_read("wbc_scratch", (B, 432))
    print(json.dumps({n: round(float(cyc[:, i].mean())) for i, n in enumerate(names)}, indent=1))
    if stop == -1: print("total cycles/instance", cyc[:, :12].sum(1).mean())
itf.debug_set("wbc_stop", 0)
dbg = None
# active-set iterations per level (the kernel's run time is that of the slowest instance: one wave per SIMD)
itf.debug_set("wbc_stop", -4)
mpc.control_step_resident(cfg["horizon"], cfg["period"], cfg["time"]); itf.synchronize()
its = itf.debug_read("wbc_scratch", (B, 432))[:, 13:15]
print("active-set iterations level 1: mean %.1f max %d;  level 2: mean %.1f max %d;  per instance max %d" % (its[:, 0].mean(), its[:, 0].max(), its[:, 1].mean(), its[:, 1].max(), its.sum(1).max()))
itf.debug_set("wbc_stop", 0)
