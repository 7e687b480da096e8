"""tools/gpu_phase_probe.py — time K3 with phases skipped (profiling only)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from qm_control_amd import api, scenarios
B = int(os.environ.get("QM_PROBE_B", "1024"))
cfg = scenarios.make_config("C4", batch=B)
itf = api.QMInterface(blobs=scenarios.load_blobs(), max_batch=B, max_nodes=128, max_ref_knots=2, max_events=cfg["ev"].shape[1])
mpc = api.SqpMpc(itf); mpc.set_problem(cfg["t0"], cfg["x0"], cfg["ref_t"], cfg["ref_x"], cfg["ev"], cfg["modes"])
out = {}
for name, mask in (("full", 0), ("no_chol_solve", 1), ("no_products", 2), ("no_forward", 4), ("no_symmetrise", 8), ("no_chol,no_products", 3), ("backward loads only", 11), ("only forward", 11 | 16), ("nothing (launch + terminal)", 15 | 16)):
    itf.debug_set("riccati_skip", mask)
    mpc.solve_resident(cfg["horizon"]); itf.synchronize()
    itf.set_profiling(True); itf.reset_kernel_ms()
    for _ in range(3): mpc.solve_resident(cfg["horizon"])
    ms, n = itf.kernel_ms("riccati"); out[name] = ms / n; itf.set_profiling(False)
print(json.dumps(out, indent=1))
# K4 (WBC): cumulative time up to each phase boundary
wout = {}
for name, stop in (("full", 0), ("rigid_body", 1), ("+level0", 2), ("+level1", 3), ("+level2 (no output stage)", 4)):
    itf.debug_set("riccati_skip", 0); itf.debug_set("wbc_stop", stop)
    mpc.control_step_resident(cfg["horizon"], cfg["period"], cfg["time"]); itf.synchronize()
    itf.set_profiling(True); itf.reset_kernel_ms()
    for _ in range(3): mpc.control_step_resident(cfg["horizon"], cfg["period"], cfg["time"])
    ms, n = itf.kernel_ms("wbc"); wout[name] = ms / n; itf.set_profiling(False)
print(json.dumps(wout, indent=1))
# K4 in-kernel cycle counters (lane 0 of each instance), mean over instances
itf.debug_set("wbc_stop", -1)
mpc.control_step_resident(cfg["horizon"], cfg["period"], cfg["time"]); itf.synchronize()
import numpy as np
cyc = itf.debug_read("wbc_scratch", (B, 432))
names = ["init+rigid_body", "task build + AZ/g0", "L0 G build + active rows", "L0 QR solve", "L0 Z_times/d0_apply/c0c1", "L0 line search", "L0 iteration tail", "null space", "L>=1 factor + DZ", "L>=1 solves / appends / drops", "L>=1 iteration rest / level tail", "output"]
print(json.dumps({n: float(cyc[:, i].mean()) for i, n in enumerate(names)}, indent=1)); print("total cycles/instance", cyc[:, :12].sum(1).mean())
itf.debug_set("wbc_stop", 0)
itf.debug_set("wbc_stop", -2)
mpc.control_step_resident(cfg["horizon"], cfg["period"], cfg["time"]); itf.synchronize()
cyc = itf.debug_read("wbc_scratch", (B, 432))
fn = ["tq_solve", "tq_mult", "tq_drop", "tq_append", "  append 1-2 t, coefficients", "  append 3 column sweep", "  append 4 row rotations", "  append 5 row + y", "(other)"]
print(json.dumps({n: float(cyc[:, i].mean()) for i, n in enumerate(fn)}, indent=1))
itf.debug_set("wbc_stop", 0)
# K1b in-kernel stamps (thread 0 of each node's workgroup)
itf.debug_set("lq_prof", 1)
mpc.solve_resident(cfg["horizon"]); itf.synchronize()
nm = 128; SR = 7360
stage = itf.debug_read("stage", (B * nm, SR))
rows = stage[np.arange(B)[:, None] * nm + np.arange(5, 95)[None, :]].reshape(-1, SR)[:, 4752:4761]
ln = ["P0 stage inputs", "I jacobian columns", "I RK2 composition", "II constraint rows", "II projector", "II projected dynamics", "III cost model (input / state terms, barriers)", "III EE term, [Q | q], R assembly", "III projected cost + stores"]
print(json.dumps({n: float(rows[:, i].mean()) for i, n in enumerate(ln)}, indent=1)); print("LQ cycles/node between the first and the last stamp", rows[:, :9].sum(1).mean())
itf.debug_set("lq_prof", 0)
itf.debug_set("wbc_stop", -3)
mpc.control_step_resident(cfg["horizon"], cfg["period"], cfg["time"]); itf.synchronize()
cyc = itf.debug_read("wbc_scratch", (B, 432))
nn_ = ["L0 load", "L0 pivoted QR", "L0 backward accumulation", "L0 write Zp", "(start -> rigid-body passes)", "L1 load", "L1 pivoted QR", "L1 backward accumulation", "(start -> end of rigid-body passes)", "L1 Zp <- Zp Q2"]
print(json.dumps({n: float(cyc[:, i].mean()) for i, n in enumerate(nn_)}, indent=1))
itf.debug_set("wbc_stop", 0)
