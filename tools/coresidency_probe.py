"""tools/coresidency_probe.py — what is sharing SIMDs between a narrow latency-bound solver wave and the issue-bound LQ kernel worth?  (profiling only)
A filler kernel (<= 256 VGPR, 20 KB LDS, 1024 waves, dependent MFMA / FMA / LDS chains) runs on the second stream beside MPC iterations of the benchmark batch."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from qm_control_amd import api, scenarios
B = 1024
cfg = scenarios.make_config("C4", batch=B)
itf = api.QMInterface(blobs=scenarios.load_blobs(), max_batch=B, max_nodes=128, max_ref_knots=2, max_events=cfg["ev"].shape[1])
mpc = api.SqpMpc(itf); mpc.set_problem(cfg["t0"], cfg["x0"], cfg["ref_t"], cfg["ref_x"], cfg["ev"], cfg["modes"])
for _ in range(12): itf.microbench_fp64(True)
for _ in range(3): mpc.solve_resident(cfg["horizon"])
itf.synchronize()
res = {}
for iters in (2000, 4000):
    alone = min(itf.debug_filler(1024, iters) for _ in range(3)); res["filler_alone_ms_%d" % iters] = alone
def mpc_time(n, filler_iters=0):
    itf.synchronize(); t = time.perf_counter()
    for _ in range(n):
        if filler_iters: itf.debug_filler(1024, filler_iters, wait=False)
        mpc.solve_resident(cfg["horizon"])
    itf.synchronize(); return (time.perf_counter() - t) / n * 1e3
res["mpc_alone_ms"] = mpc_time(10)
for iters in (2000, 4000):
    res["mpc_with_filler_ms_%d" % iters] = mpc_time(10, iters)
    res["serial_sum_ms_%d" % iters] = res["mpc_alone_ms"] + res["filler_alone_ms_%d" % iters]
for iters in (0, 2000, 3500):
    r = min((itf.debug_lq_with_filler(B, cfg["horizon"], 1024 if iters else 0, max(1, iters)) for _ in range(3)), key=lambda v: v[0])
    res["lq_beside_filler_%d" % iters] = {"lq_ms": r[0], "filler_ms": r[1], "riccati_ms": r[2]}
print(json.dumps(res, indent=1))
