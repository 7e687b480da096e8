"""tools/batch_sweep.py — the benchmark step (C4 instances, trot N = 100) at other batch sizes; prints ms/step and steps/s."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from qm_control_amd import api, scenarios
for B in (1, 16, 100, 256, 512, 1024, 2048, 4096):
    cfg = scenarios.make_config("C4", batch=B)
    itf = api.QMInterface(blobs=scenarios.load_blobs(), max_batch=B, max_nodes=128, max_ref_knots=2, max_events=cfg["ev"].shape[1])
    mpc = api.SqpMpc(itf); wbc = api.HierarchicalWbc(itf)
    mpc.set_problem(cfg["t0"], cfg["x0"], cfg["ref_t"], cfg["ref_x"], cfg["ev"], cfg["modes"])
    for _ in range(2): wbc.reset(); mpc.control_step_resident(cfg["horizon"], cfg["period"], cfg["time"])
    itf.synchronize(); n = 10 if B <= 1024 else 5
    t = time.perf_counter()
    for _ in range(n): wbc.reset(); mpc.control_step_resident(cfg["horizon"], cfg["period"], cfg["time"])
    itf.synchronize(); t = (time.perf_counter() - t) / n
    res = mpc.download(); out, qps = wbc.download(B)
    print("B %5d  ms/step %7.3f  steps/s %9.0f  ok %s" % (B, t * 1e3, B / t, bool((res["status"] == 0).all() and (qps == 0).all())))
    itf.close()
