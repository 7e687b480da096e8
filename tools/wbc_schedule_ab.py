"""tools/wbc_schedule_ab.py — one variant of the WBC scheduling experiment (round 6): the pipelined cold control step of the benchmark workload with
QM_WBC_DEFER=1 (WBC(k) launched behind K1a(k + 1) instead of right after the policy of step k) and / or QM_WBC_STREAM_PRIORITY=-1|0|1 (read by libqmhip at context creation).
Prints one JSON line; tools/gpu_call.sh alternates the variants on one box."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from qm_control_amd import api, scenarios
B = 1024; cfg = scenarios.make_config("C3", batch=B)
itf = api.QMInterface(blobs=scenarios.load_blobs(), max_batch=B, max_nodes=128, max_ref_knots=2, max_events=cfg["ev"].shape[1])
mpc = api.SqpMpc(itf); wbc = api.HierarchicalWbc(itf)
if os.environ.get("QM_WBC_DEFER"): itf.debug_set("wbc_defer", int(os.environ["QM_WBC_DEFER"]))
for _ in range(12): itf.microbench_fp64(True)
mpc.set_problem(cfg["t0"], cfg["x0"], cfg["ref_t"], cfg["ref_x"], cfg["ev"], cfg["modes"]); wbc.reset()
for _ in range(5): mpc.control_step_resident(cfg["horizon"], cfg["period"], cfg["time"])
itf.synchronize(); ts = []
for rep in range(3):
    t = time.perf_counter()
    for _ in range(30): mpc.control_step_resident(cfg["horizon"], cfg["period"], cfg["time"])
    itf.synchronize(); ts.append((time.perf_counter() - t) / 30 * 1e3)
itf.set_profiling(True); itf.reset_kernel_ms()
for _ in range(10): mpc.control_step_resident(cfg["horizon"], cfg["period"], cfg["time"])
itf.synchronize(); itf.set_profiling(False)
ms = {k: round(itf.kernel_ms(k)[0] / max(1, itf.kernel_ms(k)[1]), 4) for k in ("grid", "lq_kin", "lq", "riccati", "ls_eval", "ls_misc", "policy", "wbc")}
res = mpc.download(); out, qps = wbc.download(B)
print(json.dumps({"defer": os.environ.get("QM_WBC_DEFER", "0"), "prio": os.environ.get("QM_WBC_STREAM_PRIORITY", "none"), "ms_per_step": [round(v, 4) for v in ts], "kernel_ms_profiled": ms,
                  "ok": bool((res["status"] >= 0).all() and (qps == 0).all()), "tau_checksum": float(np.abs(out[:, 36:]).sum())}))
