"""tools/quick_kernel_ms.py — per-kernel HIP-event times and the step rate of the benchmark workload in a few seconds (iteration aid on the GPU box)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from qm_control_amd import api, scenarios
if os.environ.get("QM_AB_LIB"): api.LIB_PATH = os.path.join(ROOT, os.environ["QM_AB_LIB"])       # A/B runs of two builds on one box (tools/ab_kernel_ms.sh)
B = int(os.environ.get("QM_B", "1024"))
cfg = scenarios.make_config("C4", batch=B)
itf = api.QMInterface(blobs=scenarios.load_blobs(), max_batch=B, max_nodes=int(os.environ.get("QM_NMAX", "128")), max_ref_knots=2, max_events=cfg["ev"].shape[1])
mpc = api.SqpMpc(itf); wbc = api.HierarchicalWbc(itf)
mpc.set_problem(cfg["t0"], cfg["x0"], cfg["ref_t"], cfg["ref_x"], cfg["ev"], cfg["modes"])
MPC_ONLY = bool(os.environ.get("QM_MPC_ONLY"))
if os.environ.get("QM_LQ_SLICES"): itf.debug_set("lq_slices", int(os.environ["QM_LQ_SLICES"]))
def step():
    if MPC_ONLY: mpc.solve_resident(cfg["horizon"])       # no WBC on the second stream: per-kernel times without cross-stream waiting
    else: wbc.reset(); mpc.control_step_resident(cfg["horizon"], cfg["period"], cfg["time"])
for _ in range(12): itf.microbench_fp64(True)
for _ in range(5): step()
itf.synchronize(); itf.set_profiling(True); itf.reset_kernel_ms()
for _ in range(10): step()
itf.synchronize(); itf.set_profiling(False)
ms = {k: round(itf.kernel_ms(k)[0] / max(1, itf.kernel_ms(k)[1]), 4) for k in ("grid", "lq_kin", "lq", "riccati", "ls_eval", "ls_misc", "policy", "wbc")}
itf.synchronize(); t = time.perf_counter()
for _ in range(30): step()
itf.synchronize(); dt = (time.perf_counter() - t) / 30
res = mpc.download(); out, qps = wbc.download(B)
if MPC_ONLY: qps = qps * 0
print(json.dumps({"B": B, "ms_per_step": round(dt * 1e3, 4), "steps_per_s": round(B / dt), "ok": bool((res["status"] == 0).all() and (qps == 0).all()), "kernel_ms": ms, "tau_checksum": float(np.abs(out[:, 36:]).sum())}))
