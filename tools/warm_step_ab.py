"""tools/warm_step_ab.py [rounds] — same-box A/B of the line search's two drivers on the benchmark workload (1024 instances, trot, N = 100): `ls_device_tail` 1 (round 6: the
trials after the first in ONE launch, no host round trip) against 0 (rounds 1-5: one host round trip per trial), alternating on one context.  Per variant: the pipelined
COLD control step the headline times (one trial), and the WARM-started closed loop (`closed_loop_warm_start` of the bench line: 10 receding-horizon steps, ~ 15 % of the
instances backtrack) with a checksum of the torques and of the last primal solution — the two drivers must agree bit for bit."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from qm_control_amd import api, scenarios
B = int(os.environ.get("QM_B", "1024")); rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cfg = scenarios.make_config("C3", batch=B)
itf = api.QMInterface(blobs=scenarios.load_blobs(), max_batch=B, max_nodes=128, max_ref_knots=2, max_events=cfg["ev"].shape[1])
mpc = api.SqpMpc(itf); wbc = api.HierarchicalWbc(itf)
for _ in range(12): itf.microbench_fp64(True)


def cold(n):
    mpc.set_problem(cfg["t0"], cfg["x0"], cfg["ref_t"], cfg["ref_x"], cfg["ev"], cfg["modes"]); wbc.reset()
    for _ in range(3): mpc.control_step_resident(cfg["horizon"], cfg["period"], cfg["time"])
    itf.synchronize(); t = time.perf_counter()
    for _ in range(n): mpc.control_step_resident(cfg["horizon"], cfg["period"], cfg["time"])
    itf.synchronize(); return (time.perf_counter() - t) / n * 1e3


def warm(n):
    mpc.set_problem(cfg["t0"], cfg["x0"], cfg["ref_t"], cfg["ref_x"], cfg["ev"], cfg["modes"]); wbc.reset()
    mpc.closed_loop_resident(2, 0.01, cfg["horizon"], cfg["period"], cfg["time"]); itf.synchronize(); t = time.perf_counter()
    mpc.closed_loop_resident(n, 0.01, cfg["horizon"], cfg["period"], cfg["time"]); itf.synchronize(); dt = (time.perf_counter() - t) / n * 1e3
    res = mpc.download(); out, qps = wbc.download(B)
    return dt, int(res["ls_trials"]), float(np.abs(out[:, 36:]).sum()), float(np.abs(res["x"]).sum() + np.abs(res["u"]).sum()), bool((res["status"] >= 0).all() and (qps == 0).all())


for r in range(rounds):
    for tail, fused in ((0, 0), (1, 0), (1, 1)):      # rounds 1-5 | device-side line-search tail | + policy at t0 from the deciding kernels, apply beside the WBC
        itf.debug_set("ls_device_tail", tail); itf.debug_set("fused_policy", fused)
        c = cold(20); w = warm(10)
        print(json.dumps({"round": r, "ls_device_tail": tail, "fused_policy": fused, "cold_ms_per_step": round(c, 4), "warm_ms_per_step": round(w[0], 4), "warm_ls_trials_last": w[1], "tau_checksum": w[2], "primal_checksum": w[3], "ok": w[4]}), flush=True)
itf.close()
