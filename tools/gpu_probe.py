"""tools/gpu_probe.py — first-light measurements on an MI355X: FP64 micro-benchmarks, per-kernel timing."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from qm_control_amd import api, scenarios

def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    blobs = scenarios.load_blobs()
    cfg = scenarios.make_config("C4", batch=B)
    itf = api.QMInterface(blobs=blobs, max_batch=B, max_nodes=128, max_ref_knots=2, max_events=cfg["ev"].shape[1])
    out = {"B": B}
    out["fp64_fma_tflops"] = itf.microbench_fp64(False); out["fp64_mfma_tflops"] = itf.microbench_fp64(True)
    mpc = api.SqpMpc(itf)
    mpc.set_problem(cfg["t0"], cfg["x0"], cfg["ref_t"], cfg["ref_x"], cfg["ev"], cfg["modes"])
    mpc.solve_resident(cfg["horizon"]); itf.synchronize()
    t = time.time(); reps = 3
    for _ in range(reps): mpc.solve_resident(cfg["horizon"])
    itf.synchronize(); out["mpc_ms_per_batch"] = (time.time() - t) / reps * 1e3
    itf.set_profiling(True); itf.reset_kernel_ms()
    for _ in range(reps): mpc.solve_resident(cfg["horizon"])
    for k in ("grid", "lq", "riccati", "ls_eval", "ls_misc"):
        ms, n = itf.kernel_ms(k); out["ms_" + k] = ms / reps; out["launches_" + k] = n / reps
    res = mpc.download(); out["status_bad"] = int((res["status"] != 0).sum()); out["nodes_mean"] = float(res["num_nodes"].mean()); out["ls_trials"] = res["ls_trials"]
    out["alpha_hist"] = {str(a): int(c) for a, c in zip(*np.unique(res["perf"][:, 8], return_counts=True))}
    print(json.dumps(out, indent=1))

if __name__ == "__main__":
    main()
