"""tools/gpu_phase_probe.py — time K3 with phases skipped (profiling only)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from qm_control_amd import api, scenarios
B = 1024
cfg = scenarios.make_config("C4", batch=B)
itf = api.QMInterface(blobs=scenarios.load_blobs(), max_batch=B, max_nodes=128, max_ref_knots=2, max_events=cfg["ev"].shape[1])
mpc = api.SqpMpc(itf); mpc.set_problem(cfg["t0"], cfg["x0"], cfg["ref_t"], cfg["ref_x"], cfg["ev"], cfg["modes"])
out = {}
for name, mask in (("full", 0), ("no_chol_solve", 1), ("no_closed_loop", 2), ("no_forward", 4), ("no_SA_SB", 8), ("no_Hux_Huu", 16), ("events_only(no stage work)", 32), ("no_chol,no_cl,no_fwd", 7), ("all gemm+chol+cl+fwd off", 31)):
    itf.debug_set("riccati_skip", mask)
    mpc.solve_resident(cfg["horizon"]); itf.synchronize()
    itf.set_profiling(True); itf.reset_kernel_ms()
    for _ in range(3): mpc.solve_resident(cfg["horizon"])
    ms, n = itf.kernel_ms("riccati"); out[name] = ms / n; itf.set_profiling(False)
print(json.dumps(out, indent=1))
