"""tools/coresidency_footprint_probe.py — round 6: what would a WBC at <= 344 registers / <= 24 KB of LDS (one LQ wave per SIMD beside it) be worth, measured BEFORE rewriting it?
A latency-bound stand-in (qm_filler_wide*_kernel: dependent f64-MFMA / FMA / LDS chains like the WBC's, 1024 one-wave workgroups) takes the WBC's place on the second stream of
back-to-back MPC steps of the benchmark batch, with
   (a) today's WBC footprint       386 registers, 40 KB  -> nothing with LDS fits beside it (4 x 40 KB = the CU's 160 KB)           [calibration: must behave like the real WBC]
   (b) the footprint asked for      339 registers, 24 KB  -> one 168-register / 15.6 KB LQ wave per SIMD fits beside it
   (c) a narrow wave                 25 registers, 20 KB  -> the LQ kernel keeps (nearly) its three waves per SIMD: the upper bound of any co-residency
its length tuned to the WBC kernel's ~ 0.42 ms when alone.  Reports ms per step of: MPC alone, MPC + real WBC (the product's pipelined control step), MPC + each stand-in."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from qm_control_amd import api, scenarios
B = 1024; cfg = scenarios.make_config("C3", batch=B)
itf = api.QMInterface(blobs=scenarios.load_blobs(), max_batch=B, max_nodes=128, max_ref_knots=2, max_events=cfg["ev"].shape[1])
mpc = api.SqpMpc(itf); wbc = api.HierarchicalWbc(itf)
mpc.set_problem(cfg["t0"], cfg["x0"], cfg["ref_t"], cfg["ref_x"], cfg["ev"], cfg["modes"]); wbc.reset()
for _ in range(12): itf.microbench_fp64(True)
TARGET = float(os.environ.get("QM_FILLER_MS", "0.42"))


def timed(fn, n=30, reps=3):
    for _ in range(5): fn()
    itf.synchronize(); out = []
    for _ in range(reps):
        t = time.perf_counter()
        for _ in range(n): fn()
        itf.synchronize(); out.append((time.perf_counter() - t) / n * 1e3)
    return [round(v, 4) for v in out]


res = {"mpc_alone_ms": timed(lambda: mpc.solve_resident(cfg["horizon"])), "mpc_plus_real_wbc_ms": timed(lambda: mpc.control_step_resident(cfg["horizon"], cfg["period"], cfg["time"]))}
itf.set_profiling(True); itf.reset_kernel_ms()
for _ in range(5): mpc.control_step_resident(cfg["horizon"], cfg["period"], cfg["time"])
itf.synchronize(); itf.set_profiling(False); res["real_wbc_kernel_ms_in_pipeline"] = round(itf.kernel_ms("wbc")[0] / max(1, itf.kernel_ms("wbc")[1]), 4)
for name, live, lds in (("a_wide_386vgpr_40KB", 184, 40 * 1024), ("b_wide_339vgpr_24KB", 160, 24 * 1024), ("b2_wide_339vgpr_40KB", 160, 40 * 1024), ("b3_wide_386vgpr_24KB", 184, 24 * 1024), ("c_narrow_25vgpr_20KB", 0, 20 * 1024)):
    itf.debug_set("filler_live", live); itf.debug_set("filler_lds", lds)
    iters = 1500
    for _ in range(4):      # tune the length to the target when alone
        alone = min(itf.debug_filler(1024, iters) for _ in range(3)); iters = max(50, int(iters * TARGET / alone))
    alone = min(itf.debug_filler(1024, iters) for _ in range(3))

    def step():
        itf.debug_filler(1024, iters, wait=False); mpc.solve_resident(cfg["horizon"])
    res[name] = {"filler_alone_ms": round(alone, 4), "iters": iters, "mpc_plus_filler_ms": timed(step)}
    # the same stand-in launched BEHIND K1a of the next step (K1a, 254 registers, fits beside neither wide footprint and precedes the LQ kernel in its stream: in the product's order
    # nothing of step k + 1 but the grid kernels can start before the wide waves retire) — beside the LQ kernel only
    itf.debug_set("filler_at_lq", 1); res[name]["mpc_plus_filler_behind_K1a_ms"] = timed(step); itf.debug_set("filler_at_lq", 0); itf.synchronize()
print(json.dumps(res, indent=1))
