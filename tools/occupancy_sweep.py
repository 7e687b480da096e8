"""tools/occupancy_sweep.py — BASELINE.json config 3's "per-node wavefront occupancy sweep" (SURVEY.md §8(d) C3): kernel time against the number of wavefronts
resident per CU, on the benchmark workload (trot, N = 100, random initial states).  Run on the GPU box; writes gpurun_out/occupancy_sweep.json
(kept as profiles/r03_occupancy_sweep.json).

Two ways of moving the occupancy, both without touching the kernels:
  * LDS padding (qmhip_debug_set "lds_pad:<kernel>"): extra dynamic LDS per workgroup caps the workgroups a CU can hold (160 KB of LDS per CU) —
    `waves_per_cu` = min(register limit, LDS limit) x waves per workgroup; times are MPC-only steps (no WBC on the second stream) at B = 1024;
  * batch size for the one-wave-per-instance solvers (K3, WBC): B / 256 waves per CU are OFFERED, 4 per CU (one per SIMD) fit their register budget.
"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from qm_control_amd import api, scenarios

LDS_CU = 160 * 1024
# kernel group -> (own LDS bytes per workgroup, waves per workgroup, waves per SIMD allowed by its registers); from tools/kernel_resources.py / the launch code
GROUPS = {"lq": (None, 1, 3), "lq_kin": (None, 1, 1), "ls_eval": (None, 1, 1), "riccati": (None, 1, 1), "wbc": (None, 1, 1)}


def engine(B):
    cfg = scenarios.make_config("C4", batch=B)
    itf = api.QMInterface(blobs=scenarios.load_blobs(), max_batch=B, max_nodes=128, max_ref_knots=2, max_events=cfg["ev"].shape[1])
    mpc = api.SqpMpc(itf); wbc = api.HierarchicalWbc(itf)
    mpc.set_problem(cfg["t0"], cfg["x0"], cfg["ref_t"], cfg["ref_x"], cfg["ev"], cfg["modes"])
    return cfg, itf, mpc, wbc


def kernel_ms(itf, step, names, reps=6):
    for _ in range(2): step()
    itf.synchronize(); itf.set_profiling(True); itf.reset_kernel_ms()
    for _ in range(reps): step()
    itf.synchronize(); itf.set_profiling(False)
    return {k: itf.kernel_ms(k)[0] / max(1, itf.kernel_ms(k)[1]) for k in names}


def main():
    out = {"workload": "C4 (trot, N = 100, seed 1235)", "lds_per_cu": LDS_CU, "lds_padding_sweep_B1024": {}, "batch_sweep": {}}
    sizes = {"lq": 16000, "lq_kin": 20480, "ls_eval": 16384, "riccati": 38272, "wbc": 40944}      # own LDS per workgroup: LQ_LDS_BYTES, LQ_KIN_LDS_BYTES, LS_EVAL_LDS_BYTES, RW_LDS_BYTES, WBC_LDS_BYTES
    batch_only = "--batch-only" in sys.argv
    cfg, itf, mpc, wbc = engine(1024)
    for _ in range(12): itf.microbench_fp64(True)
    mpc_step = lambda: mpc.solve_resident(cfg["horizon"])
    def wbc_step():
        wbc.reset(); mpc.control_step_resident(cfg["horizon"], cfg["period"], cfg["time"]); itf.synchronize()
    for g, own in ({} if batch_only else sizes).items():
        rows = []
        for target in (8, 4, 2, 1):                                   # workgroups (= waves: every kernel here is one wave per workgroup) per CU allowed by LDS
            reg_cap = 4 * GROUPS[g][2]; natural = min(reg_cap, LDS_CU // own)
            if target > natural: continue
            pad = 0 if target == natural else LDS_CU // target - own - 256          # `target` workgroups fit a CU's LDS, target + 1 do not
            per_cu = min(reg_cap, LDS_CU // (own + pad))
            try:
                itf.debug_set("lds_pad:" + g, pad)
                ms = kernel_ms(itf, wbc_step if g == "wbc" else mpc_step, [g])[g]
                rows.append({"waves_per_cu": per_cu, "lds_bytes_per_workgroup": own + pad, "kernel_ms": round(ms, 4)})
            except api.QmhipError as e:
                rows.append({"waves_per_cu": per_cu, "lds_bytes_per_workgroup": own + pad, "error": str(e)[:200]})
            itf.debug_set("lds_pad:" + g, 0)
        out["lds_padding_sweep_B1024"]["qm_%s_kernel" % g] = rows
        print(g, rows, flush=True)
    itf.close()
    for B in (256, 512, 1024, 2048, 4096, 8192):
        cfg, itf, mpc, wbc = engine(B)
        def step():
            wbc.reset(); mpc.control_step_resident(cfg["horizon"], cfg["period"], cfg["time"])
        ms = kernel_ms(itf, lambda: (step(), itf.synchronize()), ["lq_kin", "lq", "riccati", "ls_eval", "wbc"], reps=4)
        itf.synchronize(); t = time.perf_counter()
        for _ in range(10): step()
        itf.synchronize(); dt = (time.perf_counter() - t) / 10
        out["batch_sweep"][str(B)] = {"instances_per_simd": B / 1024.0, "solver_waves_offered_per_cu": B / 256.0, "ms_per_step_pipelined": round(dt * 1e3, 4), "steps_per_s": round(B / dt), "ls_trials": int(mpc.download()["ls_trials"]),
                                      "kernel_ms_unpipelined": {k: round(v, 4) for k, v in ms.items()}, "kernel_us_per_instance": {k: round(1e3 * v / B, 4) for k, v in ms.items()}}
        print(B, out["batch_sweep"][str(B)], flush=True)
        itf.close()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "occupancy_sweep.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
