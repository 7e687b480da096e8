"""tools/lq_residency_probe.py — what a wavefront of K1b (one node) spends outside its stamped phases, and how well the 2048 wave slots of the chip stay filled.
The instrumented instance (qm_lq_dbg_kernel, debug switch lq_prof) stamps every wave's entry and exit in shader-clock cycles AND in ticks of the constant 100 MHz
reference clock, with the hardware slot it ran on (HW_ID / XCC_ID): shader clock under this kernel's load, resident waves per SIMD, idle gaps between consecutive
waves of one slot, prologue (entry -> first phase stamp: the one memory round trip of the node's inputs) and store drain (last stamp -> vmcnt 0)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from qm_control_amd import api, scenarios
if os.environ.get("QM_AB_LIB"): api.LIB_PATH = os.path.join(ROOT, os.environ["QM_AB_LIB"])
sys.path.insert(0, os.path.join(ROOT, "tests")); import lq_record_check as LC
B = 1024; nm = 128; SR = LC.SR["SR_SIZE"]; SRK = LC.SR["SR_K"]      # (record layout from the header: the stamps live in SR_K of every stage record)
cfg = scenarios.make_config("C4", batch=B)
itf = api.QMInterface(blobs=scenarios.load_blobs(), max_batch=B, max_nodes=nm, max_ref_knots=2, max_events=cfg["ev"].shape[1])
mpc = api.SqpMpc(itf); mpc.set_problem(cfg["t0"], cfg["x0"], cfg["ref_t"], cfg["ref_x"], cfg["ev"], cfg["modes"])
mpc.solve_resident(cfg["horizon"]); itf.synchronize()
itf.debug_set("lq_prof", 1); itf.debug_set("riccati_skip", 20)      # (K3 would overwrite nothing here, but its backward stage is skipped so that the records stay as K1b left them)
mpc.solve_resident(cfg["horizon"]); itf.synchronize()
rows = itf.debug_read("stage", (B * nm, SR)).reshape(B, nm, SR)[:, :, SRK:SRK + 18]
valid = (rows[:, :, 10] > 1e2) & (rows[:, :, 10] < 1e6) & (rows[:, :, 9] > 1e3) & (rows[:, :, 9] < 1e7) & (rows[:, :, 12] > rows[:, :, 11])
r = rows[valid]
cyc, real, r0, r1, hw, xcc = r[:, 9], r[:, 10], r[:, 11], r[:, 12], r[:, 13].astype(np.int64), r[:, 14].astype(np.int64)
slot = (xcc & 15) * (1 << 20) + (hw & 0xFFFFF)                       # XCC, then SE / SH / CU / SIMD / wave slot of HW_ID
order = np.lexsort((r0, slot)); ss = slot[order]; a0 = r0[order]; a1 = r1[order]
gaps = (a0[1:] - a1[:-1])[ss[1:] == ss[:-1]]
span = r1.max() - r0.min(); nsimd = len(np.unique(slot >> 4))
names = ["P0 stage inputs, defect, tracking terms", "II constraint rows", "II projector (G, Px, Pe, Pu descriptors)", "I jacobian columns -> tile", "I RK2 composition, B_d transposed",
         "projected dynamics + Bp", "III cost model (R0 (u - unom), barriers)", "III EE term, [Q | q], R assembly", "III projected cost + stores"]      # order of the round-4 kernel (constraints before Jacobians)
print(json.dumps({
    "waves_stamped": int(len(r)), "wave_slots_used": int(len(np.unique(slot))), "simds_used": int(nsimd),
    "shader_clock_GHz_under_K1b": round(float(cyc.sum() / real.sum() * 0.1), 3),
    "kernel_span_us": round(float(span / 100.0), 1),
    "wave_lifetime_cycles_mean": round(float(cyc.mean())), "wave_lifetime_us_mean": round(float(real.mean() / 100.0), 2),
    "resident_waves_per_simd_mean": round(float(real.sum() / (nsimd * span)), 3),
    "slot_gap_us": {"mean": round(float(gaps.mean() / 100), 2), "median": round(float(np.median(gaps) / 100), 2), "p90": round(float(np.percentile(gaps, 90) / 100), 2)},
    "prologue_cycles_mean": round(float(r[:, 15].mean())), "store_drain_cycles": {"mean": round(float(r[:, 16].mean())), "median": round(float(np.median(r[:, 16]))), "p90": round(float(np.percentile(r[:, 16], 90)))},
    "phase_cycles_mean": {n: round(float(r[:, k].mean())) for k, n in enumerate(names)},
    "phases_sum_cycles": round(float(r[:, :9].sum(1).mean())),
    "note": "instrumented instance (a few per cent slower than qm_lq_kernel: debug branches, cycle stamps, the closing s_waitcnt)"}, indent=1))
