"""tools/lq_prof.py — cycles per phase of K1b from the instrumented instance's stamps (qm_lq_dbg_kernel with the debug switch lq_prof: SR_K of every stage record; the
instance runs at two waves per SIMD with the debug branches compiled in: the SHARES are what this is for), next to the instructions the product instance issues per phase."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from qm_control_amd import api, scenarios
import lq_record_check as LC
if os.environ.get("QM_AB_LIB"): api.LIB_PATH = os.path.join(ROOT, os.environ["QM_AB_LIB"])
B = 1024
cfg = scenarios.make_config("C4", batch=B)
itf = api.QMInterface(blobs=scenarios.load_blobs(), max_batch=B, max_nodes=128, max_ref_knots=2, max_events=cfg["ev"].shape[1])
mpc = api.SqpMpc(itf); mpc.set_problem(cfg["t0"], cfg["x0"], cfg["ref_t"], cfg["ref_x"], cfg["ev"], cfg["modes"])
mpc.solve_resident(cfg["horizon"]); itf.synchronize()
itf.debug_set("lq_prof", 1); itf.debug_set("riccati_skip", 20)
mpc.solve_resident(cfg["horizon"]); itf.synchronize()
nm = 128; SR = LC.SR["SR_SIZE"]; K0 = LC.SR["SR_K"]
stage = itf.debug_read("stage", (B * nm, SR))
rows = stage[np.arange(B)[:, None] * nm + np.arange(5, 95)[None, :]].reshape(-1, SR)[:, K0:K0 + 17]
rows = rows[(rows[:, 9] > 1e3) & (rows[:, 9] < 1e7)]
ln = ["P0 inputs -> LDS, defect, tracking terms", "II constraint rows", "II projector (G, Px, Pe, Pu descriptors)", "I jacobian columns -> tile", "I RK2 composition, B_d transposed", "projected dynamics + Bp",
      "III cost model (R0 (u - unom), barriers)", "III R / EE / Q assembly", "III projected cost + stores"]
out = {n: round(float(rows[:, i].mean())) for i, n in enumerate(ln)}
out["sum of the phases"] = round(float(rows[:, :9].sum(1).mean())); out["wave lifetime (entry -> stores acknowledged)"] = round(float(rows[:, 9].mean()))
out["entry -> first stamp (prologue loads)"] = round(float(rows[:, 15].mean())); out["last stamp -> vmcnt 0 (store drain)"] = round(float(rows[:, 16].mean()))
print(json.dumps(out, indent=1))
