"""tools/riccati_prof.py — in-kernel cycle counters of K3 (profiling only)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from qm_control_amd import api, scenarios
if os.environ.get("QM_AB_LIB"): api.LIB_PATH = os.path.join(ROOT, os.environ["QM_AB_LIB"])
B = 1024
cfg = scenarios.make_config("C4", batch=B)
itf = api.QMInterface(blobs=scenarios.load_blobs(), max_batch=B, max_nodes=128, max_ref_knots=2, max_events=cfg["ev"].shape[1])
mpc = api.SqpMpc(itf); mpc.set_problem(cfg["t0"], cfg["x0"], cfg["ref_t"], cfg["ref_x"], cfg["ev"], cfg["modes"])
mpc.solve_resident(cfg["horizon"]); itf.synchronize()
import sys as _s
itf.debug_set("riccati_skip", 32 | int(os.environ.get("RSKIP", "0")))
mpc.solve_resident(cfg["horizon"]); itf.synchronize()
sys.path.insert(0, os.path.join(ROOT, 'tests')); import lq_record_check as LC
nm = 128; SR = LC.SR['SR_SIZE']
stage = itf.debug_read("stage", (B * nm, SR))
rows = stage[np.arange(B) * nm][:, LC.SR['SR_K']:LC.SR['SR_K'] + 15]
names = ["operands LDS->frag (+dma wait)", "5 products", "stage to LDS + columns", "Cholesky loop", "scale, stores, W reload", "WtW", "symmetrise", "BACKWARD total", "FORWARD total", "fwd: loop head + dx store", "fwd: record regs -> LDS", "fwd: next-record fetch issue", "fwd: A dx, W dx", "fwd: triangular solve", "fwd: Pu ut, B ut, du store"]
print(json.dumps({n: float(rows[:, i].mean()) for i, n in enumerate(names)}, indent=1))
itf.debug_set("riccati_skip", 0)
itf.set_profiling(True); itf.reset_kernel_ms()
for _ in range(3): mpc.solve_resident(cfg["horizon"])
print({k: round(itf.kernel_ms(k)[0] / max(1, itf.kernel_ms(k)[1]), 3) for k in ("lq_kin", "lq", "riccati", "ls_eval")})
