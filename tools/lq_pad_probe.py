import os, sys, json
ROOT='/root/repo'; sys.path.insert(0, ROOT)
import numpy as np
from qm_control_amd import api, scenarios
api.LIB_PATH = os.path.join(ROOT, os.environ["QM_AB_LIB"])
cfg = scenarios.make_config("C4", batch=1024)
itf = api.QMInterface(blobs=scenarios.load_blobs(), max_batch=1024, max_nodes=128, max_ref_knots=2, max_events=cfg["ev"].shape[1])
mpc = api.SqpMpc(itf); mpc.set_problem(cfg["t0"], cfg["x0"], cfg["ref_t"], cfg["ref_x"], cfg["ev"], cfg["modes"])
for _ in range(12): itf.microbench_fp64(True)
out = {}
for pad in [0, 512, 2048, 4096, 10240, 24576]:
    itf.debug_set("lds_pad:lq", pad)
    for _ in range(3): mpc.solve_resident(cfg["horizon"])
    itf.synchronize(); itf.set_profiling(True); itf.reset_kernel_ms()
    for _ in range(8): mpc.solve_resident(cfg["horizon"])
    itf.synchronize(); itf.set_profiling(False)
    own = {"a.so": 16896, "v5.so": 16000}.get(os.path.basename(os.environ["QM_AB_LIB"]), 16384)
    out[pad] = (160 * 1024 // (own + pad), round(itf.kernel_ms("lq")[0] / itf.kernel_ms("lq")[1], 4))
print(os.environ["QM_AB_LIB"], json.dumps(out))
