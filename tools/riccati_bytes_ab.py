"""tools/riccati_bytes_ab.py — round-5 review item 5: what could K3 gain from moving fewer bytes?  The instrumented instance of the Riccati kernel with skip bit 64 leaves out
three of the twelve 1 KB fragment chunks per stage (the 360 doubles that packed triangles of Qp(0,0), Qp(1,1), Rp would save, 7 % of the stage's bytes) WITHOUT paying for any
unpacking (results meaningless): an upper bound of the packing's gain.  Alternates with bit 128 (the same instance, nothing skipped) on one box; MPC-only steps."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from qm_control_amd import api, scenarios
B = 1024; cfg = scenarios.make_config("C3", batch=B)
itf = api.QMInterface(blobs=scenarios.load_blobs(), max_batch=B, max_nodes=128, max_ref_knots=2, max_events=cfg["ev"].shape[1])
mpc = api.SqpMpc(itf); mpc.set_problem(cfg["t0"], cfg["x0"], cfg["ref_t"], cfg["ref_x"], cfg["ev"], cfg["modes"])
for _ in range(12): itf.microbench_fp64(True)
for rnd in range(3):
    for bit, name in ((128, "all 22 chunks"), (64, "19 chunks (lean)"), (0, "product instance")):
        itf.debug_set("riccati_skip", bit)
        for _ in range(3): mpc.solve_resident(cfg["horizon"])
        itf.synchronize(); itf.set_profiling(True); itf.reset_kernel_ms()
        for _ in range(10): mpc.solve_resident(cfg["horizon"])
        itf.synchronize(); itf.set_profiling(False)
        print(json.dumps({"round": rnd, "variant": name, "riccati_ms": round(itf.kernel_ms("riccati")[0] / max(1, itf.kernel_ms("riccati")[1]), 4)}), flush=True)
itf.debug_set("riccati_skip", 0); itf.close()
