"""tools/cold_step_dump.py <tag> — one cold control step of the bench batch (C3, 1024 instances, seed 1235) on the build QM_AB_LIB names; every output array saved bit-exactly to
gpurun_out/cold_<tag>.npz so that two builds are compared ENTRY BY ENTRY (the 16-digit torque checksum of bench.py cannot see per-entry differences of 1e-15)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from qm_control_amd import api, scenarios
if os.environ.get("QM_AB_LIB"): api.LIB_PATH = os.path.join(ROOT, os.environ["QM_AB_LIB"])
import bench
cfg = scenarios.make_config("C3", batch=1024, n_intervals=100)
e = bench.HipEngine(cfg, 0); e.step(); e.sync()
res = e.mpc.download(); out, st = e.wbc.download(1024)
np.savez(os.path.join(os.environ.get("QM_DUMP_DIR", "gpurun_out"), "cold_%s.npz" % sys.argv[1]), x=res["x"], u=res["u"], t=res["t"], perf=res["perf"], status=res["status"], wbc=out, wbc_status=st)
print(sys.argv[1], os.path.relpath(api.LIB_PATH, ROOT), "torque checksum %.16g" % float(np.asarray(out)[:, -18:].sum()), "status ok", bool((res["status"] >= 0).all()))
e.close()
