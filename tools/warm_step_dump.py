"""tools/warm_step_dump.py <tag> <n> — the bench batch (C3, 256 instances) run as n receding-horizon steps on the device (warm-started from the previous solution, observation advanced along the
policy; n = 1 is the cold step), outputs saved bit-exactly to $QM_DUMP_DIR/cold_<tag>.npz for tools/cold_step_compare.py.
QM_DEBUG_SET=key=value[,key=value]: debug switches of the context (qmhip_debug_set) set before the run."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from qm_control_amd import api, scenarios
if os.environ.get("QM_AB_LIB"): api.LIB_PATH = os.path.join(ROOT, os.environ["QM_AB_LIB"])
import bench
B = 256; n = int(sys.argv[2]); cfg = scenarios.make_config("C3", batch=B, n_intervals=100)
e = bench.HipEngine(cfg, 0)
for kv in filter(None, os.environ.get("QM_DEBUG_SET", "").split(",")): e.itf.debug_set(kv.split("=")[0], int(kv.split("=")[1]))      # e.g. QM_DEBUG_SET=r_dense=1
e.wbc.reset(); e.mpc.closed_loop_resident(n, 0.01, cfg["horizon"], cfg["period"], cfg["time"]); e.itf.synchronize()
res = e.mpc.download(); out, st = e.wbc.download(B)
np.savez(os.path.join(os.environ.get("QM_DUMP_DIR", "gpurun_out"), "cold_%s.npz" % sys.argv[1]), x=res["x"], u=res["u"], t=res["t"], perf=res["perf"], status=res["status"], wbc=out, wbc_status=st)
e.close()
