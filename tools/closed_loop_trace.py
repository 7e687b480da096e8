"""tools/closed_loop_trace.py <tag> [seconds] — the README experiment's device loop (tools/readme_experiment_gpu.py) with the plant's rigid-body state recorded after EVERY 20-tick chunk,
written bit-exactly to gpurun_out/cl_trace_<tag>.npy: two builds (QM_AB_LIB) or two runs of one build are compared chunk by chunk to find where — if anywhere — they part."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from qm_control_amd import api, scenarios
import tools.readme_experiment_gpu as R
if os.environ.get("QM_AB_LIB"): api.LIB_PATH = os.path.join(ROOT, os.environ["QM_AB_LIB"])       # another build on the same box

tag = sys.argv[1]; secs = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
trace = []
orig = api.QMInterface.debug_read
def spy(self, name, shape, dtype=np.float64):
    a = orig(self, name, shape, dtype)
    if name == "sim_rbd": trace.append(np.array(a, copy=True))
    return a
api.QMInterface.debug_read = spy
r = R.run(scenarios.load_blobs(), -0.1, 0.0, 4, secs)
os.makedirs("gpurun_out", exist_ok=True); np.save("gpurun_out/cl_trace_%s.npy" % tag, np.stack(trace))
print(tag, "lib", os.path.relpath(api.LIB_PATH, ROOT), "chunks", len(trace), "ee_dev_max_mm", r["ee_dev_max_mm"][0], "deg", r["ee_dev_max_deg"][0], "travel", r["base_travel_m"][0], "ok", r["all_status_ok"])
