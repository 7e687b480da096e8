"""tools/c4_node_capacity.py — BASELINE config 4 (8192 instances, N = 100) on ONE device at a node capacity of 116 instead of bench.py's generic N + 28:
runs two steps, reports the largest node count any instance used, the status histogram and the measured device footprint.  Writes gpurun_out/c4_node_capacity.json."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
import bench
from qm_control_amd import scenarios
out = {}
for cap in (116, 128):
    cfg = scenarios.make_config("C4", batch=8192, n_intervals=100)
    e = bench.HipEngine(cfg, 0, max_nodes=cap)
    e.step(); e.step(); e.sync(); res = e.mpc.download()
    st = np.asarray(res["status"]); nn = np.asarray(res["num_nodes"])
    out[str(cap)] = {"max_nodes": cap, "device_gb_measured": round(e.device_bytes / 1e9, 3), "status_histogram": {int(k): int(v) for k, v in zip(*np.unique(st, return_counts=True))},
                     "n_nodes_max": None if nn is None else int(nn.max()), "n_nodes_min": None if nn is None else int(nn.min())}
    print(out[str(cap)], flush=True); e.close(); torch.cuda.synchronize()
json.dump(out, open("gpurun_out/c4_node_capacity.json", "w"), indent=1)
