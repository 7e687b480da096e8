"""tools/context_footprint.py — MEASURED device memory of a qmhip context: free bytes (hipMemGetInfo through torch.cuda.mem_get_info, device-wide, so the library's own
allocations count) before and after creating the interface + MPC + WBC contexts at the sizes include/qmhip.h and DESIGN.md §3 quote.  Writes gpurun_out/context_footprint.json."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from qm_control_amd import api, scenarios, record_model

out = {"kernel_source_hash": record_model.kernel_source_hash(), "cases": []}
torch.cuda.init(); torch.zeros(1, device="cuda:0"); torch.cuda.synchronize()
for B, nodes in ((1, 128), (1024, 128), (512, 192), (8192, 116), (8192, 128)):
    free0 = torch.cuda.mem_get_info(0)[0]
    itf = api.QMInterface(blobs=scenarios.load_blobs(), device=0, max_batch=B, max_nodes=nodes, max_ref_knots=2, max_events=8)
    mpc = api.SqpMpc(itf); wbc = api.HierarchicalWbc(itf); itf.synchronize()
    used = free0 - torch.cuda.mem_get_info(0)[0]
    out["cases"].append({"max_batch": B, "max_nodes": nodes, "device_bytes": int(used), "GB": round(used / 1e9, 3), "KB_per_instance_node": round(used / (B * nodes) / 1e3, 2)})
    print(out["cases"][-1], flush=True)
    del mpc, wbc; itf.close(); torch.cuda.synchronize()
os.makedirs("gpurun_out", exist_ok=True); json.dump(out, open("gpurun_out/context_footprint.json", "w"), indent=1)
