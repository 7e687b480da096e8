"""tools/hbm_traffic_digest.py — fold the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE summaries written by rocpd_pmc_summary.py) into
profiles/hbm_traffic.json, the per-kernel HBM bytes per launch that bench.py reports as roofline.traffic.
usage: python tools/hbm_traffic_digest.py <fetch.csv> <write.csv> <round tag> [out.json]"""
import csv, json, os, re, sys


def _stamp():
    """hash of the kernel sources the profiled run was built from: written on the GPU box by tools/gpu_round_profile.sh (gpurun_out/kernel_source_hash.txt)"""
    p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "kernel_source_hash.txt")
    return open(p).read().strip() if os.path.exists(p) else None

def main(fetch_csv, write_csv, tag="?", out="profiles/hbm_traffic.json"):
    sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
    from qm_control_amd import record_model as rm
    f = {r["Kernel"]: float(r["AvgPerDispatch"]) for r in csv.DictReader(open(fetch_csv))}
    w = {r["Kernel"]: float(r["AvgPerDispatch"]) for r in csv.DictReader(open(write_csv))}
    short = lambda k: (re.match(r"_Z\d+(qm_\w+_kernel)", k) or [None, k])[1]
    res = {"_comment": "HBM bytes per launch from two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; units KB) of `bench.py --steps 3 --warmup 1` at B=1024 on one MI355X "
                       "(tools/gpu_round_profile.sh). FETCH_SIZE is doubled as MI355X_MICROARCH.md (HBM section) prescribes for gfx950 (the counter tallies 128-B requests at 64 B); "
                       "WRITE_SIZE is taken as reported. Cross-check against the byte model of qm_control_amd/record_model.py (the figures bench.py and DESIGN.md §4 use): %s "
                       "(~105.7k non-event intervals per launch at B = 1024, N = 100)." % json.dumps(rm.summary()),
           "round": tag, "kernel_source_hash": _stamp(), "source": [fetch_csv, write_csv], "kernels": {}}
    for k in f:
        if k.startswith("_Z"):
            res["kernels"][short(k)] = {"fetch_size_kb_raw": f[k], "write_size_kb": w.get(k, 0.0), "fetch_bytes_corrected": 2 * 1024 * f[k], "write_bytes": 1024 * w.get(k, 0.0),
                                         "traffic_bytes": 2 * 1024 * f[k] + 1024 * w.get(k, 0.0)}
    json.dump(res, open(out, "w"), indent=1)
    for k, v in res["kernels"].items(): print("%-28s %.3f GB" % (k, v["traffic_bytes"] / 1e9))

if __name__ == "__main__":
    main(*sys.argv[1:])
