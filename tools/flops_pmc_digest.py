"""tools/flops_pmc_digest.py — FP64 flops the kernels ISSUE per launch, from the two SQ instruction-count passes of tools/gpu_round_profile.sh
(rocpd_pmc_summary.py CSVs of pmc_flops_a / pmc_flops_b) -> profiles/flops_pmc.json.

The SQ counters are reported per shader engine (one row per SE and dispatch): a launch's total is the per-dispatch average times the number of SEs,
taken from SQ_WAVES (waves per SE) against the launch's grid.  flops = 512 x MFMA_MOPS_F64 + 64 lanes x (2 x FMA_F64 + MUL_F64 + ADD_F64 + TRANS_F64)
— wave-level instruction counts priced at a full exec mask, i.e. an UPPER bound of the vector part (masked lanes count) and the matrix part includes
tile padding / rank-1 updates: this is the "issued" view that sits next to the algorithmic model of bench.py.
usage: python tools/flops_pmc_digest.py <flops_a.csv> <flops_b.csv> <round tag> [n_se=32] [out.json]"""
import csv, json, os, re, sys


def _stamp():
    """hash of the kernel sources the profiled run was built from: written on the GPU box by tools/gpu_round_profile.sh (gpurun_out/kernel_source_hash.txt)"""
    p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "kernel_source_hash.txt")
    return open(p).read().strip() if os.path.exists(p) else None


def main(a_csv, b_csv, tag, n_se="32", out="profiles/flops_pmc.json"):
    n_se = int(n_se); rows = {}
    for f in (a_csv, b_csv):
        for r in csv.DictReader(open(f)):
            rows.setdefault(r["Kernel"], {})[r["Counter"]] = float(r["AvgPerDispatch"])
    short = lambda k: (re.match(r"_Z\d+(qm_\w+_kernel)", k) or [None, k])[1]
    res = {"round": tag, "kernel_source_hash": _stamp(), "_comment": __doc__.split("usage")[0].strip(), "n_shader_engines": n_se, "kernels": {}}
    for k, c in rows.items():
        if not k.startswith("_Z"):
            continue
        g = lambda n: c.get(n, 0.0) * n_se
        mfma = 512.0 * g("SQ_INSTS_VALU_MFMA_MOPS_F64")
        valu = 64.0 * (2.0 * g("SQ_INSTS_VALU_FMA_F64") + g("SQ_INSTS_VALU_MUL_F64") + g("SQ_INSTS_VALU_ADD_F64") + g("SQ_INSTS_VALU_TRANS_F64"))
        f64_insts = g("SQ_INSTS_VALU_FMA_F64") + g("SQ_INSTS_VALU_MUL_F64") + g("SQ_INSTS_VALU_ADD_F64") + g("SQ_INSTS_VALU_TRANS_F64")
        res["kernels"][short(k)] = {"flops_per_launch": mfma + valu, "mfma_flops": mfma, "valu_f64_flops_full_mask": valu, "mfma_instructions": g("SQ_INSTS_MFMA"),
                                    "valu_instructions": g("SQ_INSTS_VALU"), "valu_f64_instructions": f64_insts, "waves": g("SQ_WAVES"),
                                    "valu_non_f64_share": 1.0 - (f64_insts + g("SQ_INSTS_MFMA")) / max(1.0, g("SQ_INSTS_VALU"))}
    json.dump(res, open(out, "w"), indent=1)
    for k, v in sorted(res["kernels"].items(), key=lambda kv: -kv[1]["flops_per_launch"])[:8]:
        print("%-28s %.2f Gflop issued/launch (MFMA %.2f, VALU %.2f); VALU instr %.3g of which non-FP64 %.0f %%" % (k, v["flops_per_launch"] / 1e9, v["mfma_flops"] / 1e9, v["valu_f64_flops_full_mask"] / 1e9, v["valu_instructions"], 100 * v["valu_non_f64_share"]))


if __name__ == "__main__":
    main(*sys.argv[1:])
