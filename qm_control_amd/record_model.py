"""Byte model of the three big kernels, in f64 counts per unit — the ONE set of figures bench.py, DESIGN.md §4 and
tools/hbm_traffic_digest.py use (unit: one non-event shooting interval for K1b / K3, one instance for the WBC).

Stage record written by K1b and read by K3 (csrc/kernels/qm_dev_common.h, SR_*): only what is structurally non-zero is moved; K3's backward operands travel in
fragment order (SR_FRAG: whole 16 x 16 tiles, the padding of a 30-wide block included).
"""
# ---- K1b qm_lq_kernel ----
LQ_WRITE_DOUBLES = 2665      # rows 0..11 of Ap 360 and Bp 216 (joint rows are e_j + dt Px[j] resp. dt Pu[j]: rebuilt by K3), [Qp | qp] [Pp | rp] Rp in fragment order 768 + 512 + 256 (SR_FRAG; m <= 16), the 12 non-zero rows of Px 360, vectors 108+30, swing blocks 24+2 (Pu is implied by the contact mode)
LQ_READ_DOUBLES = 493        # kin record (377 of the 384-double record: round 5 dropped the stage-2 state, the joint rows of the flow values and the stage-2 arm block; rounds 1-4: 504) + x, u, references, node descriptors
# ---- K3 qm_riccati_kernel ----
RICCATI_BWD_READ_DOUBLES = 2576    # rows 0..11 of Ap 360 and Bp 216, the fragment-order operands 1536 (three upper tiles of [Qp | qp], tile row 0 of [Pp | rp], tile (0,0) of Rp; + 448 when m > 16), rows 12..23 of Px 360, bp qp rp 78, swing blocks + mode + dt 26
RICCATI_BWD_WRITE_DOUBLES = 558    # the gain K = -L^-T W (540) and the offset k = -L^-T y (18), formed on the matrix core for the forward rollout
RICCATI_FWD_READ_DOUBLES = 1628    # rows 0..11 of Ap (360) and Bp (216), K 540, rows 12..23 of Px 360, bp qp rp Pe 108, k 18, swing blocks + mode + dt 26
RICCATI_FWD_IO_DOUBLES = 120       # x (2 nodes' worth of defect reads at events amortised), dx 30 + du 30 written, x0
# ---- K6/K7 qm_wbc_kernel ----
WBC_IO_DOUBLES = 900         # inputs 30 + 30 + 55, outputs 54, tip / Jacobian scratch ~ 730


def summary():
    lq = LQ_WRITE_DOUBLES + LQ_READ_DOUBLES
    ric = RICCATI_BWD_READ_DOUBLES + RICCATI_BWD_WRITE_DOUBLES + RICCATI_FWD_READ_DOUBLES + RICCATI_FWD_IO_DOUBLES
    return {"lq_doubles_per_interval": lq, "riccati_doubles_per_interval": ric, "wbc_doubles_per_instance": WBC_IO_DOUBLES,
            "lq_bytes_per_interval": 8 * lq, "riccati_bytes_per_interval": 8 * ric}


def kernel_source_hash():
    """sha256 (first 16 hex digits) over the device + launch sources a counter profile depends on: csrc/kernels/*.h, the launch sequences csrc/host/*_pipeline.h and
    include/qmhip_layout.h (file names included, sorted) — not the C ABI glue or header comments, which cannot move a kernel's counters.
    tools/gpu_round_profile.sh records it next to the PMC passes, the digests stamp it into profiles/flops_pmc.json / hbm_traffic.json, and bench.py labels a roofline
    priced on counters of OTHER sources `stale`."""
    import hashlib
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for d, keep in (("qm_control_amd/csrc/kernels", lambda f: f.endswith(".h")), ("qm_control_amd/csrc/host", lambda f: f.endswith("pipeline.h")), ("include", lambda f: f == "qmhip_layout.h")):
        full = os.path.join(root, d)
        for f in sorted(os.listdir(full)):
            if keep(f):
                h.update((d + "/" + f + "\n").encode())
                with open(os.path.join(full, f), "rb") as fh:
                    h.update(fh.read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(kernel_source_hash())
