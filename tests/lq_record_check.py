"""tests/lq_record_check.py — TEST INFRASTRUCTURE: per-entry parity of what K1a/K1b leave in HBM for one shooting interval — the unprojected LQ model
(debug record) and the projected stage record K3 consumes — against the oracle's NodeLQ of the same interval (<= 1e-10).

The product projects with a CLOSED-FORM null-space basis, the oracle with its own; `du = Pe + Px dx + Pu ut` spans the same affine set, so
the basis-dependent blocks are compared after the change of basis T = pinv(Pu_oracle) Pu_product (checked to reproduce Pu_product exactly):
    Bp = Bp_o T,  Pp = Tᵀ Pp_o,  Rp = Tᵀ Rp_o T,  rp = Tᵀ rp_o;   Px, Pe, Ap, bp, Qp, qp are basis independent."""
import os
import re
import numpy as np

_HDR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "qm_control_amd", "csrc", "kernels")


def _defines(path, prefix):
    out = {}
    for line in open(path):
        m = re.match(r"#define\s+(%s\w+)\s+(.+?)\s*(/\*.*)?$" % prefix, line)
        if m:
            try:
                out[m.group(1)] = int(eval(m.group(2), {}, dict(out)))
            except Exception:
                pass
    return out


SR = _defines(os.path.join(_HDR, "qm_dev_common.h"), "SR_")
DBG = _defines(os.path.join(_HDR, "k_lq.h"), "LQ_DBG_")


def _err(a, b, floor=1e-3):
    a = np.asarray(a, float); b = np.asarray(b, float)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.abs(a - b).max() / max(float(np.abs(b).max()), floor))


def frag_tile(rec, off, regs=4):
    """16 x 16 tile stored in fragment order at rec[off:]: register r, lane l = 16 g + c holds entry (g + 4 r, c); `regs` < 4: only the first registers exist (rest zero)"""
    T = np.zeros((16, 16))
    for r in range(regs):
        T[np.arange(4)[:, None] + 4 * r, np.arange(16)[None, :]] = np.asarray(rec[off + 64 * r:off + 64 * r + 64]).reshape(4, 16)
    return T


def frag_operands(rec, m):
    """the fragment-order region of a stage record (SR_FRAG): [Qp | qp] (32 x 32, lower-left tile mirrored), [Pp | rp] (m x 32), Rp (m x m, lower-left mirrored)"""
    F = SR["SR_FRAG"]
    Q = np.zeros((32, 32)); Q[:16, :16] = frag_tile(rec, F + SR["SR_F_QP"]); Q[:16, 16:] = frag_tile(rec, F + SR["SR_F_QP"] + 256); Q[16:, 16:] = frag_tile(rec, F + SR["SR_F_QP"] + 512)
    Q[16:, :16] = Q[:16, 16:].T
    P = np.zeros((32, 32)); P[:16, :16] = frag_tile(rec, F + SR["SR_F_PP"]); P[:16, 16:] = frag_tile(rec, F + SR["SR_F_PP"] + 256)
    R = np.zeros((32, 32)); R[:16, :16] = frag_tile(rec, F + SR["SR_F_RP"])
    if m > 16:
        P[16:, :16] = frag_tile(rec, F + SR["SR_F_PP1"], 1); P[16:, 16:] = frag_tile(rec, F + SR["SR_F_PP1"] + 64, 1)
        R[:16, 16:] = frag_tile(rec, F + SR["SR_F_RP01"]); R[16:, 16:] = frag_tile(rec, F + SR["SR_F_RP11"], 1); R[16:, :16] = R[:16, 16:].T
    return Q, P, R


def check_interval(rec, dbg, lq, pr, after_riccati=False):
    """rec: stage record as K1b wrote it (SR_SIZE doubles); dbg: debug record; lq / pr: oracle.node_lq(i) / node_proj(i).
    after_riccati: the record has been through K3, which replaced Pp by the feedback gain K and left the feed-forward k in SR_KFF (ut = K dx + k): those are then
    compared instead, K = T⁻¹ K_o, k = T⁻¹ k_o.  Returns {block name: relative error on the block's own scale}."""
    e = {}
    nc = lq["nc"]; m = pr["m"]
    assert int(dbg[DBG["LQ_DBG_nc"]]) == nc and int(rec[SR["SR_SCAL"]]) == m and m == 30 - nc
    g = lambda off, *shape: np.asarray(dbg[off:off + int(np.prod(shape))]).reshape(shape)
    # ---- unprojected: discrete dynamics (RK2 sensitivities), cost model, equality rows ----
    e["A_d"] = _err(g(DBG["LQ_DBG_A"], 30, 30), lq["A"]); e["B_d"] = _err(g(DBG["LQ_DBG_B"], 30, 30), lq["B"]); e["b_d"] = _err(g(DBG["LQ_DBG_b"], 30), lq["b"])
    e["Q"] = _err(g(DBG["LQ_DBG_Q"], 30, 30), lq["Q"]); e["R"] = _err(g(DBG["LQ_DBG_R"], 30, 30), lq["R"]); e["q"] = _err(g(DBG["LQ_DBG_q"], 30), lq["q"]); e["r"] = _err(g(DBG["LQ_DBG_r"], 30), lq["r"])
    e["C"] = _err(g(DBG["LQ_DBG_C"], 16, 30)[:nc], lq["C"][:nc]); e["D"] = _err(g(DBG["LQ_DBG_D"], 16, 30)[:nc], lq["D"][:nc]); e["e"] = _err(g(DBG["LQ_DBG_e"], 16)[:nc], lq["e"][:nc])
    # ---- projected stage record ----
    s = lambda off, *shape: np.asarray(rec[off:off + int(np.prod(shape))]).reshape(shape)
    Px = np.zeros((30, 30)); Px[12:24] = s(SR["SR_PX"] + 360, 12, 30)           # only rows 12..23 (leg joint velocities) exist in the record (SR_PX is their virtual base); the others are zero by construction
    Pu = g(DBG["LQ_DBG_PU"], 30, 18)[:, :m]; Pe = s(SR["SR_PE"], 30)               # Pu: debug record only (K3 rebuilds Pu ut from the mode and the swing blocks)
    Pu_o = pr["Pu"][:, :m]
    T = np.linalg.pinv(Pu_o) @ Pu
    e["Px"] = _err(Px, pr["Px"]); e["Pe"] = _err(Pe, pr["Pe"]); e["range(Pu)"] = _err(Pu_o @ T, Pu)
    assert np.linalg.cond(T) < 1e6
    # the projector really is one: D Pu = 0, D Px = -C, D Pe = -e (rows of the equality constraint)
    Dm = lq["D"][:nc]; Cm = lq["C"][:nc]
    e["D Pu"] = float(np.abs(Dm @ Pu).max() / max(1.0, np.abs(Dm).max())); e["D Px + C"] = float(np.abs(Dm @ Px + Cm).max() / max(1.0, np.abs(Cm).max())); e["D Pe + e"] = float(np.abs(Dm @ Pe + lq["e"][:nc]).max())
    # rows 0..11 of Ap / Bp are stored; joint rows are rebuilt by K3 as e_j + dt Px[12 + j] and dt Pu[12 + j]
    dt = rec[SR["SR_MODEF"] + 1]
    Ap = np.zeros((30, 30)); Bp = np.zeros((30, m)); Ap[:12] = s(SR["SR_AP"], 12, 30); Bp[:12] = s(SR["SR_BP"], 12, 18)[:, :m]
    Ap[12:] = np.eye(30)[12:] + dt * Px[12:]; Bp[12:] = dt * Pu[12:]
    e["Ap"] = _err(Ap, pr["Ap"]); e["Bp"] = _err(Bp, pr["Bp"][:, :m] @ T); e["bp"] = _err(s(SR["SR_BPV"], 30), pr["bp"])
    # [Qp | qp], [Pp | rp], Rp: the fragment-order region K3's backward sweep reads (upper tiles of the symmetric blocks; vectors in column 30), which K3 leaves untouched
    Qf, Pf, Rf = frag_operands(rec, m)
    e["Qp"] = _err(Qf[:30, :30], pr["Qp"]); e["qp"] = _err(Qf[:30, 30], pr["qp"]); e["qp (vector copy)"] = _err(s(SR["SR_QPV"], 30), pr["qp"])
    Rp = Rf[:m, :m]
    e["Rp"] = _err(Rp if m > 16 else np.triu(Rp) + np.triu(Rp, 1).T, T.T @ pr["Rp"][:m, :m] @ T)
    e["rp"] = _err(Pf[:m, 30], T.T @ pr["rp"][:m]); e["rp (vector copy)"] = _err(s(SR["SR_RPV"], 18)[:m], T.T @ pr["rp"][:m])
    e["Pp"] = _err(Pf[:m, :30], T.T @ pr["Pp"][:m])
    # what K3 relies on in the padding: rows >= m of [Pp | rp] and everything outside [0, m) x [0, m) of Rp are exactly zero
    e["padding"] = float(max(np.abs(Pf[m:]).max(initial=0.0), np.abs(Rf[m:]).max(initial=0.0), np.abs(Rf[:, m:]).max(initial=0.0)))
    if after_riccati:
        Ti = np.linalg.inv(T)
        e["K"] = _err(s(SR["SR_PP"], 18, 30)[:m], Ti @ pr["K"][:m]); e["k"] = _err(s(SR["SR_KFF"], 18)[:m], Ti @ pr["kff"][:m])
    return e
