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


def check_interval(rec, dbg, lq, pr, after_riccati=False):
    """rec: stage record as K1b wrote it (SR_SIZE doubles, debug mode: SR_PU and the zero rows of Px present); dbg: debug record; lq / pr: oracle.node_lq(i) / node_proj(i).
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
    Px = s(SR["SR_PX"], 30, 30); Pu = s(SR["SR_PU"], 30, 18)[:, :m]; Pe = s(SR["SR_PE"], 30)
    Pu_o = pr["Pu"][:, :m]
    T = np.linalg.pinv(Pu_o) @ Pu
    e["Px"] = _err(Px, pr["Px"]); e["Pe"] = _err(Pe, pr["Pe"]); e["range(Pu)"] = _err(Pu_o @ T, Pu)
    assert np.linalg.cond(T) < 1e6
    # the projector really is one: D Pu = 0, D Px = -C, D Pe = -e (rows of the equality constraint)
    Dm = lq["D"][:nc]; Cm = lq["C"][:nc]
    e["D Pu"] = float(np.abs(Dm @ Pu).max() / max(1.0, np.abs(Dm).max())); e["D Px + C"] = float(np.abs(Dm @ Px + Cm).max() / max(1.0, np.abs(Cm).max())); e["D Pe + e"] = float(np.abs(Dm @ Pe + lq["e"][:nc]).max())
    # rows 0..11 of Ap / Bp are stored; joint rows are rebuilt by K3 as e_j + dt Px[12 + j] and dt Pu[12 + j]
    dt = rec[SR["SR_MODEF"] + 1]
    Ap = s(SR["SR_AP"], 30, 30).copy(); Bp = s(SR["SR_BP"], 30, 18)[:, :m].copy()
    Ap[12:] = np.eye(30)[12:] + dt * Px[12:]; Bp[12:] = dt * Pu[12:]
    e["Ap"] = _err(Ap, pr["Ap"]); e["Bp"] = _err(Bp, pr["Bp"][:, :m] @ T); e["bp"] = _err(s(SR["SR_BPV"], 30), pr["bp"])
    # symmetric blocks: the upper 16 x 16 tiles are stored
    def sym(M):
        M = M.copy(); n = M.shape[0]
        for i in range(16, n):
            M[i, :16] = M[:16, i]
        return M
    e["Qp"] = _err(sym(s(SR["SR_QP"], 30, 30)), pr["Qp"]); e["qp"] = _err(s(SR["SR_QPV"], 30), pr["qp"])
    Rp = s(SR["SR_RP"], 18, 18)[:m, :m]
    e["Rp"] = _err(sym(Rp) if m > 16 else np.triu(Rp) + np.triu(Rp, 1).T, T.T @ pr["Rp"][:m, :m] @ T)
    e["rp"] = _err(s(SR["SR_RPV"], 18)[:m], T.T @ pr["rp"][:m])
    if after_riccati:
        Ti = np.linalg.inv(T)
        e["K"] = _err(s(SR["SR_PP"], 18, 30)[:m], Ti @ pr["K"][:m]); e["k"] = _err(s(SR["SR_KFF"], 18)[:m], Ti @ pr["kff"][:m])
    else:
        e["Pp"] = _err(s(SR["SR_PP"], 18, 30)[:m], T.T @ pr["Pp"][:m])
    return e
