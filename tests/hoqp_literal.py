"""tests/hoqp_literal.py — TEST INFRASTRUCTURE: the hierarchical QP cascade of the reference built LITERALLY and solved by a
generic method, as an independent check of the oracle's / the product's inequality-constrained least-squares reformulation.

Every level forms h_, c_, d_, f_ exactly as qm_wbc/src/HoQp.cpp:57-124 does (decision vector [z; w] with the slack w PRESENT,
`zᵀ(AZ)ᵀ(AZ)z + 1e-12 I`, rows stacked [−w <= 0; D_prev Z z <= f_prev − D_prev x_prev + w_prev*; D Z z − w <= f − D x_prev]) and solves

        min ½ yᵀ h y + cᵀ y   s.t.  d y <= f

with a textbook primal-dual interior-point method (a guess of the active set) followed by dense KKT solves on the active set whose
result is CERTIFIED by the KKT conditions of that convex QP (stationarity, primal feasibility, multiplier signs).  The literal Hessian
has eigenvalues 1e-12 next to 1e3: in f64 the directions only the regulariser sees come out as round-off / 1e-12 = O(1) noise (the
reference lives with that: the next level re-optimises exactly those directions), which makes an f64 active-set loop wander; the KKT
systems are therefore formed and solved in 80-bit extended precision (own LU with partial pivoting, numpy has no longdouble LAPACK).  Nothing here eliminates the
slack, stacks `sqrt(rho) I` rows, or uses a QR / null-space active-set step: it shares no reformulation and no code with
oracle/src/wbc.h:solveHoLevel or with csrc/kernels/k_wbc.h.  The null-space basis (HoQp.cpp:126-133 uses Eigen's
FullPivLU::kernel()) is scipy's SVD basis — x is basis independent.
"""
import warnings
import numpy as np
import scipy.linalg


def _ipm(H, c, D, f, iters=60):
    """Mehrotra predictor-corrector for min ½yᵀHy + cᵀy, Dy + s = f, s >= 0 (H positive semi-definite). Returns y, lam, s."""
    n, m = H.shape[0], D.shape[0]
    y = np.zeros(n)
    s = np.maximum(f - D @ y, 1.0); lam = np.ones(m)
    with np.errstate(all="ignore"), warnings.catch_warnings():
      warnings.simplefilter("ignore")                  # a singular barrier system ends the guess phase, nothing else
      for _ in range(iters):
          rd = H @ y + c + D.T @ lam; rp = D @ y + s - f; mu = float(s @ lam) / m
          if max(np.abs(rd).max(), np.abs(rp).max()) < 1e-11 * (1.0 + np.abs(f).max()) and mu < 1e-13:
              break
          W = lam / s
          K = H + D.T @ (W[:, None] * D) + 1e-9 * np.eye(n)     # (the guess generator may regularise; the certified solve below does not)
          if not np.isfinite(K).all():
              break                                     # the polish below does the exact work from the last finite iterate
          cf = scipy.linalg.lu_factor(K, check_finite=False)

          def solve(r3):
              # [H Dᵀ 0; D 0 I; 0 S L] [dy dl ds] = [−rd, −rp, r3]
              rhs = -rd - D.T @ ((r3 + lam * rp) / s) + 0.0
              dy = scipy.linalg.lu_solve(cf, rhs, check_finite=False)
              ds = -rp - D @ dy
              dl = (r3 - lam * ds) / s
              return dy, dl, ds

          dy, dl, ds = solve(-s * lam)
          if not (np.isfinite(dy).all() and np.isfinite(dl).all() and np.isfinite(ds).all()):
              break

          def maxstep(v, dv):
              neg = dv < 0
              return min(1.0, float((-v[neg] / dv[neg]).min())) if neg.any() else 1.0

          ap, ad = maxstep(s, ds), maxstep(lam, dl)
          mu_aff = float((s + ap * ds) @ (lam + ad * dl)) / m
          sigma = (mu_aff / mu) ** 3 if mu > 0 else 0.0
          dy, dl, ds = solve(-s * lam - ds * dl + sigma * mu)
          ap, ad = 0.995 * maxstep(s, ds), 0.995 * maxstep(lam, dl)
          if not (np.isfinite(dy).all() and np.isfinite(dl).all() and np.isfinite(ds).all()):
              break
          y = y + ap * dy; s = np.maximum(s + ap * ds, 1e-300); lam = np.maximum(lam + ad * dl, 1e-300)
    return y, lam, s


LD = np.longdouble


def _lu_solve_ld(K, rhs):
    """Gaussian elimination with partial pivoting in extended precision; K (n x n) and rhs are overwritten"""
    n = K.shape[0]
    for k in range(n):
        p = k + int(np.argmax(np.abs(K[k:, k])))
        if K[p, k] == 0:
            raise np.linalg.LinAlgError("singular KKT matrix (dependent active rows)")
        if p != k:
            K[[k, p]] = K[[p, k]]; rhs[[k, p]] = rhs[[p, k]]
        if k + 1 < n:
            l = K[k + 1:, k] / K[k, k]
            K[k + 1:, k + 1:] -= np.outer(l, K[k, k + 1:]); rhs[k + 1:] -= l * rhs[k]
    x = np.zeros(n, LD)
    for k in range(n - 1, -1, -1):
        x[k] = (rhs[k] - K[k, k + 1:] @ x[k + 1:]) / K[k, k]
    return x


def _kkt_on_set(H, c, D, f, act):
    """equality-constrained QP on the rows `act` (extended precision): returns y, multipliers of those rows"""
    n = H.shape[0]; Da = D[act]; k = Da.shape[0]
    K = np.zeros((n + k, n + k), LD); K[:n, :n] = H; K[:n, n:] = Da.T; K[n:, :n] = Da
    sol = _lu_solve_ld(K, np.concatenate([-c, f[act]]).astype(LD))
    return sol[:n], sol[n:]


def solve_qp_certified(H, c, D, f):
    """generic convex QP solve; the returned point satisfies the KKT conditions (asserted)."""
    n, m = H.shape[0], D.shape[0]
    if m == 0:
        return _lu_solve_ld(np.array(H, LD), -np.array(c, LD)), np.zeros(0, LD)
    # presolve: vacuous rows 0·y <= f_i with f_i >= 0 (the unpopulated friction rows of WbcBase.cpp:329-331) have no interior and a zero multiplier
    vac = (np.abs(D).max(axis=1) == 0.0)
    assert (f[vac] >= 0.0).all(), "infeasible vacuous row"
    keep = np.nonzero(~vac)[0]
    if keep.size < m:
        yk, lk = solve_qp_certified(H, c, D[keep], f[keep])
        lam_full = np.zeros(m); lam_full[keep] = lk
        return yk, lam_full
    scale = 1.0 + float(np.abs(f).max())
    y, lam, s = _ipm(np.asarray(H, float), np.asarray(c, float), np.asarray(D, float), np.asarray(f, float) + 1e-7 * scale)   # relaxed: a strict interior for the guess only
    act = lam > s                                     # strict complementarity guess
    for _ in range(100):
        idx = np.nonzero(act)[0]
        try:
            yk, lk = _kkt_on_set(H, c, D, f, idx)
        except np.linalg.LinAlgError:                 # dependent rows in the guess: drop the one with the smallest IPM multiplier
            act[idx[int(np.argmin(lam[idx]))]] = False; continue
        viol = D @ yk - f; viol[idx] = 0.0
        worst_p = int(np.argmax(viol)); lscale = 1.0 + (float(np.abs(lk).max()) if lk.size else 0.0)
        if viol[worst_p] > 1e-11 * scale:
            act[worst_p] = True; continue
        if lk.size and lk.min() < -1e-11 * lscale:
            act[idx[int(np.argmin(lk))]] = False; continue
        # KKT certificate of the LITERAL problem: stationarity residual, primal feasibility, dual feasibility
        lam_full = np.zeros(m, LD); lam_full[idx] = lk
        stat = np.abs(H @ yk + c + D.T @ lam_full).max()
        assert stat <= 1e-12 * (1.0 + float(np.abs(c).max())), stat
        return yk, lam_full
    raise AssertionError("active-set polish did not converge")


def hoqp_literal(tasks, tau=None, eps_reg=0.0):
    """tasks: list of dict(A, b, D, f) from the highest priority down.  Returns x of the last level and per-level records.
    tau / eps_reg (tools/qpoases_termination_study.py): a model of how qpOASES' stopping rule and Hessian regularisation can move a level's solution — the level QP is
    solved for the homotopy data of parameter tau < 1 (gradient tau c; bounds that are violated at the cold start y = 0 only a fraction tau of the way in) and with
    eps_reg I added to the Hessian; None / 0: the exact problem."""
    nx = tasks[0]["A"].shape[1]
    Zp = np.eye(nx, dtype=LD); xp = np.zeros(nx, LD); Dst = np.zeros((0, nx), LD); fst = np.zeros(0, LD); wst = np.zeros(0, LD)
    levels = []
    for t in tasks:
        A, b, Dc, fc = (np.asarray(t[k], LD) for k in ("A", "b", "D", "f"))
        nz, ns, nps = Zp.shape[1], Dc.shape[0], Dst.shape[0]
        has_eq = A.shape[0] > 0
        # HoQp::buildHMatrix / buildCVector (HoQp.cpp:57-91)
        if has_eq:
            AZ = A @ Zp
            zTaTaz = AZ.T @ AZ + LD(1e-12) * np.eye(nz, dtype=LD)
            ctop = AZ.T @ (A @ xp - b)
        else:
            zTaTaz = np.zeros((nz, nz), LD); ctop = np.zeros(nz, LD)
        H = np.block([[zTaTaz, np.zeros((nz, ns), LD)], [np.zeros((ns, nz), LD), np.eye(ns, dtype=LD)]])
        c = np.concatenate([ctop, np.zeros(ns, LD)])
        # HoQp::buildDMatrix / buildFVector (HoQp.cpp:93-124)
        Dm = np.block([[np.zeros((ns, nz), LD), -np.eye(ns, dtype=LD)],
                       [Dst @ Zp, np.zeros((nps, ns), LD)],
                       [Dc @ Zp if ns else np.zeros((0, nz), LD), -np.eye(ns, dtype=LD)]])
        fv = np.concatenate([np.zeros(ns, LD), fst - Dst @ xp + wst, fc - Dc @ xp if ns else np.zeros(0, LD)])
        if eps_reg:
            H = H + LD(eps_reg) * np.eye(H.shape[0], dtype=LD)
        if tau is not None:
            c = LD(tau) * c; fv = np.where(fv < 0, LD(tau) * fv, fv)
        y, lam = solve_qp_certified(H, c, Dm, fv)
        z, w = y[:nz], y[nz:]
        x = xp + Zp @ z                                             # HoQp::getSolutions (HoQp.h:30-33)
        Znew = Zp @ scipy.linalg.null_space(np.asarray(A @ Zp, float), rcond=1e-10).astype(LD) if has_eq else Zp      # HoQp::buildZMatrix (HoQp.cpp:126-133)
        levels.append(dict(x=x, z=z, w=w, nz=nz, ns=ns, n_active=int((lam > 0).sum())))
        # stackedTasks_ = task_ + stackedTasksPrev_ (current rows first), slack solutions appended after the previous ones (HoQp.cpp:46,152-158)
        Dst, fst, wst = np.vstack([Dc, Dst]), np.concatenate([fc, fst]), np.concatenate([wst, w])
        Zp, xp = Znew, x
    return np.asarray(xp, float), levels
