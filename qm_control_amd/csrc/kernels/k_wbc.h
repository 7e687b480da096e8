// k_wbc.h — K6/K7: hierarchical whole-body controller, one THREAD per instance (v1).
//
// Restates qm_wbc (WbcBase.cpp:118-563, HierarchicalWbc.cpp:18-44, HierarchicalMpcWbc.cpp:18-34, HoQp.cpp:12-158,
// Task.h:17-66) — SURVEY.md §8 a13–a19:
//   updateMeasured / updateDesired  -> recursive rigid-body passes (qm_dev_rbd.h)
//   13 task formulators             -> rows of A_k x = b_k and the structured inequality block D0 x <= f0
//   HoQp cascade (3 levels)         -> each level is an inequality-constrained least-squares problem
//        min ½|A Zp z + A xp − b|² + ½ rho |z|² + ½|w|²  s.t.  w >= 0, D Zp z − w <= f − D xp,  Dp Zp z <= fp − Dp xp + wp*
//      solved exactly by an active-set method on orthogonal factorisations (stands in for qpOASES, whose return code
//      the reference ignores, HoQp.cpp:143-146; we report qp_status instead).  rho = 1e-12 (HoQp.cpp:66).
//   updateCmd                        -> tau = [M_j, −J_jᵀ] x + h_j
// D0 (torque limits ± and friction pyramids) is never materialised: products D0·x use tau(x) and the 5x3 pyramid.
// Only the shipped hierarchy shapes are supported (own inequality rows only at level 0); anything else -> qp_status −3.
#pragma once
#include "qm_dev_rbd.h"

struct QmWbcArgs {
  const double* mb; const double* st;
  int B;
  const double* x_des; const double* u_des;   // [B][30]
  const double* rbd;                          // [B][55]
  const int* mode;                            // [B]
  const double* time;                         // [B]
  double period; int variant;                 // 0: HierarchicalWbc, 1: HierarchicalMpcWbc
  double* input_last;                         // [B][30] state (WbcBase.cpp:212-213)
  double* out;                                // [B][54]
  int* qp_status;                             // [B][3]
  double* scratch; int sstride;               // lane-interleaved workspace [WBC_SCRATCH][sstride]
  double* dbg;                                // optional [B][WBC_DBG_SIZE]: qMeas vMeas qDes vDes baseAcc nle x0 x1 x2 M J dJv
};
#define WNV 36
#define WMAXA 22        /* max equality-task rows of one level */
#define WMAXACT 20      /* cap on simultaneously active inequality rows */
#define WMAXINEQ 56
#define WG_ROWS (WMAXA + WNV + WMAXACT)
#define WRHO 1e-12
// lane-interleaved HBM workspace per instance (doubles)
#define WS_M     0
#define WS_JF    (WS_M + 576)
#define WS_JARM  (WS_JF + 288)
#define WS_A     (WS_JARM + 144)
#define WS_G0    (WS_A + WMAXA * WNV)
#define WS_GW    (WS_G0 + (WMAXA + WNV) * WNV)
#define WS_AZ    (WS_GW + WG_ROWS * WNV)
#define WS_ZP    (WS_AZ + WMAXA * WNV)
#define WS_ZN    (WS_ZP + WNV * WNV)
#define WS_VQ    (WS_ZN + WNV * WNV)
#define WS_RQ    (WS_VQ + WMAXACT * WNV)
#define WS_EROWS (WS_RQ + WMAXACT * WMAXACT)
#define WBC_SCRATCH (WS_EROWS + WMAXACT * WNV)
#define WBC_DBG_SIZE (24 * 4 + 6 + 24 + 36 * 3 + 576 + 288 + 12)
#define WBC_LDS_BYTES 0
#define WBC_BLOCK 64


// rotation error log(R_l R_rᵀ) [upstream rotationErrorInWorld]
__device__ __forceinline__ void dev_rot_error(const double* Rl, const double* Rr, double* err) {
  double R[9]; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R[3 * i + j] = Rl[3 * i] * Rr[3 * j] + Rl[3 * i + 1] * Rr[3 * j + 1] + Rl[3 * i + 2] * Rr[3 * j + 2];
  const double tr = R[0] + R[4] + R[8]; const double v[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]}; const double tmp = 0.5 * (tr - 3.0); double s;
  if (tmp > -1e-2) s = 0.5 - (tr - 3.0) / 12.0;
  else { double c = 0.5 * (tr - 1.0); c = fmax(-1.0, fmin(1.0, c)); const double th = acos(c); s = th / (2.0 * sin(th)); }
  for (int i = 0; i < 3; ++i) err[i] = s * v[i];
}

// ---- dense helpers on thread-private row-major arrays ----
// Householder least squares: min |G z − g|, G is rows x n (leading dim ld), rows >= n; G and g are overwritten
template <class PG>
__device__ __forceinline__ void dev_ls_qr(PG G, int ld, int rows, int n, double* g, double* z) {
  for (int k = 0; k < n; ++k) {
    double nrm = 0.0; for (int i = k; i < rows; ++i) nrm += G[i * ld + k] * G[i * ld + k]; nrm = sqrt(nrm);
    if (nrm == 0.0) continue;
    const double alpha = G[k * ld + k] > 0.0 ? -nrm : nrm;
    const double vk = G[k * ld + k] - alpha; double vn = vk * vk; for (int i = k + 1; i < rows; ++i) vn += G[i * ld + k] * G[i * ld + k];
    if (vn == 0.0) continue;
    for (int j = k + 1; j <= n; ++j) {   // j == n: the rhs
      double s = vk * ((j < n) ? G[k * ld + j] : g[k]); for (int i = k + 1; i < rows; ++i) s += G[i * ld + k] * ((j < n) ? G[i * ld + j] : g[i]);
      s *= 2.0 / vn;
      if (j < n) { G[k * ld + j] -= s * vk; for (int i = k + 1; i < rows; ++i) G[i * ld + j] -= s * G[i * ld + k]; }
      else { g[k] -= s * vk; for (int i = k + 1; i < rows; ++i) g[i] -= s * G[i * ld + k]; }
    }
    G[k * ld + k] = alpha; for (int i = k + 1; i < rows; ++i) G[i * ld + k] = 0.0;
  }
  for (int i = n - 1; i >= 0; --i) { double s = g[i]; for (int j = i + 1; j < n; ++j) s -= G[i * ld + j] * z[j]; z[i] = s / G[i * ld + i]; }
}
// Householder QR of Eᵀ (n x me) given E (me x n, ld): stores reflectors v_k in V (me x n, v_k[i] for i>=k) with beta_k = 2/|v_k|², R (upper me x me)
template <class PE, class PV, class PR>
__device__ __forceinline__ void dev_qr_Et(PE E, int ld, int me, int n, PV V, double* beta, PR R) {
  // work on W = Eᵀ column by column: column c of W = row c of E
  for (int c = 0; c < me; ++c) for (int i = 0; i < n; ++i) V[c * n + i] = E[c * ld + i];   // V temporarily holds W columns
  for (int k = 0; k < me; ++k) {
    PV wk = V + k * n;
    double nrm = 0.0; for (int i = k; i < n; ++i) nrm += wk[i] * wk[i]; nrm = sqrt(nrm);
    const double alpha = wk[k] > 0.0 ? -nrm : nrm;
    for (int i = 0; i < k; ++i) R[i * me + k] = wk[i];
    R[k * me + k] = alpha;
    wk[k] -= alpha; double vn = 0.0; for (int i = k; i < n; ++i) vn += wk[i] * wk[i];
    beta[k] = (vn > 0.0) ? 2.0 / vn : 0.0;
    for (int i = 0; i < k; ++i) wk[i] = 0.0;
    for (int c = k + 1; c < me; ++c) { PV wc = V + c * n; double s = 0.0; for (int i = k; i < n; ++i) s += wk[i] * wc[i]; s *= beta[k]; for (int i = k; i < n; ++i) wc[i] -= s * wk[i]; }
  }
}
template <class PV>
__device__ __forceinline__ void dev_apply_Qt(PV V, const double* beta, int me, int n, double* x) { for (int k = 0; k < me; ++k) { PV v = V + k * n; double s = 0.0; for (int i = k; i < n; ++i) s += v[i] * x[i]; s *= beta[k]; for (int i = k; i < n; ++i) x[i] -= s * v[i]; } }   // x <- Qᵀ x
template <class PV>
__device__ __forceinline__ void dev_apply_Q(PV V, const double* beta, int me, int n, double* x) { for (int k = me - 1; k >= 0; --k) { PV v = V + k * n; double s = 0.0; for (int i = k; i < n; ++i) s += v[i] * x[i]; s *= beta[k]; for (int i = k; i < n; ++i) x[i] -= s * v[i]; } }   // x <- Q x
// rows of G (rows x n) <- rows · Q   (right multiplication by H_0 H_1 ... H_{me-1})
template <class PV, class PG>
__device__ __forceinline__ void dev_right_Q(PV V, const double* beta, int me, int n, PG G, int ld, int rows) {
  for (int r = 0; r < rows; ++r) { PG g = G + r * ld; for (int k = 0; k < me; ++k) { PV v = V + k * n; double s = 0.0; for (int i = k; i < n; ++i) s += g[i] * v[i]; s *= beta[k]; for (int i = k; i < n; ++i) g[i] -= s * v[i]; } }
}

struct WbcCtx {   // everything the D0 block and the torque map need
  QmSPtr M; QmSPtr Jf; const double* nle; double tauMax[18]; int nc; int contactOf[4]; double mu; int nIneq; bool fl[4];
};
__device__ __forceinline__ void wbc_tau_lin(const WbcCtx& c, const double* x, double* tau) {   // [M_j, −J_jᵀ] x  (no h_j)
  for (int r = 0; r < 18; ++r) { double s = 0.0; for (int k = 0; k < 24; ++k) s += c.M[(6 + r) * 24 + k] * x[k]; for (int k = 0; k < 12; ++k) s -= c.Jf[k * 24 + 6 + r] * x[24 + k]; tau[r] = s; }
}
__device__ __forceinline__ void wbc_d0_apply(const WbcCtx& c, const double* x, double* out) {   // out = D0 x
  double tau[18]; wbc_tau_lin(c, x, tau);
  for (int r = 0; r < 18; ++r) { out[r] = tau[r]; out[18 + r] = -tau[r]; }
  int row = 36;
  for (int j = 0; j < c.nc; ++j) { const double* F = x + 24 + 3 * c.contactOf[j]; out[row] = -F[2]; out[row + 1] = F[0] - c.mu * F[2]; out[row + 2] = -F[0] - c.mu * F[2]; out[row + 3] = F[1] - c.mu * F[2]; out[row + 4] = -F[1] - c.mu * F[2]; row += 5; }
  for (; row < c.nIneq; ++row) out[row] = 0.0;
}
__device__ __forceinline__ void wbc_d0_row(const WbcCtx& c, int i, double* row) {
  for (int k = 0; k < WNV; ++k) row[k] = 0.0;
  if (i < 36) { const int r = (i < 18) ? i : i - 18; const double sg = (i < 18) ? 1.0 : -1.0; for (int k = 0; k < 24; ++k) row[k] = sg * c.M[(6 + r) * 24 + k]; for (int k = 0; k < 12; ++k) row[24 + k] = -sg * c.Jf[k * 24 + 6 + r]; }
  else if (i < 36 + 5 * c.nc) { const int j = (i - 36) / 5, r = (i - 36) - 5 * j; double* F = row + 24 + 3 * c.contactOf[j];
    if (r == 0) F[2] = -1.0; else if (r == 1) { F[0] = 1.0; F[2] = -c.mu; } else if (r == 2) { F[0] = -1.0; F[2] = -c.mu; } else if (r == 3) { F[1] = 1.0; F[2] = -c.mu; } else { F[1] = -1.0; F[2] = -c.mu; } }
}
__device__ __forceinline__ void wbc_d0_f(const WbcCtx& c, double* f) {
  for (int r = 0; r < 18; ++r) { f[r] = c.tauMax[r] - c.nle[6 + r]; f[18 + r] = c.tauMax[r] + c.nle[6 + r]; }
  for (int r = 36; r < c.nIneq; ++r) f[r] = 0.0;
}

// min |G0 z − g0|² (+ equality rows E z = e): returns z (and multipliers lam for the equality rows)
// G0: rows0 x n (ld = WNV) original (not modified); Gw: workspace (WG_ROWS x WNV)
__device__ __forceinline__ void wbc_eq_ls(QmSPtr G0, const double* g0, int rows0, int n, QmSPtr E, const double* e, int me,
                                          QmSPtr Gw, double* gw, QmSPtr V, double* beta, QmSPtr Rr, double* z, double* lam) {
  for (int r = 0; r < rows0; ++r) { for (int k = 0; k < n; ++k) Gw[r * WNV + k] = G0[r * WNV + k]; gw[r] = g0[r]; }
  if (me == 0) { dev_ls_qr(Gw, WNV, rows0, n, gw, z); return; }
  dev_qr_Et(E, WNV, me, n, V, beta, Rr);
  double y[WNV];
  for (int i = 0; i < me; ++i) { double s = e[i]; for (int k = 0; k < i; ++k) s -= Rr[k * me + i] * y[k]; y[i] = s / Rr[i * me + i]; }   // Rᵀ y1 = e
  dev_right_Q(V, beta, me, n, Gw, WNV, rows0);                                   // Gw <- G Q = [G Y | G N]
  for (int r = 0; r < rows0; ++r) { double s = 0.0; for (int k = 0; k < me; ++k) s += Gw[r * WNV + k] * y[k]; gw[r] -= s; }
  if (n - me > 0) dev_ls_qr(Gw + me, WNV, rows0, n - me, gw, y + me);
  for (int k = 0; k < n; ++k) z[k] = y[k];
  dev_apply_Q(V, beta, me, n, z);                                                // z = Q [y1; y2]
  // multipliers: R lam = −(Qᵀ Gᵀ (G z − g))[0:me]
  double w[WNV]; for (int k = 0; k < n; ++k) w[k] = 0.0;
  for (int r = 0; r < rows0; ++r) { double s = -g0[r]; for (int k = 0; k < n; ++k) s += G0[r * WNV + k] * z[k]; for (int k = 0; k < n; ++k) w[k] += G0[r * WNV + k] * s; }
  dev_apply_Qt(V, beta, me, n, w);
  for (int i = me - 1; i >= 0; --i) { double s = -w[i]; for (int j = i + 1; j < me; ++j) s -= Rr[i * me + j] * lam[j]; lam[i] = s / Rr[i * me + i]; }
}

// orthonormal null-space basis of AZ (r x n): Znew (36 x (n − rank)) = Zp (36 x n) · Q[:, rank:], Householder QR with column pivoting of (AZ)ᵀ
__device__ __forceinline__ int wbc_null_space(QmSPtr AZ, int r, int n, QmSPtr Zp, QmSPtr Znew, QmSPtr W /*n x r workspace*/, QmSPtr ZQ /*36 x n workspace*/) {
  for (int i = 0; i < n; ++i) for (int j = 0; j < r; ++j) W[i * r + j] = AZ[j * WNV + i];
  for (int i = 0; i < WNV * n; ++i) ZQ[i] = Zp[i];
  int rank = 0; double maxnorm0 = 0.0; const int steps = (n < r) ? n : r;
  for (int k = 0; k < steps; ++k) {
    int best = k; double bn = -1.0;
    for (int j = k; j < r; ++j) { double s = 0.0; for (int i = k; i < n; ++i) s += W[i * r + j] * W[i * r + j]; if (s > bn) { bn = s; best = j; } }
    if (k == 0) maxnorm0 = sqrt(bn);
    if (sqrt(bn) <= 1e-9 * fmax(1.0, maxnorm0)) break;
    if (best != k) for (int i = 0; i < n; ++i) { const double t = W[i * r + k]; W[i * r + k] = W[i * r + best]; W[i * r + best] = t; }
    const double nrm = sqrt(bn); const double alpha = W[k * r + k] > 0.0 ? -nrm : nrm;
    double v[WNV]; for (int i = 0; i < n; ++i) v[i] = (i >= k) ? W[i * r + k] : 0.0; v[k] -= alpha;
    double vn = 0.0; for (int i = k; i < n; ++i) vn += v[i] * v[i];
    if (vn > 0.0) {
      for (int j = k; j < r; ++j) { double s = 0.0; for (int i = k; i < n; ++i) s += v[i] * W[i * r + j]; s *= 2.0 / vn; for (int i = k; i < n; ++i) W[i * r + j] -= s * v[i]; }
      for (int row = 0; row < WNV; ++row) { double s = 0.0; for (int i = k; i < n; ++i) s += ZQ[row * n + i] * v[i]; s *= 2.0 / vn; for (int i = k; i < n; ++i) ZQ[row * n + i] -= s * v[i]; }
    }
    ++rank;
  }
  const int nn = n - rank;
  for (int row = 0; row < WNV; ++row) for (int j = 0; j < nn; ++j) Znew[row * nn + j] = ZQ[row * n + rank + j];
  return nn;
}

__global__ void qm_wbc_kernel(QmWbcArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.B) return;
  const double* mb = a.mb; const double* st = a.st;
  const double* xDes = a.x_des + (size_t)b * 30; const double* uDes = a.u_des + (size_t)b * 30; const double* rbd = a.rbd + (size_t)b * QM_NRBD;
  const int mode = a.mode[b]; const double time = a.time[b];
  WbcCtx C; C.nc = 0; for (int k = 0; k < 4; ++k) { C.fl[k] = mode_flag(mode, k); if (C.fl[k]) C.contactOf[C.nc++] = k; }
  C.mu = st[ST_WBC_FRIC]; C.nIneq = 36 + 5 * C.nc + 3 * (4 - C.nc);
  for (int l = 0; l < 4; ++l) for (int k = 0; k < 3; ++k) C.tauMax[3 * l + k] = mb[MB_TAUMAX + k]; for (int k = 0; k < 6; ++k) C.tauMax[12 + k] = mb[MB_TAUMAX + 12 + k];
  // ---- updateMeasured (WbcBase.cpp:134-191) ----
  double q[24], v[24];
  for (int i = 0; i < 3; ++i) { q[i] = rbd[3 + i]; q[3 + i] = rbd[i]; v[i] = rbd[27 + i]; }
  { const double sz = sin(q[3]), cz = cos(q[3]), sy = sin(q[4]), cy = cos(q[4]); const double wx = rbd[24], wy = rbd[25], wz = rbd[26]; const double tmp = cz * wx / cy + sz * wy / cy; v[3] = sy * tmp + wz; v[4] = -sz * wx + cz * wy; v[5] = tmp; }
  for (int j = 0; j < 18; ++j) { q[6 + j] = rbd[6 + j]; v[6 + j] = rbd[30 + j]; }
  auto ws = [&](int off) { QmSPtr r; r.p = a.scratch + (size_t)off * a.sstride + b; r.s = a.sstride; return r; };
  QmSPtr M = ws(WS_M), Jf = ws(WS_JF), Jarm = ws(WS_JARM);
  double nle[24]; RbdBase Bm; RbdTip fM[4], aM;
  rbd_tree<QmSPtr, QmSPtr>(mb, q, v, Bm, M, nle, true, fM, &aM, Jf, Jarm, true, nullptr);
  C.M = M; C.Jf = Jf; C.nle = nle;
  // ---- updateDesired (WbcBase.cpp:193-226) ----
  double qd[24], vd[24]; for (int i = 0; i < 24; ++i) qd[i] = xDes[6 + i];
  double baseAcc[6]; RbdTip fD[4], aD;
  {
    double K[KW_SIZE]; kin_base(mb, xDes, K);                 // SRBD: thd, omega, r_w
    double wr[3]; v3_cross(K + KW_OM, K + KW_RW, wr);
    for (int k = 0; k < 3; ++k) { vd[k] = xDes[k] + wr[k]; vd[3 + k] = K[KW_THD + k]; }
    for (int j = 0; j < 18; ++j) vd[6 + j] = uDes[12 + j];
    double* il = a.input_last + (size_t)b * 30; double w2[24];
    for (int k = 0; k < 6; ++k) w2[k] = 0.0;
    for (int j = 0; j < 18; ++j) { w2[6 + j] = (uDes[12 + j] - il[12 + j]) / a.period; }
    for (int k = 0; k < 30; ++k) il[k] = uDes[k];
    RbdBase Bd; RbdSums Sd, Sa;
    double* nullp = nullptr;
    rbd_tree<double*, double*>(mb, qd, vd, Bd, nullp, nullptr, false, fD, &aD, nullp, nullp, false, &Sd);          // Adot·v (bias momentum rate), true COM, desired frame velocities
    RbdBase Ba; rbd_tree<double*, double*>(mb, qd, w2, Ba, nullp, nullptr, false, nullptr, nullptr, nullp, nullp, false, &Sa);   // A_j · jointAccel (full CMM joint columns)
    const double m = mb[MB_ROBOTMASS]; const double com[3] = {Sd.mc[0] / m, Sd.mc[1] / m, Sd.mc[2] / m};
    double rate[6] = {0.0, 0.0, -9.81 * m, 0.0, 0.0, 0.0};
    for (int k = 0; k < 4; ++k) { const double r[3] = {fD[k].p[0] - com[0], fD[k].p[1] - com[1], fD[k].p[2] - com[2]}; double t[3]; v3_cross(r, uDes + 3 * k, t); for (int i = 0; i < 3; ++i) { rate[i] += uDes[3 * k + i]; rate[3 + i] += t[i]; } }
    double cF[3], cH[3]; v3_cross(com, Sd.Fb, cF); v3_cross(com, Sa.hl, cH);
    for (int i = 0; i < 3; ++i) { rate[i] -= Sd.Fb[i] + Sa.hl[i]; rate[3 + i] -= (Sd.NbO[i] - cF[i]) + (Sa.hO[i] - cH[i]); }
    // A_b⁻¹ (SRBD) : thdd = A22inv ra ; lin = rate_lin/m − (skew(r_w) E thdd)
    const double ra[3] = {rate[3], rate[4], rate[5]}; double wdd[3], thdd[3], t[3];
    m3_mulv(K + KW_IINV, ra, wdd); m3_mulv(K + KW_EINV, wdd, thdd); v3_cross(K + KW_RW, wdd, t);
    for (int i = 0; i < 3; ++i) { baseAcc[i] = rate[i] / m - t[i]; baseAcc[3 + i] = thdd[i]; }
  }
  // ---- tasks ----
  const int nc = C.nc;
  QmSPtr A = ws(WS_A); double bb[WMAXA];          // current level's equality task
  double f0[WMAXINEQ]; wbc_d0_f(C, f0);
  double dJv[12]; for (int k = 0; k < 4; ++k) for (int r = 0; r < 3; ++r) dJv[3 * k + r] = fM[k].a[r];
  // solver state
  double x[WNV]; QmSPtr Zp = ws(WS_ZP), Zn = ws(WS_ZN); int nz = WNV;
  for (int i = 0; i < WNV; ++i) { x[i] = 0.0; for (int j = 0; j < WNV; ++j) Zp[i * WNV + j] = (i == j) ? 1.0 : 0.0; }
  double w0[WMAXINEQ]; for (int i = 0; i < WMAXINEQ; ++i) w0[i] = 0.0;
  QmSPtr G0 = ws(WS_G0), Gw = ws(WS_GW), AZ = ws(WS_AZ), Vq = ws(WS_VQ), Rq = ws(WS_RQ), Erows = ws(WS_EROWS);
  double g0[WMAXA + WNV], gw[WG_ROWS], betaq[WMAXACT], erhs[WMAXACT], lam[WMAXACT];
  double xlev[3][WNV];
  int status[3] = {0, 0, 0};
  for (int level = 0; level < 3; ++level) {
    // ---- formulate the level's equality task (rows of A, b) ----
    int ra = 0;
    for (int i = 0; i < WMAXA * WNV; ++i) A[i] = 0.0;
    auto setrow_scale = [&](int row, double s) { for (int k = 0; k < WNV; ++k) A[row * WNV + k] *= s; bb[row] *= s; };
    if (level == 0) {
      for (int r = 0; r < 6; ++r) { for (int k = 0; k < 24; ++k) A[r * WNV + k] = M[r * 24 + k]; for (int k = 0; k < 12; ++k) A[r * WNV + 24 + k] = -Jf[k * 24 + r]; bb[r] = -nle[r]; }   // floating-base EoM
      ra = 6;
      for (int k = 0; k < 4; ++k) if (C.fl[k]) { for (int r = 0; r < 3; ++r) { for (int c2 = 0; c2 < 24; ++c2) A[(ra + r) * WNV + c2] = Jf[(3 * k + r) * 24 + c2]; bb[ra + r] = -dJv[3 * k + r]; } ra += 3; }   // no contact motion
      for (int k = 0; k < 4; ++k) if (!C.fl[k]) { for (int r = 0; r < 3; ++r) { A[(ra + r) * WNV + 24 + 3 * k + r] = 1.0; bb[ra + r] = 0.0; } ra += 3; }   // swing: zero force
    } else if (level == 1) {
      const bool init = (a.variant == 0 && time < 10.0);
      if (init) {   // arm joint nominal tracking
        for (int r = 0; r < 6; ++r) { A[r * WNV + 18 + r] = 1.0; bb[r] = st[ST_KP_ARM_J + r] * (qd[18 + r] - q[18 + r]) + st[ST_KD_ARM_J + r] * (vd[18 + r] - v[18 + r]); }
        ra = 6;
      } else {
        A[2] = 1.0; bb[0] = baseAcc[2] + st[ST_KP_BASE_H] * (qd[2] - q[2]) + st[ST_KD_BASE_H] * (vd[2] - v[2]); ra = 1;     // base height
        {   // base angular
          double wMeas[3], wDes[3]; const double thm[3] = {v[3], v[4], v[5]}, thdv[3] = {vd[3], vd[4], vd[5]}; m3_mulv(Bm.E, thm, wMeas); m3_mulv(Bm.E, thdv, wDes);
          double Rdes[9]; rot_zyx(qd[3], qd[4], qd[5], Rdes); double err[3]; dev_rot_error(Rdes, Bm.R, err);
          // E(theta_meas) thdd_des + Edot(theta_meas, thd_des) thd_des
          double acc[3]; { const double tdd[3] = {baseAcc[3], baseAcc[4], baseAcc[5]}; m3_mulv(Bm.E, tdd, acc); const double z[3] = {0.0, 0.0, 1.0}; double t0[3]; v3_cross(z, wDes, t0);
            const double c1[3] = {Bm.E[1], Bm.E[4], Bm.E[7]}, c2[3] = {Bm.E[2], Bm.E[5], Bm.E[8]}; double t1[3]; v3_cross(c1, c2, t1); for (int i = 0; i < 3; ++i) acc[i] += thdv[0] * t0[i] + thdv[1] * thdv[2] * t1[i]; }
          for (int r = 0; r < 3; ++r) { for (int k = 0; k < 3; ++k) A[(ra + r) * WNV + 3 + k] = Bm.E[3 * r + k]; bb[ra + r] = acc[r] + st[ST_KP_BASE_ANG] * err[r] + st[ST_KD_BASE_ANG] * (wDes[r] - wMeas[r]) - Bm.al[r]; }
          ra += 3;
        }
        if (a.variant == 0) {
          for (int r = 0; r < 3; ++r) { for (int k = 0; k < 24; ++k) A[(ra + r) * WNV + k] = Jarm[r * 24 + k]; bb[ra + r] = st[ST_KP_EE_LIN + r] * (aD.p[r] - aM.p[r]) + st[ST_KD_EE_LIN + r] * (aD.v[r] - aM.v[r]) - aM.a[r]; }   // EE linear
          ra += 3;
          double err[3]; dev_rot_error(aD.R, aM.R, err);
          for (int r = 0; r < 3; ++r) { for (int k = 0; k < 24; ++k) A[(ra + r) * WNV + k] = (k >= 3 && k < 6) ? 0.0 : Jarm[(3 + r) * 24 + k]; bb[ra + r] = st[ST_KP_EE_ANG + r] * err[r] + st[ST_KD_EE_ANG + r] * (-aM.w[r]) - (aM.al[r] - Bm.al[r]); }   // EE angular
          ra += 3;
        } else {
          for (int r = 0; r < 2; ++r) { A[(ra + r) * WNV + r] = 1.0; bb[ra + r] = baseAcc[r] + st[ST_KP_BASE_LIN] * (qd[r] - q[r]) + st[ST_KD_BASE_LIN] * (vd[r] - v[r]); }
          ra += 2;
        }
        for (int k = 0; k < 4; ++k) if (!C.fl[k]) {   // swing legs, x100
          for (int r = 0; r < 3; ++r) { for (int c2 = 0; c2 < 24; ++c2) A[(ra + r) * WNV + c2] = Jf[(3 * k + r) * 24 + c2]; bb[ra + r] = st[ST_KP_SWING] * (fD[k].p[r] - fM[k].p[r]) + st[ST_KD_SWING] * (fD[k].v[r] - fM[k].v[r]) - dJv[3 * k + r]; setrow_scale(ra + r, 100.0); }
          ra += 3;
        }
      }
    } else {
      for (int r = 0; r < 12; ++r) { A[r * WNV + 24 + r] = 1.0; bb[r] = uDes[r]; } ra = 12;      // contact force
      if (a.variant == 0) { for (int r = 0; r < 2; ++r) { A[(ra + r) * WNV + r] = 1.0; bb[ra + r] = baseAcc[r] + st[ST_KP_BASE_LIN] * (qd[r] - q[r]) + st[ST_KD_BASE_LIN] * (vd[r] - v[r]); } ra += 2; }
    }
    // ---- stacked LS rows G0 = [A Zp; sqrt(rho) I], g0 = [b − A xp; 0] ----
    const int n = nz;
    for (int r = 0; r < ra; ++r) { for (int k = 0; k < n; ++k) { double s = 0.0; for (int c2 = 0; c2 < WNV; ++c2) s += A[r * WNV + c2] * Zp[c2 * n + k]; AZ[r * WNV + k] = s; G0[r * WNV + k] = s; } double s = bb[r]; for (int c2 = 0; c2 < WNV; ++c2) s -= A[r * WNV + c2] * x[c2]; g0[r] = s; }
    for (int r = 0; r < n; ++r) { for (int k = 0; k < n; ++k) G0[(ra + r) * WNV + k] = (r == k) ? sqrt(WRHO) : 0.0; g0[ra + r] = 0.0; }
    const int rows0 = ra + n;
    double z[WNV]; for (int k = 0; k < n; ++k) z[k] = 0.0;
    double dx[WNV], Dx[WMAXINEQ];
    if (level == 0) {
      // own (soft) inequality rows: Newton on the active set with exact line search (phi is convex piecewise quadratic)
      double fb[WMAXINEQ]; wbc_d0_apply(C, x, Dx); for (int i = 0; i < C.nIneq; ++i) fb[i] = f0[i] - Dx[i];
      bool act[WMAXINEQ]; for (int i = 0; i < C.nIneq; ++i) act[i] = (0.0 - fb[i] > 0.0);
      int it = 0;
      for (; it < 100; ++it) {
        int na = 0;
        for (int r = 0; r < rows0; ++r) { for (int k = 0; k < n; ++k) Gw[r * WNV + k] = G0[r * WNV + k]; gw[r] = g0[r]; }
        for (int i = 0; i < C.nIneq && na < WMAXACT; ++i) if (act[i]) { double row[WNV]; wbc_d0_row(C, i, row); for (int k = 0; k < n; ++k) { double s = 0.0; for (int c2 = 0; c2 < WNV; ++c2) s += row[c2] * Zp[c2 * n + k]; Gw[(rows0 + na) * WNV + k] = s; } gw[rows0 + na] = fb[i]; ++na; }
        double zn[WNV]; dev_ls_qr(Gw, WNV, rows0 + na, n, gw, zn);
        double p[WNV]; for (int k = 0; k < n; ++k) p[k] = zn[k] - z[k];
        // directional derivative along p: dphi(a) = (G0(z + a p) − g0)·G0 p + sum_{active at a} (d_i(z + a p) − fb_i) d_i p
        double Zz[WNV], Zpv[WNV], Dz[WMAXINEQ], Dp[WMAXINEQ];
        for (int r = 0; r < WNV; ++r) { double s1 = 0.0, s2 = 0.0; for (int k = 0; k < n; ++k) { s1 += Zp[r * n + k] * z[k]; s2 += Zp[r * n + k] * p[k]; } Zz[r] = s1; Zpv[r] = s2; }
        wbc_d0_apply(C, Zz, Dz); wbc_d0_apply(C, Zpv, Dp);
        double c0 = 0.0, c1 = 0.0;   // smooth part: derivative = c0 + a c1
        for (int r = 0; r < rows0; ++r) { double gz = -g0[r], gp = 0.0; for (int k = 0; k < n; ++k) { gz += G0[r * WNV + k] * z[k]; gp += G0[r * WNV + k] * p[k]; } c0 += gz * gp; c1 += gp * gp; }
        auto dphi = [&](double al) { double s = c0 + al * c1; for (int i = 0; i < C.nIneq; ++i) { const double vv = Dz[i] + al * Dp[i] - fb[i]; if (vv > 0.0) s += vv * Dp[i]; } return s; };
        double al = 1.0;
        if (dphi(1.0) > 0.0) { double lo = 0.0, hi = 1.0; for (int bi = 0; bi < 200; ++bi) { const double mid = 0.5 * (lo + hi); if (dphi(mid) > 0.0) hi = mid; else lo = mid; } al = 0.5 * (lo + hi); }
        for (int k = 0; k < n; ++k) z[k] += al * p[k];
        bool same = true;
        for (int i = 0; i < C.nIneq; ++i) { const bool nai = (Dz[i] + al * Dp[i] - fb[i] > 0.0); if (nai != act[i]) same = false; act[i] = nai; }
        if (same && al == 1.0) break;
        double pn = 0.0, zs = 1.0; for (int k = 0; k < n; ++k) { pn = fmax(pn, fabs(al * p[k])); zs = fmax(zs, fabs(z[k])); }
        if (pn <= 1e-12 * zs) break;                      // minimiser sits on a kink: both active sets give the same z
      }
      if (it >= 100) status[0] = 1;
      for (int r = 0; r < WNV; ++r) { double s = 0.0; for (int k = 0; k < n; ++k) s += Zp[r * n + k] * z[k]; dx[r] = s; }
      wbc_d0_apply(C, dx, Dx); for (int i = 0; i < C.nIneq; ++i) w0[i] = fmax(0.0, Dx[i] - fb[i]);
    } else {
      // hard rows of level 0: primal active set (Nocedal & Wright 16.3) from the feasible z = 0
      double fb[WMAXINEQ]; wbc_d0_apply(C, x, Dx); for (int i = 0; i < C.nIneq; ++i) fb[i] = f0[i] - Dx[i] + w0[i];
      int W[WMAXACT]; int nw = 0; int it = 0; bool degenerate = false; double pscale = 0.0;
      for (; it < 100; ++it) {
        for (int q2 = 0; q2 < nw; ++q2) { double row[WNV]; wbc_d0_row(C, W[q2], row); for (int k = 0; k < n; ++k) { double s = 0.0; for (int c2 = 0; c2 < WNV; ++c2) s += row[c2] * Zp[c2 * n + k]; Erows[q2 * WNV + k] = s; } erhs[q2] = fb[W[q2]]; }
        double zn[WNV]; wbc_eq_ls(G0, g0, rows0, n, Erows, erhs, nw, Gw, gw, Vq, betaq, Rq, zn, lam);
        double p[WNV], pn = 0.0, zs = 1.0; for (int k = 0; k < n; ++k) { p[k] = zn[k] - z[k]; pn = fmax(pn, fabs(p[k])); zs = fmax(zs, fabs(z[k])); }
        pscale = fmax(pscale, pn);
        if (pn <= 1e-9 * fmax(zs, pscale)) {             // relative to the largest step seen: the problem's own length scale
          // stationary on the working set: drop a row with a negative multiplier (most negative; lowest index after a degenerate step — Bland)
          int worst = -1; double lw = 0.0, lscale = 1.0; for (int q2 = 0; q2 < nw; ++q2) lscale = fmax(lscale, fabs(lam[q2]));
          for (int q2 = 0; q2 < nw; ++q2) if (lam[q2] < -1e-9 * lscale) { if (degenerate) { if (worst < 0 || W[q2] < W[worst]) worst = q2; } else if (lam[q2] < lw) { lw = lam[q2]; worst = q2; } }
          if (worst < 0) break;
          for (int q2 = worst; q2 < nw - 1; ++q2) W[q2] = W[q2 + 1]; --nw;
        } else {
          double Zz[WNV], Zpv[WNV], Dz[WMAXINEQ], Dp[WMAXINEQ];
          for (int r = 0; r < WNV; ++r) { double s1 = 0.0, s2 = 0.0; for (int k = 0; k < n; ++k) { s1 += Zp[r * n + k] * z[k]; s2 += Zp[r * n + k] * p[k]; } Zz[r] = s1; Zpv[r] = s2; }
          wbc_d0_apply(C, Zz, Dz); wbc_d0_apply(C, Zpv, Dp);
          double al = 1.0; int block = -1;
          for (int i = 0; i < C.nIneq; ++i) { bool inW = false; for (int q2 = 0; q2 < nw; ++q2) if (W[q2] == i) inW = true; if (inW) continue; if (Dp[i] > 1e-10 * fmax(1.0, pn)) { const double aa = fmax(0.0, (fb[i] - Dz[i]) / Dp[i]); if (aa < al) { al = aa; block = i; } } }   // relative threshold: E p = 0 only to round-off
          for (int k = 0; k < n; ++k) z[k] += al * p[k];
          degenerate = (al <= 1e-12);
          if (block >= 0) { if (nw < n && nw < WMAXACT) W[nw++] = block; else { status[level] = 2; break; } }
        }
      }
      if (it >= 100 && status[level] == 0) status[level] = 1;
      for (int r = 0; r < WNV; ++r) { double s = 0.0; for (int k = 0; k < n; ++k) s += Zp[r * n + k] * z[k]; dx[r] = s; }
    }
    for (int r = 0; r < WNV; ++r) { x[r] += dx[r]; xlev[level][r] = x[r]; }
    if (level < 2) { nz = wbc_null_space(AZ, ra, n, Zp, Zn, Gw, Gw + WMAXA * WNV); for (int i = 0; i < WNV * nz; ++i) Zp[i] = Zn[i]; }
    if (level > 0 && status[level] == 0 && status[level - 1] != 0) status[level] = status[level - 1];
  }
  // ---- updateCmd (WbcBase.cpp:548-563) ----
  double* out = a.out + (size_t)b * QM_NWBC_OUT; double tau[18]; wbc_tau_lin(C, x, tau);
  for (int i = 0; i < WNV; ++i) out[i] = x[i];
  for (int r = 0; r < 18; ++r) out[WNV + r] = tau[r] + nle[6 + r];
  for (int l = 0; l < 3; ++l) a.qp_status[b * 3 + l] = status[l];
  if (a.dbg) {
    double* d = a.dbg + (size_t)b * WBC_DBG_SIZE; int o = 0;
    for (int i = 0; i < 24; ++i) d[o++] = q[i]; for (int i = 0; i < 24; ++i) d[o++] = v[i]; for (int i = 0; i < 24; ++i) d[o++] = qd[i]; for (int i = 0; i < 24; ++i) d[o++] = vd[i];
    for (int i = 0; i < 6; ++i) d[o++] = baseAcc[i]; for (int i = 0; i < 24; ++i) d[o++] = nle[i];
    for (int l = 0; l < 3; ++l) for (int i = 0; i < 36; ++i) d[o++] = xlev[l][i];
    for (int i = 0; i < 576; ++i) d[o++] = M[i]; for (int i = 0; i < 288; ++i) d[o++] = Jf[i]; for (int i = 0; i < 12; ++i) d[o++] = dJv[i];
  }
}
