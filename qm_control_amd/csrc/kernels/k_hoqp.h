// k_hoqp.h — qm::HoQp on ARBITRARY task hierarchies: the general stacking of qm_wbc/src/HoQp.cpp:12-158, batched, one wavefront per problem.
//
// The whole-body controller's own cascade (k_wbc.h) is specialised to the two hierarchies the reference ships: inequality rows in the first level only, slack
// eliminated analytically, everything in registers and 40 KB of LDS.  This kernel is the GENERAL form behind qmhip_hoqp_solve: any number of levels, equality rows
// A x = b and inequality rows D x <= f at every level (own inequality rows BELOW the first level included), for a WbcBase subclass that stacks its tasks differently.
// It is not on the benchmark's path and is written for clarity, not speed: dense Householder factorisations on a per-problem workspace in HBM, lanes striding over
// rows / columns, a workgroup-scope fence between dependent phases.
//
// Per level (HoQp.cpp:57-133), with Zp / xp the stacked null-space basis and solution of the higher levels:
//   minimise ½|A Zp z + A xp − b|² + ½ 1e-12 |z|² + ½|w|²
//   s.t.     −w <= 0,   Dp Zp z <= fp − Dp xp + wp*  (rows of the higher levels, relaxed by their slack solutions),   D Zp z − w <= f − D xp
// The slack w stays a VARIABLE (y = [z; w], the rows in HoQp::buildDMatrix's order) and the level is solved by the primal active set of oracle/src/wbc.h
// (primalActiveSetLSI / eqConstrainedLS: QR of the working rows' transpose, reduced least squares on their null space) from the feasible point z = 0,
// w = max(0, −(f − D xp)).  The reference's pairing quirk is reproduced by construction: the stacked rows are kept current-level-first (stackedTasks_ = task_ +
// stackedTasksPrev_, HoQp.cpp:46), the stacked slack solutions previous-level-first (HoQp.cpp:152-158).  status per level: 0 ok, 1 iteration limit (qpOASES'
// nWSR = 100), 2 working set larger than the problem, 3 the higher levels' rows do not hold at the previous solution (the quirk at work: two higher levels with
// inequality rows and a non-zero slack).
#pragma once
#include "qm_dev_common.h"

#define HQ_NMAX   36      /* decision variables */
#define HQ_MAMAX  36      /* equality rows of one level */
#define HQ_MDMAX  64      /* inequality rows of one level */
#define HQ_MHMAX  128     /* stacked inequality rows of all higher levels */
#define HQ_NYMAX  (HQ_NMAX + HQ_MDMAX)                    /* [z; w] */
#define HQ_NCMAX  (2 * HQ_MDMAX + HQ_MHMAX)               /* constraint rows of a level */
#define HQ_LEVELS 8
#define HQ_RHO 1.0e-12    /* HoQp.cpp:66 */

// workspace carve (doubles) of one problem
#define HQW_Z     0                                        /* [36][36] stacked null-space basis (n x nz, ld 36) */
#define HQW_ZN    (HQW_Z + HQ_NMAX * HQ_NMAX)
#define HQW_X     (HQW_ZN + HQ_NMAX * HQ_NMAX)              /* [36] */
#define HQW_DST   (HQW_X + HQ_NMAX)                        /* [128][36] stacked rows, current level first */
#define HQW_FST   (HQW_DST + HQ_MHMAX * HQ_NMAX)
#define HQW_WST   (HQW_FST + HQ_MHMAX)                     /* stacked slack solutions, previous levels first */
#define HQW_AZ    (HQW_WST + HQ_MHMAX)                     /* [36][36] */
#define HQW_G     (HQW_AZ + HQ_MAMAX * HQ_NMAX)            /* [(36 + 36)][37]: [A Zp; sqrt(rho) I | g0] -> [R0 | c0] */
#define HQW_DZ    (HQW_G + (HQ_MAMAX + HQ_NMAX) * (HQ_NMAX + 1))      /* [64][36] own rows times Zp */
#define HQW_HZ    (HQW_DZ + HQ_MDMAX * HQ_NMAX)            /* [128][36] higher rows times Zp */
#define HQW_FBO   (HQW_HZ + HQ_MHMAX * HQ_NMAX)            /* [64] */
#define HQW_FBH   (HQW_FBO + HQ_MDMAX)                     /* [128] */
#define HQW_Y     (HQW_FBH + HQ_MHMAX)                     /* [100] */
#define HQW_YN    (HQW_Y + HQ_NYMAX)
#define HQW_P     (HQW_YN + HQ_NYMAX)
#define HQW_LAM   (HQW_P + HQ_NYMAX)
#define HQW_V     (HQW_LAM + HQ_NYMAX)                     /* [256] scratch vectors */
#define HQW_ET    (HQW_V + 256)                            /* [100][100] working rows transposed -> R_E */
#define HQW_QE    (HQW_ET + HQ_NYMAX * HQ_NYMAX)           /* [100][100] explicit Q of that factorisation */
#define HQW_RN    (HQW_QE + HQ_NYMAX * HQ_NYMAX)           /* [100][101] [R N | rhs] */
#define HQW_SIZE  (HQW_RN + HQ_NYMAX * (HQ_NYMAX + 1))

struct QmHoqpArgs {
  int B, n_levels, n;
  int ma[HQ_LEVELS], md[HQ_LEVELS];
  const double* A; const double* b; const double* D; const double* f;      // [B][sum ma][n], [B][sum ma], [B][sum md][n], [B][sum md]
  int sum_ma, sum_md;
  double* ws;                                                            // [B][HQW_SIZE]
  int* wlist;                                                            // [B][HQ_NCMAX] working-set lists
  double* x; int* status;                                                // [B][n], [B][n_levels]
};

__device__ __forceinline__ void hq_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); }

// Householder QR of the m x n matrix W (ld) in place, first `steps` columns; `extra` further columns (right-hand sides) ride along.  Q (m x m, ld ldq) is formed
// explicitly when given.  Lanes stride over the columns / the rows of Q.  (Same arithmetic as oracle/src/la.h: householderQR.)
__device__ __forceinline__ void hq_house(double* W, int m, int n, int ld, int steps, double* Q, int ldq, double* v) {
  const int l = threadIdx.x & 63;
  if (Q) { for (int idx = l; idx < m * m; idx += 64) { const int i = idx / m, j = idx - i * m; Q[i * ldq + j] = (i == j) ? 1.0 : 0.0; } }
  hq_sync();
  for (int k = 0; k < steps && k < m; ++k) {
    double part = 0.0; for (int i = k + l; i < m; i += 64) part += W[i * ld + k] * W[i * ld + k];
    const double norm = sqrt(qm_wave_sum(part));
    if (norm == 0.0) continue;                                            // wave-uniform
    const double wkk = W[k * ld + k], alpha = wkk > 0.0 ? -norm : norm;
    for (int i = k + l; i < m; i += 64) v[i] = W[i * ld + k] - ((i == k) ? alpha : 0.0);
    hq_sync();
    double vp = 0.0; for (int i = k + l; i < m; i += 64) vp += v[i] * v[i];
    const double vn = qm_wave_sum(vp);
    if (vn == 0.0) continue;
    const double sc = 2.0 / vn;
    for (int j = k + l; j < n; j += 64) { double s = 0.0; for (int i = k; i < m; ++i) s += v[i] * W[i * ld + j]; s *= sc; for (int i = k; i < m; ++i) W[i * ld + j] -= s * v[i]; }
    if (Q) for (int j = l; j < m; j += 64) { double s = 0.0; for (int i = k; i < m; ++i) s += Q[j * ldq + i] * v[i]; s *= sc; for (int i = k; i < m; ++i) Q[j * ldq + i] -= s * v[i]; }
    hq_sync();
  }
}
// x = R⁻¹ c for the upper-triangular n x n block of W (ld), c = column `cc` of W; lane 0 works, everybody sees the result after the sync
__device__ __forceinline__ void hq_backsolve(const double* W, int n, int ld, int cc, double* x) {
  if ((threadIdx.x & 63) == 0) for (int i = n - 1; i >= 0; --i) { double s = W[i * ld + cc]; for (int j = i + 1; j < n; ++j) s -= W[i * ld + j] * x[j]; x[i] = s / W[i * ld + i]; }
  hq_sync();
}

// min |R y − c|² s.t. E y = e  with R = blkdiag(R0 (nz x nz, upper), I (ms x ms)), c = [c0; 0];  E: me rows given by `rowfn(a, j)` / `rhsfn(a)`.
// Returns yn and the multipliers lam (oracle/src/wbc.h: eqConstrainedLS).
template <class RowFn, class RhsFn>
__device__ __forceinline__ void hq_eq_ls(double* ws, int nz, int ms, int me, RowFn rowfn, RhsFn rhsfn, double* yn, double* lam) {
  const int l = threadIdx.x & 63, ny = nz + ms;
  const double* R0 = ws + HQW_G; const int ldg = HQ_NMAX + 1;
  double* ET = ws + HQW_ET; double* QE = ws + HQW_QE; double* RN = ws + HQW_RN; double* v = ws + HQW_V;
  auto Rent = [&](int i, int j) { return (i < nz) ? ((j < nz && j >= i) ? R0[i * ldg + j] : 0.0) : ((i == j) ? 1.0 : 0.0); };
  auto cent = [&](int i) { return (i < nz) ? R0[i * ldg + nz] : 0.0; };
  if (me == 0) {
    hq_backsolve(R0, nz, ldg, nz, yn);
    for (int i = nz + l; i < ny; i += 64) yn[i] = 0.0;
    hq_sync(); return;
  }
  // Eᵀ = Q [R_E; 0]
  for (int idx = l; idx < ny * me; idx += 64) { const int i = idx / me, a = idx - i * me; ET[i * HQ_NYMAX + a] = rowfn(a, i); }
  hq_sync();
  hq_house(ET, ny, me, HQ_NYMAX, me, QE, HQ_NYMAX, v);
  // R_Eᵀ y1 = e
  double* y1 = v + 128;
  if (l == 0) for (int i = 0; i < me; ++i) { double s = rhsfn(i); for (int k = 0; k < i; ++k) s -= ET[k * HQ_NYMAX + i] * y1[k]; y1[i] = s / ET[i * HQ_NYMAX + i]; }
  hq_sync();
  // yp = Y y1 (Y = first me columns of Q)
  for (int i = l; i < ny; i += 64) { double s = 0.0; for (int a = 0; a < me; ++a) s += QE[i * HQ_NYMAX + a] * y1[a]; yn[i] = s; }
  hq_sync();
  const int nn = ny - me;
  if (nn > 0) {
    // [R N | c − R yp], N = last nn columns of Q
    for (int idx = l; idx < ny * (nn + 1); idx += 64) {
      const int i = idx / (nn + 1), j = idx - i * (nn + 1); double s;
      if (j < nn) { s = 0.0; if (i < nz) { for (int k = i; k < nz; ++k) s += R0[i * ldg + k] * QE[k * HQ_NYMAX + me + j]; } else s = QE[i * HQ_NYMAX + me + j]; }
      else { s = cent(i); if (i < nz) { for (int k = i; k < nz; ++k) s -= R0[i * ldg + k] * yn[k]; } else s -= yn[i]; }
      RN[i * (HQ_NYMAX + 1) + j] = s;
    }
    hq_sync();
    hq_house(RN, ny, nn + 1, HQ_NYMAX + 1, nn, nullptr, 0, v);
    double* y2 = v + 128;
    hq_backsolve(RN, nn, HQ_NYMAX + 1, nn, y2);
    for (int i = l; i < ny; i += 64) { double s = yn[i]; for (int j = 0; j < nn; ++j) s += QE[i * HQ_NYMAX + me + j] * y2[j]; yn[i] = s; }
    hq_sync();
  }
  // multipliers: R_E lam = −Yᵀ Rᵀ (R y − c)
  double* r = v; double* gr = v + 128;
  for (int i = l; i < ny; i += 64) { double s = -cent(i); if (i < nz) { for (int k = i; k < nz; ++k) s += R0[i * ldg + k] * yn[k]; } else s += yn[i]; r[i] = s; }
  hq_sync();
  for (int j = l; j < ny; j += 64) { double s = 0.0; if (j < nz) { for (int i = 0; i <= j; ++i) s += R0[i * ldg + j] * r[i]; } else s = r[j]; gr[j] = s; }
  hq_sync();
  for (int a = l; a < me; a += 64) { double s = 0.0; for (int i = 0; i < ny; ++i) s += QE[i * HQ_NYMAX + a] * gr[i]; lam[a] = -s; }
  hq_sync();
  if (l == 0) for (int i = me - 1; i >= 0; --i) { double s = lam[i]; for (int j = i + 1; j < me; ++j) s -= ET[i * HQ_NYMAX + j] * lam[j]; lam[i] = s / ET[i * HQ_NYMAX + i]; }
  hq_sync(); (void)Rent;
}

__global__ void __launch_bounds__(64) qm_hoqp_kernel(QmHoqpArgs a) {
  const int b = blockIdx.x, l = threadIdx.x & 63;
  if (b >= a.B) return;
  const int n = a.n;
  double* ws = a.ws + (size_t)b * HQW_SIZE; int* W = a.wlist + (size_t)b * HQ_NCMAX;
  double* Z = ws + HQW_Z; double* ZN = ws + HQW_ZN; double* x = ws + HQW_X; double* Dst = ws + HQW_DST; double* fst = ws + HQW_FST; double* wst = ws + HQW_WST;
  double* AZ = ws + HQW_AZ; double* G = ws + HQW_G; double* DZ = ws + HQW_DZ; double* HZ = ws + HQW_HZ; double* fbo = ws + HQW_FBO; double* fbh = ws + HQW_FBH;
  double* y = ws + HQW_Y; double* yn = ws + HQW_YN; double* p = ws + HQW_P; double* lam = ws + HQW_LAM; double* v = ws + HQW_V;
  const int ldg = HQ_NMAX + 1;
  for (int idx = l; idx < n * n; idx += 64) { const int i = idx / n, j = idx - i * n; Z[i * HQ_NMAX + j] = (i == j) ? 1.0 : 0.0; }
  for (int i = l; i < n; i += 64) x[i] = 0.0;
  hq_sync();
  int nz = n, mh = 0, nws = 0, oa = 0, od = 0, prev_status = 0;
  for (int lev = 0; lev < a.n_levels; ++lev) {
    const int ma = a.ma[lev], ms = a.md[lev];
    const double* Al = a.A + ((size_t)b * a.sum_ma + oa) * n; const double* bl = a.b + (size_t)b * a.sum_ma + oa;
    const double* Dl = a.D + ((size_t)b * a.sum_md + od) * n; const double* fl = a.f + (size_t)b * a.sum_md + od;
    int status = 0;
    // A Zp, [A Zp; sqrt(rho) I | b − A xp; 0] -> [R0 | c0]
    for (int idx = l; idx < (ma + nz) * (nz + 1); idx += 64) {
      const int i = idx / (nz + 1), j = idx - i * (nz + 1); double s = 0.0;
      if (i < ma) { if (j < nz) { for (int k = 0; k < n; ++k) s += Al[i * n + k] * Z[k * HQ_NMAX + j]; AZ[i * HQ_NMAX + j] = s; } else { s = bl[i]; for (int k = 0; k < n; ++k) s -= Al[i * n + k] * x[k]; } }
      else s = (j == i - ma) ? sqrt(HQ_RHO) : 0.0;
      G[i * ldg + j] = s;
    }
    hq_sync();
    hq_house(G, ma + nz, nz + 1, ldg, nz, nullptr, 0, v);
    // own rows and the higher levels' rows in the current coordinates
    for (int idx = l; idx < ms * nz; idx += 64) { const int i = idx / nz, j = idx - i * nz; double s = 0.0; for (int k = 0; k < n; ++k) s += Dl[i * n + k] * Z[k * HQ_NMAX + j]; DZ[i * HQ_NMAX + j] = s; }
    for (int i = l; i < ms; i += 64) { double s = fl[i]; for (int k = 0; k < n; ++k) s -= Dl[i * n + k] * x[k]; fbo[i] = s; }
    for (int idx = l; idx < mh * nz; idx += 64) { const int i = idx / nz, j = idx - i * nz; double s = 0.0; for (int k = 0; k < n; ++k) s += Dst[i * HQ_NMAX + k] * Z[k * HQ_NMAX + j]; HZ[i * HQ_NMAX + j] = s; }
    for (int i = l; i < mh; i += 64) { double s = fst[i] + ((i < nws) ? wst[i] : 0.0); for (int k = 0; k < n; ++k) s -= Dst[i * HQ_NMAX + k] * x[k]; fbh[i] = s; }
    hq_sync();
    const int ny = nz + ms, nc = 2 * ms + mh;
    for (int i = l; i < ny; i += 64) y[i] = (i < nz) ? 0.0 : fmax(0.0, -fbo[i - nz]);
    // the higher levels' rows must hold at the previous solution (hardRowsHoldAtPrevious)
    { double sc = 1.0, mn = 0.0; for (int i = l; i < mh; i += 64) { sc = fmax(sc, fabs(fbh[i])); mn = fmin(mn, fbh[i]); }
      sc = qm_wave_max(sc); mn = -qm_wave_max(-mn); if (mn < -1e-9 * sc) status = 3; }
    hq_sync();
    // constraint row i of the level: [0 −I] (i < ms), [HZ 0] (ms <= i < ms + mh), [DZ −I] (the rest); c: 0, fbh, fbo
    auto crow = [&](int i, int j) -> double {
      if (i < ms) return (j == nz + i) ? -1.0 : 0.0;
      if (i < ms + mh) return (j < nz) ? HZ[(i - ms) * HQ_NMAX + j] : 0.0;
      const int q = i - ms - mh; return (j < nz) ? DZ[q * HQ_NMAX + j] : ((j == nz + q) ? -1.0 : 0.0);
    };
    auto crhs = [&](int i) -> double { return (i < ms) ? 0.0 : ((i < ms + mh) ? fbh[i - ms] : fbo[i - ms - mh]); };
    int nw = 0, it = 0; bool degenerate = false; double pscale = 0.0;
    if (status == 0) {
      for (; it < 100; ++it) {
        hq_eq_ls(ws, nz, ms, nw, [&](int aidx, int j) { return crow(W[aidx], j); }, [&](int aidx) { return crhs(W[aidx]); }, yn, lam);
        double pn = 0.0, zs = 1.0;
        for (int i = l; i < ny; i += 64) { const double pi = yn[i] - y[i]; p[i] = pi; pn = fmax(pn, fabs(pi)); zs = fmax(zs, fabs(y[i])); }
        pn = qm_wave_max(pn); zs = qm_wave_max(zs); pscale = fmax(pscale, pn);
        hq_sync();
        if (pn <= 1e-9 * fmax(zs, pscale)) {
          // stationary on the working set: drop the row with the most negative multiplier (lowest constraint index after a degenerate step: Bland)
          double lscale = 1.0; for (int q = l; q < nw; q += 64) lscale = fmax(lscale, fabs(lam[q])); lscale = qm_wave_max(lscale);
          int worst = -1;
          if (l == 0) { double lw = 0.0; for (int q = 0; q < nw; ++q) if (lam[q] < -1e-9 * lscale) { if (degenerate) { if (worst < 0 || W[q] < W[worst]) worst = q; } else if (lam[q] < lw) { lw = lam[q]; worst = q; } } v[255] = (double)worst; }
          hq_sync(); worst = (int)v[255];
          if (worst < 0) break;
          if (l == 0) for (int q = worst; q + 1 < nw; ++q) W[q] = W[q + 1];
          --nw; hq_sync();
        } else {
          // ratio test over the rows outside the working set (ties: lowest index)
          double amin = 1.0; int block = -1;
          if (l == 0) {
            for (int i = 0; i < nc; ++i) {
              bool inw = false; for (int q = 0; q < nw; ++q) inw = inw || (W[q] == i);
              if (inw) continue;
              double dp = 0.0, dy = 0.0;
              if (i < ms) { dp = -p[nz + i]; dy = -y[nz + i]; }
              else { for (int j = 0; j < nz; ++j) { const double e = crow(i, j); dp += e * p[j]; dy += e * y[j]; } if (i >= ms + mh) { const int q = i - ms - mh; dp -= p[nz + q]; dy -= y[nz + q]; } }
              if (dp > 1e-10 * fmax(1.0, pn)) { const double al = fmax(0.0, (crhs(i) - dy) / dp); if (al < amin) { amin = al; block = i; } }
            }
            v[254] = amin; v[255] = (double)block;
          }
          hq_sync(); amin = v[254]; block = (int)v[255];
          for (int i = l; i < ny; i += 64) y[i] += amin * p[i];
          degenerate = (amin <= 1e-12);
          hq_sync();
          if (block >= 0) { if (nw < ny) { if (l == 0) W[nw] = block; ++nw; hq_sync(); } else { status = 2; break; } }
        }
      }
      if (it >= 100 && status == 0) status = 1;
    }
    // x += Zp z; slack solution; next null-space basis Zp <- Zp null(A Zp); stacked rows / slacks for the levels below
    for (int i = l; i < n; i += 64) { double s = x[i]; for (int j = 0; j < nz; ++j) s += Z[i * HQ_NMAX + j] * y[j]; v[i] = s; }      // (status 3: y = [0; initial slack], x stays)
    hq_sync();
    for (int i = l; i < n; i += 64) x[i] = v[i];
    int nzn = nz;
    if (ma > 0) {
      // QR with column pivoting of (A Zp)ᵀ (nz x ma), explicit Q: its last nz − rank columns span the kernel (oracle/src/la.h: nullSpace)
      double* T = ws + HQW_ET; double* Q = ws + HQW_QE;
      for (int idx = l; idx < nz * ma; idx += 64) { const int i = idx / ma, j = idx - i * ma; T[i * HQ_NYMAX + j] = AZ[j * HQ_NMAX + i]; }
      for (int idx = l; idx < nz * nz; idx += 64) { const int i = idx / nz, j = idx - i * nz; Q[i * HQ_NYMAX + j] = (i == j) ? 1.0 : 0.0; }
      hq_sync();
      int rank = 0; double maxnorm0 = 0.0; const int steps = (nz < ma) ? nz : ma;
      for (int k = 0; k < steps; ++k) {
        // pivot: the remaining column of largest norm
        double bn = -1.0; int best = k;
        for (int j = k; j < ma; ++j) { double part = 0.0; for (int i = k + l; i < nz; i += 64) part += T[i * HQ_NYMAX + j] * T[i * HQ_NYMAX + j]; const double s = qm_wave_sum(part); if (s > bn) { bn = s; best = j; } }
        if (k == 0) maxnorm0 = sqrt(bn);
        if (sqrt(bn) <= 1e-9 * fmax(1.0, maxnorm0)) break;
        if (best != k) { for (int i = l; i < nz; i += 64) { const double t = T[i * HQ_NYMAX + k]; T[i * HQ_NYMAX + k] = T[i * HQ_NYMAX + best]; T[i * HQ_NYMAX + best] = t; } hq_sync(); }
        const double norm = sqrt(bn), tkk = T[k * HQ_NYMAX + k], alpha = tkk > 0.0 ? -norm : norm;
        for (int i = k + l; i < nz; i += 64) v[i] = T[i * HQ_NYMAX + k] - ((i == k) ? alpha : 0.0);
        hq_sync();
        double vp = 0.0; for (int i = k + l; i < nz; i += 64) vp += v[i] * v[i];
        const double vn = qm_wave_sum(vp);
        if (vn > 0.0) {
          const double sc = 2.0 / vn;
          for (int j = k + l; j < ma; j += 64) { double s = 0.0; for (int i = k; i < nz; ++i) s += v[i] * T[i * HQ_NYMAX + j]; s *= sc; for (int i = k; i < nz; ++i) T[i * HQ_NYMAX + j] -= s * v[i]; }
          for (int j = l; j < nz; j += 64) { double s = 0.0; for (int i = k; i < nz; ++i) s += Q[j * HQ_NYMAX + i] * v[i]; s *= sc; for (int i = k; i < nz; ++i) Q[j * HQ_NYMAX + i] -= s * v[i]; }
        }
        hq_sync();
        ++rank;
      }
      nzn = nz - rank;
      for (int idx = l; idx < n * nzn; idx += 64) { const int i = idx / nzn, j = idx - i * nzn; double s = 0.0; for (int k = 0; k < nz; ++k) s += Z[i * HQ_NMAX + k] * Q[k * HQ_NYMAX + rank + j]; ZN[i * HQ_NMAX + j] = s; }
      hq_sync();
      for (int idx = l; idx < n * nzn; idx += 64) { const int i = idx / nzn, j = idx - i * nzn; Z[i * HQ_NMAX + j] = ZN[i * HQ_NMAX + j]; }
      hq_sync();
    }
    // stackedTasks_ = task_ + stackedTasksPrev_: the level's rows go IN FRONT; its slack solutions BEHIND the previous ones
    if (ms > 0) {
      for (int i = mh - 1; i >= 0; --i) { for (int k = l; k < n; k += 64) Dst[(i + ms) * HQ_NMAX + k] = Dst[i * HQ_NMAX + k]; if (l == 0) fst[i + ms] = fst[i]; hq_sync(); }
      for (int idx = l; idx < ms * n; idx += 64) { const int i = idx / n, k = idx - i * n; Dst[i * HQ_NMAX + k] = Dl[i * n + k]; }
      for (int i = l; i < ms; i += 64) { fst[i] = fl[i]; wst[nws + i] = fmax(0.0, y[nz + i]); }
      mh += ms; nws += ms;
      hq_sync();
    }
    if (prev_status != 0 && status == 0) status = prev_status;
    if (l == 0) a.status[(size_t)b * a.n_levels + lev] = status;
    prev_status = status; nz = nzn; oa += ma; od += ms;
  }
  for (int i = l; i < n; i += 64) a.x[(size_t)b * n + i] = x[i];
}
