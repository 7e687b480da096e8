// k_wbc.h — K6/K7: hierarchical whole-body controller, ONE WAVEFRONT PER INSTANCE, all matrices in LDS.
//
// Restates qm_wbc (WbcBase.cpp:118-563, HierarchicalWbc.cpp:18-44, HierarchicalMpcWbc.cpp:18-34, HoQp.cpp:12-158,
// Task.h:17-66) — SURVEY.md §8 a13–a19:
//   updateMeasured / updateDesired  -> recursive rigid-body passes (qm_dev_rbd.h); the 5 chains of the measured pass,
//                                      of the desired pass and of the joint-acceleration pass run on 15 lanes at once
//   13 task formulators             -> rows of A_k x = b_k and the structured inequality block D0 x <= f0
//   HoQp cascade (3 levels)         -> each level is an inequality-constrained least-squares problem
//        min ½|A Zp z + A xp − b|² + ½ rho |z|² + ½|w|²  s.t.  w >= 0, D Zp z − w <= f − D xp,  Dp Zp z <= fp − Dp xp + wp*
//      solved exactly, lane = matrix column / row (stands in for qpOASES, whose return code the reference ignores, HoQp.cpp:143-146; we report qp_status).
//      rho = 1e-12 (HoQp.cpp:66).  Level 0 (own soft rows): Newton on the active set; the task rows are factored once (Householder, rq_house_tri), an iteration folds
//      its active soft rows into a copy and searches the step length exactly.  Levels >= 1 (hard rows of level 0): primal active set (Nocedal & Wright 16.3) on a TQ
//      factorisation that is UPDATED by plane rotations when a working row comes or goes (tq_append / tq_drop / tq_solve / tq_mult) — no factorisation is repeated.
//   updateCmd                        -> tau = [M_j, −J_jᵀ] x + h_j
// D0 (torque limits ± and friction pyramids) is never materialised: products D0·x use tau(x) and the 5x3 pyramid.
// Control flow is wave-uniform: every decision is taken on values all 64 lanes read from LDS or get from a wave reduction,
// and a wave's DS operations retire in order, so no s_barrier is needed (qm_wave_sync = compiler fence only).
// Only the shipped hierarchy shapes are supported (own inequality rows only at level 0).
#pragma once
#include "qm_dev_rbd.h"
#include "qm_dev_kin.h"

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
  int* qp_status;                             // [B][3]  0 ok, 1 iteration limit (nWSR=100), 2 working set overflow
  double* scratch;                            // [B][WBC_SCRATCH] per-instance HBM scratch (WS_* below): tip records, arm Jacobian, cycle counters
  int stop;                                   // profiling only: 1 return after the rigid-body phase, 2/3/4 after level 0/1/2 (no outputs)
  double* dbg;                                // optional [B][WBC_DBG_SIZE]: qMeas vMeas qDes vDes baseAcc nle x0 x1 x2 M J dJv
};
#define WBC_SCRATCH 432  /* per-instance HBM scratch, WS_* below */
#define WBC_DBG_SIZE (24 * 4 + 6 + 24 + 36 * 3 + 576 + 288 + 12)
#define WBC_BLOCK 64

#define WNV 36
#define WMAXA 22        /* max equality-task rows of one level */
#define WMAXACT 20      /* cap on simultaneously active inequality rows */
#define WMAXINEQ 56
#define WRHO 1e-12

// ---- LDS carve (doubles): 40 KB per wavefront -> four instances per CU ----
#define WVLD 18                                   /* leading dim of the small (n <= 18) level >= 1 work arrays */
#define WTLD 19                                   /* [R | c] and its scratch copy at levels >= 1 */
#define WL_M      0
#define WL_NLE    (WL_M + 576)
#define WL_JF     (WL_NLE + 24)
#define WL_BB     (WL_JF + 288)                   /* [22] */
#define WL_AZ     (WL_BB + WMAXA)                 /* [22][36]: level 0 task rows (Zp = I), A Zp at levels >= 1 */
#define WL_G      (WL_AZ + WMAXA * WNV)           /* 720: level 0 packed triangular [R | c] (702); levels >= 1 T = [R | c] updated in place (18 x 19) + the append's row / coefficients (WL_TQ_*);
                                                     null-space workspace (A Zp)ᵀ; RBD sums/accumulators before the cascade */
#define WG_SIZE   720
#define WL_ZP     (WL_G + WG_SIZE)                /* [36][n], n <= 18 (level 0 works with Zp = I implicitly) */
#define WL_HV     (WL_ZP + WNV * WVLD)            /* [48] pivot column broadcast */
#define WL_X      (WL_HV + 48)
#define WL_Z      (WL_X + WNV)
#define WL_ZN     (WL_Z + WNV)
#define WL_P      (WL_ZN + WNV)
#define WL_ZZ     (WL_P + WNV)                    /* Zp z */
#define WL_ZPV    (WL_ZZ + WNV)                   /* Zp p */
#define WL_G0RHS  (WL_ZPV + WNV)                  /* [58] g0 */
#define WL_F0     (WL_G0RHS + 64)
#define WL_W0     (WL_F0 + WMAXINEQ)
#define WL_FB     (WL_W0 + WMAXINEQ)
#define WL_DZ     (WL_FB + WMAXINEQ)
#define WL_DP     (WL_DZ + WMAXINEQ)
#define WL_TAU    (WL_DP + WMAXINEQ)              /* [18] */
#define WL_V      (WL_TAU + 24)                   /* level 0: copy of the task rows' factor [R0 | c0] (702, spans V .. R); levels >= 1: Q [18][18] of the TQ factorisation */
#define WL_A      WL_V                            /* [22][36] task rows of levels >= 1: only live while A Zp / g0 are formed, aliases V..EROWS */
#define WL_BETA   (WL_V + WMAXACT * WVLD)
#define WL_R      (WL_BETA + WMAXACT)             /* [20][20] */
#define WL_EROWS  (WL_R + WMAXACT * WMAXACT)      /* [20][18] L = E Q: the working rows in the rotated coordinates (before the loop: hand-over tile of D0 Zp) */
#define WL_ERHS   (WL_EROWS + WMAXACT * WVLD)
#define WL_LAM    (WL_ERHS + WMAXACT)
#define WL_Y      (WL_LAM + WMAXACT)              /* [36] y = Qᵀ z of the current working set (zero beyond n) */
#define WL_W36    (WL_Y + WNV)
#define WL_WLIST  (WL_W36 + WNV)                  /* [20] ints: working set */
#define WL_ACC    (WL_G + 18 * 16)                /* chain accumulators: 3 passes x 6 slots x 20 (inside G, after the momentum sums) */
#define WL_MISC   (WL_WLIST + 12)                 /* q v qd vd w2 (5 x 24), baseAcc(6) */
#define WL_BM     (WL_MISC + 5 * 24 + 8)          /* [21] measured root state for the level-1 tasks: E(9) R(9) al(3) (42 registers less across the level-0 solver) */
#define WL_TOTAL  (WL_BM + 24)
/* rigid-body phase only: per-joint subtree composites of the 18 chain lanes (19 x 6 doubles each, lane interleaved), behind the arm Jacobian that is
   being built at the head of the Zp region; spans Zp .. R, all of which the cascade (re)initialises itself — the region is cleared again after the passes */
#define WL_RBDWS  (WL_ZP + 144)
#define WL_RBDWS_SIZE (RBD_COMP * 6 * 18)
// per-instance HBM scratch (doubles): cycle counters, arm Jacobian, the ten tip records (read a handful of times while the tasks are built)
#define WS_TIME   0
#define WS_JARM   16                              /* [6][24] */
#define WS_TIPS   (WS_JARM + 144)                 /* 10 tips x 27 doubles: measured feet 0-3, arm 4, desired feet 5-8, arm 9 */
#define WS_SIZE   (WS_TIPS + 270 + 2)
#define WBC_LDS_BYTES (WL_TOTAL * 8)

__device__ __forceinline__ double wv_sum(double v) { return qm_wave_sum(v); }
__device__ __forceinline__ double wv_max(double v) { return qm_wave_max(v); }

// rotation error log(R_l R_rᵀ) [upstream rotationErrorInWorld]
__device__ __forceinline__ void dev_rot_error(const double* Rl, const double* Rr, double* err) {
  double R[9]; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R[3 * i + j] = Rl[3 * i] * Rr[3 * j] + Rl[3 * i + 1] * Rr[3 * j + 1] + Rl[3 * i + 2] * Rr[3 * j + 2];
  const double tr = R[0] + R[4] + R[8]; const double v[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]}; const double tmp = 0.5 * (tr - 3.0); double s;
  if (tmp > -1e-2) s = 0.5 - (tr - 3.0) / 12.0;
  else { double c = 0.5 * (tr - 1.0); c = fmax(-1.0, fmin(1.0, c)); const double th = acos(c); s = th / (2.0 * sin(th)); }
  for (int i = 0; i < 3; ++i) err[i] = s * v[i];
}

// Householder scalars of a pivot column with squared norm nrm2 and pivot entry g: alpha = −sign(g)|x|, vk = g − alpha, b2 = 2 / (v·v).
// v·v = 2 |x| (|x| + |g|), so b2 = (1/|x|) · 1/(|x| + |g|): one reciprocal square root and one reciprocal instead of a sqrt and a division.
// This sits on the critical path of every Householder step of a lone wave (≈ 16 cycles per dependent f64 op), so the chain is kept short:
// the hardware estimates (v_rsq_f64 / v_rcp_f64, 2^-24 on gfx950, measured) get ONE third-order correction each (error e³ ≈ 2^-70), and the
// reciprocal's estimate is started from the ESTIMATED norm so that it overlaps the refinement of the rsqrt.  ok = false for a null column.
__device__ __forceinline__ bool qm_house_scalars(double nrm2, double g, double& alpha, double& vk, double& b2) {
  const bool ok = nrm2 > 0.0;
  const double x = ok ? nrm2 : 1.0, ag = fabs(g);
  const double y0 = __builtin_amdgcn_rsq(x);                        // 1/|x| (2^-24)
  const double h = x * y0;                                          // |x| (2^-24)
  const double q0 = __builtin_amdgcn_rcp(h + ag);                   // 1/(|x| + |g|) from the estimated norm: off the rsqrt's chain
  const double e = fma(-h, y0, 1.0), t = fma(0.375, e, 0.5);        // 1/sqrt(x) = y0 (1 + e/2 + 3 e²/8 + O(e³)),  e = 1 − x y0²
  const double r = fma(y0 * e, t, y0), nrm = fma(h * e, t, h);      // 1/|x| and |x| = x / |x|, both to ≈ 1 ulp
  const double den = nrm + ag, e2 = fma(-den, q0, 1.0);
  const double q = fma(fma(e2, e2, e2), q0, q0);                    // q0 (1 + e2 + e2²): 1/den to ≈ 1 ulp
  alpha = g > 0.0 ? -nrm : nrm; vk = g - alpha; b2 = ok ? r * q : 0.0;
  return ok;
}

// ---- wave-cooperative dense helpers ----
// QR of [T; D]: T (n x (n+1), upper triangular with the rhs in column n, LDS, leading dim ldT) stacked on MRD dense rows held
// by column in registers (lane j: d[0..MRD) = column j of D, lane n = its rhs, unused rows zero).  Step k only touches row k of T
// and the dense rows, so a least-squares matrix [sqrt(rho) I; A] costs (1 + rows(A)) per column instead of n + rows(A).
// On return T holds [R | Qᵀ rhs].
// ldT == 0 selects PACKED storage of the upper triangle + rhs: row k holds columns k..n at offset k (2n + 3 − k) / 2.
__device__ __forceinline__ int wv_tidx(int k, int c, int n, int ld) { return ld ? k * ld + c : ((k * (2 * n + 3 - k)) >> 1) + (c - k); }
template <int MRD>
__device__ __forceinline__ void rq_house_tri(double (&d)[MRD], double* T, int ldT, int n, double* hv) {
  // (skipping the unpopulated rows of d with per-row uniform branches was measured 2x SLOWER: the branches serialise the LDS reads)
  const int l = threadIdx.x & 63;
  for (int k = 0; k < n; ++k) {
    // pivot column to every lane: through v_readlane when it fits the scalar registers (MRD <= 24), through LDS otherwise
    double pv[MRD <= 24 ? MRD : 1];
    if (MRD <= 24) {
#pragma unroll
      for (int i = 0; i < (MRD <= 24 ? MRD : 1); ++i) pv[i] = qm_bcast(d[i], k);
    } else {
      qm_wave_sync();
      if (l == k) {
#pragma unroll
        for (int i = 0; i < MRD; ++i) hv[i] = d[i];
      }
      qm_wave_sync();
    }
#define RQ_V(i) (MRD <= 24 ? pv[(MRD <= 24) ? (i) : 0] : hv[i])
    const double tkk = T[wv_tidx(k, k, n, ldT)]; const double tl = (l >= k && l <= n) ? T[wv_tidx(k, l, n, ldT)] : 0.0;
    double nq[4] = {tkk * tkk, 0.0, 0.0, 0.0}, dq[4] = {0.0, 0.0, 0.0, 0.0};   // four partial sums: the dependent chain is what costs on a lone wave
#pragma unroll
    for (int i = 0; i < MRD; ++i) { const double vi = RQ_V(i); nq[i & 3] += vi * vi; dq[i & 3] += vi * d[i]; }
    const double nrm2 = (nq[0] + nq[1]) + (nq[2] + nq[3]), dot = (dq[0] + dq[1]) + (dq[2] + dq[3]);
    double alpha, vk, b2; const bool ok = qm_house_scalars(nrm2, tkk, alpha, vk, b2);
    const double s = (dot + vk * tl) * b2;
    if (ok && l > k && l <= n) {
      T[wv_tidx(k, l, n, ldT)] = tl - s * vk;
#pragma unroll
      for (int i = 0; i < MRD; ++i) d[i] -= s * RQ_V(i);
    }
    qm_wave_sync();                                       // every lane has read the pivot before it is replaced
    if (ok && l == k) T[wv_tidx(k, k, n, ldT)] = alpha;
  }
  qm_wave_sync();
#undef RQ_V
}
// Triangular solves without a wave reduction in the dependency chain: lane j keeps ROW j of the triangular matrix in registers, its
// own right-hand side and 1 / diagonal; the unknowns are resolved one by one (z_i = r_i / R_ii on lane i), broadcast with v_readlane,
// and every other lane folds R[j][i] z_i into its r_j.  `at(j, i)` reads element (j, i) of the matrix (LDS); returns lane l's z_l.
// upper: R z = r, i = n-1 .. 0, row j holds i > j.
template <int MAXN, class At>
__device__ __forceinline__ double wv_solve_upper(At at, int n, double rj) {
  const int l = threadIdx.x & 63;
  double row[MAXN];
#pragma unroll
  for (int i = 0; i < MAXN; ++i) row[i] = (l < n && i > l && i < n) ? at(l, i) : 0.0;
  const double rinv = (l < n) ? 1.0 / at(l, l) : 0.0; double z = 0.0;
#pragma unroll
  for (int i = MAXN - 1; i >= 0; --i) if (i < n) { const double zi = qm_bcast(rj * rinv, i); if (l == i) z = zi; rj -= row[i] * zi; }
  return z;
}
// R z = c on [R | c] held like rq_house_tri leaves it (ld == 0: packed); z -> LDS vector
template <int MAXN>
__device__ __forceinline__ void wv_backsub_tri(const double* T, int ld, int n, double* z) {
  const int l = threadIdx.x & 63;
  const double zl = wv_solve_upper<MAXN>([&](int j, int i) { return T[wv_tidx(j, i, n, ld)]; }, n, (l < n) ? T[wv_tidx(l, n, n, ld)] : 0.0);
  qm_wave_sync();
  if (l < n) z[l] = zl;
  qm_wave_sync();
}

struct WbcCtx {   // everything the D0 block and the torque map need (all wave-uniform)
  const double* M; const double* Jf; const double* nle; int nc; int mode; double mu; int nIneq;     // no arrays: a runtime-indexed member array would live in the private segment (scratch)
  __device__ __forceinline__ bool fl(int k) const { return mode_flag(mode, k); }
  // contact index (LF RF LH RH) of the j-th stance foot
  __device__ __forceinline__ int contactOf(int j) const { int cnt = 0, res = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const bool f = mode_flag(mode, k); res = (f && cnt == j) ? k : res; cnt += f ? 1 : 0; } return res; }
};
// out = D0 x ; lanes cooperate, tau scratch in LDS
__device__ __forceinline__ void wv_d0_apply(const WbcCtx& c, const double* x, double* tau, double* out) {
  const int l = threadIdx.x & 63;
  if (l < 18) {                                             // four partial sums: a dependent f64 chain costs ≈ 16 cycles per link on a lone wave
    double s4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < 24; ++k) s4[k & 3] += c.M[(6 + l) * 24 + k] * x[k];
#pragma unroll
    for (int k = 0; k < 12; ++k) s4[k & 3] -= c.Jf[k * 24 + 6 + l] * x[24 + k];
    const double s = (s4[0] + s4[1]) + (s4[2] + s4[3]); tau[l] = s; out[l] = s; out[18 + l] = -s; }
  if (l >= 36 && l < c.nIneq) {
    const int r = l - 36; double v = 0.0;
    if (r < 5 * c.nc) { const int j = r / 5, q = r - 5 * j; const double* F = x + 24 + 3 * c.contactOf(j); v = (q == 0) ? -F[2] : (q == 1) ? F[0] - c.mu * F[2] : (q == 2) ? -F[0] - c.mu * F[2] : (q == 3) ? F[1] - c.mu * F[2] : -F[1] - c.mu * F[2]; }
    out[l] = v;
  }
  qm_wave_sync();
}
__device__ __forceinline__ double wbc_d0_entry(const WbcCtx& c, int i, int k) {   // D0[i][k]
  if (i < 36) { const int r = (i < 18) ? i : i - 18; const double sg = (i < 18) ? 1.0 : -1.0; return (k < 24) ? sg * c.M[(6 + r) * 24 + k] : -sg * c.Jf[(k - 24) * 24 + 6 + r]; }
  if (i < 36 + 5 * c.nc) { const int j = (i - 36) / 5, q = (i - 36) - 5 * j; const int k0 = 24 + 3 * c.contactOf(j); if (k < k0 || k >= k0 + 3) return 0.0; const int a = k - k0;
    if (q == 0) return a == 2 ? -1.0 : 0.0; if (a == 2) return -c.mu; if (q == 1) return a == 0 ? 1.0 : 0.0; if (q == 2) return a == 0 ? -1.0 : 0.0; if (q == 3) return a == 1 ? 1.0 : 0.0; return a == 1 ? -1.0 : 0.0; }
  return 0.0;
}
// y(36) = Zp (36 x n) z
__device__ __forceinline__ void wv_Z_times(const double* Zp, int n, const double* z, double* y) {
  const int l = threadIdx.x & 63;
  if (l < WNV) { double s = 0.0;
    if (n == WNV) s = z[l];                                 // level 0: Zp = I
    else { double s4[4] = {0.0, 0.0, 0.0, 0.0};             // n <= 18: static loop, the entries past the row end (the next row's, finite) are masked
#pragma unroll
      for (int k = 0; k < WVLD; ++k) { const double zk = Zp[l * n + k] * z[k]; s4[k & 3] += (k < n) ? zk : 0.0; }
      s = (s4[0] + s4[1]) + (s4[2] + s4[3]); }
    y[l] = s; }
  qm_wave_sync();
}

// ---- levels >= 1: the equality-constrained solves of the active-set loop on an UPDATED factorisation ----
// min |R z − c|² s.t. E z = e, E = the working rows.  [R | c] is the once-per-level QR factor of G0 = [A Zp; sqrt(rho) I | g0].  Kept across the iterations:
//   Q (n x n orthogonal), T = Uᵀ R Q upper triangular with ct = Uᵀ c in column n (U is never needed), L = E Q.
// The first k = n − me columns of Q span the null space of E; the direction of working row a (in order of addition) is column n−1−a, so L[a][j] = 0 for j < n−1−a
// (the TQ factorisation of Gill & Murray's null-space active-set methods).  With z = Q y:
//   y[k..n) from the working rows alone (L is triangular), y[0..k) from the leading k x k triangle of T — a solve is two triangular substitutions and a product with Q;
//   appending a row rotates the free columns so that the row's free part lands in column k−1, and k−1 row rotations make T triangular again;
//   dropping working row a moves the directions of the rows behind it one column to the right (one column + one row rotation each), which frees column k.
// Every step of the active-set loop therefore costs O(n²) — no factorisation is ever repeated (rounds 1–3: a QR of the n x (n − me) reduced matrix per step, 13 k cycles).
// Layouts: T (ld WTLD) in the first half of G, Q (ld WVLD) where the reflectors used to be, L rows in WL_EROWS, e in WL_ERHS, y in WL_Y; the second half of G holds the
// new row and the rotation coefficients of an append.
#define WL_TQ_T   WL_G
#define WL_TQ_Q   WL_V
#define WL_TQ_L   WL_EROWS
#define WL_TQ_CO  (WL_G + WVLD * WTLD)            /* t, alpha, sigma, kappa, free part of t: 5 x 20 */
#define WL_TQ_ROW (WL_TQ_CO + 100)                /* [18] the row to append (zero beyond n) */
#define WL_TQ_H   (WL_TQ_ROW + 20)                /* [18] residual / Qᵀ gradient of the multiplier solve */
// Givens pair with c a + s b = |(a, b)|, −s a + c b = 0 (identity for a null pair); the reciprocal root is the hardware estimate + one third-order correction (qm_house_scalars)
__device__ __forceinline__ void qm_givens(double a, double b, double& c, double& s) {
  const double h2 = fma(a, a, b * b); const bool ok = h2 > 0.0; const double x = ok ? h2 : 1.0;
  const double y0 = __builtin_amdgcn_rsq(x), h = x * y0, e = fma(-h, y0, 1.0), r = fma(y0 * e, fma(0.375, e, 0.5), y0);
  c = ok ? a * r : 1.0; s = ok ? b * r : 0.0;
}
// 1 / x from the hardware estimate (2^-24) and two Newton steps: ≈ 1 ulp, a third of the dependent chain of the IEEE division sequence
__device__ __forceinline__ double qm_recip(double x) {
  const double q0 = __builtin_amdgcn_rcp(x), e = fma(-x, q0, 1.0), q1 = fma(fma(e, e, e), q0, q0);
  return fma(fma(-x, q1, 1.0), q1, q1);
}
// 1 / sqrt(x), x > 0 (same correction as qm_givens)
__device__ __forceinline__ double qm_rsqrt(double x) { const double y0 = __builtin_amdgcn_rsq(x), h = x * y0, e = fma(-h, y0, 1.0); return fma(y0 * e, fma(0.375, e, 0.5), y0); }
// wv_solve_upper / wv_solve_lower for the factors of the TQ update: every lane loads its whole row UNCONDITIONALLY (`at` must be readable for every (lane row, i < MAXN)) and
// masks in registers — a load under a wave-uniform condition becomes a scalar branch around each ds_read, which serialises them — and the diagonal is inverted by qm_recip
template <int MAXN, class At>
__device__ __forceinline__ double tq_solve_upper(At at, int n, double rj) {
  const int l = threadIdx.x & 63, lr = (l < n) ? l : 0;
  double row[MAXN];
#pragma unroll
  for (int i = 0; i < MAXN; ++i) { const double v = at(lr, i); row[i] = (l < n && i > l && i < n) ? v : 0.0; }
  const double rinv = (l < n) ? qm_recip(at(lr, lr)) : 0.0; double z = 0.0;
#pragma unroll
  for (int i = MAXN - 1; i >= 0; --i) if (i < n) { const double zi = qm_bcast(rj * rinv, i); if (l == i) z = zi; rj -= row[i] * zi; }
  return z;
}
template <int MAXN, class At>
__device__ __forceinline__ double tq_solve_lower(At at, int n, double rj) {
  const int l = threadIdx.x & 63, lr = (l < n) ? l : 0;
  double row[MAXN];
#pragma unroll
  for (int i = 0; i < MAXN; ++i) { const double v = at(lr, i); row[i] = (l < n && i < l) ? v : 0.0; }
  const double rinv = (l < n) ? qm_recip(at(lr, lr)) : 0.0; double z = 0.0;
#pragma unroll
  for (int i = 0; i < MAXN; ++i) if (i < n) { const double zi = qm_bcast(rj * rinv, i); if (l == i) z = zi; rj -= row[i] * zi; }
  return z;
}
// [R | c] has just been left in T by rq_house_tri (strictly lower part zero): Q = I, no working rows; rows n.. of T are cleared (the column sweeps load 18 rows)
__device__ __forceinline__ void tq_init(double* S, int n) {
  const int l = threadIdx.x & 63;
  for (int idx = l; idx < WVLD * WVLD; idx += 64) { const int r = idx / WVLD, c = idx - WVLD * r; S[WL_TQ_Q + idx] = (r == c && r < n) ? 1.0 : 0.0; }
  for (int idx = l; idx < WVLD * WTLD; idx += 64) { const int r = idx / WTLD, c = idx - WTLD * r; if (r >= n || c > n) S[WL_TQ_T + idx] = 0.0; }   // (entries right of ct are read by the 18-wide sweeps)
  if (l < 20) S[WL_Y + l] = 0.0;                                          // y is kept zero beyond n, and its free part is cleared before it is solved for: no sum below needs a bound
  qm_wave_sync();
}
// z = Q y for the current working set (me rows): zout[0..n)
__device__ __forceinline__ void tq_solve(double* S, int n, int me, double* zout) {
  const int l = threadIdx.x & 63, k = n - me;
  const double* T = S + WL_TQ_T; const double* Q = S + WL_TQ_Q; double* y = S + WL_Y;
  if (l < k) y[l] = 0.0;
  qm_wave_sync();
  { const int lr = (l < n) ? l : 0;
    double rhs[2] = {T[lr * WTLD + n], 0.0};
#pragma unroll
    for (int j = 0; j < WVLD; ++j) rhs[j & 1] -= T[lr * WTLD + j] * y[j];                 // y[0..k) = 0, y[n..) = 0 (column n of T is ct when n < 18)
    const double yl = tq_solve_upper<WVLD>([&](int j, int i) { return T[j * WTLD + i]; }, k, (l < k) ? rhs[0] + rhs[1] : 0.0);
    qm_wave_sync();
    if (l < k) y[l] = yl;
    qm_wave_sync();
  }
  { const int lr = (l < n) ? l : 0; double sp[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int j = 0; j < WVLD; ++j) sp[j & 3] += Q[lr * WVLD + j] * y[j];                  // Q and y are zero beyond n
    if (l < n) zout[l] = (sp[0] + sp[1]) + (sp[2] + sp[3]); }
  qm_wave_sync();
}
// append the row S[WL_TQ_ROW] (zero beyond n) with right-hand side rhs as working row `me` (k = n − me >= 1 free columns before the call)
template <bool PROF>
__device__ __forceinline__ void tq_append(double* S, int n, int me, double rhs, long long* tf) {
  const int l = threadIdx.x & 63, k = n - me;
  long long tl_ = PROF ? (long long)__builtin_readcyclecounter() : 0;
#define WF(k) { if (PROF) { const long long now_ = (long long)__builtin_readcyclecounter(); tf[k] += now_ - tl_; tl_ = now_; } }
  double* T = S + WL_TQ_T; double* Q = S + WL_TQ_Q; double* L = S + WL_TQ_L; double* co = S + WL_TQ_CO; const double* drow = S + WL_TQ_ROW; double* y = S + WL_Y;
  // 1. t = rowᵀ Q (lane = column); co[0..) = t, co[80..) = its free part (zero from column k on)
  if (l < k) y[l] = 0.0;                                                  // the free part of y is solved for again after the append; cleared, the sum of step 5 needs no bound
  { const int lc = (l < n) ? l : 0; double sp[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < WVLD; ++i) sp[i & 3] += drow[i] * Q[i * WVLD + lc];
    const double tl = (l < n) ? (sp[0] + sp[1]) + (sp[2] + sp[3]) : 0.0;
    if (l < 20) { co[l] = tl; co[80 + l] = (l < k) ? tl : 0.0; } }
  qm_wave_sync();
  // 2. the rotations (0,1), (1,2), … (k−2,k−1) that carry the free part et = t[0..k) into column k−1 have a closed-form product: with the prefix norms r_j = |et[0..j]|,
  //    column_j' = alpha_j P_j − sigma_j column_{j+1},  P_j = sum_{c <= j} et_c column_c,  alpha_j = et_{j+1} / (r_j r_{j+1}),  sigma_j = r_j / r_{j+1}   (j < k−1; r_j = 0: unchanged)
  //    column_{k−1}' = P_{k−1} / r_{k−1};   columns >= k are untouched (kappa = 1).   Lane j computes the coefficients of column j.
  double rlast, rlinv;
  { double q4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < WVLD; ++i) { const double ti = co[80 + i]; q4[i & 3] += (i <= l) ? ti * ti : 0.0; }
    const double r2 = (q4[0] + q4[1]) + (q4[2] + q4[3]);
    const double tn = co[80 + ((l < 19) ? l + 1 : 0)];                       // et_{l+1} (zero from column k on)
    const double r2n = fma(tn, tn, r2);
    const bool pos = r2 > 1.0e-280;                                     // (a prefix of exact zeros is the usual case; squared norms in the denormal range count as zero: error < 1e-140)
    const double ij = qm_rsqrt(pos ? r2 : 1.0), in = qm_rsqrt(r2n > 1.0e-280 ? r2n : 1.0);  // 1 / r_j, 1 / r_{j+1}
    double al = 0.0, sg = 0.0, ka = 1.0;
    if (l < k - 1) { if (pos) { al = tn * ij * in; sg = r2 * ij * in; ka = 0.0; } }
    else if (l == k - 1) { al = ij; ka = 0.0; }
    rlast = qm_bcast(r2 * ij, k - 1); rlinv = qm_bcast(ij, k - 1);
    if (l < 20) { co[20 + l] = al; co[40 + l] = sg; co[60 + l] = ka; }
  }
  qm_wave_sync();
  WF(4)
  // 3. columns of T and Q (lane = row: rows of T on lanes 0.., rows of Q on lanes 32..); P runs along the row
  { const bool isT = l < 32; const int i = isT ? l : l - 32; const bool act = i < n;
    double* rowp = isT ? T + (act ? i : 0) * WTLD : Q + (act ? i : 0) * WVLD;
    double x[WVLD + 1];
#pragma unroll
    for (int j = 0; j < WVLD; ++j) x[j] = rowp[j];
    x[WVLD] = 0.0;
    double P = 0.0;
#pragma unroll
    for (int j = 0; j < WVLD; ++j) { P = fma(co[j], x[j], P); const double o = fma(co[20 + j], P, fma(-co[40 + j], x[j + 1], co[60 + j] * x[j])); if (act) rowp[j] = o; }
  }
  qm_wave_sync();
  WF(5)
  // 4. T is upper Hessenberg in its first k−1 columns now: rotate rows (j, j+1), j = 0 .. k−2 (lane = column, ct rides in lane n; the pivot pair comes from lane j)
  { const int c = (l <= n) ? l : 0; double tt[WVLD];
#pragma unroll
    for (int r = 0; r < WVLD; ++r) tt[r] = T[r * WTLD + c];
#pragma unroll
    for (int j = 0; j < WVLD - 1; ++j) {
      if (j < k - 1) {
        const double a = qm_bcast(tt[j], j), b = qm_bcast(tt[j + 1], j);
        double cg, sn; qm_givens(a, b, cg, sn);
        const double u = tt[j], v = tt[j + 1];
        tt[j] = fma(cg, u, sn * v); tt[j + 1] = (l == j) ? 0.0 : fma(cg, v, -sn * u);
      }
    }
    if (l <= n) {
#pragma unroll
      for (int r = 0; r < WVLD; ++r) T[r * WTLD + l] = tt[r];
    }
  }
  WF(6)
  // 5. the new working row in the rotated coordinates and its y
  if (l < WVLD) L[me * WVLD + l] = (l < k - 1) ? 0.0 : (l == k - 1 ? rlast : co[l]);
  { double sp[2] = {rhs, 0.0};
#pragma unroll
    for (int j = 0; j < WVLD; ++j) sp[j & 1] -= co[j] * y[j];                            // y[0..k) = 0
    qm_wave_sync();
    if (l == 0) { y[k - 1] = (sp[0] + sp[1]) * rlinv; S[WL_ERHS + me] = rhs; } }
  qm_wave_sync();
  WF(7)
#undef WF
}
// remove working row ad (me rows before the call)
__device__ __forceinline__ void tq_drop(double* S, int n, int me, int ad) {
  const int l = threadIdx.x & 63;
  double* T = S + WL_TQ_T; double* Q = S + WL_TQ_Q; double* L = S + WL_TQ_L; double* y = S + WL_Y; double* e = S + WL_ERHS;
  // row holders of the column rotations: rows of T on lanes 0..17, rows of Q on 18..35, rows of L on 36..53
  const int g = (l >= 36) ? 2 : (l >= 18 ? 1 : 0), i = l - 18 * g; const bool act = (g == 2) ? (i < me && l < 54) : (i < n);
  double* rowp = (g == 0) ? T + (act ? i : 0) * WTLD : (g == 1) ? Q + (act ? i : 0) * WVLD : L + (act ? i : 0) * WVLD;
  for (int a = ad + 1; a < me; ++a) {
    const int j = n - 1 - a;                                              // pivot column of row a: moves to j + 1
    double cg, sn; qm_givens(L[a * WVLD + j + 1], L[a * WVLD + j], cg, sn);
    qm_wave_sync();
    if (act) { const double u = rowp[j], v = rowp[j + 1]; rowp[j + 1] = fma(cg, v, sn * u); rowp[j] = (g == 2 && i == a) ? 0.0 : fma(cg, u, -sn * v); }
    qm_wave_sync();
    double c2, s2; qm_givens(T[j * WTLD + j], T[(j + 1) * WTLD + j], c2, s2);   // the column rotation left one entry below the diagonal
    qm_wave_sync();
    if (l <= n) { const double u = T[j * WTLD + l], v = T[(j + 1) * WTLD + l]; T[j * WTLD + l] = fma(c2, u, s2 * v); T[(j + 1) * WTLD + l] = (l == j) ? 0.0 : fma(c2, v, -s2 * u); }
    qm_wave_sync();
  }
  for (int a = ad; a + 1 < me; ++a) { if (l < WVLD) L[a * WVLD + l] = L[(a + 1) * WVLD + l]; }      // a wave's DS operations retire in order
  { const bool mv = (l >= ad && l + 1 < me); const double en = mv ? e[l + 1] : 0.0; qm_wave_sync(); if (mv) e[l] = en; }
  qm_wave_sync();
  // y of the working rows from the first one on (the rows behind the dropped one changed): equation a reads sum_{a' <= a} L[a][n−1−a'] y[n−1−a'] = e[a]
  const int m1 = me - 1;
  if (l < n - m1) y[l] = 0.0;
  if (m1 > 0) {
    const double yl = tq_solve_lower<WVLD>([&](int jj, int ii) { return L[jj * WVLD + ((ii < n) ? n - 1 - ii : 0)]; }, m1, (l < m1) ? e[l] : 0.0);
    qm_wave_sync();
    if (l < m1) y[n - 1 - l] = yl;
  }
  qm_wave_sync();
}
// multipliers of the me working rows at the solution of the last tq_solve: grad + Eᵀ lam = 0.  In the rotated coordinates Qᵀ grad = Tᵀ (T y − ct) =: h, whose first k
// entries vanish, and h[n−1−a] = −sum_{a' >= a} L[a'][n−1−a] lam[a']: a triangular solve from the last working row back
__device__ __forceinline__ void tq_mult(double* S, int n, int me) {
  const int l = threadIdx.x & 63, k = n - me;
  const double* T = S + WL_TQ_T; const double* L = S + WL_TQ_L; const double* y = S + WL_Y; double* h = S + WL_TQ_H; double* lam = S + WL_LAM;
  if (me == 0) return;
  { const int lr = (l < n) ? l : 0; double sp[2] = {-T[lr * WTLD + n], 0.0};
#pragma unroll
    for (int j = 0; j < WVLD; ++j) sp[j & 1] += T[lr * WTLD + j] * y[j];                  // y is zero beyond n
    if (l < 20) h[l] = (l >= k && l < n) ? sp[0] + sp[1] : 0.0; }
  qm_wave_sync();
  double hj;
  { const int lc = (l < n) ? l : 0; double sp[2] = {0.0, 0.0};
#pragma unroll
    for (int r = 0; r < WVLD; ++r) sp[r & 1] += T[r * WTLD + lc] * h[r];
    hj = sp[0] + sp[1]; }
  qm_wave_sync();
  if (l < n) h[l] = hj;
  qm_wave_sync();
  const double mu = tq_solve_upper<WVLD>([&](int jj, int ii) { return L[ii * WVLD + n - 1 - jj]; }, me, (l < me) ? h[n - 1 - l] : 0.0);
  if (l < me) lam[l] = -mu;
  qm_wave_sync();
}

// orthonormal null space of AZ (ra x n): Zp (36 x n) <- Zp · Q2, Q2 = Q[:, rank:] from the Householder QR with column pivoting of
// (AZ)ᵀ (n x ra); returns n − rank.  MAXN = 36 at level 0 (n == 36, Zp = I), 18 afterwards.  Lane j holds column j of (AZ)ᵀ in
// registers; rows move up one slot per step so every register index is static, and pivoting only marks lanes — nothing is swapped,
// R is never needed.  Reflector k is stored SHIFTED (Vs[k][i] = v_k[k + i], zero padded to MAXN), so neither sweep needs an index
// bound: Q2 = H_0 … H_{r-1} [0; I] is accumulated backwards with one column of Q2 per lane in a register window that moves down one
// row per reflector (the row that enters is a zero of [0; I]).
template <int MAXN, bool PROF>
__device__ __forceinline__ int wv_null_space(double* S, int ra, int n, long long* tn) {
  const int l = threadIdx.x & 63;
  long long tl_ = PROF ? (long long)__builtin_readcyclecounter() : 0;
#define WN(k) { if (PROF) { const long long now_ = (long long)__builtin_readcyclecounter(); tn[k] += now_ - tl_; tl_ = now_; } }
  double* Vs = S + WL_G; double* Zp = S + WL_ZP; const double* AZ = S + WL_AZ; double* bet = S + WL_R; double* vd = S + WL_R + 40;   // Vs: [rank][MAXN] (<= 18 x 36); bet, vd: 2 / (v·v) and the pivot entry of each reflector
  double col[MAXN];
#pragma unroll
  for (int i = 0; i < MAXN; ++i) col[i] = (l < ra && i < n) ? AZ[l * WNV + i] : 0.0;
  const int steps = (n < ra) ? n : ra;
  bool done = (l >= ra);
  int rank = 0; double maxnorm0 = 0.0;
  WN(0)
  for (int k = 0; k < steps; ++k) {
    double nq[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < MAXN; ++i) nq[i & 3] += col[i] * col[i];
    const double cn = done ? -1.0 : (nq[0] + nq[1]) + (nq[2] + nq[3]);
    const double bn = wv_max(cn);
    const unsigned long long cand = __ballot(!done && cn == bn);
    if (cand == 0ull) break;
    const int best = __ffsll((long long)cand) - 1;
    if (k == 0) maxnorm0 = bn;                                           // squared norms throughout: |x| <= 1e-9 max(1, |x0|)  <=>  |x|² <= 1e-18 max(1, |x0|²)
    if (bn <= 1e-18 * fmax(1.0, maxnorm0)) break;
    // pivot column to every lane: v_readlane when it fits the scalar registers, else through its (final) place in Vs
    double v[MAXN];
    if (MAXN <= 18) {
#pragma unroll
      for (int i = 0; i < MAXN; ++i) v[i] = qm_bcast(col[i], best);
    } else {
      qm_wave_sync();
      if (l == best) {
#pragma unroll
        for (int i = 0; i < MAXN; ++i) Vs[k * MAXN + i] = col[i];
      }
      qm_wave_sync();
#pragma unroll
      for (int i = 0; i < MAXN; ++i) v[i] = Vs[k * MAXN + i];
    }
    double alpha, vk, b2; qm_house_scalars(bn, v[0], alpha, vk, b2);
    double dq[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 1; i < MAXN; ++i) dq[i & 3] += v[i] * col[i];
    const double s = ((dq[0] + dq[1]) + (dq[2] + dq[3]) + vk * col[0]) * b2;
    if (l == best) {
      done = true; bet[k] = b2; vd[k] = vk;               // (the pivot slot Vs[k][0] still holds the unreflected value other lanes are reading)
      if (MAXN <= 18) {
#pragma unroll
        for (int i = 1; i < MAXN; ++i) Vs[k * MAXN + i] = col[i];
      }
    }
#pragma unroll
    for (int i = 1; i < MAXN; ++i) col[i - 1] = col[i] - s * v[i];      // reflect and move up one row (finished lanes carry garbage, never read)
    col[MAXN - 1] = 0.0;
    ++rank;
  }
  WN(1)
  if (n - rank > WVLD) rank = n - WVLD;                                // the level >= 1 work arrays hold at most 18 null-space directions
  const int nn = n - rank;
  qm_wave_sync();
  // backward accumulation: lane j < nn carries column j of Q2 in a window w[i] = Q2[k + i][j] that starts at k = rank
  double w[MAXN];
#pragma unroll
  for (int i = 0; i < MAXN; ++i) w[i] = (i == l && l < nn) ? 1.0 : 0.0;
  for (int k = rank - 1; k >= 0; --k) {
#pragma unroll
    for (int i = MAXN - 1; i > 0; --i) w[i] = w[i - 1];
    w[0] = 0.0;
    const double* vs = Vs + k * MAXN; double sq[4] = {0.0, 0.0, 0.0, 0.0}; double vv[MAXN];
#pragma unroll
    for (int i = 0; i < MAXN; ++i) { vv[i] = (i == 0) ? vd[k] : vs[i]; sq[i & 3] += vv[i] * w[i]; }
    const double sacc = ((sq[0] + sq[1]) + (sq[2] + sq[3])) * bet[k];
#pragma unroll
    for (int i = 0; i < MAXN; ++i) w[i] -= sacc * vv[i];
  }
  WN(2)
  if (MAXN == WNV) {                                                   // level 0: Q2 IS the new Zp (36 x nn)
    if (l < nn) {
#pragma unroll
      for (int i = 0; i < MAXN; ++i) Zp[i * nn + l] = w[i];
    }
    qm_wave_sync();
    WN(3)
    return nn;
  }
  double* Q2 = Vs + steps * MAXN;                                      // [n][nn] behind the reflectors (n <= 18 here: at most 612 doubles together)
  if (l < nn) {
#pragma unroll
    for (int i = 0; i < (MAXN <= WVLD ? MAXN : WVLD); ++i) Q2[i * nn + l] = w[i];          // rows >= n are zeros that stay inside the workspace
  }
  // Zp (36 x n) <- Zp Q2 (36 x nn), row by row in place (every lane holds its old row in registers before anything is written)
  double row[WVLD];
  if (l < WNV) {
#pragma unroll
    for (int i = 0; i < WVLD; ++i) row[i] = (i < n) ? Zp[l * n + i] : 0.0;
  }
  qm_wave_sync();
  if (l < WNV) for (int j = 0; j < nn; ++j) { double sacc = 0.0;
#pragma unroll
    for (int i = 0; i < WVLD; ++i) sacc += row[i] * Q2[i * nn + j];                            // row[i >= n] == 0
    Zp[l * nn + j] = sacc; }
  qm_wave_sync();
  WN(4)
#undef WN
  return nn;
}

__device__ __forceinline__ void tip_store(double* dst, const RbdTip& t) { for (int i = 0; i < 3; ++i) { dst[i] = t.p[i]; dst[12 + i] = t.v[i]; dst[15 + i] = t.w[i]; dst[18 + i] = t.a[i]; dst[21 + i] = t.al[i]; } for (int i = 0; i < 9; ++i) dst[3 + i] = t.R[i]; }
#define TIP_P(t) (t)
#define TIP_R(t) ((t) + 3)
#define TIP_V(t) ((t) + 12)
#define TIP_W(t) ((t) + 15)
#define TIP_A(t) ((t) + 18)
#define TIP_AL(t) ((t) + 21)

// PROF: in-kernel cycle counters of the phases (tools/wbc_prof.py; a.stop < 0 selects which set is written out).  The production instance carries none of
// them: 31 64-bit accumulators live across the whole cascade were the kernel's last private-segment spills.
template <bool PROF>
__device__ __forceinline__ void qm_wbc_body(const QmWbcArgs& a) {
  extern __shared__ double qm_smem[];
  double* S = qm_smem;
  const int b = blockIdx.x, l = threadIdx.x & 63;
  if (b >= a.B) return;
  long long tacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; long long tfine[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; long long tnull[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; long long tlast = PROF ? (long long)__builtin_readcyclecounter() : 0; long long tfl = tlast;
#define WT(k) { if (PROF) { const long long now_ = (long long)__builtin_readcyclecounter(); tacc[k] += now_ - tlast; tlast = now_; } }
#define TF(k) { if (PROF) { const long long now_ = (long long)__builtin_readcyclecounter(); tfine[k] += now_ - tfl; tfl = now_; } }
  const double* mb = qm_table(a.mb); const double* st = qm_table(a.st);
  const double* xDes = a.x_des + (size_t)b * 30; const double* uDes = a.u_des + (size_t)b * 30; const double* rbd = a.rbd + (size_t)b * QM_NRBD;
  const int mode = a.mode[b]; const double time = a.time[b];
  for (int i = l; i < WL_TOTAL; i += 64) S[i] = 0.0;
  qm_wave_sync();
  double* M = S + WL_M; double* nle = S + WL_NLE; double* Jf = S + WL_JF;
  double* gs = a.scratch + (size_t)b * WBC_SCRATCH; double* Jarm = S + WL_ZP; double* tips = gs + WS_TIPS;   // Jarm: built in the (still unused) Zp region, parked in HBM scratch for level 1
  WbcCtx C; C.mode = mode; C.nc = 0; for (int k = 0; k < 4; ++k) C.nc += mode_flag(mode, k) ? 1 : 0;
  C.mu = st[ST_WBC_FRIC]; C.nIneq = 36 + 5 * C.nc + 3 * (4 - C.nc); C.M = M; C.Jf = Jf; C.nle = nle;
  // ---- generalized coordinates of the three passes: measured (q,v), desired (qd,vd), joint-acceleration (qd, w2) ----
  double* q = S + WL_MISC; double* v = q + 24; double* qd = v + 24; double* vd = qd + 24; double* w2 = vd + 24; double* baseAcc = w2 + 24;
  if (l < 3) { q[l] = rbd[3 + l]; q[3 + l] = rbd[l]; v[l] = rbd[27 + l]; }
  if (l == 3) { const double z = rbd[0], y = rbd[1]; double sz, cz, sy, cy; qm_sincos(z, sz, cz); qm_sincos(y, sy, cy); const double wx = rbd[24], wy = rbd[25], wz = rbd[26]; const double tmp = cz * wx / cy + sz * wy / cy; v[3] = sy * tmp + wz; v[4] = -sz * wx + cz * wy; v[5] = tmp; }
  if (l >= 6 && l < 24) { q[l] = rbd[l]; v[l] = rbd[24 + l]; }
  if (l >= 32 && l < 56) qd[l - 32] = xDes[6 + (l - 32)];
  {
    double Kd[KW_LEG];                                 // SRBD quantities at the desired state (every lane: cheap, avoids a broadcast); NOT kept across the
    kin_base<true>(mb, xDes, Kd);                      // rigid-body passes (21 of them are needed again for baseAccDesired: recomputed there, 42 registers less here)
    if (l == 0) { double wr[3]; v3_cross(Kd + KW_OM, Kd + KW_RW, wr); for (int k = 0; k < 3; ++k) { vd[k] = xDes[k] + wr[k]; vd[3 + k] = Kd[KW_THD + k]; } }
  }
  if (l >= 6 && l < 24) { vd[l] = uDes[6 + l]; const double* il = a.input_last + (size_t)b * 30; w2[l] = (uDes[6 + l] - il[6 + l]) / a.period; }
  qm_wave_sync();
  if (l < 30) a.input_last[(size_t)b * 30 + l] = uDes[l];
  if (PROF) { const long long now_ = (long long)__builtin_readcyclecounter(); tnull[4] = now_ - tlast; }
  // ---- rigid-body passes: lanes 0-5 measured, 8-13 desired, 16-21 joint-acceleration; slot 0-3 legs, 4 arm, 5 root body ----
  {
    const int pass = l >> 3, slot = l & 7;
    if (pass < 3 && slot < 6) {
      const double* qq = (pass == 0) ? q : qd; const double* vv = (pass == 0) ? v : (pass == 1 ? vd : w2);
      RbdBase Bb; rbd_base(qq, vv, Bb);
      double cm = 0.0, ch[3] = {0, 0, 0}, cI[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, F[3] = {0, 0, 0}, NO[3] = {0, 0, 0};
      RbdSums* Sm = S + WL_G + (pass * 6 + slot) * 16;          // momentum sums of this slot, accumulated in place in LDS (G is free at this point and was cleared at the
                                                                // start; a private array of 16 is not promoted to registers by the compiler and would live in scratch memory)
      RbdTip tip; const bool meas = (pass == 0);
      if (slot < 5) {
        // legs (3 joints) and the arm (6 joints) share one instruction stream: slot 4 runs all six joint slots, the legs mask the last three
        const bool leg = slot < 4; const int contact = leg ? chain_to_contact(slot) : 4;
        RbdJsink Jt; Jt.rows = leg ? Jf + 3 * contact * QM_NQ : Jarm; Jt.nrows = leg ? 3 : 6; Jt.dummy = 0.0;   // Jacobian columns land in place (LDS was cleared at the start)
        RbdCompLds comp; comp.base = S + WL_RBDWS + (pass * 6 + slot); comp.stride = 18;
        rbd_chain<6, double*, RbdJsink, RbdCompLds>(mb, leg ? 3 * slot : 12, contact, qq, vv, Bb, M, nle, meas, cm, ch, cI, F, NO, Sm, tip, Jt, meas, leg ? 3 : 6, comp);
        if (meas) {
          double* Jr = Jt.rows;
#pragma unroll
          for (int r = 0; r < 3; ++r) Jr[r * QM_NQ + r] = 1.0;
#pragma unroll
          for (int k = 0; k < 3; ++k) { const double e[3] = {Bb.E[k], Bb.E[3 + k], Bb.E[6 + k]}, d[3] = {tip.p[0] - Bb.p[0], tip.p[1] - Bb.p[1], tip.p[2] - Bb.p[2]}; double cr[3]; v3_cross(e, d, cr);
#pragma unroll
            for (int r = 0; r < 3; ++r) { Jr[r * QM_NQ + 3 + k] = cr[r]; if (!leg) Jr[(3 + r) * QM_NQ + 3 + k] = e[r]; } }
        }
        if (pass < 2) tip_store(tips + 27 * (5 * pass + contact), tip);
      } else {
        const double zero3[3] = {0.0, 0.0, 0.0}; double c[3], Iw[9], vc[3], ac[3];
        body_state(mb, 0, Bb.R, Bb.p, Bb.vlin, Bb.w, zero3, Bb.al, c, Iw, vc, ac);
        add_body(mb[MB_MASS], c, Iw, vc, Bb.w, ac, Bb.al, cm, ch, cI, F, NO, Sm);
      }
      double* acc = S + WL_ACC + (pass * 6 + slot) * 20;       // cm ch(3) cI(9) F(3) NO(3)
      acc[0] = cm; for (int i = 0; i < 3; ++i) { acc[1 + i] = ch[i]; acc[13 + i] = F[i]; acc[16 + i] = NO[i]; } for (int i = 0; i < 9; ++i) acc[4 + i] = cI[i];
    }
  }
  qm_wave_sync();
  for (int i = l; i < 144; i += 64) gs[WS_JARM + i] = Jarm[i];
  Jarm = gs + WS_JARM;
  for (int i = l; i < WL_RBDWS_SIZE; i += 64) S[WL_RBDWS + i] = 0.0;      // the cascade expects its work arrays as the kernel's initial clear left them
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");    // the tip records and the arm Jacobian go through HBM scratch (same CU: no L2 maintenance needed)
  qm_wave_sync();
  if (PROF) { const long long now_ = (long long)__builtin_readcyclecounter(); tnull[8] = now_ - tlast; }
  RbdBase Bm; rbd_base(q, v, Bm);                        // measured root state (every lane)
  // base block of M and base rows of nle from the whole-tree composite (lane d = base dof)
  {
    double cm = 0.0, ch[3] = {0, 0, 0}, cI[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, F[3] = {0, 0, 0}, NO[3] = {0, 0, 0};
    for (int s2 = 0; s2 < 6; ++s2) { const double* acc = S + WL_ACC + s2 * 20; cm += acc[0]; for (int i = 0; i < 3; ++i) { ch[i] += acc[1 + i]; F[i] += acc[13 + i]; NO[i] += acc[16 + i]; } for (int i = 0; i < 9; ++i) cI[i] += acc[4 + i]; }
    if (l < 6) {
      double w[3], vO[3]; rbd_S_base(Bm, l, w, vO);
      double wh[3], hv3[3], Iw_[3]; v3_cross(w, ch, wh); v3_cross(ch, vO, hv3); m3_mulv(cI, w, Iw_);
      const double f[3] = {cm * vO[0] + wh[0], cm * vO[1] + wh[1], cm * vO[2] + wh[2]}, nO[3] = {Iw_[0] + hv3[0], Iw_[1] + hv3[1], Iw_[2] + hv3[2]};
      for (int e = 0; e < 6; ++e) { double w2b[3], vO2[3]; rbd_S_base(Bm, e, w2b, vO2); M[e * QM_NQ + l] = w2b[0] * nO[0] + w2b[1] * nO[1] + w2b[2] * nO[2] + vO2[0] * f[0] + vO2[1] * f[1] + vO2[2] * f[2]; }
      nle[l] = w[0] * NO[0] + w[1] * NO[1] + w[2] * NO[2] + vO[0] * F[0] + vO[1] * F[1] + vO[2] * F[2];
    }
  }
  if (l == 0) { for (int i = 0; i < 9; ++i) { S[WL_BM + i] = Bm.E[i]; S[WL_BM + 9 + i] = Bm.R[i]; } for (int i = 0; i < 3; ++i) S[WL_BM + 18 + i] = Bm.al[i]; }
  // ---- baseAccDesired (WbcBase.cpp:215-225; SURVEY.md a14 aliasing: A_b SRBD, Adot & A_j full CMM, true COM) ----
  const double* tipsM = tips; const double* tipsD = tips + 27 * 5;
  if (l == 0) {
    double Sd[16], Sa[16]; for (int i = 0; i < 16; ++i) { Sd[i] = 0.0; Sa[i] = 0.0; }
    for (int s2 = 0; s2 < 6; ++s2) for (int i = 0; i < 16; ++i) { Sd[i] += S[WL_G + (6 + s2) * 16 + i]; Sa[i] += S[WL_G + (12 + s2) * 16 + i]; }
    const double m = mb[MB_ROBOTMASS]; const double com[3] = {Sd[1] / m, Sd[2] / m, Sd[3] / m};
    double rate[6] = {0.0, 0.0, -9.81 * m, 0.0, 0.0, 0.0};
    for (int k = 0; k < 4; ++k) { const double* pf = TIP_P(tipsD + 27 * k); const double r[3] = {pf[0] - com[0], pf[1] - com[1], pf[2] - com[2]}; double t[3]; v3_cross(r, uDes + 3 * k, t); for (int i = 0; i < 3; ++i) { rate[i] += uDes[3 * k + i]; rate[3 + i] += t[i]; } }
    double cF[3], cH[3]; v3_cross(com, Sd + 10, cF); v3_cross(com, Sa + 4, cH);
    for (int i = 0; i < 3; ++i) { rate[i] -= Sd[10 + i] + Sa[4 + i]; rate[3 + i] -= (Sd[13 + i] - cF[i]) + (Sa[7 + i] - cH[i]); }
    const double ra3[3] = {rate[3], rate[4], rate[5]}; double wdd[3], thdd[3], t[3];
    double Kd[KW_LEG]; kin_base<true>(mb, xDes, Kd);
    m3_mulv(Kd + KW_IINV, ra3, wdd); m3_mulv(Kd + KW_EINV, wdd, thdd); v3_cross(Kd + KW_RW, wdd, t);
    for (int i = 0; i < 3; ++i) { baseAcc[i] = rate[i] / m - t[i]; baseAcc[3 + i] = thdd[i]; }
  }
  qm_wave_sync();
  if (PROF && a.stop == 1) return;
  WT(0)
  // ---- cascade ----
  double* A = S + WL_AZ; double* bb = S + WL_BB; double* AZ = S + WL_AZ; double* Zp = S + WL_ZP; double* x = S + WL_X; double* z = S + WL_Z; double* zn = S + WL_ZN; double* p = S + WL_P;
  double* f0 = S + WL_F0; double* w0 = S + WL_W0; double* fb = S + WL_FB; double* Dz = S + WL_DZ; double* Dp = S + WL_DP; double* tau = S + WL_TAU; double* g0 = S + WL_G0RHS; double* G = S + WL_G;
  double* Zz = S + WL_ZZ; double* Zpv = S + WL_ZPV; double* lam = S + WL_LAM;
  if (l < 18) { const double tmax = a.mb[MB_TAUMAX + ((l < 12) ? (l % 3) : l)];      // torque limits: the first leg's three for every leg (WbcBase.cpp:565-578), the arm's own; lane-indexed: a plain global load
                f0[l] = tmax - nle[6 + l]; f0[18 + l] = tmax + nle[6 + l]; }
  qm_wave_sync();
  int nz = WNV; int status0 = 0, status1 = 0, status2 = 0;      // per-level qp status (three scalars: an array indexed by `level` would live in scratch memory)
  auto get_status = [&](int lv) { return lv == 0 ? status0 : (lv == 1 ? status1 : status2); };
  auto set_status = [&](int lv, int v) { status0 = (lv == 0) ? v : status0; status1 = (lv == 1) ? v : status1; status2 = (lv == 2) ? v : status2; };
  for (int level = 0; level < 3; ++level) {
    // ---- the level's equality task ----
    A = (level == 0) ? AZ : S + WL_A;                       // level 0: Zp = I, the task rows are A Zp
    for (int idx = l; idx < WMAXA * WNV; idx += 64) A[idx] = 0.0;
    qm_wave_sync();
    int ra = 0;
    if (level == 0) {
      for (int idx = l; idx < 6 * 36; idx += 64) { const int r = idx / 36, k = idx - 36 * r; A[r * WNV + k] = (k < 24) ? M[r * 24 + k] : -Jf[(k - 24) * 24 + r]; }
      if (l < 6) bb[l] = -nle[l];
      ra = 6;
      for (int k = 0; k < 4; ++k) if (C.fl(k)) { for (int idx = l; idx < 72; idx += 64) { const int r = idx / 24, c2 = idx - 24 * r; A[(ra + r) * WNV + c2] = Jf[(3 * k + r) * 24 + c2]; } if (l < 3) bb[ra + l] = -TIP_A(tipsM + 27 * k)[l]; ra += 3; }
      for (int k = 0; k < 4; ++k) if (!C.fl(k)) { if (l < 3) { A[(ra + l) * WNV + 24 + 3 * k + l] = 1.0; bb[ra + l] = 0.0; } ra += 3; }
    } else if (level == 1) {
      const bool init = (a.variant == 0 && time < 10.0);
      if (init) { if (l < 6) { A[l * WNV + 18 + l] = 1.0; bb[l] = st[ST_KP_ARM_J + l] * (qd[18 + l] - q[18 + l]) + st[ST_KD_ARM_J + l] * (vd[18 + l] - v[18 + l]); } ra = 6; }
      else {
        if (l == 0) { A[2] = 1.0; bb[0] = baseAcc[2] + st[ST_KP_BASE_H] * (qd[2] - q[2]) + st[ST_KD_BASE_H] * (vd[2] - v[2]); }
        ra = 1;
        if (l == 0) {   // base angular
          const double* BmE = S + WL_BM; const double* BmR = BmE + 9; const double* Bmal = BmE + 18;
          double wMeas[3], wDes[3]; const double thm[3] = {v[3], v[4], v[5]}, thdv[3] = {vd[3], vd[4], vd[5]}; m3_mulv(BmE, thm, wMeas); m3_mulv(BmE, thdv, wDes);
          double Rdes[9]; rot_zyx<true>(qd[3], qd[4], qd[5], Rdes); double err[3]; dev_rot_error(Rdes, BmR, err);
          double acc[3]; { const double tdd[3] = {baseAcc[3], baseAcc[4], baseAcc[5]}; m3_mulv(BmE, tdd, acc); const double zz[3] = {0.0, 0.0, 1.0}; double t0[3]; v3_cross(zz, wDes, t0);
            const double c1[3] = {BmE[1], BmE[4], BmE[7]}, c2[3] = {BmE[2], BmE[5], BmE[8]}; double t1[3]; v3_cross(c1, c2, t1); for (int i = 0; i < 3; ++i) acc[i] += thdv[0] * t0[i] + thdv[1] * thdv[2] * t1[i]; }
          for (int r = 0; r < 3; ++r) { for (int k = 0; k < 3; ++k) A[(ra + r) * WNV + 3 + k] = BmE[3 * r + k]; bb[ra + r] = acc[r] + st[ST_KP_BASE_ANG] * err[r] + st[ST_KD_BASE_ANG] * (wDes[r] - wMeas[r]) - Bmal[r]; }
        }
        ra += 3;
        if (a.variant == 0) {
          const double* aM = tipsM + 27 * 4; const double* aD = tipsD + 27 * 4;
          for (int idx = l; idx < 72; idx += 64) { const int r = idx / 24, k = idx - 24 * r; A[(ra + r) * WNV + k] = Jarm[r * 24 + k]; A[(ra + 3 + r) * WNV + k] = (k >= 3 && k < 6) ? 0.0 : Jarm[(3 + r) * 24 + k]; }
          if (l < 3) bb[ra + l] = st[ST_KP_EE_LIN + l] * (TIP_P(aD)[l] - TIP_P(aM)[l]) + st[ST_KD_EE_LIN + l] * (TIP_V(aD)[l] - TIP_V(aM)[l]) - TIP_A(aM)[l];
          if (l == 0) { double err[3]; dev_rot_error(TIP_R(aD), TIP_R(aM), err); for (int r = 0; r < 3; ++r) bb[ra + 3 + r] = st[ST_KP_EE_ANG + r] * err[r] + st[ST_KD_EE_ANG + r] * (-TIP_W(aM)[r]) - (TIP_AL(aM)[r] - S[WL_BM + 18 + r]); }
          ra += 6;
        } else {
          if (l < 2) { A[(ra + l) * WNV + l] = 1.0; bb[ra + l] = baseAcc[l] + st[ST_KP_BASE_LIN] * (qd[l] - q[l]) + st[ST_KD_BASE_LIN] * (vd[l] - v[l]); }
          ra += 2;
        }
        for (int k = 0; k < 4; ++k) if (!C.fl(k)) {   // swing legs, x100
          const double* fM = tipsM + 27 * k; const double* fD = tipsD + 27 * k;
          for (int idx = l; idx < 72; idx += 64) { const int r = idx / 24, c2 = idx - 24 * r; A[(ra + r) * WNV + c2] = 100.0 * Jf[(3 * k + r) * 24 + c2]; }
          if (l < 3) bb[ra + l] = 100.0 * (st[ST_KP_SWING] * (TIP_P(fD)[l] - TIP_P(fM)[l]) + st[ST_KD_SWING] * (TIP_V(fD)[l] - TIP_V(fM)[l]) - TIP_A(fM)[l]);
          ra += 3;
        }
      }
    } else {
      if (l < 12) { A[l * WNV + 24 + l] = 1.0; bb[l] = uDes[l]; }
      ra = 12;
      if (a.variant == 0) { if (l < 2) { A[(ra + l) * WNV + l] = 1.0; bb[ra + l] = baseAcc[l] + st[ST_KP_BASE_LIN] * (qd[l] - q[l]) + st[ST_KD_BASE_LIN] * (vd[l] - v[l]); } ra += 2; }
    }
    qm_wave_sync();
    // ---- AZ = A Zp, g0 = [b − A xp; 0] ----
    const int n = nz;
    if (level > 0) for (int idx = l; idx < ra * n; idx += 64) { const int r = idx / n, k = idx - r * n; double s4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int c2 = 0; c2 < WNV; ++c2) s4[c2 & 3] += A[r * WNV + c2] * Zp[c2 * n + k];
      AZ[r * WNV + k] = (s4[0] + s4[1]) + (s4[2] + s4[3]); }
    for (int r = l; r < ra + n; r += 64) { double s = 0.0; if (r < ra) { double s4[4] = {bb[r], 0.0, 0.0, 0.0};
#pragma unroll
        for (int c2 = 0; c2 < WNV; ++c2) s4[c2 & 3] -= A[r * WNV + c2] * x[c2];
        s = (s4[0] + s4[1]) + (s4[2] + s4[3]); } g0[r] = s; }
    if (l < WNV) z[l] = 0.0;
    qm_wave_sync();
    const int rows0 = ra + n;
    WT(1)
    if (level == 0) {
      // own (soft) inequality rows: Newton on the active set with exact line search (phi is convex piecewise quadratic)
      wv_d0_apply(C, x, tau, Dz);
      if (l < C.nIneq) fb[l] = f0[l] - Dz[l];
      qm_wave_sync();
      unsigned long long actmask = __ballot((l < C.nIneq) && (0.0 - fb[l] > 0.0));
      // The task rows do not change over the iterations: [sqrt(rho) I; A | b] is factored ONCE (packed triangle [R0 | c0] in G, a copy kept in the region the levels >= 1
      // use for their own factors), and an iteration only folds its active soft rows into a fresh copy — 6 dense rows per Householder step instead of 24.  (Rounds 1–3 factored
      // the whole stack in every iteration; with no active row — the usual first iteration — the solve is the back substitution alone.)
      double* G0 = S + WL_V;                              // [702] [R0 | c0]
      for (int idx = l; idx < 702; idx += 64) G[idx] = 0.0;
      qm_wave_sync();
      if (l < n) G[wv_tidx(l, l, n, 0)] = sqrt(WRHO);
      qm_wave_sync();
      if (ra <= 18) {
        double d[18];
#pragma unroll
        for (int r = 0; r < 18; ++r) d[r] = (r < ra && l <= n) ? ((l == n) ? g0[r] : AZ[r * WNV + l]) : 0.0;
        rq_house_tri<18>(d, G, 0, n, S + WL_HV);
      } else {
        double d[WMAXA];
#pragma unroll
        for (int r = 0; r < WMAXA; ++r) d[r] = (r < ra && l <= n) ? ((l == n) ? g0[r] : AZ[r * WNV + l]) : 0.0;
        rq_house_tri<WMAXA>(d, G, 0, n, S + WL_HV);
      }
      for (int idx = l; idx < 702; idx += 64) G0[idx] = G[idx];
      qm_wave_sync();
      bool dirty = false;                                 // G holds something else than [R0 | c0]
      WT(2)
      int it = 0;
      for (; it < 100; ++it) {
        int* alist = (int*)(S + WL_WLIST);
        int na = __popcll(actmask); if (na > WMAXACT) na = WMAXACT;
        { const int rank = __popcll(actmask & ((1ull << l) - 1ull)); if (((actmask >> l) & 1ull) && rank < WMAXACT) alist[rank] = l; }
        if (dirty) { for (int idx = l; idx < 702; idx += 64) G[idx] = G0[idx]; dirty = false; }
        qm_wave_sync();
        if (na > 0) {
          if (na <= 6) {
            // the usual case (a handful of active soft rows): pivot column broadcast by v_readlane
            double d[6];
#pragma unroll
            for (int q2 = 0; q2 < 6; ++q2) { double v = 0.0; if (q2 < na && l <= n) { const int i = alist[q2]; v = (l == n) ? fb[i] : wbc_d0_entry(C, i, l); } d[q2] = v; }
            rq_house_tri<6>(d, G, 0, n, S + WL_HV);
          } else {
            double d[WMAXACT];
#pragma unroll
            for (int q2 = 0; q2 < WMAXACT; ++q2) { double v = 0.0; if (q2 < na && l <= n) { const int i = alist[q2]; v = (l == n) ? fb[i] : wbc_d0_entry(C, i, l); } d[q2] = v; }
            rq_house_tri<WMAXACT>(d, G, 0, n, S + WL_HV);
          }
          dirty = true;
        }
        wv_backsub_tri<WNV>(G, 0, n, zn);
        if (l < n) p[l] = zn[l] - z[l];
        qm_wave_sync();
        WT(3)
        wv_Z_times(Zp, n, z, Zz); wv_Z_times(Zp, n, p, Zpv);
        wv_d0_apply(C, Zz, tau, Dz); wv_d0_apply(C, Zpv, tau, Dp);
        double c0p = 0.0, c1p = 0.0;   // smooth part of dphi: c0 + a c1
        for (int r = l; r < rows0; r += 64) { double gz = -g0[r], gp = 0.0;
          if (r < ra) { double z2[2] = {0.0, 0.0}, p2[2] = {0.0, 0.0};      // n = 36 here (level 0)
#pragma unroll
            for (int k = 0; k < WNV; ++k) { const double ak = AZ[r * WNV + k]; z2[k & 1] += ak * z[k]; p2[k & 1] += ak * p[k]; }
            gz += z2[0] + z2[1]; gp = p2[0] + p2[1]; }
          else { gz += sqrt(WRHO) * z[r - ra]; gp = sqrt(WRHO) * p[r - ra]; } c0p += gz * gp; c1p += gp * gp; }
        const double c0 = wv_sum(c0p), c1 = wv_sum(c1p);
        WT(4)
        // exact line search on the convex piecewise-quadratic phi (lane i carries soft row i): dphi is piecewise LINEAR and non-decreasing, so Newton's iteration on it
        // — value and slope from one pass over the rows, kept inside the bracket [lo, hi] — lands on the root exactly as soon as a step stays within one linear piece
        // (the set of rows with a positive residual does not change); rounds 1–3 bisected the bracket down to one ulp, ≈ 52 wave reductions
        const bool mine = (l < C.nIneq); const double myDz = mine ? Dz[l] : 0.0, myDp = mine ? Dp[l] : 0.0, myfb = mine ? fb[l] : 1.0;
        auto dphi = [&](double a, double& f, double& sl, unsigned long long& msk) { const double vv = myDz + a * myDp - myfb; const bool on = vv > 0.0; msk = __ballot(on);
          f = c0 + a * c1 + wv_sum(on ? vv * myDp : 0.0); sl = c1 + wv_sum(on ? myDp * myDp : 0.0); };
        double al = 1.0;
        { double f, sl; unsigned long long m0; dphi(1.0, f, sl, m0);
          if (f > 0.0) {
            double lo = 0.0, hi = 1.0;
            for (int bi = 0; bi < 200; ++bi) {
              double an = al - f / sl; bool newton = true;
              if (!(an > lo && an < hi)) { an = 0.5 * (lo + hi); newton = false; }
              if (an == lo || an == hi) { al = an; break; }                      // the bracket cannot shrink any more
              double fn, sn; unsigned long long mn; dphi(an, fn, sn, mn);
              if (fn > 0.0) hi = an; else lo = an;
              const bool done = (newton && mn == m0) || fn == 0.0;
              al = an; f = fn; sl = sn; m0 = mn;
              if (done) break;
            }
          }
        }
        WT(5)
        const double pn = wv_max((l < n) ? fabs(al * p[l]) : 0.0);
        if (l < n) z[l] += al * p[l];
        const unsigned long long nm = __ballot((l < C.nIneq) && (Dz[l] + al * Dp[l] - fb[l] > 0.0));
        const bool same = (nm == actmask); actmask = nm;
        qm_wave_sync();
        WT(6)
        if (same && al == 1.0) break;
        const double zs = fmax(1.0, wv_max((l < n) ? fabs(z[l]) : 0.0));
        if (pn <= 1e-12 * zs) break;
      }
      if (it >= 100) status0 = 1;
      wv_Z_times(Zp, n, z, Zz); wv_d0_apply(C, Zz, tau, Dz);
      if (l < C.nIneq) w0[l] = fmax(0.0, Dz[l] - fb[l]);
      qm_wave_sync();
    } else {
      // hard rows of level 0: primal active set (Nocedal & Wright 16.3) from the feasible z = 0.  Factor G0 = [A Zp; sqrt(rho) I | g0]
      // = Q [R | c] once (|G0 z − g0|² = |R z − c|² + const) and form DZ = D0 Zp once; an iteration then only touches n x n data.
      wv_d0_apply(C, x, tau, Dz);
      if (l < C.nIneq) fb[l] = f0[l] - Dz[l] + w0[l];
      for (int idx = l; idx < n * (n + 1); idx += 64) { const int r = idx / (n + 1), k = idx - r * (n + 1); G[r * WTLD + k] = (r == k) ? sqrt(WRHO) : 0.0; }
      qm_wave_sync();
      {
        double d[WMAXA];
#pragma unroll
        for (int r = 0; r < WMAXA; ++r) d[r] = (r < ra && l <= n) ? ((l == n) ? g0[r] : AZ[r * WNV + l]) : 0.0;
        rq_house_tri<WMAXA>(d, G, WTLD, n, S + WL_HV);
      }
      tq_init(S, n);                                     // [R | c] stays in the first 18 x 19 block of G and is updated in place from here on
      double dz[WVLD];                                   // row l of D0 Zp lives in the registers of lane l
#pragma unroll
      for (int k = 0; k < WVLD; ++k) dz[k] = 0.0;
      {
        // torque rows (+/- the same 18 x 36 block [M_j | −J_jᵀ]) times Zp on the matrix core: P = Zᵀ Y, Z[k][i] = D0[i][k], Y = Zp
        const int g4 = l >> 4, c4 = l & 15;
        qm_d4 Zt[3][2], Yz[3][2], Pz[2][2];
#pragma unroll
        for (int K = 0; K < 3; ++K)
#pragma unroll
          for (int J = 0; J < 2; ++J)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int k = 16 * K + g4 + 4 * r, i = 16 * J + c4;
              Zt[K][J][r] = (k < WNV && i < 18) ? ((k < 24) ? M[(6 + i) * 24 + k] : -Jf[(k - 24) * 24 + 6 + i]) : 0.0;
              Yz[K][J][r] = (k < WNV && i < n) ? Zp[k * n + i] : 0.0;
            }
        qm_frag_zero<2, 2>(Pz);
        qm_gemm_tn<3, 2, 2>(Zt, Yz, Pz, 0, 9, false);
        double* tmp = S + WL_EROWS;                       // [18][18] hand-over: fragments -> one row per lane
        qm_wave_sync();
#pragma unroll
        for (int I = 0; I < 2; ++I)
#pragma unroll
          for (int J = 0; J < 2; ++J)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const int row = 16 * I + g4 + 4 * r, col = 16 * J + c4; if (row < 18 && col < WVLD) tmp[row * WVLD + col] = Pz[I][J][r]; }
        qm_wave_sync();
        if (l < 36) {
          const int rr = (l < 18) ? l : l - 18; const double sg = (l < 18) ? 1.0 : -1.0;
#pragma unroll
          for (int k = 0; k < WVLD; ++k) dz[k] = (k < n) ? sg * tmp[rr * WVLD + k] : 0.0;
        } else if (l < 36 + 5 * C.nc) {                   // friction pyramid rows touch one force triple
          const int k0 = 24 + 3 * C.contactOf((l - 36) / 5);
          for (int a3 = 0; a3 < 3; ++a3) { const double e = wbc_d0_entry(C, l, k0 + a3); if (e != 0.0) {
#pragma unroll
            for (int k = 0; k < WVLD; ++k) if (k < n) dz[k] += e * Zp[(k0 + a3) * n + k]; } }
        }
        qm_wave_sync();
      }
      qm_wave_sync();
      WT(8)
      int* Wi = (int*)(S + WL_WLIST);                    // working-set list lives in LDS (wave-uniform reads)
      unsigned long long wmask = 0ull; int nw = 0; int it = 0; bool degenerate = false, vertex = false; double pscale = 0.0;
      bool atopt = false;                                // z is already the optimum on the working set (a full step has just been taken): nothing to solve, go to the drop test
      for (; it < 100; ++it) {
        double pn = 0.0, zs = 1.0;
        if (!atopt) {
          WT(10)
          TF(8)
          tq_solve(S, n, nw, zn);
          TF(0)
          if (l < n) p[l] = zn[l] - z[l];
          qm_wave_sync();
          pn = wv_max((l < n) ? fabs(p[l]) : 0.0); zs = fmax(1.0, wv_max((l < n) ? fabs(z[l]) : 0.0));
          pscale = fmax(pscale, pn);
          WT(9)
        }
#ifdef QM_WBC_TRACE
        if (l == 0) { printf("L%d it %d nw %d pn %.3e zs %.3e pscale %.3e W:", level, it, nw, pn, zs, pscale); for (int q2 = 0; q2 < nw; ++q2) printf(" %d", Wi[q2]); printf("\n"); }
#endif
        if (atopt || pn <= 1e-9 * fmax(zs, pscale) || vertex) {
          vertex = false; atopt = false;
          // stationary on the working set: drop a row with a negative multiplier (most negative; lowest constraint index after a degenerate step — Bland)
          TF(8)
          tq_mult(S, n, nw);                              // multipliers at zn, the solution of the last solve (== z here)
          TF(1)
          const double mylam = (l < nw) ? lam[l] : 0.0; const double lscale = fmax(1.0, wv_max(fabs(mylam)));
          const bool cand = (l < nw) && (mylam < -1e-9 * lscale);
          int worst;
          // (index minima as DPP maxima of small integers held in doubles: a ds_bpermute butterfly costs several times more on a lone wave)
          if (degenerate) { const int m = (int)wv_max(cand ? (double)(4095 - (Wi[(l < nw) ? l : 0] * 64 + l)) : -1.0); worst = (m < 0) ? -1 : ((4095 - m) & 63); }
          else { const double lmin = -wv_max(cand ? -mylam : -1e300); const int m = (int)wv_max((cand && mylam == lmin) ? (double)(63 - l) : -1.0); worst = (m < 0) ? -1 : 63 - m; }
          if (worst < 0) break;
          wmask &= ~(1ull << Wi[worst]);
          TF(8)
          tq_drop(S, n, nw, worst);
          TF(2)
          const int nxt = (l + 1 < nw) ? Wi[l + 1] : 0;
          qm_wave_sync();
          if (l >= worst && l + 1 < nw) Wi[l] = nxt;
          --nw;
          qm_wave_sync();
          WT(9)
        } else {
          double aa = 1e300;
          if (l < C.nIneq && !((wmask >> l) & 1ull)) {
            double dp = 0.0, dzz = 0.0;
#pragma unroll
            for (int k = 0; k < WVLD; ++k) { dp += dz[k] * p[k]; dzz += dz[k] * z[k]; }            // dz[k >= n] == 0, p / z hold finite leftovers there
            if (dp > 1e-10 * fmax(1.0, pn)) aa = fmax(0.0, (fb[l] - dzz) / dp);
          }
          const double amin = -wv_max(-aa);
          double al = 1.0; int block = -1;
          if (amin < 1.0) { al = amin; block = 63 - (int)wv_max((aa == amin) ? (double)(63 - l) : -1.0); }
          qm_wave_sync();
          if (l < n) z[l] += al * p[l];
#ifdef QM_WBC_TRACE
          if (l == 0) printf("   step al %.6e block %d\n", al, block);
#endif
          degenerate = (al <= 1e-12);
          atopt = (block < 0);                            // full step: z = zn; solving the same rows again would return p = 0 and the multipliers of this solve
          if (block >= 0) {
            if (nw < n && nw < WMAXACT) {
              if (l == block) {
#pragma unroll
                for (int k = 0; k < WVLD; ++k) S[WL_TQ_ROW + k] = dz[k];      // dz is zero beyond n: no per-index bound (uniform branches would serialise the LDS traffic)
              }
              if (l == 0) Wi[nw] = block;
              qm_wave_sync();
              WT(10)
              TF(8)
              tq_append<PROF>(S, n, nw, fb[block], tfine);
              TF(3)
              WT(9)
              wmask |= (1ull << block); ++nw;
            }
            else if (nw >= n) { if (al <= 1e-12) vertex = true; }   // n rows are active already (the working-set rows are numerically dependent, else p would vanish): the set
                                                                 // cannot grow beyond the dimension.  A step blocked at once means z is a degenerate vertex: decide by the multipliers
                                                                 // (drop by Bland's rule).  After a partial step the same rows are solved again and the remainder ends up here.
            else { set_status(level, 2); qm_wave_sync(); break; }
          }
          qm_wave_sync();
        }
      }
      if (it >= 100 && get_status(level) == 0) set_status(level, 1);
      if (PROF && a.stop == -4 && l == 0) gs[WS_TIME + 12 + level] = (double)it;      // profiling: active-set iterations of this level
    }
    wv_Z_times(Zp, n, z, Zz);
    if (l < WNV) { x[l] += Zz[l]; if (PROF && a.dbg) a.dbg[(size_t)b * WBC_DBG_SIZE + 126 + level * WNV + l] = x[l]; }
    qm_wave_sync();
    if (PROF && a.stop == 2 + level) return;
    WT(10)
    if (level < 2) nz = (level == 0) ? wv_null_space<WNV, PROF>(S, ra, n, tnull) : wv_null_space<WVLD, PROF>(S, ra, n, tnull + 5);
    WT(7)
    if (level > 0 && get_status(level) == 0 && get_status(level - 1) != 0) set_status(level, get_status(level - 1));
  }
  // ---- updateCmd (WbcBase.cpp:548-563) ----
  wv_d0_apply(C, x, tau, Dz);
  double* out = a.out + (size_t)b * QM_NWBC_OUT;
  if (l < WNV) out[l] = x[l];
  if (l < 18) out[WNV + l] = tau[l] + nle[6 + l];
  if (l < 3) a.qp_status[b * 3 + l] = get_status(l);
  WT(11)
  if (PROF && a.stop < 0 && l == 0) for (int k = 0; k < 12; ++k) gs[WS_TIME + k] = (double)tacc[k];
  if (PROF && a.stop == -2 && l == 0) for (int k = 0; k < 9; ++k) gs[WS_TIME + k] = (double)tfine[k];
  if (PROF && a.stop == -3 && l == 0) for (int k = 0; k < 10; ++k) gs[WS_TIME + k] = (double)tnull[k];
#undef WT
#undef TF
  if (PROF && a.dbg) {
    double* d = a.dbg + (size_t)b * WBC_DBG_SIZE;
    if (l < 24) { d[l] = q[l]; d[24 + l] = v[l]; d[48 + l] = qd[l]; d[72 + l] = vd[l]; d[102 + l] = nle[l]; }
    if (l < 6) d[96 + l] = baseAcc[l];
    for (int idx = l; idx < 576; idx += 64) d[234 + idx] = M[idx];
    for (int idx = l; idx < 288; idx += 64) d[810 + idx] = Jf[idx];
    if (l < 12) d[1098 + l] = TIP_A(tipsM + 27 * (l / 3))[l % 3];
  }
}
__global__ void QM_UNPAIRED_LDS __launch_bounds__(WBC_BLOCK) qm_wbc_kernel(QmWbcArgs a) { qm_wbc_body<false>(a); }
__global__ void QM_UNPAIRED_LDS __launch_bounds__(WBC_BLOCK) qm_wbc_prof_kernel(QmWbcArgs a) { qm_wbc_body<true>(a); }      // profiling only (a.stop < 0)
