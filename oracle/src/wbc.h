// oracle/src/wbc.h — TEST INFRASTRUCTURE (CPU oracle). Hierarchical whole-body controller of
// qm_wbc (WbcBase.cpp, HoQp.cpp, HierarchicalWbc.cpp, Task.h) restated; SURVEY.md §8 a13–a19.
// Rigid-body quantities that the reference takes from Pinocchio (crba, nonLinearEffects, frame
// Jacobians and their time variation, dccrba) are restated from first principles: body Jacobians +
// Lagrange's equations, with time derivatives taken by forward-mode AD along (q, v).  The product
// uses recursive algorithms instead, so the two are independent.  PARITY UNPINNED.
#pragma once
#include "ocp.h"
#include <cstdio>
#include <cstdlib>

// 6x24 LOCAL_WORLD_ALIGNED Jacobian [lin; ang] of a point rigidly attached to body b
template <class T> inline void pointJacobian(const Model& M, const Kin<T>& k, const T* q, int b, const V3<T>& pt, T J[6][QM_NQ]) {
  for (int i = 0; i < 6; ++i) for (int j = 0; j < QM_NQ; ++j) J[i][j] = T(0.0);
  for (int i = 0; i < 3; ++i) J[i][i] = T(1.0);
  M3<T> E = eulerZyxE(q[3], q[4]);
  for (int c = 0; c < 3; ++c) {
    V3<T> e = v3<T>(E(0, c), E(1, c), E(2, c)); V3<T> l = cross(e, pt - k.p[0]);
    for (int i = 0; i < 3; ++i) { J[i][3 + c] = l[i]; J[3 + i][3 + c] = e[i]; }
  }
  for (int bb = b; bb > 0; bb = M.parent(bb - 1)) {
    const int j = bb - 1; V3<T> aw = k.R[bb] * v3d<T>(M.axis(j)); V3<T> l = cross(aw, pt - k.p[bb]);
    for (int i = 0; i < 3; ++i) { J[i][6 + j] = l[i]; J[3 + i][6 + j] = aw[i]; }
  }
}
template <class T> inline void frameJacobian(const Model& M, const T* q, int f, T J[6][QM_NQ]) {   // f = 0..4, or -1 for the "base" frame
  Kin<T> k; forwardKinematics(M, q, k);
  if (f < 0) pointJacobian(M, k, q, 0, k.p[0], J); else pointJacobian(M, k, q, M.fparent(f), k.fp[f], J);
}
// joint-space inertia M(q) (pinocchio::crba, symmetrised at WbcBase.cpp:153-155)
template <class T> inline void massMatrix(const Model& M, const T* q, T Mq[QM_NQ][QM_NQ]) {
  Kin<T> k; forwardKinematics(M, q, k);
  for (int i = 0; i < QM_NQ; ++i) for (int j = 0; j < QM_NQ; ++j) Mq[i][j] = T(0.0);
  for (int b = 0; b < QM_NB; ++b) {
    V3<T> c = k.p[b] + k.R[b] * v3d<T>(M.com(b));
    T J[6][QM_NQ]; pointJacobian(M, k, q, b, c, J);
    M3<T> Iw = k.R[b] * m3d<T>(M.inertia(b)) * transpose(k.R[b]);
    T IJ[3][QM_NQ];
    for (int r = 0; r < 3; ++r) for (int j = 0; j < QM_NQ; ++j) IJ[r][j] = Iw(r, 0) * J[3][j] + Iw(r, 1) * J[4][j] + Iw(r, 2) * J[5][j];
    for (int i = 0; i < QM_NQ; ++i) for (int j = 0; j < QM_NQ; ++j) {
      T s = (J[0][i] * J[0][j] + J[1][i] * J[1][j] + J[2][i] * J[2][j]) * M.mass(b);
      s += J[3][i] * IJ[0][j] + J[4][i] * IJ[1][j] + J[5][i] * IJ[2][j];
      Mq[i][j] += s;
    }
  }
}
template <class T> inline T potentialEnergy(const Model& M, const T* q) {
  Kin<T> k; forwardKinematics(M, q, k); T V = T(0.0);
  for (int b = 0; b < QM_NB; ++b) { V3<T> c = k.p[b] + k.R[b] * v3d<T>(M.com(b)); V += c[2] * (9.81 * M.mass(b)); }
  return V;
}
// centroidal momentum matrix about the true COM, rows [lin; ang] (pinocchio ccrba/dccrba data.Ag), and COM
template <class T> inline void centroidalMomentumMatrix(const Model& M, const T* q, T A[6][QM_NQ], V3<T>& com) {
  Kin<T> k; forwardKinematics(M, q, k);
  V3<T> cb[QM_NB]; com = v3<T>(T(0.0), T(0.0), T(0.0));
  for (int b = 0; b < QM_NB; ++b) { cb[b] = k.p[b] + k.R[b] * v3d<T>(M.com(b)); com = com + scale(cb[b], M.mass(b)); }
  com = scale(com, 1.0 / M.robotMass());
  for (int i = 0; i < 6; ++i) for (int j = 0; j < QM_NQ; ++j) A[i][j] = T(0.0);
  for (int b = 0; b < QM_NB; ++b) {
    T J[6][QM_NQ]; pointJacobian(M, k, q, b, cb[b], J);
    M3<T> Iw = k.R[b] * m3d<T>(M.inertia(b)) * transpose(k.R[b]);
    M3<T> S = skew(cb[b] - com);
    for (int j = 0; j < QM_NQ; ++j) {
      for (int r = 0; r < 3; ++r) {
        A[r][j] += J[r][j] * M.mass(b);
        A[3 + r][j] += (S(r, 0) * J[0][j] + S(r, 1) * J[1][j] + S(r, 2) * J[2][j]) * M.mass(b) + Iw(r, 0) * J[3][j] + Iw(r, 1) * J[4][j] + Iw(r, 2) * J[5][j];
      }
    }
  }
}

// rotation error log(R_lhs R_rhsᵀ)  [upstream ocs2_robotic_tools rotationErrorInWorld / rotationMatrixToRotationVector]
inline void rotationErrorInWorld(const M3<double>& Rl, const M3<double>& Rr, double err[3]) {
  M3<double> R = Rl * transpose(Rr);
  const double tr = R(0, 0) + R(1, 1) + R(2, 2);
  const double v[3] = {R(2, 1) - R(1, 2), R(0, 2) - R(2, 0), R(1, 0) - R(0, 1)};
  const double tmp = 0.5 * (tr - 3.0);
  double s;
  if (tmp > -1e-2) s = 0.5 - (tr - 3.0) / 12.0;                      // small-angle Taylor expansion
  else { double c = 0.5 * (tr - 1.0); c = std::max(-1.0, std::min(1.0, c)); const double th = std::acos(c); s = th / (2.0 * std::sin(th)); }
  for (int i = 0; i < 3; ++i) err[i] = s * v[i];
}

struct Task { Mat A; Vec b; Mat D; Vec f; };   // A x = b ; D x <= f   (qm_wbc/include/qm_wbc/Task.h:17-66)
inline Task operator+(const Task& l, const Task& r) { return {vstack(l.A, r.A), vcat(l.b, r.b), vstack(l.D, r.D), vcat(l.f, r.f)}; }
inline Task operator*(const Task& t, double s) { return {scaled(t.A, s), vscaled(t.b, s), scaled(t.D, s), vscaled(t.f, s)}; }

// ------------------------------------------------------------------------------------------------
// one priority level of the cascade (qm_wbc/src/HoQp.cpp:12-158) solved as an inequality-constrained
// least-squares problem by an exact active-set method with orthogonal factorisations.  This stands in for
// qpOASES (@268b2f2, setToMPC, nWSR=100, cold start; HoQp.cpp:135-150): any exact convex-QP method gives
// the same x.
//   minimise ½|A Zp z + A xp − b|² + ½ rho |z|² + ½|w|²
//   s.t.     w >= 0,  D Zp z − w <= f − D xp (own rows),  Dp Zp z <= fp − Dp xp + wp* (rows of higher levels)
// The slack is eliminated analytically: w = max(0, D(xp + Zp z) − f).
// ------------------------------------------------------------------------------------------------
struct HoLevel { Mat Z; Vec x; Vec w; Mat Dstack; Vec fstack; Vec wstack; int status = 0; int iters = 0; };

// min |G z − g|² s.t. E z = e  (E rows assumed independent); returns z and multipliers lam (Gᵀ(Gz−g) + Eᵀ lam = 0)
inline void eqConstrainedLS(const Mat& G, const Vec& g, const Mat& E, const Vec& e, Vec& z, Vec& lam) {
  const int n = G.c, me = E.r;
  if (me == 0) {
    Mat Qf, Rf; householderQR(G, Qf, Rf);
    Vec qtg = matvecT(Qf, g); z.assign(n, 0.0);
    for (int i = n - 1; i >= 0; --i) { double s = qtg[i]; for (int j = i + 1; j < n; ++j) s -= Rf(i, j) * z[j]; z[i] = s / Rf(i, i); }
    lam.clear(); return;
  }
  Mat Qe, Re; householderQR(transpose(E), Qe, Re);      // Eᵀ = [Y N][Re;0]
  Vec y1(me);                                           // Reᵀ y1 = e
  for (int i = 0; i < me; ++i) { double s = e[i]; for (int k = 0; k < i; ++k) s -= Re(k, i) * y1[k]; y1[i] = s / Re(i, i); }
  Mat Y(n, me), N(n, n - me);
  for (int i = 0; i < n; ++i) { for (int j = 0; j < me; ++j) Y(i, j) = Qe(i, j); for (int j = 0; j < n - me; ++j) N(i, j) = Qe(i, me + j); }
  Vec zp = matvec(Y, y1);
  z = zp;
  if (n - me > 0) {
    Mat GN = matmul(G, N); Vec rhs = vsub(g, matvec(G, zp));
    Mat Qf, Rf; householderQR(GN, Qf, Rf); Vec qtg = matvecT(Qf, rhs); const int nn = n - me; Vec y2(nn, 0.0);
    for (int i = nn - 1; i >= 0; --i) { double s = qtg[i]; for (int j = i + 1; j < nn; ++j) s -= Rf(i, j) * y2[j]; y2[i] = s / Rf(i, i); }
    z = vadd(zp, matvec(N, y2));
  }
  // multipliers: Eᵀ lam = −Gᵀ(Gz − g)  ->  Re lam = −Yᵀ Gᵀ r
  Vec r = vsub(matvec(G, z), g); Vec gr = matvecT(G, r); Vec ytg = matvecT(Y, gr);
  lam.assign(me, 0.0);
  for (int i = me - 1; i >= 0; --i) { double s = -ytg[i]; for (int j = i + 1; j < me; ++j) s -= Re(i, j) * lam[j]; lam[i] = s / Re(i, i); }
}

// min |G y − g|² s.t. C y <= c, from a feasible y: primal active set (Nocedal & Wright alg. 16.3) on the QR-based equality-constrained solves above.
// status: 0 converged, 1 iteration limit (qpOASES' nWSR = 100, HoQp.cpp:141)
inline void primalActiveSetLSI(const Mat& G0, const Vec& g0, const Mat& DZ, const Vec& fb, Vec& z, int& status, int& iters) {
  const int n = G0.c, mh = DZ.r;
  std::vector<int> W; bool degenerate = false, vertex = false; double pscale = 0.0;
  for (iters = 0; iters < 100; ++iters) {
    Mat E((int)W.size(), n); Vec e(W.size());
    for (size_t a = 0; a < W.size(); ++a) { for (int j = 0; j < n; ++j) E((int)a, j) = DZ(W[a], j); e[a] = fb[W[a]]; }
    Vec zn, lam; eqConstrainedLS(G0, g0, E, e, zn, lam);
    Vec p = vsub(zn, z); double pn = 0; for (double v : p) pn = std::max(pn, std::fabs(v));
    double zs = 1.0; for (double v : z) zs = std::max(zs, std::fabs(v));
    pscale = std::max(pscale, pn);
    if (vertex || pn <= 1e-9 * std::max(zs, pscale)) {      // relative to the largest step seen: the problem's own length scale
      vertex = false;
      // stationary on the working set: drop a row with a negative multiplier (most negative; lowest index after a degenerate step — Bland)
      int worst = -1; double lw = 0.0; double lscale = 1.0; for (double v : lam) lscale = std::max(lscale, std::fabs(v));
      for (size_t a = 0; a < W.size(); ++a) if (lam[a] < -1e-9 * lscale) { if (degenerate) { if (worst < 0 || W[a] < W[worst]) worst = (int)a; } else if (lam[a] < lw) { lw = lam[a]; worst = (int)a; } }
      if (worst < 0) break;
      W.erase(W.begin() + worst);
    } else {
      double alpha = 1.0; int block = -1; Vec Dz = matvec(DZ, z), Dp = matvec(DZ, p);
      for (int i = 0; i < mh; ++i) {
        if (std::find(W.begin(), W.end(), i) != W.end()) continue;
        if (Dp[i] > 1e-10 * std::max(1.0, pn)) { const double a = std::max(0.0, (fb[i] - Dz[i]) / Dp[i]); if (a < alpha) { alpha = a; block = i; } }   // relative threshold: E p = 0 only to round-off; ties: lowest index
      }
      for (int j = 0; j < n; ++j) z[j] += alpha * p[j];
      degenerate = (alpha <= 1e-12);
      if (block >= 0) {
        if ((int)W.size() < n) W.push_back(block);
        else if (alpha <= 1e-12) vertex = true;       // n rows are active already (they are numerically dependent, else p would vanish): the set cannot grow beyond the dimension.
                                                      // A step blocked at once means z is a degenerate vertex: the multipliers decide (Bland's rule) — what the device kernel does
                                                      // (k_wbc.h); after a partial step the same rows are simply solved again.  (Rounds 1-4 reported status 2 here: which side of
                                                      // this branch a stress case lands on depends on rounding, i.e. on the compiler's flags.)
      }
    }
  }
  if (iters >= 100 && status == 0) status = 1;
}

// The higher levels' rows, relaxed by their slack solutions, hold at the previous solution by construction — unless SURVEY.md a17's quirk has paired rows and slacks
// of two higher levels wrongly (inequality rows on two higher levels, a non-zero slack among them): the reference then hands qpOASES a problem whose intended
// feasible point is infeasible and ignores the return code.  Status 3; the level keeps the previous solution.
inline bool hardRowsHoldAtPrevious(const Vec& fb) { double sc = 1.0; for (double v : fb) sc = std::max(sc, std::fabs(v)); for (double v : fb) if (v < -1e-9 * sc) return false; return true; }

inline HoLevel solveHoLevel(const Task& task, const HoLevel* prev, int nx) {
  const double rho = 1e-12;                              // HoQp.cpp:66
  HoLevel L;
  Mat Zp = prev ? prev->Z : Mat::identity(nx); Vec xp = prev ? prev->x : Vec(nx, 0.0);
  const int n = Zp.c;
  const bool hasEq = task.A.r > 0, hasIneq = task.D.r > 0;
  const bool hasPrevIneq = prev && prev->Dstack.r > 0;
  // stacked LS rows: [A Zp; sqrt(rho) I] z ≈ [b − A xp; 0]
  Mat AZ = hasEq ? matmul(task.A, Zp) : Mat(0, n);
  Vec rb = hasEq ? vsub(task.b, matvec(task.A, xp)) : Vec();
  Mat G0(AZ.r + n, n); Vec g0(AZ.r + n, 0.0);
  for (int i = 0; i < AZ.r; ++i) { for (int j = 0; j < n; ++j) G0(i, j) = AZ(i, j); g0[i] = rb[i]; }
  for (int j = 0; j < n; ++j) G0(AZ.r + j, j) = std::sqrt(rho);
  Vec z(n, 0.0);
  // General stacking (HoQp.cpp:92-124): OWN inequality rows at a level below the first one.  The shipped hierarchies never build it (their levels 1, 2 carry
  // equalities only); a WbcBase subclass with, say, torque limits one level below the friction cones does.  The slack stays a VARIABLE here: y = [z; w],
  //   minimise ½|G0 z − g0|² + ½|w|²   s.t.  −w <= 0,   Dp Zp z <= fp − Dp xp + wp*  (rows of the higher levels),   D Zp z − w <= f − D xp
  // — the rows in HoQp::buildDMatrix's order — solved by the same primal active set from the feasible point z = 0, w = max(0, −(f − D xp)).
  // SURVEY.md a17's quirk is reproduced by construction: prev->Dstack / fstack hold the higher levels' rows CURRENT-FIRST (stackedTasks_ = task_ + stackedTasksPrev_,
  // HoQp.cpp:46) while prev->wstack holds their slack solutions PREVIOUS-FIRST (HoQp.cpp:152-158): with inequality rows on two higher levels of different
  // sizes the slacks are added to the wrong rows, exactly as the reference does.
  if (hasIneq && hasPrevIneq) {
    Mat DZ = matmul(task.D, Zp); Vec fbo = vsub(task.f, matvec(task.D, xp)); const int ms = DZ.r;
    Mat HZ = matmul(prev->Dstack, Zp); Vec fbh = vadd(vsub(prev->fstack, matvec(prev->Dstack, xp)), prev->wstack); const int mh = HZ.r;
    const int ny = n + ms;
    Mat G(G0.r + ms, ny); Vec g(G0.r + ms, 0.0);
    for (int i = 0; i < G0.r; ++i) { for (int j = 0; j < n; ++j) G(i, j) = G0(i, j); g[i] = g0[i]; }
    for (int i = 0; i < ms; ++i) G(G0.r + i, n + i) = 1.0;
    Mat C(2 * ms + mh, ny); Vec c(2 * ms + mh, 0.0);
    for (int i = 0; i < ms; ++i) C(i, n + i) = -1.0;
    for (int i = 0; i < mh; ++i) { for (int j = 0; j < n; ++j) C(ms + i, j) = HZ(i, j); c[ms + i] = fbh[i]; }
    for (int i = 0; i < ms; ++i) { for (int j = 0; j < n; ++j) C(ms + mh + i, j) = DZ(i, j); C(ms + mh + i, n + i) = -1.0; c[ms + mh + i] = fbo[i]; }
    Vec y(ny, 0.0); for (int i = 0; i < ms; ++i) y[n + i] = std::max(0.0, -fbo[i]);
    if (!hardRowsHoldAtPrevious(fbh)) L.status = 3; else
    primalActiveSetLSI(G, g, C, c, y, L.status, L.iters);
    for (int j = 0; j < n; ++j) z[j] = y[j];
    L.w.assign(ms, 0.0); for (int i = 0; i < ms; ++i) L.w[i] = std::max(0.0, y[n + i]);
  } else if (hasIneq) {
    // soft rows: phi(z) = ½|G0 z − g0|² + ½ sum (d_i z − f_i)_+²  — Newton on the active set with exact line search
    Mat DZ = matmul(task.D, Zp); Vec fb = vsub(task.f, matvec(task.D, xp)); const int ms = DZ.r;
    std::vector<char> act(ms, 0);
    for (int i = 0; i < ms; ++i) act[i] = (0.0 - fb[i] > 0.0);
    for (L.iters = 0; L.iters < 100; ++L.iters) {
      int na = 0; for (int i = 0; i < ms; ++i) na += act[i];
      Mat G(G0.r + na, n); Vec g(G0.r + na);
      for (int i = 0; i < G0.r; ++i) { for (int j = 0; j < n; ++j) G(i, j) = G0(i, j); g[i] = g0[i]; }
      int r = G0.r; for (int i = 0; i < ms; ++i) if (act[i]) { for (int j = 0; j < n; ++j) G(r, j) = DZ(i, j); g[r] = fb[i]; ++r; }
      Vec zn, lam; eqConstrainedLS(G, g, Mat(0, n), Vec(), zn, lam);
      Vec p = vsub(zn, z);
      // exact line search of the convex piecewise quadratic along p on [0,1]
      Vec Dz = matvec(DZ, z), Dp = matvec(DZ, p); Vec G0z = vsub(matvec(G0, z), g0), G0p = matvec(G0, p);
      auto dphi = [&](double a) { double s = 0; for (int i = 0; i < G0.r; ++i) s += (G0z[i] + a * G0p[i]) * G0p[i]; for (int i = 0; i < ms; ++i) { const double v = Dz[i] + a * Dp[i] - fb[i]; if (v > 0) s += v * Dp[i]; } return s; };
      double a = 1.0;
      if (dphi(1.0) > 0.0) {   // minimiser inside (0,1): bisection on the monotone derivative
        double lo = 0.0, hi = 1.0; for (int it = 0; it < 200; ++it) { const double mid = 0.5 * (lo + hi); if (dphi(mid) > 0) hi = mid; else lo = mid; } a = 0.5 * (lo + hi);
      }
      for (int j = 0; j < n; ++j) z[j] += a * p[j];
      bool same = true; Vec Dzn = matvec(DZ, z);
      for (int i = 0; i < ms; ++i) { const char na_i = (Dzn[i] - fb[i] > 0.0); if (na_i != act[i]) same = false; act[i] = na_i; }
      if (same && a == 1.0) break;
      double pn = 0, zs = 1.0; for (int j = 0; j < n; ++j) { pn = std::max(pn, std::fabs(a * p[j])); zs = std::max(zs, std::fabs(z[j])); }
      if (pn <= 1e-12 * zs) break;                        // minimiser sits on a kink: both active sets give the same z
    }
    if (L.iters >= 100) L.status = 1;                      // nWSR exhausted
    Vec Dzn = matvec(DZ, z); L.w.assign(ms, 0.0); for (int i = 0; i < ms; ++i) L.w[i] = std::max(0.0, Dzn[i] - fb[i]);
  } else if (hasPrevIneq) {
    // hard rows of the higher levels: primal active-set from the feasible z = 0
    Mat DZ = matmul(prev->Dstack, Zp); Vec fb = vadd(vsub(prev->fstack, matvec(prev->Dstack, xp)), prev->wstack);
    if (!hardRowsHoldAtPrevious(fb)) L.status = 3; else
    primalActiveSetLSI(G0, g0, DZ, fb, z, L.status, L.iters);
  } else {
    Vec lam; eqConstrainedLS(G0, g0, Mat(0, n), Vec(), z, lam);
  }
  L.x = vadd(xp, matvec(Zp, z));
  L.Z = hasEq ? matmul(Zp, nullSpace(AZ)) : Zp;            // HoQp::buildZMatrix
  // stacked inequality rows / slacks handed to the next level (HoQp.cpp:46,152-158)
  L.Dstack = vstack(task.D, prev ? prev->Dstack : Mat()); L.fstack = vcat(task.f, prev ? prev->fstack : Vec());
  L.wstack = vcat(prev ? prev->wstack : Vec(), L.w);
  if (prev && prev->status != 0 && L.status == 0) L.status = prev->status;
  return L;
}

// ------------------------------------------------------------------------------------------------
// WbcBase restated
// ------------------------------------------------------------------------------------------------
struct WbcDebug;
struct WbcState { Vec inputLast; WbcState() : inputLast(QM_NU, 0.0) {} };
struct WbcDebug { Vec qMeas, vMeas, qDes, vDes, baseAcc, nle, x0, x1, x2; Mat Mq, J, dJ; int status[3]; int iters[3]; Task task[3]; };   // task[k]: level k exactly as handed to HoQp (tests/test_hoqp_literal.py)

inline Vec wbcUpdate(const Model& M, WbcState& S, const Vec& xDes, const Vec& uDes, const Vec& rbd, int mode, double period, double time, bool mpcVariant, WbcDebug* dbg) {
  const double* st = M.st; const int nq = QM_NQ, nv = QM_NWBC;
  bool fl[4]; modeToFlags(mode, fl); int nc = 0; for (int i = 0; i < 4; ++i) nc += fl[i];
  // ---- updateMeasured (WbcBase.cpp:134-191) ----
  Vec q(nq), v(nq);
  for (int i = 0; i < 3; ++i) { q[i] = rbd[3 + i]; q[3 + i] = rbd[i]; v[i] = rbd[nq + 3 + i]; }
  { // getEulerAnglesZyxDerivativesFromGlobalAngularVelocity
    const double sz = std::sin(q[3]), cz = std::cos(q[3]), sy = std::sin(q[4]), cy = std::cos(q[4]);
    const double wx = rbd[nq], wy = rbd[nq + 1], wz = rbd[nq + 2]; const double tmp = cz * wx / cy + sz * wy / cy;
    v[3] = sy * tmp + wz; v[4] = -sz * wx + cz * wy; v[5] = tmp;
  }
  for (int j = 0; j < QM_NJ; ++j) { q[6 + j] = rbd[6 + j]; v[6 + j] = rbd[nq + 6 + j]; }
  // M(q) and nle = C v + g via Lagrange: h_i = sum_jk dM_ij/dq_k v_j v_k − ½ sum_jk dM_jk/dq_i v_j v_k + dV/dq_i
  Mat Mq(nq, nq); Vec nle(nq, 0.0);
  {
    typedef Dual<QM_NQ> D; static thread_local D Md[QM_NQ][QM_NQ]; D qd[QM_NQ];
    for (int i = 0; i < nq; ++i) qd[i] = D::seed(q[i], i);
    massMatrix<D>(M, qd, Md); D V = potentialEnergy<D>(M, qd);
    for (int i = 0; i < nq; ++i) for (int j = 0; j < nq; ++j) Mq(i, j) = Md[i][j].v;
    for (int i = 0; i < nq; ++i) {
      double s = V.d[i];
      for (int j = 0; j < nq; ++j) for (int k = 0; k < nq; ++k) s += (Md[i][j].d[k] - 0.5 * Md[j][k].d[i]) * v[j] * v[k];
      nle[i] = s;
    }
  }
  // frame Jacobians and time variation (d/dt along (q,v) by AD with a single direction)
  typedef Dual<1> D1; D1 q1[QM_NQ]; for (int i = 0; i < nq; ++i) { q1[i] = D1(q[i]); q1[i].d[0] = v[i]; }
  Mat J(12, nq), dJ(12, nq), baseJ(6, nq), baseDJ(6, nq), armJ(6, nq), armDJ(6, nq);
  for (int f = -1; f < QM_NF; ++f) {
    D1 Jf[6][QM_NQ]; frameJacobian<D1>(M, q1, f, Jf);
    for (int r = 0; r < 6; ++r) for (int c = 0; c < nq; ++c) {
      if (f >= 0 && f < 4 && r < 3) { J(3 * f + r, c) = Jf[r][c].v; dJ(3 * f + r, c) = Jf[r][c].d[0]; }
      if (f < 0) { baseJ(r, c) = Jf[r][c].v; baseDJ(r, c) = Jf[r][c].d[0]; }
      if (f == 4) { armJ(r, c) = Jf[r][c].v; armDJ(r, c) = Jf[r][c].d[0]; }
    }
  }
  Kin<double> kM; forwardKinematics(M, q.data(), kM);
  // ---- updateDesired (WbcBase.cpp:193-226) ----
  Vec qd(nq), vd(nq); for (int i = 0; i < nq; ++i) qd[i] = xDes[6 + i];
  Kin<double> kD; forwardKinematics(M, qd.data(), kD);
  Srbd<double> cd; srbd(M, qd.data(), cd);
  baseVelocity(M, cd, xDes.data(), vd.data()); for (int j = 0; j < QM_NJ; ++j) vd[6 + j] = uDes[12 + j];
  Vec jointAccel(QM_NJ); for (int j = 0; j < QM_NJ; ++j) jointAccel[j] = (uDes[12 + j] - S.inputLast[12 + j]) / period;
  S.inputLast = uDes;
  Vec baseAcc(6, 0.0);
  {
    // A_b (SRBD) inverse, full CMM joint columns A_j, full dCCRBA Adot, true COM in the momentum rate
    double Af[6][QM_NQ]; V3<double> com; centroidalMomentumMatrix<double>(M, qd.data(), Af, com);
    D1 q1d[QM_NQ]; for (int i = 0; i < nq; ++i) { q1d[i] = D1(qd[i]); q1d[i].d[0] = vd[i]; }
    D1 Ad[6][QM_NQ]; V3<D1> comd; centroidalMomentumMatrix<D1>(M, q1d, Ad, comd);
    const double m = M.robotMass();
    double rate[6] = {0, 0, -9.81 * m, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
      V3<double> F = v3<double>(uDes[3 * i], uDes[3 * i + 1], uDes[3 * i + 2]); V3<double> r = kD.fp[i] - com; V3<double> t = cross(r, F);
      for (int k = 0; k < 3; ++k) { rate[k] += F[k]; rate[3 + k] += t[k]; }
    }
    for (int r = 0; r < 6; ++r) { double s = 0; for (int c = 0; c < nq; ++c) s += Ad[r][c].d[0] * vd[c]; for (int j = 0; j < QM_NJ; ++j) s += Af[r][6 + j] * jointAccel[j]; rate[r] -= s; }
    // AbInv = [[I/m, −(1/m) A12 A22inv],[0, A22inv]]
    V3<double> ra = v3<double>(rate[3], rate[4], rate[5]); V3<double> thdd = cd.A22inv * ra; V3<double> corr = (cd.A12 * cd.A22inv) * ra;
    for (int k = 0; k < 3; ++k) { baseAcc[k] = rate[k] / m - corr[k] / m; baseAcc[3 + k] = thdd[k]; }
  }
  // measured / desired frame positions and velocities (PinocchioEndEffectorKinematics getPosition/getVelocity)
  V3<double> pM[QM_NF], vM[QM_NF], wM[QM_NF], pD[QM_NF], vDd[QM_NF], wD[QM_NF];
  for (int f = 0; f < QM_NF; ++f) { pM[f] = kM.fp[f]; pD[f] = kD.fp[f]; frameVelocity(M, kM, q.data(), v.data(), f, vM[f], wM[f]); frameVelocity(M, kD, qd.data(), vd.data(), f, vDd[f], wD[f]); }
  // ---- task formulators (WbcBase.cpp:228-546) ----
  auto zerosTask = [&](int rows) { Task t; t.A = Mat(rows, nv); t.b.assign(rows, 0.0); return t; };
  Vec tauMax(QM_NJ); for (int l = 0; l < 4; ++l) for (int k = 0; k < 3; ++k) tauMax[3 * l + k] = M.mb[MB_TAUMAX + k]; for (int k = 0; k < 6; ++k) tauMax[12 + k] = M.mb[MB_TAUMAX + 12 + k];
  Task eom = zerosTask(6);
  for (int r = 0; r < 6; ++r) { for (int c = 0; c < nq; ++c) eom.A(r, c) = Mq(r, c); for (int c = 0; c < 12; ++c) eom.A(r, nq + c) = -J(c, r); eom.b[r] = -nle[r]; }
  Task torque; torque.D = Mat(36, nv); torque.f.assign(36, 0.0);
  for (int r = 0; r < QM_NJ; ++r) {
    for (int c = 0; c < nq; ++c) { torque.D(r, c) = Mq(6 + r, c); torque.D(18 + r, c) = -Mq(6 + r, c); }
    for (int c = 0; c < 12; ++c) { torque.D(r, nq + c) = -J(c, 6 + r); torque.D(18 + r, nq + c) = J(c, 6 + r); }
    torque.f[r] = tauMax[r] - nle[6 + r]; torque.f[18 + r] = tauMax[r] + nle[6 + r];
  }
  Vec dJv = matvec(dJ, v);
  Task noContact = zerosTask(3 * nc);
  { int j = 0; for (int i = 0; i < 4; ++i) if (fl[i]) { for (int r = 0; r < 3; ++r) { for (int c = 0; c < nq; ++c) noContact.A(3 * j + r, c) = J(3 * i + r, c); noContact.b[3 * j + r] = -dJv[3 * i + r]; } ++j; } }
  Task friction = zerosTask(3 * (4 - nc));
  { int j = 0; for (int i = 0; i < 4; ++i) if (!fl[i]) { for (int r = 0; r < 3; ++r) friction.A(3 * j + r, nq + 3 * i + r) = 1.0; ++j; } }
  friction.D = Mat(5 * nc + 3 * (4 - nc), nv); friction.f.assign(friction.D.r, 0.0);
  { const double mu = st[ST_WBC_FRIC]; const double pyr[5][3] = {{0, 0, -1}, {1, 0, -mu}, {-1, 0, -mu}, {0, 1, -mu}, {0, -1, -mu}};
    int j = 0; for (int i = 0; i < 4; ++i) if (fl[i]) { for (int r = 0; r < 5; ++r) for (int c = 0; c < 3; ++c) friction.D(5 * j + r, nq + 3 * i + c) = pyr[r][c]; ++j; } }
  Task baseHeight = zerosTask(1); baseHeight.A(0, 2) = 1.0;
  baseHeight.b[0] = baseAcc[2] + st[ST_KP_BASE_H] * (qd[2] - q[2]) + st[ST_KD_BASE_H] * (vd[2] - v[2]);
  Task baseAng = zerosTask(3);
  {
    M3<double> E = eulerZyxE(q[3], q[4]);
    V3<double> wMeas = E * v3<double>(v[3], v[4], v[5]), wDes = E * v3<double>(vd[3], vd[4], vd[5]);
    double err[3]; rotationErrorInWorld(rotZyx(qd[3], qd[4], qd[5]), rotZyx(q[3], q[4], q[5]), err);
    // getGlobalAngularAccelerationFromEulerAnglesZyxDerivatives(eulerMeasured, eulerRatesDesired, eulerAccDesired) = E thdd + Edot thd
    D1 z1(q[3]), y1(q[4]); z1.d[0] = vd[3]; y1.d[0] = vd[4]; M3<D1> Ed = eulerZyxE(z1, y1);
    double acc[3]; for (int r = 0; r < 3; ++r) { acc[r] = 0; for (int c = 0; c < 3; ++c) acc[r] += E(r, c) * baseAcc[3 + c] + Ed(r, c).d[0] * vd[3 + c]; }
    Vec bdj = matvec(baseDJ, v);
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < nq; ++c) baseAng.A(r, c) = baseJ(3 + r, c); baseAng.b[r] = acc[r] + st[ST_KP_BASE_ANG] * err[r] + st[ST_KD_BASE_ANG] * (wDes[r] - wMeas[r]) - bdj[3 + r]; }
  }
  Task baseLin = zerosTask(2);
  for (int r = 0; r < 2; ++r) { baseLin.A(r, r) = 1.0; baseLin.b[r] = baseAcc[r] + st[ST_KP_BASE_LIN] * (qd[r] - q[r]) + st[ST_KD_BASE_LIN] * (vd[r] - v[r]); }
  Task swing = zerosTask(3 * (4 - nc));
  { int j = 0; for (int i = 0; i < 4; ++i) if (!fl[i]) { for (int r = 0; r < 3; ++r) { for (int c = 0; c < nq; ++c) swing.A(3 * j + r, c) = J(3 * i + r, c);
      swing.b[3 * j + r] = st[ST_KP_SWING] * (pD[i][r] - pM[i][r]) + st[ST_KD_SWING] * (vDd[i][r] - vM[i][r]) - dJv[3 * i + r]; } ++j; } }
  Task armJoint = zerosTask(6);
  for (int r = 0; r < 6; ++r) { armJoint.A(r, 18 + r) = 1.0; armJoint.b[r] = st[ST_KP_ARM_J + r] * (qd[18 + r] - q[18 + r]) + st[ST_KD_ARM_J + r] * (vd[18 + r] - v[18 + r]); }
  Vec adjv = matvec(armDJ, v);
  Task eeLin = zerosTask(3);
  for (int r = 0; r < 3; ++r) { for (int c = 0; c < nq; ++c) eeLin.A(r, c) = armJ(r, c); eeLin.b[r] = st[ST_KP_EE_LIN + r] * (pD[4][r] - pM[4][r]) + st[ST_KD_EE_LIN + r] * (vDd[4][r] - vM[4][r]) - adjv[r]; }
  Task eeAng = zerosTask(3);
  {
    double err[3]; rotationErrorInWorld(kD.fR[4], kM.fR[4], err);
    for (int r = 0; r < 3; ++r) {
      double djv = 0; for (int c = 0; c < nq; ++c) { const bool z = (c >= 3 && c < 6); eeAng.A(r, c) = z ? 0.0 : armJ(3 + r, c); djv += (z ? 0.0 : armDJ(3 + r, c)) * v[c]; }
      eeAng.b[r] = st[ST_KP_EE_ANG + r] * err[r] + st[ST_KD_EE_ANG + r] * (-wM[4][r]) - djv;
    }
  }
  Task contactForce = zerosTask(12); for (int r = 0; r < 12; ++r) { contactForce.A(r, nq + r) = 1.0; contactForce.b[r] = uDes[r]; }
  // ---- hierarchy (HierarchicalWbc.cpp:18-44 / HierarchicalMpcWbc.cpp:18-34) ----
  Task task0 = eom + torque + noContact + friction, task1, task2;
  if (mpcVariant) { task1 = baseHeight + baseAng + baseLin + swing * 100.0; task2 = contactForce; }
  else { task1 = (time < 10.0) ? armJoint : (baseHeight + baseAng + eeLin + eeAng + swing * 100.0); task2 = contactForce + baseLin; }
  HoLevel l0 = solveHoLevel(task0, nullptr, nv); HoLevel l1 = solveHoLevel(task1, &l0, nv); HoLevel l2 = solveHoLevel(task2, &l1, nv);
  // ---- updateCmd (WbcBase.cpp:548-563) ----
  Vec out(QM_NWBC_OUT, 0.0);
  for (int i = 0; i < nv; ++i) out[i] = l2.x[i];
  for (int r = 0; r < QM_NJ; ++r) { double s = nle[6 + r]; for (int c = 0; c < nq; ++c) s += Mq(6 + r, c) * l2.x[c]; for (int c = 0; c < 12; ++c) s -= J(c, 6 + r) * l2.x[nq + c]; out[nv + r] = s; }
  if (dbg) { dbg->task[0] = task0; dbg->task[1] = task1; dbg->task[2] = task2; dbg->qMeas = q; dbg->vMeas = v; dbg->qDes = qd; dbg->vDes = vd; dbg->baseAcc = baseAcc; dbg->nle = nle; dbg->Mq = Mq; dbg->J = J; dbg->dJ = dJ; dbg->x0 = l0.x; dbg->x1 = l1.x; dbg->x2 = l2.x; dbg->status[0] = l0.status; dbg->status[1] = l1.status; dbg->status[2] = l2.status; dbg->iters[0] = l0.iters; dbg->iters[1] = l1.iters; dbg->iters[2] = l2.iters; }
  return out;
}
