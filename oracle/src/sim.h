// oracle/src/sim.h — TEST INFRASTRUCTURE (CPU oracle).  Batched-plant restatement: the hybrid joint command law with
// the command delay of qm_gazebo::QMHWSim::writeSim (qm_gazebo/src/QMHWSim.cpp:98-116, delay from qm_gazebo/config/default.yaml:2) in front of the
// floating-base forward dynamics  M(q) vdot = Sᵀ tau − nle(q, v) + Σ J_iᵀ f_i  with a penalty ground contact of the four feet and semi-implicit
// Euler integration.  M and nle come from the Lagrangian / forward-mode AD formulation of wbc.h (the product uses recursive world-frame passes, so
// the two are independent); the contact model is the product's own stated model (Gazebo's ODE solver is not in the reference's sources): PARITY of
// the contact physics is therefore against this restatement only; the command law and the delay buffer follow the reference line by line.
#pragma once
#include <deque>
#include "wbc.h"

struct SimParams { double k_n = 4.0e4, d_n = 200.0, mu = 0.8, v_eps = 1.0e-2, foot_radius = 0.02, delay = 0.009; int saturate = 1; };
struct SimCmd { double stamp = 0.0; double pos[QM_NJ] = {0}, vel[QM_NJ] = {0}, kp[QM_NJ] = {0}, kd[QM_NJ] = {0}, ff[QM_NJ] = {0}; };
struct SimState { double q[QM_NQ] = {0}, v[QM_NQ] = {0}, time = 0.0; std::deque<SimCmd> buffer; SimCmd held; double force[12] = {0}; int contact[4] = {0, 0, 0, 0}; int status = 0; };

// joint-space inertia and non-linear effects (gravity included) by Lagrange's equations — same construction as wbcUpdate
inline void massAndNle(const Model& M, const double* q, const double* v, Mat& Mq, Vec& nle) {
  const int nq = QM_NQ; typedef Dual<QM_NQ> D; static thread_local D Md[QM_NQ][QM_NQ]; D qd[QM_NQ];
  for (int i = 0; i < nq; ++i) qd[i] = D::seed(q[i], i);
  massMatrix<D>(M, qd, Md); D V = potentialEnergy<D>(M, qd);
  Mq = Mat(nq, nq); nle.assign(nq, 0.0);
  for (int i = 0; i < nq; ++i) for (int j = 0; j < nq; ++j) Mq(i, j) = Md[i][j].v;
  for (int i = 0; i < nq; ++i) {
    double s = V.d[i];
    for (int j = 0; j < nq; ++j) for (int k = 0; k < nq; ++k) s += (Md[i][j].d[k] - 0.5 * Md[j][k].d[i]) * v[j] * v[k];
    nle[i] = s;
  }
}

// one simulation step: writeSim(time, period) + nsub integration sub-steps of period / nsub
inline void simStep(const Model& M, const SimParams& P, SimState& S, double period, int nsub) {
  // QMHWSim.cpp:100-110 — drop commands older than the delay, push the held command, apply the oldest survivor
  while (!S.buffer.empty() && S.buffer.back().stamp + P.delay < S.time) S.buffer.pop_back();
  SimCmd c = S.held; c.stamp = S.time; S.buffer.push_front(c);
  const SimCmd cmd = S.buffer.back();
  const double h = period / nsub; const int nq = QM_NQ;
  for (int s = 0; s < nsub; ++s) {
    S.time += h;
    Mat Mq; Vec nle; massAndNle(M, S.q, S.v, Mq, nle);
    Kin<double> k; forwardKinematics(M, S.q, k);
    Mat J(12, nq);
    for (int f = 0; f < 4; ++f) { double Jf[6][QM_NQ]; frameJacobian<double>(M, S.q, f, Jf); for (int r = 0; r < 3; ++r) for (int cc = 0; cc < nq; ++cc) J(3 * f + r, cc) = Jf[r][cc]; }
    Vec rhs(nq, 0.0);
    for (int j = 0; j < QM_NJ; ++j) {
      double t = cmd.kp[j] * (cmd.pos[j] - S.q[6 + j]) + cmd.kd[j] * (cmd.vel[j] - S.v[6 + j]) + cmd.ff[j];   // QMHWSim.cpp:112-113
      const double tm = M.mb[MB_TAUMAX + j];
      if (P.saturate) t = std::min(tm, std::max(-tm, t));                                                         // [upstream] DefaultRobotHWSim effort saturation
      rhs[6 + j] = t;
    }
    for (int f = 0; f < 4; ++f) {
      double vf[3] = {0, 0, 0}; for (int r = 0; r < 3; ++r) for (int cc = 0; cc < nq; ++cc) vf[r] += J(3 * f + r, cc) * S.v[cc];
      const double pen = P.foot_radius - k.fp[f][2]; double fx = 0.0, fy = 0.0, fz = 0.0;
      if (pen > 0.0) {
        fz = std::max(0.0, P.k_n * pen - P.d_n * vf[2]);
        const double sc = -P.mu * fz / std::sqrt(vf[0] * vf[0] + vf[1] * vf[1] + P.v_eps * P.v_eps);
        fx = sc * vf[0]; fy = sc * vf[1];
      }
      S.force[3 * f] = fx; S.force[3 * f + 1] = fy; S.force[3 * f + 2] = fz;
    }
    for (int i = 0; i < nq; ++i) { rhs[i] -= nle[i]; for (int r = 0; r < 12; ++r) rhs[i] += J(r, i) * S.force[r]; }
    Mat L; if (!cholesky(Mq, L)) { S.status = 1; return; }
    Vec a = cholSolve(L, rhs);
    for (int i = 0; i < nq; ++i) { S.v[i] += h * a[i]; S.q[i] += h * S.v[i]; }
  }
  Kin<double> k; forwardKinematics(M, S.q, k);
  for (int f = 0; f < 4; ++f) S.contact[f] = (P.foot_radius - k.fp[f][2] > 0.0) ? 1 : 0;
}
