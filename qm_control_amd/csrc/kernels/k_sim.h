// k_sim.h — K8: batched rigid-body plant behind the controller, ONE WAVEFRONT PER INSTANCE.
//
// Stands where Gazebo + qm_gazebo::QMHWSim stand in the reference:
//   * hybrid joint command with the command delay of the simulated hardware interface — QMHWSim::writeSim
//     (qm_gazebo/src/QMHWSim.cpp:98-116): every simulation step pushes the current command (stamp, posDes, velDes, kp, kd, ff) into a
//     per-joint buffer, drops the commands older than `delay` (qm_gazebo/config/default.yaml:2, 9 ms) and applies the OLDEST remaining one:
//     tau_j = kp (posDes − q_j) + kd (velDes − qd_j) + ff; the effort is saturated at the URDF limit as gazebo_ros_control's
//     DefaultRobotHWSim::writeSim [upstream] does through its joint-limits interface;
//   * forward dynamics of the 24-dof floating-base tree:  M(q) vdot = Sᵀ tau − nle(q, v) + Σ_i J_iᵀ f_i  (same recursive world-frame
//     passes as the WBC, qm_dev_rbd.h; generalized velocity = [pdot_world, zyx rates, joint rates]), dense Cholesky of M per step;
//   * ground contact of the four feet.  Gazebo's ODE contact solver is not part of the reference's sources and is not reproduced: the plant
//     uses a stated penalty model on the plane z = 0 — normal force max(0, k pen − d vz) with pen = foot_radius − p_z > 0, regularised Coulomb
//     friction −mu f_n v_t / sqrt(|v_t|² + v_eps²) — parameters in QmSimParams;
//   * semi-implicit Euler: v+ = v + h vdot, q+ = q + h v+ (the coordinates' rates ARE the velocities: Translation + SphericalZYX root);
//   * QMHWSim::readSim (QMHWSim.cpp:60-75): joint / base state and the contact flags; the state is handed over in the estimator's
//     rbd layout (qm_estimation/src/StateEstimateBase.cpp:41-103) so that the WBC and the MPC observation read it like the reference's
//     "ground truth" estimator (FromTopiceEstimate).
// Joint position limits and self collision are not modelled.
#pragma once
#include "qm_dev_rbd.h"
#include "qm_dev_kin.h"

#define QM_SIM_SLOTS 16                 /* command delay ring: delay / period + 1 entries are ever live (9 ms / 1 ms); a full ring drops its oldest entry */
#define QM_SIM_CMD 91                   /* stamp, posDes(18), velDes(18), kp(18), kd(18), ff(18) */
struct QmSimParams { double k_n, d_n, mu, v_eps, foot_radius, delay; int saturate; };

struct QmSimArgs {
  const double* mb;
  int B; int nsub; double h;            // one call = one writeSim + nsub integration sub-steps of length h
  QmSimParams p;
  double* q; double* v; double* time;   // [B][24], [B][24], [B] plant state (Pinocchio coordinates)
  const double* cmd;                    // [B][QM_SIM_CMD - 1] command the controller holds (HybridJointHandle::setCommand): posDes velDes kp kd ff
  double* ring; int* ring_n;            // [B][QM_SIM_SLOTS][QM_SIM_CMD] delay buffer, newest first in ring order; [B][2] = head, count
  double* rbd; int* contact;            // [B][55] state in the estimator's layout, [B][4] contact flags (LF RF LH RH)
  double* force;                        // [B][12] contact forces of the last sub-step (world frame)
  int* status;                          // [B] 0 ok, 1 mass matrix not positive definite
};

// ---- LDS carve (doubles) ----
#define SL_M     0                      /* [24][24] */
#define SL_NLE   576
#define SL_JF    600                    /* [12][24] */
#define SL_ACC   888                    /* [6][20] */
#define SL_TIP   1008                   /* feet p(3) v(3) x 4, arm p(3) R(9) */
#define SL_Q     1044
#define SL_V     1068
#define SL_RHS   1092
#define SL_F     1116                   /* [12] */
#define SL_TAU   1128                   /* [18] */
#define SL_JDUM  1146                   /* [6][24] sink of the arm Jacobian (not needed by the plant) */
#define SL_COMP  1290                   /* [19 x 6][6] per-joint subtree composites of the six chain lanes (lane interleaved; registers would spill) */
#define SL_TOTAL (SL_COMP + 19 * 6 * 6)
#define SIM_LDS_BYTES (SL_TOTAL * 8)

// measured pass of one chain per lane (lanes 0-3 legs, 4 arm, 5 root body) + base block: M, nle, foot Jacobians, tips
__device__ __forceinline__ void sim_dynamics_terms(const double* mb, double* S, const int l, const bool want_m) {
  double* M = S + SL_M; double* nle = S + SL_NLE; double* Jf = S + SL_JF; const double* q = S + SL_Q; const double* v = S + SL_V;
  for (int i = l; i < SL_TIP; i += 64) S[i] = 0.0;
  qm_wave_sync();
  if (l < 6) {
    RbdBase Bb; rbd_base(q, v, Bb);
    double cm = 0.0, ch[3] = {0, 0, 0}, cI[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, F[3] = {0, 0, 0}, NO[3] = {0, 0, 0};
    if (l < 5) {
      const bool leg = l < 4; const int contact = leg ? chain_to_contact(l) : 4;
      RbdJsink Jt; Jt.rows = leg ? Jf + 3 * contact * QM_NQ : S + SL_JDUM; Jt.nrows = leg ? 3 : 6; Jt.dummy = 0.0;
      RbdTip tip;
      RbdCompLds comp; comp.base = S + SL_COMP + l; comp.stride = 6;
      rbd_chain<6, double*, RbdJsink, RbdCompLds>(mb, leg ? 3 * l : 12, contact, q, v, Bb, M, nle, want_m, cm, ch, cI, F, NO, (RbdSums*)nullptr, tip, Jt, want_m && leg, leg ? 3 : 6, comp);
      if (leg) {
        if (want_m) {
          double* Jr = Jt.rows;
          for (int r = 0; r < 3; ++r) Jr[r * QM_NQ + r] = 1.0;
          for (int k = 0; k < 3; ++k) { const double e[3] = {Bb.E[k], Bb.E[3 + k], Bb.E[6 + k]}, d[3] = {tip.p[0] - Bb.p[0], tip.p[1] - Bb.p[1], tip.p[2] - Bb.p[2]}; double cr[3]; v3_cross(e, d, cr);
            for (int r = 0; r < 3; ++r) Jr[r * QM_NQ + 3 + k] = cr[r]; }
        }
        for (int i = 0; i < 3; ++i) { S[SL_TIP + 6 * contact + i] = tip.p[i]; S[SL_TIP + 6 * contact + 3 + i] = tip.v[i]; }
      } else { for (int i = 0; i < 3; ++i) S[SL_TIP + 24 + i] = tip.p[i]; for (int i = 0; i < 9; ++i) S[SL_TIP + 27 + i] = tip.R[i]; }
    } else {
      const double zero3[3] = {0.0, 0.0, 0.0}; double c[3], Iw[9], vc[3], ac[3];
      body_state(mb, 0, Bb.R, Bb.p, Bb.vlin, Bb.w, zero3, Bb.al, c, Iw, vc, ac);
      add_body(mb[MB_MASS], c, Iw, vc, Bb.w, ac, Bb.al, cm, ch, cI, F, NO, (RbdSums*)nullptr);
    }
    double* acc = S + SL_ACC + l * 20;
    acc[0] = cm; for (int i = 0; i < 3; ++i) { acc[1 + i] = ch[i]; acc[13 + i] = F[i]; acc[16 + i] = NO[i]; } for (int i = 0; i < 9; ++i) acc[4 + i] = cI[i];
  }
  qm_wave_sync();
  if (want_m && l < 6) {   // base block of M and base rows of nle from the whole-tree composite (lane = base dof)
    RbdBase Bm; rbd_base(q, v, Bm);
    double cm = 0.0, ch[3] = {0, 0, 0}, cI[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, F[3] = {0, 0, 0}, NO[3] = {0, 0, 0};
    for (int s2 = 0; s2 < 6; ++s2) { const double* acc = S + SL_ACC + s2 * 20; cm += acc[0]; for (int i = 0; i < 3; ++i) { ch[i] += acc[1 + i]; F[i] += acc[13 + i]; NO[i] += acc[16 + i]; } for (int i = 0; i < 9; ++i) cI[i] += acc[4 + i]; }
    double w[3], vO[3]; rbd_S_base(Bm, l, w, vO);
    double wh[3], hv3[3], Iw_[3]; v3_cross(w, ch, wh); v3_cross(ch, vO, hv3); m3_mulv(cI, w, Iw_);
    const double f[3] = {cm * vO[0] + wh[0], cm * vO[1] + wh[1], cm * vO[2] + wh[2]}, nO[3] = {Iw_[0] + hv3[0], Iw_[1] + hv3[1], Iw_[2] + hv3[2]};
    for (int e = 0; e < 6; ++e) { double w2b[3], vO2[3]; rbd_S_base(Bm, e, w2b, vO2); M[e * QM_NQ + l] = w2b[0] * nO[0] + w2b[1] * nO[1] + w2b[2] * nO[2] + vO2[0] * f[0] + vO2[1] * f[1] + vO2[2] * f[2]; }
    nle[l] = w[0] * NO[0] + w[1] * NO[1] + w[2] * NO[2] + vO[0] * F[0] + vO[1] * F[1] + vO[2] * F[2];
  }
  qm_wave_sync();
}

__global__ void __launch_bounds__(64) qm_sim_kernel(QmSimArgs a) {
  extern __shared__ double qm_smem[];
  double* S = qm_smem;
  const int b = blockIdx.x, l = threadIdx.x & 63;
  if (b >= a.B) return;
  const double* mb = qm_table(a.mb);
  double* M = S + SL_M; double* nle = S + SL_NLE; double* Jf = S + SL_JF; double* q = S + SL_Q; double* v = S + SL_V; double* rhs = S + SL_RHS; double* fc = S + SL_F; double* tau = S + SL_TAU;
  if (l < 24) { q[l] = a.q[(size_t)b * 24 + l]; v[l] = a.v[(size_t)b * 24 + l]; }
  double time = a.time[b];
  int head = a.ring_n[b * 2], cnt = a.ring_n[b * 2 + 1];
  double* ring = a.ring + (size_t)b * QM_SIM_SLOTS * QM_SIM_CMD;
  int bad = 0;
  qm_wave_sync();
  // ---- delay buffer (QMHWSim.cpp:100-110), once per call like writeSim once per simulation step: drop what is older than `delay`, push the held
  //      command with the current time stamp, apply the oldest survivor during this step ----
  if (a.nsub > 0) {                       // nsub == 0: hand-over of the current state only (after a reset), not a simulation step
    while (cnt > 0 && ring[((head + cnt - 1) % QM_SIM_SLOTS) * QM_SIM_CMD] + a.p.delay < time) --cnt;
    head = (head + QM_SIM_SLOTS - 1) % QM_SIM_SLOTS; if (cnt < QM_SIM_SLOTS) ++cnt;
    { double* slot = ring + head * QM_SIM_CMD; if (l == 0) slot[0] = time; for (int i = l; i < QM_SIM_CMD - 1; i += 64) slot[1 + i] = a.cmd[(size_t)b * (QM_SIM_CMD - 1) + i]; }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    qm_wave_sync();
  }
  const double* use = ring + ((head + (cnt > 0 ? cnt : 1) - 1) % QM_SIM_SLOTS) * QM_SIM_CMD;
  for (int s = 0; s < a.nsub; ++s) {
    time += a.h;
    sim_dynamics_terms(mb, S, l, true);
    if (l >= 6 && l < 24) {
      const int j = l - 6;
      double t = use[1 + 36 + j] * (use[1 + j] - q[l]) + use[1 + 54 + j] * (use[1 + 18 + j] - v[l]) + use[1 + 72 + j];   // kp (posDes − q) + kd (velDes − qd) + ff
      const double tm = mb[MB_TAUMAX + j];
      if (a.p.saturate) t = fmin(tm, fmax(-tm, t));
      tau[j] = t;
    }
    if (l < 4) {
      const double* tp = S + SL_TIP + 6 * l; const double pen = a.p.foot_radius - tp[2];
      double fx = 0.0, fy = 0.0, fz = 0.0;
      if (pen > 0.0) {
        fz = fmax(0.0, a.p.k_n * pen - a.p.d_n * tp[5]);
        const double sc = -a.p.mu * fz / sqrt(tp[3] * tp[3] + tp[4] * tp[4] + a.p.v_eps * a.p.v_eps);
        fx = sc * tp[3]; fy = sc * tp[4];
      }
      fc[3 * l] = fx; fc[3 * l + 1] = fy; fc[3 * l + 2] = fz;
    }
    qm_wave_sync();
    if (l < 24) { double r = ((l >= 6) ? tau[l - 6] : 0.0) - nle[l]; for (int k = 0; k < 12; ++k) r += Jf[k * QM_NQ + l] * fc[k]; rhs[l] = r; }
    qm_wave_sync();
    // ---- M = L Lᵀ in place (lower triangle), right-looking; lane i owns row i ----
    for (int k = 0; k < 24; ++k) {
      const double dkk = M[k * 24 + k];
      if (!(dkk > 0.0)) bad = 1;
      const double inv = 1.0 / sqrt(dkk > 0.0 ? dkk : 1.0);
      qm_wave_sync();
      if (l >= k && l < 24) M[l * 24 + k] *= inv;                     // column k (the diagonal becomes sqrt(dkk))
      qm_wave_sync();
      if (l > k && l < 24) { const double lik = M[l * 24 + k]; for (int j = k + 1; j <= l; ++j) M[l * 24 + j] -= lik * M[j * 24 + k]; }
      qm_wave_sync();
    }
    // L y = rhs, Lᵀ x = y: lane i keeps its own entry, the resolved ones are published through LDS
    { double r = (l < 24) ? rhs[l] : 0.0;
      for (int k = 0; k < 24; ++k) { if (l == k) rhs[k] = r / M[k * 24 + k]; qm_wave_sync(); if (l > k && l < 24) r -= M[l * 24 + k] * rhs[k]; }
      qm_wave_sync();
      r = (l < 24) ? rhs[l] : 0.0;
      for (int k = 23; k >= 0; --k) { if (l == k) rhs[k] = r / M[k * 24 + k]; qm_wave_sync(); if (l < k) r -= M[k * 24 + l] * rhs[k]; }
      qm_wave_sync(); }
    if (l < 24) { const double vn = v[l] + a.h * rhs[l]; v[l] = vn; q[l] += a.h * vn; }
    qm_wave_sync();
  }
  // ---- hand-over: plant state, estimator-layout rbd state (FK of the arm tip at the final state), contact flags ----
  sim_dynamics_terms(mb, S, l, false);
  if (l < 24) { a.q[(size_t)b * 24 + l] = q[l]; a.v[(size_t)b * 24 + l] = v[l]; }
  double* r = a.rbd + (size_t)b * QM_NRBD;
  if (l < 3) { r[l] = q[3 + l]; r[3 + l] = q[l]; r[27 + l] = v[l]; }
  if (l == 3) { double E[9]; euler_E(q[3], q[4], E); const double thd[3] = {v[3], v[4], v[5]}; double w[3]; m3_mulv(E, thd, w); for (int i = 0; i < 3; ++i) r[24 + i] = w[i]; }
  if (l >= 6 && l < 24) { r[l] = q[l]; r[24 + l] = v[l]; }
  if (l == 4) { double qq[4]; mat_to_quat(S + SL_TIP + 27, qq); for (int i = 0; i < 3; ++i) r[48 + i] = S[SL_TIP + 24 + i]; for (int i = 0; i < 4; ++i) r[51 + i] = qq[i]; }
  if (l < 4) a.contact[b * 4 + l] = (a.p.foot_radius - S[SL_TIP + 6 * l + 2] > 0.0) ? 1 : 0;
  if (l < 12) a.force[(size_t)b * 12 + l] = (a.nsub > 0) ? fc[l] : 0.0;
  if (l == 0) { a.time[b] = time; a.ring_n[b * 2] = head; a.ring_n[b * 2 + 1] = cnt; a.status[b] = bad; }
}
