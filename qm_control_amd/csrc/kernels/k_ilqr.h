// k_ilqr.h — forward ROLLOUTS of the discrete iLQR behind the MPC entry points: the alternative solver whose settings block the
// reference loads (`ddp { algorithm … }`, qm_controllers/config/task.info:33-71, qm_interface/src/QMInterface.cpp:70) but never instantiates
// (QMController::setupMpc installs SqpMpc, QMController.cpp:287-288).  Restates the structure of [upstream ocs2_ddp] ILQR + LineSearchStrategy on the SQP's
// own transcription: grid (K0), LQ model + projection (K1a / K1b) and Riccati factors (K3) are shared; what iLQR adds is
//   mode 0  the NOMINAL rollout  x_{i+1} = RK2(x_i, u_i)  that makes the initial guess dynamically consistent (single shooting), and
//   mode 1  the line search's NONLINEAR rollout with feedback at step length a:
//             ũ_i = −L_i⁻ᵀ (W_i (x~_i − x_i) + a y_i),   u~_i = u_i + a Pe_i + Px_i (x~_i − x_i) + Pu_i ũ_i,   x~_{i+1} = RK2(x~_i, u~_i)
//           (K_i = −L⁻ᵀ W, kff_i = −L⁻ᵀ y: the gains are applied through the factors K3 left in the stage record, never formed).
// One THREAD per instance, nodes in sequence: a rollout is a serial chain of nonlinear steps.  This is a functional path (parity-tested against the oracle's
// oracle/src/ilqr.h), not a tuned one: the per-node record reads are strided and the kernel spills; DESIGN.md says so.
#pragma once
#include "qm_dev_kin.h"
#include "k_grid.h"

struct QmRolloutArgs {
  const double* mb; const double* st;
  int B, nmax, mode, trial;
  const int* n_nodes; const double* node_dt; const int* node_ev;   // [B], [nmax][B]
  const double* x0;                                                 // [B][30]
  double* x; const double* u;                                       // [nmax][B][30] nominal trajectory (mode 0 writes x)
  const double* stage;                                              // [B][nmax][SR_SIZE] (mode 1)
  double* alpha; const int* done;                                   // [B]
  double* xt; double* ut;                                           // [nmax][B][30] trial trajectory (mode 1)
};

// x+ = x + dt/2 (f(x, u) + f(x + dt f(x, u), u))  — Heun step of the SRBD flow map (the transcription's discretisation, k_lq.h)
__device__ __forceinline__ void ilqr_rk2(const double* mb, const double* x, const double* u, double dt, double* xn) {
  double K[KW_SIZE], f1[30], x2[30], f2[30];
  kin_base(mb, x, K); for (int c = 0; c < 4; ++c) kin_leg(mb, c, x, u, K);
  flow_from_kin(mb, x, u, K, f1);
  for (int q = 0; q < 30; ++q) x2[q] = x[q] + dt * f1[q];
  kin_base(mb, x2, K); for (int c = 0; c < 4; ++c) kin_leg(mb, c, x2, u, K);
  flow_from_kin(mb, x2, u, K, f2);
  for (int q = 0; q < 30; ++q) xn[q] = x[q] + 0.5 * dt * f1[q] + 0.5 * dt * f2[q];
}

__global__ void __launch_bounds__(64) qm_ilqr_rollout_kernel(QmRolloutArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.B) return;
  const int n = a.n_nodes[b]; const double* mb = a.mb;
  double xc[30], uc[30], xn[30];
  if (a.mode == 0) {
    for (int q = 0; q < 30; ++q) { xc[q] = a.x0[(size_t)b * 30 + q]; a.x[(size_t)b * 30 + q] = xc[q]; }
    for (int i = 0; i + 1 < n; ++i) {
      const size_t nb = (size_t)i * a.B + b, nbn = (size_t)(i + 1) * a.B + b;
      if (a.node_ev[nb] != QM_EV_PRE) { for (int q = 0; q < 30; ++q) uc[q] = a.u[nb * 30 + q]; ilqr_rk2(mb, xc, uc, a.node_dt[nb], xn); for (int q = 0; q < 30; ++q) xc[q] = xn[q]; }
      for (int q = 0; q < 30; ++q) a.x[nbn * 30 + q] = xc[q];
    }
    return;
  }
  if (a.trial == 0) a.alpha[b] = a.st[ST_DDP_MAX_STEP];                 // the line search starts at ddp.lineSearch.maxStepLength (task.info:67)
  if (a.done[b] != 0) return;
  const double al = a.alpha[b];
  for (int q = 0; q < 30; ++q) { xc[q] = a.x0[(size_t)b * 30 + q]; a.xt[(size_t)b * 30 + q] = xc[q]; }
  for (int i = 0; i + 1 < n; ++i) {
    const size_t nb = (size_t)i * a.B + b, nbn = (size_t)(i + 1) * a.B + b;
    if (a.node_ev[nb] == QM_EV_PRE) { for (int q = 0; q < 30; ++q) { a.ut[nb * 30 + q] = 0.0; a.xt[nbn * 30 + q] = xc[q]; } continue; }
    const double* rec = a.stage + ((size_t)b * a.nmax + i) * SR_SIZE;
    const int m = (int)rec[SR_SCAL]; const int md = (int)rec[SR_MODEF];
    double dxi[30]; for (int q = 0; q < 30; ++q) dxi[q] = xc[q] - a.x[nb * 30 + q];
    // t = W dx + a y ;  v = L⁻ᵀ t (the record holds L⁻¹, lower triangle) ;  ũ = −v
    double tv[QM_MMAX], v[QM_MMAX];
    for (int r = 0; r < QM_MMAX; ++r) { double s = 0.0; if (r < m) { s = al * rec[SR_KFF + r]; for (int q = 0; q < 30; ++q) s += rec[SR_PP + r * 30 + q] * dxi[q]; } tv[r] = s; }
    for (int r = 0; r < QM_MMAX; ++r) { double s = 0.0; for (int q = r; q < QM_MMAX; ++q) if (q < m) s += rec[SR_RP + q * QM_MMAX + r] * tv[q]; v[r] = s; }
    for (int r = 0; r < QM_MMAX; ++r) v[r] = (r < m) ? -v[r] : 0.0;
    // du = a Pe + Px dx + Pu ũ : Px has the 12 leg-joint-velocity rows; Pu's columns are unit vectors (stance force components, arm joint velocities) and one
    // 3 x 2 null-space block per swing leg (SR_SWG), in the column order K1b's projector uses (k_riccati.h forward rollout)
    int nst = 0; for (int k = 0; k < 4; ++k) nst += mode_flag(md, k) ? 1 : 0;
    for (int r = 0; r < 30; ++r) {
      double s = al * rec[SR_PE + r];
      if (r >= 12 && r < 24) for (int q = 0; q < 30; ++q) s += rec[SR_PX + r * 30 + q] * dxi[q];
      const int kk = (r < 12) ? r / 3 : ((r < 24) ? chain_to_contact((r - 12) / 3) : 0), r3 = (r < 12) ? r % 3 : ((r < 24) ? (r - 12) % 3 : r - 24);
      int before_st = 0, before_sw = 0; for (int k = 0; k < 4; ++k) if (k < kk) { before_st += mode_flag(md, k) ? 1 : 0; before_sw += mode_flag(md, k) ? 0 : 1; }
      const bool stf = mode_flag(md, kk);
      if (r < 12) { if (stf) s += v[3 * before_st + r3]; }
      else if (r < 24) { if (!stf) { const int col = 3 * nst + 2 * before_sw; s += rec[SR_SWG + 6 * kk + r3] * v[col] + rec[SR_SWG + 6 * kk + 3 + r3] * v[col + 1]; } }
      else s += v[3 * nst + 2 * (4 - nst) + r3];
      uc[r] = a.u[nb * 30 + r] + s;
    }
    for (int q = 0; q < 30; ++q) a.ut[nb * 30 + q] = uc[q];
    ilqr_rk2(mb, xc, uc, a.node_dt[nb], xn);
    for (int q = 0; q < 30; ++q) { xc[q] = xn[q]; a.xt[nbn * 30 + q] = xc[q]; }
  }
  if (n >= 1) { const size_t nl = (size_t)(n - 1) * a.B + b; for (int q = 0; q < 30; ++q) a.ut[nl * 30 + q] = 0.0; }
}
