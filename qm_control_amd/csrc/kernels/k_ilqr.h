// k_ilqr.h — forward ROLLOUTS of the discrete iLQR behind the MPC entry points: the alternative solver whose settings block the
// reference loads (`ddp { algorithm … }`, qm_controllers/config/task.info:33-71, qm_interface/src/QMInterface.cpp:70) but never instantiates
// (QMController::setupMpc installs SqpMpc, QMController.cpp:287-288).  Restates the structure of [upstream ocs2_ddp] ILQR + LineSearchStrategy on the SQP's
// own transcription: grid (K0), LQ model + projection (K1a / K1b) and Riccati factors (K3) are shared; what iLQR adds is
//   mode 0  the NOMINAL rollout  x_{i+1} = RK2(x_i, u_i)  that makes the initial guess dynamically consistent (single shooting), and
//   mode 1  the line search's NONLINEAR rollout with feedback at step length a:
//             ũ_i = K_i (x~_i − x_i) + a k_i,   u~_i = u_i + a Pe_i + Px_i (x~_i − x_i) + Pu_i ũ_i,   x~_{i+1} = RK2(x~_i, u~_i)
//           (K_i = −L⁻ᵀ W, k_i = −L⁻ᵀ y: formed by K3's backward sweep on the matrix core and left in the stage record).
// One WAVEFRONT per instance, nodes in sequence (a rollout is a serial chain of nonlinear steps): lane l < 30 carries component l of the state and of the input;
// the four leg chains of the flow map run on lanes 0..3, the base block and the momentum sums are wave-uniform (v_readlane), the feedback products are one matrix row
// per lane.  The arithmetic follows the scalar statement term by term (same summation orders), which the first version — one THREAD per instance, 2.4 KB of scratch,
// ≈ 10 ms per rollout at any batch size — computed.
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

// component l of the flow map f(x, u) from lane-distributed x, u (lanes >= 30: 0).  Same terms, in the same order, as kin_base + kin_chain + flow_from_kin
__device__ __forceinline__ double ilqr_flow_lane(const double* mb, double xl, double ul, int l) {
  double xb[12], Kb[KW_LEG];
#pragma unroll
  for (int i = 0; i < 12; ++i) xb[i] = qm_bcast(xl, i);                 // momentum + base pose: wave-uniform
  kin_base<true>(mb, xb, Kb);
  // the leg chain of this lane (lanes >= 4 repeat chain l & 3; their result is not read)
  const int c = l & 3, frame = chain_to_contact(c);
  double q3[3];
#pragma unroll
  for (int jj = 0; jj < 3; ++jj) q3[jj] = __shfl(xl, 12 + 3 * c + jj, 64);
  double Rp[9], pp[3], Rj[9], Rq[9], Rn[9], p[3];
#pragma unroll
  for (int i = 0; i < 9; ++i) Rp[i] = Kb[KW_RB + i];
#pragma unroll
  for (int i = 0; i < 3; ++i) pp[i] = xb[6 + i];
#pragma unroll
  for (int jj = 0; jj < 3; ++jj) {
    const int j = 3 * c + jj;
    double t[3]; m3_mulv(Rp, mb + MB_JP + 3 * j, t);
    for (int i = 0; i < 3; ++i) pp[i] += t[i];
    m3_mul(Rp, mb + MB_JR + 9 * j, Rj);
    rot_axis_angle<true>(mb + MB_AXIS + 3 * j, q3[jj], Rq);
    m3_mul(Rj, Rq, Rn);
    for (int i = 0; i < 9; ++i) Rp[i] = Rn[i];
  }
  { double t[3]; m3_mulv(Rp, mb + MB_FP + 3 * frame, t); for (int i = 0; i < 3; ++i) p[i] = pp[i] + t[i]; }
  const double m = mb[MB_ROBOTMASS], im = 1.0 / m;
  double lin[3] = {0.0, 0.0, -9.81 * m}, ang[3] = {0.0, 0.0, 0.0};
#pragma unroll
  for (int i = 0; i < 4; ++i) {                                         // contacts in order, as flow_from_kin
    const int src = contact_to_chain(i);
    const double pf[3] = {qm_bcast(p[0], src), qm_bcast(p[1], src), qm_bcast(p[2], src)};
    const double F[3] = {qm_bcast(ul, 3 * i), qm_bcast(ul, 3 * i + 1), qm_bcast(ul, 3 * i + 2)};
    const double d[3] = {pf[0] - Kb[KW_COM], pf[1] - Kb[KW_COM + 1], pf[2] - Kb[KW_COM + 2]};
    double cr[3]; v3_cross(d, F, cr);
    for (int k = 0; k < 3; ++k) { lin[k] += F[k]; ang[k] += cr[k]; }
  }
  double wr[3]; v3_cross(Kb + KW_OM, Kb + KW_RW, wr);
  double f12[12];
#pragma unroll
  for (int k = 0; k < 3; ++k) { f12[k] = lin[k] * im; f12[3 + k] = ang[k] * im; f12[6 + k] = xb[k] + wr[k]; f12[9 + k] = Kb[KW_THD + k]; }
  double fl = (l < 30) ? ul : 0.0;                                      // joint rows: qdot_j = u_j
#pragma unroll
  for (int k = 0; k < 12; ++k) fl = (l == k) ? f12[k] : fl;
  return fl;
}
// x+ = x + dt/2 (f(x, u) + f(x + dt f(x, u), u))  — Heun step of the SRBD flow map (the transcription's discretisation, k_lq.h), lane-distributed
__device__ __forceinline__ double ilqr_rk2_lane(const double* mb, double xl, double ul, double dt, int l) {
  const double f1 = ilqr_flow_lane(mb, xl, ul, l);
  const double x2 = xl + dt * f1;
  const double f2 = ilqr_flow_lane(mb, x2, ul, l);
  return xl + 0.5 * dt * f1 + 0.5 * dt * f2;
}

__global__ void __launch_bounds__(64) qm_ilqr_rollout_kernel(QmRolloutArgs a) {
  const int b = blockIdx.x, l = threadIdx.x & 63;
  if (b >= a.B) return;
  const int n = a.n_nodes[b]; const double* mb = qm_table(a.mb);
  const bool lx = l < 30;
  double xc = lx ? a.x0[(size_t)b * 30 + l] : 0.0;
  if (a.mode == 0) {
    if (lx) a.x[(size_t)b * 30 + l] = xc;
    for (int i = 0; i + 1 < n; ++i) {
      const size_t nb = (size_t)i * a.B + b, nbn = (size_t)(i + 1) * a.B + b;
      if (a.node_ev[nb] != QM_EV_PRE) { const double uc = lx ? a.u[nb * 30 + l] : 0.0; xc = ilqr_rk2_lane(mb, xc, uc, a.node_dt[nb], l); xc = lx ? xc : 0.0; }
      if (lx) a.x[nbn * 30 + l] = xc;
    }
    return;
  }
  if (a.trial == 0 && l == 0) a.alpha[b] = a.st[ST_DDP_MAX_STEP];       // the line search starts at ddp.lineSearch.maxStepLength (task.info:67)
  if (a.done[b] != 0) return;
  const double al = (a.trial == 0) ? a.st[ST_DDP_MAX_STEP] : a.alpha[b];
  if (lx) a.xt[(size_t)b * 30 + l] = xc;
  for (int i = 0; i + 1 < n; ++i) {
    const size_t nb = (size_t)i * a.B + b, nbn = (size_t)(i + 1) * a.B + b;
    if (a.node_ev[nb] == QM_EV_PRE) { if (lx) { a.ut[nb * 30 + l] = 0.0; a.xt[nbn * 30 + l] = xc; } continue; }
    const double* rec = a.stage + ((size_t)b * a.nmax + i) * SR_SIZE;
    const int m = (int)rec[SR_SCAL]; const int md = (int)rec[SR_MODEF];
    const double dxl = lx ? xc - a.x[nb * 30 + l] : 0.0;
    // ũ = K dx + a k  (lane r: row r; the record holds the gain K = −L⁻ᵀ W and the offset k = −L⁻ᵀ y the backward sweep formed)
    const int lr = (l < m) ? l : 0;
    // every record entry this lane needs is requested BEFORE the first dependent use: 60 independent loads in flight instead of one L2 round trip per term
    const int r = lx ? l : 0; const bool hasPx = r >= 12 && r < 24; const int pr = hasPx ? r : 12;
    double w[30], px[30];
#pragma unroll
    for (int q = 0; q < 30; ++q) { w[q] = rec[SR_PP + lr * 30 + q]; px[q] = rec[SR_PX + pr * 30 + q]; }
    const double kl = rec[SR_KFF + lr], pel = rec[SR_PE + r];
    double v = (l < m) ? al * kl : 0.0;
#pragma unroll
    for (int q = 0; q < 30; ++q) v += ((l < m) ? w[q] : 0.0) * qm_bcast(dxl, q);
    // du = a Pe + Px dx + Pu ũ : Px has the 12 leg-joint-velocity rows; Pu's columns are unit vectors (stance force components, arm joint velocities) and one
    // 3 x 2 null-space block per swing leg (SR_SWG), in the column order K1b's projector uses (k_riccati.h forward rollout)
    int nst = 0; for (int k = 0; k < 4; ++k) nst += mode_flag(md, k) ? 1 : 0;
    double s = al * pel;
#pragma unroll
    for (int q = 0; q < 30; ++q) s += (hasPx ? px[q] : 0.0) * qm_bcast(dxl, q);      // uniform control flow around the broadcasts: the other rows add exact zeros
    const int kk = (r < 12) ? r / 3 : ((r < 24) ? chain_to_contact((r - 12) / 3) : 0), r3 = (r < 12) ? r % 3 : ((r < 24) ? (r - 12) % 3 : r - 24);
    int before_st = 0, before_sw = 0; for (int k = 0; k < 4; ++k) if (k < kk) { before_st += mode_flag(md, k) ? 1 : 0; before_sw += mode_flag(md, k) ? 0 : 1; }
    const bool stf = mode_flag(md, kk);
    const int col = (r < 12) ? 3 * before_st + r3 : ((r < 24) ? 3 * nst + 2 * before_sw : 3 * nst + 2 * (4 - nst) + r3);
    const double v1 = __shfl(v, col & 63, 64), v2 = __shfl(v, (col + 1) & 63, 64);
    if (r < 12) { if (stf) s += v1; }
    else if (r < 24) { if (!stf) s += rec[SR_SWG + 6 * kk + r3] * v1 + rec[SR_SWG + 6 * kk + 3 + r3] * v2; }
    else s += v1;
    const double uc = lx ? a.u[nb * 30 + l] + s : 0.0;
    if (lx) a.ut[nb * 30 + l] = uc;
    xc = ilqr_rk2_lane(mb, xc, uc, a.node_dt[nb], l); xc = lx ? xc : 0.0;
    if (lx) a.xt[nbn * 30 + l] = xc;
  }
  if (n >= 1 && lx) { const size_t nl = (size_t)(n - 1) * a.B + b; a.ut[nl * 30 + l] = 0.0; }
}
