// k_ls.h — K4: filter line-search of the SQP step (SURVEY.md §8 a11 step 7, Appendix B.6 step 6;
// [upstream ocs2_sqp SqpSolver::takeStep + FilterLinesearch::acceptStep], settings task.info:81-82).
//
//   qm_ls_eval_kernel    one THREAD per (instance, node): value-only Heun defect, cost and equality residual of the
//                        trial trajectory x + alpha dx, u + alpha du   (scalar kinematics: lanes = instances)
//   qm_perf_sum_kernel   one WAVEFRONT per instance: sum over nodes (+ initial-state defect); for a trial point also the
//                        filter acceptance: accept, alpha <- alpha/2, or stop
//   qm_ls_apply_kernel   one thread per (instance, node): x += alpha dx, u += alpha du and the primal solution
//                        (u at PreEvent nodes copied from the previous node, last u repeated)
#pragma once
#include "qm_dev_kin.h"
#include "k_grid.h"
#include "k_ipm.h"

struct QmLsArgs {
  const double* mb; const double* st;
  int B, nmax;
  const int* n_nodes; const double* node_ts; const double* node_dt; const int* node_ev; const int* node_mode;
  const double* zvel; const double* zpos; const double* xref; const double* eeref;
  const double* x0;
  double* x; double* u;                 // [nmax][B][30] iterate (updated by apply)
  const double* dx; const double* du;   // [nmax][B][30]
  double* alpha;                        // [B] current trial step
  int* done;                            // [B] 0: searching, 1: accepted, 2: no step
  double* perf;                         // [nmax][B][PF_SIZE] node terms (baseline from K1, or trial)
  double* perf_sum;                     // [B][4]  merit, cost, dynSSE, eqSSE  of `perf`
  const double* base_sum;               // [B][4]
  const double* step_info;              // [B][4]  armijo, |dx|², |du|²
  double* xs; double* us;               // [nmax][B][30] primal solution out
  double* out_perf;                     // [B][10] baseline(4) after(4) alpha armijo
  int trial;                            // index of the current trial (0-based); qm_ls_tail_kernel: index of its first trial
  int max_trials;                       // qm_ls_tail_kernel: trials of this line search in all (the tail runs trial, trial + 1, ... < max_trials)
  int with_alpha;                       // perf_sum: 1 = initial-state defect of the trial iterate, 0 = of the base iterate
  int* open_cnt; int* tickets;          // [QM_LS_MAX_TRIALS] instances still searching after trial t / blocks that have passed; zeroed by the baseline sum
  volatile int* host_open;              // [QM_LS_MAX_TRIALS] host-visible copy of open_cnt[t], written by the last block of trial t (the host never copies flags)
  // policy at t0 straight from the line search's decision (round 6; null p0_x: off): what MPC_MRT_Interface::evaluatePolicy(t0) would read from the primal solution
  // x + alpha dx is written by the kernels that DECIDE alpha (qm_perf_sum for the first trial, qm_ls_tail for the later ones), so that the whole-body controller of the
  // step can start behind the decision — the batch's apply (xs, us on every node) and the policy kernel leave the critical path (qm_pipeline.h)
  const double* node_t; const double* p0_t; const double* p0_ev; const int* p0_modes; int p0_nev; double* p0_x; double* p0_u; int* p0_mode;
  // discrete iLQR (k_ilqr.h): the trial trajectory is a nonlinear ROLLOUT held in xt / ut (null for the SQP, whose trial point is x + alpha dx, u + alpha du);
  // merit = cost + rho sqrt(eqSSE), accepted when merit(a) < merit(0) + 1e-4 a armijo, a halved down to ddp.lineSearch.minStepLength
  const double* xt; const double* ut; int ilqr;
  // interior-point solver (k_ipm.h): slack and its direction [nmax][B][QM_NH], per-instance info (barrier parameter at [b * IPM_INFO]); null otherwise
  const double* ipm_s; const double* ipm_ds; const double* ipm_info;
};
#define QM_LS_MAX_TRIALS 16
#define LS_EVAL_LDS_BYTES ((64 * 31 + 64) * 8)      /* one 31-double row per thread + the wave's 64 step lengths: 16 KB per wave, eight waves per CU */

// cost value of one intermediate node (a2 + a6 + a7 + a5), not yet × dt; K must hold base, legs and arm
__device__ __forceinline__ double node_cost_value(const double* mb, const double* st, const double* x, const double* u, const double* K, int mode,
                                                   const double* xref, const double* eeref, double muPos, double muOri, bool intermediate, double* du /* [30] workspace (a per-thread LDS row) */) {
  double c = 0.0;
  if (intermediate) {
    int nst = 0; for (int k = 0; k < 4; ++k) nst += mode_flag(mode, k);
    _Pragma("unroll") for (int i = 0; i < 30; ++i) { const double d = x[i] - xref[i]; c += 0.5 * st[ST_Q + i] * d * d; du[i] = u[i]; }
    if (nst > 0) { _Pragma("unroll") for (int k = 0; k < 4; ++k) if (mode_flag(mode, k)) du[3 * k + 2] -= mb[MB_ROBOTMASS] * 9.81 / nst; }
    for (int i = 0; i < 30; ++i) { double s = 0.0; for (int j = 0; j < 30; ++j) s += st[ST_R + 30 * i + j] * du[j]; c += 0.5 * du[i] * s; }
    for (int i = 0; i < 6; ++i) {
      const double lo = mb[MB_QLO + 12 + i], hi = mb[MB_QHI + 12 + i], z = x[24 + i], mu = st[ST_JPOS_MU], de = st[ST_JPOS_DELTA];
      c += barrier_val(mu, de, z - lo) + barrier_val(mu, de, hi - z) - (barrier_val(mu, de, -lo) + barrier_val(mu, de, hi));
      const double vlo = st[ST_JVEL_LO + i], vhi = st[ST_JVEL_HI + i], w = u[24 + i], mv = st[ST_JVEL_MU], dv = st[ST_JVEL_DELTA];
      c += barrier_val(mv, dv, w - vlo) + barrier_val(mv, dv, vhi - w) - (barrier_val(mv, dv, -vlo) + barrier_val(mv, dv, vhi));
    }
    for (int k = 0; k < 4; ++k) if (mode_flag(mode, k)) {
      const double Fx = u[3 * k], Fy = u[3 * k + 1], Fz = u[3 * k + 2];
      const double h = st[ST_FRIC_COEF] * Fz - sqrt(Fx * Fx + Fy * Fy + st[ST_FRIC_REG]);
      c += barrier_val(st[ST_FRIC_MU], st[ST_FRIC_DELTA], h);
    }
  }
  double g[6], qee[4]; ee_error(K, eeref, eeref + 3, qee, g);
  for (int r = 0; r < 6; ++r) c += 0.5 * (r < 3 ? muPos : muOri) * g[r] * g[r];
  return c;
}
// the terms of an intermediate node's cost that need no kinematics (a2 tracking + input weight, a6 boxes, a7 friction cone), not yet × dt; the same arithmetic,
// in the same order, as the `intermediate` part of node_cost_value; u − u_nominal lives in registers
// RB: the input weight R has the zero pattern its construction gives it with the shipped task file (QMInterface.cpp:274-299: a diagonal R of task.info, leg block <- Jᵀ R₁₂ J with
// the block-diagonal feet Jacobian) — diagonal on the twelve force and six arm entries, one 3 x 3 block per leg; the host checks the blob entry by entry (qm_r_is_block_diagonal)
// and picks the instance.  The structured product adds the same non-zero terms in the same order as the dense one (an exact zero contributes nothing): bit-identical, 54
// instead of 900 multiply-adds and table entries per node.
// ctrack: the tracking term  sum_i 1/2 Q_i (x_i − xref_i)^2  summed in index order by the caller (the reference rows come in with the wave's cooperative loads)
// IPM: without the relaxed-barrier terms a6 / a7 (the interior-point solver carries the arm boxes and the friction cones as constraints)
template <bool RB, bool IPM = false>
__device__ __forceinline__ double node_cost_value_xu(const double* mb, const double* st, const double* x, const double* u, int mode, double ctrack) {
  double c = ctrack;
  int nst = 0; for (int k = 0; k < 4; ++k) nst += mode_flag(mode, k);
  const double wz = mb[MB_ROBOTMASS] * 9.81 / nst;                                   // weight-compensating normal force per stance foot (nst == 0: never selected)
  // u − u_nominal, entry j (static index): only the normal force of a stance foot has a nominal value.  With the structured R nothing of it is kept: every entry is
  // formed where its block needs it (the input sits in LDS) — thirty live doubles less across the barrier values below
#define QM_DU(j) ((((j) < 12) && (((j) % 3) == 2) && nst > 0 && mode_flag(mode, ((j) < 12 ? (j) / 3 : 0))) ? u[j] - wz : u[j])
  if (RB) {
    _Pragma("unroll") for (int i = 0; i < 30; ++i) {
      const int j0 = (i >= 12 && i < 24) ? 12 + 3 * ((i - 12) / 3) : i, j1 = (i >= 12 && i < 24) ? j0 + 3 : i + 1;
      double s = 0.0; _Pragma("unroll") for (int j = j0; j < j1; ++j) s += st[ST_R + 30 * i + j] * QM_DU(j);
      c += 0.5 * QM_DU(i) * s; }
  } else {
    double du[30]; _Pragma("unroll") for (int i = 0; i < 30; ++i) du[i] = QM_DU(i);
    _Pragma("unroll") for (int i = 0; i < 30; ++i) { double s = 0.0; _Pragma("unroll") for (int j = 0; j < 30; ++j) s += st[ST_R + 30 * i + j] * du[j]; c += 0.5 * du[i] * s; }
  }
#undef QM_DU
  if (IPM) return c;
  __builtin_amdgcn_sched_barrier(0);
  _Pragma("unroll") for (int i = 0; i < 6; ++i) {
    const double lo = mb[MB_QLO + 12 + i], hi = mb[MB_QHI + 12 + i], z = x[24 + i], mu = st[ST_JPOS_MU], de = st[ST_JPOS_DELTA];
    c += barrier_val(mu, de, z - lo) + barrier_val(mu, de, hi - z) - (barrier_val(mu, de, -lo) + barrier_val(mu, de, hi));
    const double vlo = st[ST_JVEL_LO + i], vhi = st[ST_JVEL_HI + i], w = u[24 + i], mv = st[ST_JVEL_MU], dv = st[ST_JVEL_DELTA];
    c += barrier_val(mv, dv, w - vlo) + barrier_val(mv, dv, vhi - w) - (barrier_val(mv, dv, -vlo) + barrier_val(mv, dv, vhi));
    __builtin_amdgcn_sched_barrier(0);      // one joint's eight barrier values at a time: scheduled together, the 52 independent logarithms of a node spill
  }
  for (int k = 0; k < 4; ++k) if (mode_flag(mode, k)) {
    const double Fx = u[3 * k], Fy = u[3 * k + 1], Fz = u[3 * k + 2];
    const double h = st[ST_FRIC_COEF] * Fz - sqrt(Fx * Fx + Fy * Fy + st[ST_FRIC_REG]);
    c += barrier_val(st[ST_FRIC_MU], st[ST_FRIC_DELTA], h);
    __builtin_amdgcn_sched_barrier(0);
  }
  return c;
}
// squared norm of the equality residual of one intermediate node (a8)
__device__ __forceinline__ double node_eq_sse(const double* st, const double* x, const double* u, const double* K, int mode, const double* zvel, const double* zpos) {
  const double gain = st[ST_POS_ERR_GAIN]; double s = 0.0;
  for (int k = 0; k < 4; ++k) {
    double v[3]; foot_velocity(x, K, k, v); const double pz = kin_foot(K, k)[2];
    if (mode_flag(mode, k)) { for (int r = 0; r < 3; ++r) { const double e = v[r] + ((r == 2 && gain != 0.0) ? gain * pz : 0.0); s += e * e; } }
    else {
      for (int r = 0; r < 3; ++r) s += u[3 * k + r] * u[3 * k + r];
      double bb = -zvel[k]; if (gain != 0.0) bb -= gain * zpos[k];
      const double e = bb + v[2] + (gain != 0.0 ? gain * pz : 0.0); s += e * e;
    }
  }
  return s;
}

// the wave's 64 consecutive rows of a node-major [nmax][B][30] array -> a [64][31] LDS tile (row r = thread r's vector), lanes running over consecutive doubles of
// the contiguous block (16 bytes each: whole 64-byte lines; rounds 1-4: every thread loaded its own row, 64 different lines per instruction — see qm_lq_kin_kernel);
// with `step`: the trial point base + alpha step, alpha per ROW from als[64]
__device__ __forceinline__ void ls_rows_in(double* tile, const double* base, const double* step, const double* als, size_t row0, size_t nrows, int l) {
  // three rounds of five 16-byte pieces per lane (a ROLLED outer loop: fully unrolled, the fifteen row / column / address sets of the three calls were kept live across the
  // whole kernel and spilled)
#pragma nounroll
  for (int t0 = 0; t0 < 15; t0 += 5) {
    double2 v[5], d[5]; int off[5]; double al[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int e = (t0 + k) * 64 + l; const int r = e / 15, c = 2 * (e - r * 15); size_t row = row0 + r; if (row >= nrows) row = nrows - 1;
      off[k] = r * 31 + c; v[k] = *(const double2*)(base + row * 30 + c);
      if (step) { d[k] = *(const double2*)(step + row * 30 + c); al[k] = als[r]; } }
#pragma unroll
    for (int k = 0; k < 5; ++k) { if (step) { v[k].x += al[k] * d[k].x; v[k].y += al[k] * d[k].y; } tile[off[k]] = v[k].x; tile[off[k] + 1] = v[k].y; }
  }
}
// Terms of ONE node at the trial point: x[30] (registers) and u (a per-thread LDS row) hold the node's trial state and input, ctrack its tracking cost; on return cost / eq
// (not yet × dt) and, for a regular node, x = the end point of the Heun step (to be compared with the next node's trial state).  Shared by the thread-per-(node, instance)
// kernel of the first trial and the per-instance tail kernel of the later ones: the same arithmetic in the same order.
template <bool RB, bool IPM>
__device__ __forceinline__ void ls_node_terms(const QmLsArgs& a, const double* mb, const double* st, double (&x)[30], const double* u, const double ctrack, const int nb, const int b, const double al,
                                               const bool active, const bool term, const bool reg, const double dt, const int mode, double& cost, double& eq, double& ipm_cost, double& ipm_res) {
  double K[KW_SIZE];
  // Ordered for a SMALL live set (256 registers, one LDS row per thread -> two waves per SIMD, every wavefront of the launch resident at once): first the cost terms
  // that need no kinematics (tracking, input weight, boxes, friction cone), then base -> arm (end-effector term), then the legs ONE AT A TIME, each consumed at once
  // by the equality residual and the flow map's momentum sums — the full workspace K[196] never exists.  A zero-length interval contributes neither cost nor
  // constraint residual and needs no second Heun stage; the guards also split this straight-line kernel into basic blocks, which bounds the scheduler's live ranges.
  if (reg && dt > 0.0) cost = node_cost_value_xu<RB, IPM>(mb, st, x, u, mode, ctrack);
  kin_base<true>(mb, x, K);
  if (active && (term || (reg && dt > 0.0))) {            // end-effector pose term: the intermediate soft constraint, or the final one at the terminal node (its only term)
    kin_arm<true>(mb, x, K); double g6[6], qee[4]; ee_error(K, a.eeref + nb * 7, a.eeref + nb * 7 + 3, qee, g6);
    const double mp = term ? st[ST_MU_EEF_POS] : st[ST_MU_EE_POS], mo = term ? st[ST_MU_EEF_ORI] : st[ST_MU_EE_ORI];
    for (int r = 0; r < 6; ++r) cost += 0.5 * (r < 3 ? mp : mo) * g6[r] * g6[r]; }
  // interior-point instance: ipm_cost = −mu Σ ln s and ipm_res = Σ (h − s)² of the trial point and its trial slacks s + alpha ds (not × dt / × dt)
  if (IPM && reg) {
    const double mub = a.ipm_info[b * IPM_INFO];
    for (int r = 0; r < QM_NH; ++r) if (ipm_row_on(r, mode)) {
      const double sr = a.ipm_s[(size_t)nb * QM_NH + r] + al * a.ipm_ds[(size_t)nb * QM_NH + r], h = ipm_row_value(mb, st, x, u, r);
      ipm_cost -= mub * log(sr); ipm_res += (h - sr) * (h - sr); }
  }
  if (reg) {
    const double mass = mb[MB_ROBOTMASS], im = 1.0 / mass, gain = st[ST_POS_ERR_GAIN];
    double f1[12], f2[12];
    // the swing-height references of the four contacts, requested TOGETHER ahead of the legs (round 6: each swing leg loaded its own inside its turn of the loop — one
    // memory round trip per swing leg, waited for on the spot, with one leg scheduled at a time)
    // (the structured product instance only: the dense and the interior-point instances, which are not on the benchmark's path, spill the eight extra doubles — 220 -> 368 B)
    constexpr bool ZPRE = RB && !IPM;
    double zv4[4] = {0.0, 0.0, 0.0, 0.0}, zp4[4] = {0.0, 0.0, 0.0, 0.0};
    if (ZPRE && dt > 0.0) { _Pragma("unroll") for (int k = 0; k < 4; ++k) zv4[k] = a.zvel[nb * 4 + k]; if (gain != 0.0) { _Pragma("unroll") for (int k = 0; k < 4; ++k) zp4[k] = a.zpos[nb * 4 + k]; } }
    { double lin[3] = {0.0, 0.0, -9.81 * mass}, ang[3] = {0.0, 0.0, 0.0};
      _Pragma("unroll") for (int c = 0; c < 4; ++c) {
        kin_leg<true>(mb, c, x, u, K); const int k = chain_to_contact(c); const double* L = K + KW_LEG + KW_LEGSZ * c;
        const double d[3] = {L[18] - K[KW_COM], L[19] - K[KW_COM + 1], L[20] - K[KW_COM + 2]};
        double t[3]; v3_cross(d, u + 3 * k, t);
        for (int q = 0; q < 3; ++q) { lin[q] += u[3 * k + q]; ang[q] += t[q]; }
        if (dt > 0.0) {                                       // equality residual of this leg's contact (a8): same terms as node_eq_sse
          double w[3]; v3_cross(K + KW_OM, d, w); const double v[3] = {x[0] + w[0] + L[21], x[1] + w[1] + L[22], x[2] + w[2] + L[23]}; const double pz = L[20];
          if (mode_flag(mode, k)) { for (int r = 0; r < 3; ++r) { const double e = v[r] + ((r == 2 && gain != 0.0) ? gain * pz : 0.0); eq += e * e; } }
          else {
            for (int r = 0; r < 3; ++r) eq += u[3 * k + r] * u[3 * k + r];
            double bb = ZPRE ? -zv4[k] : -a.zvel[nb * 4 + k]; if (gain != 0.0) bb -= gain * (ZPRE ? zp4[k] : a.zpos[nb * 4 + k]);
            const double e = bb + v[2] + (gain != 0.0 ? gain * pz : 0.0); eq += e * e;
          }
        }
        __builtin_amdgcn_sched_barrier(0);      // one leg at a time (scheduled together, the four chains' temporaries do not fit)
      }
      double wr[3]; v3_cross(K + KW_OM, K + KW_RW, wr);
      for (int q = 0; q < 3; ++q) { f1[q] = lin[q] * im; f1[3 + q] = ang[q] * im; f1[6 + q] = x[q] + wr[q]; f1[9 + q] = K[KW_THD + q]; } }
    _Pragma("unroll") for (int q = 0; q < 12; ++q) f2[q] = f1[q];
    if (dt > 0.0) {
      double x2[30];
      _Pragma("unroll") for (int q = 0; q < 30; ++q) x2[q] = x[q] + dt * ((q < 12) ? f1[q < 12 ? q : 0] : u[q]);
      kin_base<true>(mb, x2, K);
      double lin[3] = {0.0, 0.0, -9.81 * mass}, ang[3] = {0.0, 0.0, 0.0};
      _Pragma("unroll") for (int c = 0; c < 4; ++c) {
        kin_leg<true>(mb, c, x2, u, K); const int k = chain_to_contact(c); const double* L = K + KW_LEG + KW_LEGSZ * c;
        const double d[3] = {L[18] - K[KW_COM], L[19] - K[KW_COM + 1], L[20] - K[KW_COM + 2]};
        double t[3]; v3_cross(d, u + 3 * k, t);
        for (int q = 0; q < 3; ++q) { lin[q] += u[3 * k + q]; ang[q] += t[q]; }
        __builtin_amdgcn_sched_barrier(0);
      }
      double wr[3]; v3_cross(K + KW_OM, K + KW_RW, wr);
      for (int q = 0; q < 3; ++q) { f2[q] = lin[q] * im; f2[3 + q] = ang[q] * im; f2[6 + q] = x2[q] + wr[q]; f2[9 + q] = K[KW_THD + q]; }
    }
    // the Heun step's end point takes x's place (the joint rows of the flow map are the input's joint velocities in both stages); the input row is free after this
    _Pragma("unroll") for (int q = 0; q < 30; ++q) {
      const double fa = (q < 12) ? f1[q < 12 ? q : 0] : u[q], fb = (q < 12) ? f2[q < 12 ? q : 0] : u[q];
      x[q] = x[q] + 0.5 * dt * fa + 0.5 * dt * fb; }
  }
}
// One THREAD per (instance, node), the wave's inputs moved together (ls_rows_in).  No thread leaves before the last cooperative load: a thread without work (its
// instance has finished the search, a row behind the instance's last node) only helps to move data; the whole wave leaves at once when none of its threads has work.
template <bool RB, bool IPM = false> __global__ void __launch_bounds__(64, 2) qm_ls_eval_kernel_t(QmLsArgs a) {
  const int l = threadIdx.x & 63;
  const size_t g0 = (size_t)blockIdx.x * 64, nrows = (size_t)a.nmax * a.B;      // first row of this wave's block (blockDim.x == 64)
  size_t g = g0 + l; const bool inrange = g < nrows; if (!inrange) g = nrows - 1;
  const int i = (int)(g / a.B), b = (int)(g - (size_t)i * a.B);
  const int n = a.n_nodes[b];
  const bool active = inrange && i < n && a.done[b] == 0;
  if (__ballot(active) == 0ull) return;
  const int nb = (int)g; const double al = a.alpha[b];
  const double* mb = qm_table(a.mb); const double* st = qm_table(a.st);
  extern __shared__ double qm_smem[];                  // LS_EVAL_LDS_BYTES: [64][31] rows (the trial state on its way in, then each thread's trial input, at the end the next node's trial state) + alpha[64]
  double* rows = qm_smem; double* als = qm_smem + 64 * 31; double* u = rows + l * 31;
  als[l] = al; qm_wave_sync();
  double x[30];
  ls_rows_in(rows, a.xt ? a.xt : a.x, a.xt ? nullptr : a.dx, als, g0, nrows, l); qm_wave_sync();
  _Pragma("unroll") for (int q = 0; q < 30; ++q) x[q] = u[q];
  qm_wave_sync();
  double ctrack = 0.0;                                   // tracking term of the intermediate cost (a2), index order; the reference rows take the tile on their way through
  ls_rows_in(rows, a.xref, nullptr, als, g0, nrows, l); qm_wave_sync();
  _Pragma("unroll") for (int q = 0; q < 30; ++q) { const double d = x[q] - u[q]; ctrack += 0.5 * st[ST_Q + q] * d * d; }
  qm_wave_sync();
  ls_rows_in(rows, a.ut ? a.ut : a.u, a.ut ? nullptr : a.du, als, g0, nrows, l); qm_wave_sync();
  const int ev = a.node_ev[nb]; const double dt = a.node_dt[nb]; const int mode = a.node_mode[nb];
  const bool term = (i == n - 1), pre = !term && ev == QM_EV_PRE, reg = active && !term && !pre;
  double cost = 0.0, eq = 0.0, ipm_cost = 0.0, ipm_res = 0.0;
  ls_node_terms<RB, IPM>(a, mb, st, x, u, ctrack, nb, b, al, active, term, reg, dt, mode, cost, eq, ipm_cost, ipm_res);
  // the next node's trial state (the rows one node further: the same instances, the same step lengths) comes in through the tile the inputs leave; a PreEvent node's
  // defect is the jump x_i − x_{i+1} (identity jump map), unweighted
  qm_wave_sync();
  ls_rows_in(rows, a.xt ? a.xt : a.x, a.xt ? nullptr : a.dx, als, g0 + (size_t)a.B, nrows, l); qm_wave_sync();
  double s = 0.0;
  _Pragma("unroll") for (int q = 0; q < 30; ++q) { const double d = x[q] - u[q]; s += d * d; }
  if (!active) return;
  double* pf = a.perf + (size_t)nb * PF_SIZE;
  if (term) { pf[0] = cost; pf[1] = 0.0; pf[2] = 0.0; }
  else if (pre) { pf[0] = 0.0; pf[1] = s; pf[2] = 0.0; }
  else { pf[0] = cost * dt + ipm_cost; pf[1] = dt * s; pf[2] = dt * (eq + ipm_res); }
}

#define qm_ls_eval_kernel qm_ls_eval_kernel_t<true>            /* R block diagonal (the shipped task file) */
#define qm_ls_eval_dense_kernel qm_ls_eval_kernel_t<false>     /* any R */
#define qm_ls_eval_ipm_kernel qm_ls_eval_kernel_t<false, true>  /* interior-point solver (slot 3) */
// exact zero pattern of the input weight: diag(12) + four 3 x 3 leg blocks + diag(6)  (host side: which instance of qm_ls_eval to launch)
inline bool qm_r_is_block_diagonal(const double* st) {
  for (int i = 0; i < 30; ++i) for (int j = 0; j < 30; ++j) {
    const bool in = (i >= 12 && i < 24) ? (j >= 12 + 3 * ((i - 12) / 3) && j < 15 + 3 * ((i - 12) / 3)) : (j == i);
    if (!in && st[ST_R + 30 * i + j] != 0.0) return false; }
  return true;
}

// Filter line-search decision of ONE instance for the trial at step length al0 with the sums {cost c, dynamics SSE d, equality SSE e} ([upstream ocs2_sqp
// FilterLinesearch::acceptStep, recalled]): 1 = accepted (done = 1, out_perf), 2 = the search stops without a step (done = 2, alpha = 0), 0 = goes on
// at alpha[b] = al0 / 2.  One thread per instance calls it (the deciding lane of qm_perf_sum_kernel, thread 0 of a tail block).
// (the _v form takes the instance's baseline sums {merit, -, dynamics SSE, equality SSE} and step figures {Armijo metric, |dx|², |du|²} as VALUES: qm_perf_sum_kernel requests them
//  with everything else at its top instead of one memory round trip each on the deciding lane's chain)
__device__ __forceinline__ int qm_ls_filter_decide_v(const QmLsArgs& a, const int b, const double al0, const double c, const double d, const double e, const double (&bs)[4], const double (&si)[3]) {
  const double gMax = qm_ms_param(a.st, ST_G_MAX), gMin = qm_ms_param(a.st, ST_G_MIN), gammaC = 1e-6, armijoFactor = 1e-4, alphaDecay = 0.5, alphaMin = 1e-4;
  const double ps[4] = {c, c, d, e};
  const double theta0 = sqrt(bs[2] + bs[3]), theta = sqrt(ps[2] + ps[3]);
  double al = al0; const double armijo = si[0];
  bool acc;
  if (theta > gMax) acc = theta < (1.0 - gammaC) * theta0;
  else if (theta < gMin && theta0 < gMin && al * armijo < 0.0) acc = ps[0] < bs[0] + armijoFactor * al * armijo;
  else acc = ps[0] < (bs[0] - gammaC * theta0) || theta < (1.0 - gammaC) * theta0;
  if (acc) { a.done[b] = 1; for (int q = 0; q < 4; ++q) a.out_perf[b * 10 + 4 + q] = ps[q]; a.out_perf[b * 10 + 8] = al; return 1; }
  al *= alphaDecay;
  const double dxn = sqrt(si[1]), dun = sqrt(si[2]);
  if ((al * dun < qm_ms_param(a.st, ST_DELTA_TOL) && al * dxn < qm_ms_param(a.st, ST_DELTA_TOL)) || !(al >= alphaMin)) { a.done[b] = 2; a.alpha[b] = 0.0; return 2; }
  a.alpha[b] = al;
  return 0;
}
__device__ __forceinline__ int qm_ls_filter_decide(const QmLsArgs& a, const int b, const double al0, const double c, const double d, const double e) {
  const double bs[4] = {a.base_sum[b * 4], a.base_sum[b * 4 + 1], a.base_sum[b * 4 + 2], a.base_sum[b * 4 + 3]}, si[3] = {a.step_info[b * 4], a.step_info[b * 4 + 1], a.step_info[b * 4 + 2]};
  return qm_ls_filter_decide_v(a, b, al0, c, d, e, bs, si);
}
// entries of the primal solution of instance b for the step length al (0 when no step is taken) — the arithmetic of qm_ls_apply_kernel, which writes them for every node
__device__ __forceinline__ double qm_ls_primal_x(const QmLsArgs& a, const int b, const int i, const int q, const double al) { const size_t nb = (size_t)i * a.B + b; return a.x[nb * 30 + q] + al * a.dx[nb * 30 + q]; }
__device__ __forceinline__ double qm_ls_primal_u(const QmLsArgs& a, const int b, const int n, const int i, const int q, const double al) {
  int j = (i == n - 1) ? n - 2 : i; if (j < 0) j = 0;                      // input of the primal solution: own node, or the closest earlier non-event node
  while (j > 0 && a.node_ev[j * a.B + b] == QM_EV_PRE) --j;
  const size_t jb = (size_t)j * a.B + b; const bool evj = (a.node_ev[jb] == QM_EV_PRE) || n < 2;
  return evj ? 0.0 : a.u[jb * 30 + q] + al * a.du[jb * 30 + q];
}
// evaluatePolicy at t0 ([upstream] linear interpolation of the primal solution, qm_policy_body) on the primal solution the decided step length defines; one wave:
// lanes 0..29 the state, 32..61 the input, lane 0 the mode of the schedule at t0.  Bit-identical to qm_policy_kernel run behind qm_ls_apply_kernel (same products, same order).
__device__ __forceinline__ void qm_ls_policy_at_t0(const QmLsArgs& a, const int b, const int l, const double al) {
  if (!a.p0_x) return;
  const int n = a.n_nodes[b]; const double t = a.p0_t[b];
  int idx; double ai; grid_policy_segment(a.node_t, a.node_ev, n, a.B, b, t, &idx, &ai);
  const int i1 = (n > 1) ? idx + 1 : idx;
  if (l < 30) a.p0_x[(size_t)b * 30 + l] = ai * qm_ls_primal_x(a, b, idx, l, al) + (1.0 - ai) * qm_ls_primal_x(a, b, i1, l, al);
  else if (l >= 32 && l < 62) { const int q = l - 32; a.p0_u[(size_t)b * 30 + q] = ai * qm_ls_primal_u(a, b, n, idx, q, al) + (1.0 - ai) * qm_ls_primal_u(a, b, n, i1, q, al); }
  if (l == 0) a.p0_mode[b] = a.p0_modes[(size_t)b * (a.p0_nev + 1) + grid_find_index(a.p0_ev + (size_t)b * a.p0_nev, a.p0_nev, t)];
}
// One WAVEFRONT per instance.  Sum of the node terms -> perf_sum[b] = {merit, cost, dynSSE, eqSSE} (lanes stride over the
// nodes, DPP wave reduction).  with_alpha == 0: baseline of the current iterate, also arms the line search (alpha = 1, done = 0);
// with_alpha == 1: the trial point, followed by the filter line-search decision of this instance
// ([upstream ocs2_sqp SqpSolver::takeStep / FilterLinesearch]): accept, halve alpha, or give up.
// returns true when the instance is still searching after this trial; acc / al (meaningful on lane 0): the instance stands at an ACCEPTED step of length al after this
// call (what `done[b] == 1 ? alpha[b] : 0` reads back — lane 0 knows it without the round trip)
// Round 6: every load whose address depends on (b, l) only — the instance's state words, the first two node terms of the lane, its row of the initial-state defect, the
// baseline sums and step figures the decision needs — is requested at the top, TOGETHER: the kernel is one dependent chain per instance, and each of those used to be a memory
// round trip of its own on it (≈ 20 in a row: 31 µs for a few hundred flops).  Same sums in the same order.
__device__ __forceinline__ bool qm_perf_sum_body(const QmLsArgs& a, const int b, const int l, int& acc, double& al_acc) {
  const int with_alpha = a.with_alpha;
  const int done0 = with_alpha ? a.done[b] : 0;
  const int n = a.n_nodes[b];
  const double al0 = with_alpha ? a.alpha[b] : 0.0;
  const int j0 = (l < a.nmax) ? l : a.nmax - 1, j1 = (l + 64 < a.nmax) ? l + 64 : a.nmax - 1;      // (clamped: in bounds whatever n is)
  const double* pf0 = a.perf + (size_t)(j0 * a.B + b) * PF_SIZE; const double* pf1 = a.perf + (size_t)(j1 * a.B + b) * PF_SIZE;
  const double t00 = pf0[0], t01 = pf0[1], t02 = pf0[2], t10 = pf1[0], t11 = pf1[1], t12 = pf1[2];
  double xa = 0.0, xb = 0.0, xc = 0.0;
  if (l < 30) { xa = a.x0[(size_t)b * 30 + l]; if (with_alpha && a.xt) xb = a.xt[b * 30 + l]; else { xb = a.x[b * 30 + l]; xc = a.dx[b * 30 + l]; } }
  double bs[4] = {0.0, 0.0, 0.0, 0.0}, si[3] = {0.0, 0.0, 0.0};
  if (with_alpha) { bs[0] = a.base_sum[b * 4]; bs[1] = a.base_sum[b * 4 + 1]; bs[2] = a.base_sum[b * 4 + 2]; bs[3] = a.base_sum[b * 4 + 3]; si[0] = a.step_info[b * 4]; si[1] = a.step_info[b * 4 + 1]; si[2] = a.step_info[b * 4 + 2]; }
  acc = (done0 == 1) ? 1 : 0; al_acc = acc ? al0 : 0.0;
  if (with_alpha && done0 != 0) return false;
  double c = 0.0, d = 0.0, e = 0.0;
  if (l < n) { c += t00; d += t01; e += t02; }
  if (l + 64 < n) { c += t10; d += t11; e += t12; }
  for (int i = l + 128; i < n; i += 64) { const double* pf = a.perf + (size_t)(i * a.B + b) * PF_SIZE; c += pf[0]; d += pf[1]; e += pf[2]; }
  if (l < 30) { const double x0t = (with_alpha && a.xt) ? xb : xb + al0 * xc; const double dd = xa - x0t; d += dd * dd; }
  c = qm_wave_sum(c); d = qm_wave_sum(d); e = qm_wave_sum(e);
  if (l != 0) return false;
  a.perf_sum[b * 4] = c; a.perf_sum[b * 4 + 1] = c; a.perf_sum[b * 4 + 2] = d; a.perf_sum[b * 4 + 3] = e;
  if (!with_alpha) {
    a.alpha[b] = 1.0; a.done[b] = 0;
    for (int q = 0; q < 4; ++q) { a.out_perf[b * 10 + q] = a.perf_sum[b * 4 + q]; a.out_perf[b * 10 + 4 + q] = a.perf_sum[b * 4 + q]; }
    a.out_perf[b * 10 + 8] = 0.0;
    return false;
  }
  if (a.trial == 0) a.out_perf[b * 10 + 9] = si[0];
  if (a.ilqr) {
    const double rho = a.st[ST_DDP_PENALTY]; const double armijo0 = si[0];
    const double m0 = bs[1] + rho * sqrt(bs[3]), mt = c + rho * sqrt(e);
    if (a.trial == 0) { a.out_perf[b * 10] = m0; a.out_perf[b * 10 + 4] = m0; }                       // (the Riccati prologue wrote the SQP's merit = cost)
    if (mt < m0 + 1e-4 * al0 * armijo0) { a.done[b] = 1; a.out_perf[b * 10 + 4] = mt; a.out_perf[b * 10 + 5] = c; a.out_perf[b * 10 + 6] = d; a.out_perf[b * 10 + 7] = e; a.out_perf[b * 10 + 8] = al0; acc = 1; al_acc = al0; return false; }
    const double an = al0 * 0.5;
    if (!(an >= a.st[ST_DDP_MIN_STEP])) { a.done[b] = 2; a.alpha[b] = 0.0; return false; }
    a.alpha[b] = an; return true;
  }
  const int code = qm_ls_filter_decide_v(a, b, al0, c, d, e, bs, si);
  acc = (code == 1) ? 1 : 0; al_acc = acc ? al0 : 0.0;
  return code == 0;
}
__global__ void __launch_bounds__(64) qm_perf_sum_kernel(QmLsArgs a) {
  const int b = blockIdx.x, l = threadIdx.x & 63;
  if (b >= a.B) return;
  int acc = 0; double al = 0.0;
  const bool open = qm_perf_sum_body(a, b, l, acc, al);
  if (a.with_alpha && a.p0_x) {      // the policy at t0 of what the decision stands at: the accepted step, or the iterate itself while the search goes on / when it gave up (qm_ls_tail overwrites it on acceptance)
    acc = __shfl(acc, 0, 64); al = __shfl(al, 0, 64);      // (lane 0 made the decision)
    qm_ls_policy_at_t0(a, b, l, al);
  }
  // the host only needs to know whether ANY instance is still searching: count them, and let the block that arrives last publish the count
  // in host-visible memory (one stream synchronisation instead of a flag copy + two)
  if (a.with_alpha && a.open_cnt && l == 0) {
    const int t = a.trial;
    if (open) atomicAdd(a.open_cnt + t, 1);
    if (a.tickets) {      // (only the host-driven trial loop waits for the published count; with the device-side tail — tickets == nullptr — the counts are read back on demand)
      __threadfence();
      if (atomicAdd(a.tickets + t, 1) == a.B - 1) { a.host_open[t] = atomicAdd(a.open_cnt + t, 0); __threadfence_system(); }
    }
  }
}


// ---- the trials AFTER the first, on the device (round 6) ----
// The first trial (alpha = 1) is evaluated for the whole batch by qm_ls_eval + qm_perf_sum.  Rounds 1-5 then let the HOST loop: read the count of the instances still
// searching, launch both kernels again over the whole batch, spin — one round trip per trial, and every instance paid for the batch's worst one (a warm-started solve needs a
// second trial on ~ 15 % of its instances and a third on ~ 1 %: the full step of a warm start often does not reduce the constraint violation theta ~ 0.2 > g_max it starts
// from, tools/warm_ls_histogram.py).  Now ONE launch finishes the search: one 256-thread workgroup per instance, gone at once unless its instance is still searching;
// it evaluates TWO step lengths side by side (threads 0..127 the nodes at alpha, 128..255 at alpha / 2 — the decisions are taken in the filter's order, the second
// evaluation is wasted when the first is accepted), sums the node terms exactly as qm_perf_sum does (lane-strided partial sums of one wave + the DPP reduction: a trial's
// merit is bit-identical to what the two-kernel path computes), decides, repeats until the filter accepts or stops.  qm_ls_apply_kernel follows for the whole batch (it is no
// longer speculative: nothing waits for the host); the policy at t0, which is all the step's whole-body controller needs, is written by the deciding kernels.  The host never
// waits inside the line search.
#define LS_TAIL_BLOCK 256
#define LS_TAIL_SLOTS 2
#define LS_TAIL_NODES (LS_TAIL_BLOCK / LS_TAIL_SLOTS)
#define LS_TAIL_LDS_DOUBLES(nmax) (LS_TAIL_BLOCK * 31 + LS_TAIL_SLOTS * (nmax) * 3 + LS_TAIL_SLOTS * 4 + 4)      /* input rows | node terms [slot][node][3] | sums [slot][4] | control */
#define LS_TAIL_LDS_BYTES(nmax) (LS_TAIL_LDS_DOUBLES(nmax) * 8)
template <bool RB> __global__ void __launch_bounds__(LS_TAIL_BLOCK) qm_ls_tail_kernel_t(QmLsArgs a) {
  const int b = blockIdx.x, tid = threadIdx.x, l = tid & 63;
  if (b >= a.B || a.done[b] != 0) return;                                      // (uniform over the workgroup; nobody else writes done[b] during this launch)
  const int n = a.n_nodes[b]; const size_t nrows = (size_t)a.nmax * a.B;
  const double* mb = qm_table(a.mb); const double* st = qm_table(a.st);
  extern __shared__ double qm_smem[];
  double* u = qm_smem + tid * 31; double* pfl = qm_smem + LS_TAIL_BLOCK * 31; double* sums = pfl + LS_TAIL_SLOTS * a.nmax * 3; double* ctl = sums + LS_TAIL_SLOTS * 4;
  const int slot = tid / LS_TAIL_NODES, ni = tid - slot * LS_TAIL_NODES;
  double al_first = a.alpha[b]; int trial = a.trial, state = 0;
  __syncthreads();                                                             // every thread has read alpha[b] before thread 0's decisions rewrite it
  while (trial < a.max_trials) {
    const double al = slot ? al_first * 0.5 : al_first;                        // (alphaDecay = 0.5: the same product the decision forms)
    for (int i = ni; i < n; i += LS_TAIL_NODES) {
      const size_t nb = (size_t)i * a.B + b;
      double x[30]; double ctrack = 0.0;
      _Pragma("unroll") for (int q = 0; q < 30; q += 2) { const double2 v = *(const double2*)(a.x + nb * 30 + q), dv = *(const double2*)(a.dx + nb * 30 + q); double v0 = v.x, v1 = v.y; v0 += al * dv.x; v1 += al * dv.y; x[q] = v0; x[q + 1] = v1; }
      _Pragma("unroll") for (int q = 0; q < 30; q += 2) { const double2 r = *(const double2*)(a.xref + nb * 30 + q); u[q] = r.x; u[q + 1] = r.y; }
      _Pragma("unroll") for (int q = 0; q < 30; ++q) { const double d = x[q] - u[q]; ctrack += 0.5 * st[ST_Q + q] * d * d; }
      _Pragma("unroll") for (int q = 0; q < 30; q += 2) { const double2 v = *(const double2*)(a.u + nb * 30 + q), dv = *(const double2*)(a.du + nb * 30 + q); double v0 = v.x, v1 = v.y; v0 += al * dv.x; v1 += al * dv.y; u[q] = v0; u[q + 1] = v1; }
      const int ev = a.node_ev[nb]; const double dt = a.node_dt[nb]; const int mode = a.node_mode[nb];
      const bool term = (i == n - 1), pre = !term && ev == QM_EV_PRE, reg = !term && !pre;
      double cost = 0.0, eq = 0.0, ipm_cost = 0.0, ipm_res = 0.0;
      ls_node_terms<RB, false>(a, mb, st, x, u, ctrack, (int)nb, b, al, true, term, reg, dt, mode, cost, eq, ipm_cost, ipm_res);
      size_t nn = nb + (size_t)a.B; if (nn >= nrows) nn = nrows - 1;         // (the terminal node's row behind the array is never used: clamped like the tile loads)
      double sq = 0.0;
      _Pragma("unroll") for (int q = 0; q < 30; q += 2) { const double2 v = *(const double2*)(a.x + nn * 30 + q), dv = *(const double2*)(a.dx + nn * 30 + q); double v0 = v.x, v1 = v.y; v0 += al * dv.x; v1 += al * dv.y;
                                                          const double d0 = x[q] - v0; sq += d0 * d0; const double d1 = x[q + 1] - v1; sq += d1 * d1; }
      double* pf = pfl + ((size_t)slot * a.nmax + i) * 3;
      if (term) { pf[0] = cost; pf[1] = 0.0; pf[2] = 0.0; }
      else if (pre) { pf[0] = 0.0; pf[1] = sq; pf[2] = 0.0; }
      else { pf[0] = cost * dt + ipm_cost; pf[1] = dt * sq; pf[2] = dt * (eq + ipm_res); }
    }
    __syncthreads();
    if (ni < 64) {                                                             // the first wave of each slot: qm_perf_sum_body's sums, term by term
      double c = 0.0, d = 0.0, e = 0.0;
      for (int i = l; i < n; i += 64) { const double* pf = pfl + ((size_t)slot * a.nmax + i) * 3; c += pf[0]; d += pf[1]; e += pf[2]; }
      if (l < 30) { const double x0t = a.x[b * 30 + l] + al * a.dx[b * 30 + l]; const double dd = a.x0[(size_t)b * 30 + l] - x0t; d += dd * dd; }
      c = qm_wave_sum(c); d = qm_wave_sum(d); e = qm_wave_sum(e);
      if (l == 0) { sums[slot * 4] = c; sums[slot * 4 + 1] = d; sums[slot * 4 + 2] = e; }
    }
    __syncthreads();
    if (tid == 0) {
      int stt = 0, used = 0; double aln = al_first;
      for (int sidx = 0; sidx < LS_TAIL_SLOTS && stt == 0 && trial + sidx < a.max_trials; ++sidx) {
        const double c = sums[sidx * 4], d = sums[sidx * 4 + 1], e = sums[sidx * 4 + 2];
        a.perf_sum[b * 4] = c; a.perf_sum[b * 4 + 1] = c; a.perf_sum[b * 4 + 2] = d; a.perf_sum[b * 4 + 3] = e;
        stt = qm_ls_filter_decide(a, b, aln, c, d, e); ++used;
        if (stt == 0) { if (a.open_cnt) atomicAdd(a.open_cnt + trial + sidx, 1); aln *= 0.5; }      // (open_cnt[t] > 0: some instance went on to trial t + 1 — how the host learns the number of trials)
      }
      ctl[0] = (double)stt; ctl[1] = aln; ctl[2] = (double)(trial + used);
    }
    __syncthreads();
    state = (int)ctl[0]; al_first = ctl[1]; trial = (int)ctl[2];
    if (state != 0) break;
  }
  // the batch's qm_ls_apply_kernel runs BEHIND this launch (it reads the final done / alpha of every instance); what the whole-body controller of the step needs — the
  // policy at t0 — is refreshed here for an instance that accepted a shortened step (al_first is the accepted step length: not advanced on acceptance)
  if (state == 1 && tid < 64) qm_ls_policy_at_t0(a, b, tid, al_first);
}
#define qm_ls_tail_kernel qm_ls_tail_kernel_t<true>
#define qm_ls_tail_dense_kernel qm_ls_tail_kernel_t<false>

// one thread per (node, instance, component): consecutive threads touch consecutive doubles
__global__ void qm_ls_apply_kernel(QmLsArgs a) {
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int nb = (int)(g / 30), q = (int)(g - (size_t)nb * 30);
  const int i = nb / a.B, b = nb - i * a.B;
  if (i >= a.nmax) return;
  const int n = a.n_nodes[b]; if (i >= n) return;
  const double al = (a.done[b] == 1) ? a.alpha[b] : 0.0; const bool tr = a.xt && a.done[b] == 1;       // iLQR: the accepted rollout IS the new trajectory
  a.xs[nb * 30 + q] = tr ? a.xt[nb * 30 + q] : a.x[nb * 30 + q] + al * a.dx[nb * 30 + q];
  // input of the primal solution: own node, or the closest earlier non-event node
  int j = (i == n - 1) ? n - 2 : i;
  if (j < 0) j = 0;                                          // one-node grid (degenerate horizon, K0 status -1): no interval, no input
  while (j > 0 && a.node_ev[j * a.B + b] == QM_EV_PRE) --j;
  const int jb = j * a.B + b; const bool evj = (a.node_ev[jb] == QM_EV_PRE) || n < 2;
  a.us[nb * 30 + q] = evj ? 0.0 : (tr ? a.ut[jb * 30 + q] : a.u[jb * 30 + q] + al * a.du[jb * 30 + q]);
}
// commit the accepted step into the iterate (separate launch: apply reads neighbours' u)
__global__ void qm_ls_commit_kernel(QmLsArgs a) {
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int nb = (int)(g / 30), q = (int)(g - (size_t)nb * 30);
  const int i = nb / a.B, b = nb - i * a.B;
  if (i >= a.nmax) return;
  const int n = a.n_nodes[b]; if (i >= n) return;
  const double al = (a.done[b] == 1) ? a.alpha[b] : 0.0;
  const bool hasu = (i < n - 1) && a.node_ev[nb] != QM_EV_PRE;
  if (a.xt) { if (a.done[b] == 1) { a.x[nb * 30 + q] = a.xt[nb * 30 + q]; if (hasu) a.u[nb * 30 + q] = a.ut[nb * 30 + q]; } return; }
  a.x[nb * 30 + q] += al * a.dx[nb * 30 + q]; if (hasu) a.u[nb * 30 + q] += al * a.du[nb * 30 + q];
}
