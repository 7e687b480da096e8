// k_ipm.h — hard-inequality INTERIOR-POINT variant of the multiple-shooting step (solver slot 3; SURVEY.md §8(f) rank 4: "IPM for hard friction cones", settings block `ipm`,
// qm_controllers/config/task.info:94-125, loaded at qm_interface/src/QMInterface.cpp:72).  The reference registers friction cones and arm joint limits as SOFT costs only
// (QMInterface.cpp:116-131, 177-259) and instantiates no IpmMpc: this solver works on the same OCP with those terms as CONSTRAINTS h(x, u) >= 0 (QM_NH rows per node,
// qmhip_layout.h) and restates the structure of [upstream ocs2_ipm IpmSolver + IpmHelpers, recalled]; the checker is oracle/src/ipm.h, itself pinned by a dense solve of the
// horizon's primal-dual Newton system (tests/test_ipm.py).  What the interior-point method adds to the SQP's kernels:
//   qm_ipm_init_kernel     thread / (node, instance): slack s = (1 + rate) max(h, lower bound), dual lam = (1 + rate) max(mu / s, lower bound) at the initial iterate; mu = initialBarrierParameter
//   K1b, IPM instance      (k_lq.h) condensing: the cost blocks gain Hᵀ diag(lam / s) H and Hᵀ((lam ∘ h − mu) / s − lam) where the soft costs put their barrier derivatives;
//                          node merit −mu Σ ln s, node constraint term + |h − s|²; the projection and K3 run unchanged
//   qm_ipm_dir_kernel      thread / (node, instance): ds = h + Hx dx + Hu du − s, dlam = −(lam ds + (s lam − mu)) / s, the node's fraction-to-the-boundary ratios
//   qm_ipm_alpha_kernel    wave / instance: step limits alphaP, alphaD over the horizon; the line search starts at alphaP (K3 armed it at 1)
//   K4, IPM instance       (k_ls.h) trial merit with the trial slacks s + alpha ds
//   qm_ipm_commit_kernel   thread / (node, instance, row): accepted step: s += alpha ds, lam += alphaDual dlam;  qm_ipm_barrier_kernel: thread / instance: barrier update
// Rows (QM_NH = 28): arm joint k position  2k: z − lo, 2k + 1: hi − z;  arm joint k velocity  12 + 2k, 13 + 2k;  friction cone of contact c: 24 + c (inactive — s = 1, lam = 0,
// no contribution — while the foot swings, at event nodes and behind the last interval).
#pragma once
#include "qm_dev_common.h"

#define IPM_INFO 8      /* per instance: [0] barrier parameter of the current iteration, [1] alphaP, [2] alphaD, [3] dual step taken, [4] barrier parameter after the iteration */
struct QmIpmArgs {
  const double* mb; const double* st;
  int B, nmax;
  const int* n_nodes; const int* node_ev; const int* node_mode; const double* node_dt;
  const double* x; const double* u;          // [nmax][B][30] iterate
  const double* dx; const double* du;        // [nmax][B][30] step (K3)
  double* s; double* lam;                    // [nmax][B][QM_NH]
  double* ds; double* dlam;                  // [nmax][B][QM_NH]
  double* ratio;                             // [nmax][B][2] worst −ds / (margin s), −dlam / (margin lam) of the node
  double* info;                              // [B][IPM_INFO]
  double* alpha; const int* done;            // line search state (k_ls.h)
  const double* out_perf;                    // [B][10] baseline(4) after(4) alpha armijo
};
// value and input / state derivative pattern of row r at (x, u): boxes have one unit entry (sign sg on x[24 + k] resp. u[24 + k]), a cone three entries on the contact's force
__device__ __forceinline__ bool ipm_row_on(int r, int mode) { return r < 24 || mode_flag(mode, r - 24); }
__device__ __forceinline__ double ipm_row_value(const double* mb, const double* st, const double* x, const double* u, int r) {
  if (r < 12) { const int k = r >> 1; return (r & 1) ? mb[MB_QHI + 12 + k] - x[24 + k] : x[24 + k] - mb[MB_QLO + 12 + k]; }
  if (r < 24) { const int k = (r - 12) >> 1; return (r & 1) ? st[ST_JVEL_HI + k] - u[24 + k] : u[24 + k] - st[ST_JVEL_LO + k]; }
  const int c = r - 24; const double Fx = u[3 * c], Fy = u[3 * c + 1], Fz = u[3 * c + 2];
  return st[ST_FRIC_COEF] * Fz - sqrt(Fx * Fx + Fy * Fy + st[ST_FRIC_REG]);
}
// directional derivative of row r along (dx, du)
__device__ __forceinline__ double ipm_row_dir(const double* st, const double* u, const double* dx, const double* du, int r) {
  if (r < 12) { const int k = r >> 1; return (r & 1) ? -dx[24 + k] : dx[24 + k]; }
  if (r < 24) { const int k = (r - 12) >> 1; return (r & 1) ? -du[24 + k] : du[24 + k]; }
  const int c = r - 24; const double Fx = u[3 * c], Fy = u[3 * c + 1]; const double iT = 1.0 / sqrt(Fx * Fx + Fy * Fy + st[ST_FRIC_REG]);
  return -Fx * iT * du[3 * c] - Fy * iT * du[3 * c + 1] + st[ST_FRIC_COEF] * du[3 * c + 2];
}
__device__ __forceinline__ bool ipm_node_regular(const QmIpmArgs& a, int i, int b) { const int n = a.n_nodes[b]; return i < n - 1 && a.node_ev[i * a.B + b] != QM_EV_PRE; }

__global__ void qm_ipm_init_kernel(QmIpmArgs a) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x; const int i = g / a.B, b = g - i * a.B;
  if (i >= a.nmax) return;
  const int nb = i * a.B + b; const double mu = a.st[ST_IPM_MU];
  if (i == 0) { a.info[b * IPM_INFO] = mu; a.info[b * IPM_INFO + 1] = 1.0; a.info[b * IPM_INFO + 2] = 1.0; a.info[b * IPM_INFO + 3] = 0.0; a.info[b * IPM_INFO + 4] = mu; }
  const bool reg = ipm_node_regular(a, i, b); const int mode = a.node_mode[nb];
  for (int r = 0; r < QM_NH; ++r) {
    double s = 1.0, lam = 0.0;
    if (reg && ipm_row_on(r, mode)) {
      const double h = ipm_row_value(a.mb, a.st, a.x + (size_t)nb * 30, a.u + (size_t)nb * 30, r);
      s = (1.0 + a.st[ST_IPM_SLACK_MARGIN]) * fmax(h, a.st[ST_IPM_SLACK_LB]);
      lam = (1.0 + a.st[ST_IPM_DUAL_MARGIN]) * fmax(mu / s, a.st[ST_IPM_DUAL_LB]);
    }
    a.s[(size_t)nb * QM_NH + r] = s; a.lam[(size_t)nb * QM_NH + r] = lam; a.ds[(size_t)nb * QM_NH + r] = 0.0; a.dlam[(size_t)nb * QM_NH + r] = 0.0;
  }
}
__global__ void qm_ipm_dir_kernel(QmIpmArgs a) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x; const int i = g / a.B, b = g - i * a.B;
  if (i >= a.nmax) return;
  const int nb = i * a.B + b; const double mu = a.info[b * IPM_INFO], margin = a.st[ST_IPM_FTB_MARGIN];
  const bool reg = ipm_node_regular(a, i, b); const int mode = a.node_mode[nb];
  double wp = 0.0, wd = 0.0;
  for (int r = 0; r < QM_NH; ++r) {
    double ds = 0.0, dl = 0.0;
    if (reg && ipm_row_on(r, mode)) {
      const double* x = a.x + (size_t)nb * 30; const double* u = a.u + (size_t)nb * 30;
      const double s = a.s[(size_t)nb * QM_NH + r], lam = a.lam[(size_t)nb * QM_NH + r];
      ds = ipm_row_value(a.mb, a.st, x, u, r) - s + ipm_row_dir(a.st, u, a.dx + (size_t)nb * 30, a.du + (size_t)nb * 30, r);
      dl = -(lam * ds + (s * lam - mu)) / s;
      wp = fmax(wp, -ds / (margin * s)); wd = fmax(wd, -dl / (margin * lam));
    }
    a.ds[(size_t)nb * QM_NH + r] = ds; a.dlam[(size_t)nb * QM_NH + r] = dl;
  }
  a.ratio[(size_t)nb * 2] = wp; a.ratio[(size_t)nb * 2 + 1] = wd;
}
// one wavefront per instance: fraction to the boundary over the whole horizon ([upstream ipm::fractionToBoundaryStepSize]); the line search starts at the primal limit
__global__ void __launch_bounds__(64) qm_ipm_alpha_kernel(QmIpmArgs a) {
  const int b = blockIdx.x, l = threadIdx.x & 63; if (b >= a.B) return;
  const int n = a.n_nodes[b]; double wp = 0.0, wd = 0.0;
  for (int i = l; i < n; i += 64) { wp = fmax(wp, a.ratio[(size_t)(i * a.B + b) * 2]); wd = fmax(wd, a.ratio[(size_t)(i * a.B + b) * 2 + 1]); }
  for (int o = 32; o > 0; o >>= 1) { wp = fmax(wp, __shfl_xor(wp, o)); wd = fmax(wd, __shfl_xor(wd, o)); }
  if (l == 0) {
    const double ap = (wp > 0.0) ? fmin(1.0, 1.0 / wp) : 1.0, ad = (wd > 0.0) ? fmin(1.0, 1.0 / wd) : 1.0;
    a.info[b * IPM_INFO + 1] = ap; a.info[b * IPM_INFO + 2] = ad; a.alpha[b] = ap;
  }
}
// accepted step: s += alpha ds, lam += alphaDual dlam (alphaDual = usePrimalStepSizeForDual ? min(alpha, alphaD) : alphaD); one thread per (node, instance, row)
__global__ void qm_ipm_commit_kernel(QmIpmArgs a) {
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; const size_t nb = g / QM_NH; const int i = (int)(nb / a.B), b = (int)(nb - (size_t)i * a.B);
  if (i >= a.nmax) return;
  if (a.done[b] != 1) return;
  const double al = a.alpha[b], ad = a.info[b * IPM_INFO + 2]; const double adual = (a.st[ST_IPM_PRIMAL_FOR_DUAL] != 0.0) ? fmin(al, ad) : ad;
  a.s[g] += al * a.ds[g]; a.lam[g] += adual * a.dlam[g];
}
// barrier update of the instance ([upstream IpmSolver::updateBarrierParameter]) + bookkeeping of the dual step; after qm_ipm_commit_kernel
__global__ void qm_ipm_barrier_kernel(QmIpmArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x; if (b >= a.B) return;
  const double mu = a.info[b * IPM_INFO]; const bool acc = a.done[b] == 1;
  const double al = acc ? a.alpha[b] : 0.0, ad = a.info[b * IPM_INFO + 2];
  a.info[b * IPM_INFO + 3] = acc ? ((a.st[ST_IPM_PRIMAL_FOR_DUAL] != 0.0) ? fmin(al, ad) : ad) : 0.0;
  const double* p = a.out_perf + (size_t)b * 10; const double thetaAfter = sqrt(p[6] + p[7]);
  double next = mu;
  if (fabs(p[0] - p[4]) < a.st[ST_IPM_RED_COST_TOL] && thetaAfter < a.st[ST_IPM_RED_CON_TOL]) next = fmax(a.st[ST_IPM_MU_TARGET], fmin(a.st[ST_IPM_MU_LINEAR] * mu, pow(mu, a.st[ST_IPM_MU_POWER])));
  a.info[b * IPM_INFO + 4] = next; a.info[b * IPM_INFO] = next;
}
