// k_policy.h — K5: policy evaluation (MPC_MRT_Interface::evaluatePolicy [upstream], call site
// qm_controllers/src/QMController.cpp:139-142) and the synthetic measured state of the benchmark step.
// One thread per instance.
#pragma once
#include "k_grid.h"
#include "qm_dev_kin.h"

struct QmPolicyArgs {
  const double* mb;
  int B, nmax, nev;
  const int* n_nodes; const double* node_t; const int* node_ev;   // grid
  const double* xs; const double* us;                              // primal solution [nmax][B][30]
  const double* ev; const int* modes;                              // schedule [B][nev], [B][nev+1]
  const double* t;                                                 // [B] evaluation time (t0 for the benchmark step)
  double* x_des; double* u_des; int* mode;                         // [B][30], [B][30], [B]
};

__device__ __forceinline__ void qm_policy_body(const QmPolicyArgs& a, const int b) {
  const int n = a.n_nodes[b]; const double t = a.t[b];
  int idx; double al; grid_policy_segment(a.node_t, a.node_ev, n, a.B, b, t, &idx, &al);
  const int i0 = idx * a.B + b, i1 = ((n > 1 ? idx + 1 : idx)) * a.B + b;
  for (int q = 0; q < 30; ++q) { a.x_des[(size_t)b * 30 + q] = al * a.xs[i0 * 30 + q] + (1.0 - al) * a.xs[i1 * 30 + q]; a.u_des[(size_t)b * 30 + q] = al * a.us[i0 * 30 + q] + (1.0 - al) * a.us[i1 * 30 + q]; }
  a.mode[b] = a.modes[(size_t)b * (a.nev + 1) + grid_find_index(a.ev + (size_t)b * a.nev, a.nev, t)];
}
__global__ void qm_policy_kernel(QmPolicyArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.B) return;
  qm_policy_body(a, b);
}

// measured rbd state (55) built from x0: zero velocities, EE pose by FK (layout of
// qm_estimation/src/StateEstimateBase.cpp:41-103)
struct QmMeasArgs { const double* mb; int B; const double* x0; double time; double* rbd; double* time_out; };
__device__ __forceinline__ void qm_measured_body(const QmMeasArgs& a, const int b) {
  const double* x = a.x0 + (size_t)b * 30; double* r = a.rbd + (size_t)b * QM_NRBD;
  for (int q = 0; q < QM_NRBD; ++q) r[q] = 0.0;
  for (int q = 0; q < 3; ++q) { r[q] = x[9 + q]; r[3 + q] = x[6 + q]; }
  for (int j = 0; j < QM_NJ; ++j) r[6 + j] = x[12 + j];
  double K[KW_SIZE]; kin_base(a.mb, x, K); kin_arm(a.mb, x, K);
  double qq[4]; mat_to_quat(K + KW_ARM + 39, qq);
  for (int q = 0; q < 3; ++q) r[48 + q] = K[KW_ARM + 36 + q];
  for (int q = 0; q < 4; ++q) r[51 + q] = qq[q];
  a.time_out[b] = a.time;
}
__global__ void qm_measured_kernel(QmMeasArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.B) return;
  qm_measured_body(a, b);
}
// the benchmark / closed-loop step evaluates the policy at t0 and builds the measured state in one launch (one thread per instance each:
// the first B threads take the policy, the next B the measured state)
struct QmPolicyMeasArgs { QmPolicyArgs p; QmMeasArgs m; };
__global__ void qm_policy_measured_kernel(QmPolicyMeasArgs a) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g < a.p.B) qm_policy_body(a.p, g); else if (g - a.p.B < a.m.B) qm_measured_body(a.m, g - a.p.B);
}

// hand-over of the solver's node-major arrays [nmax][B][k] in the C ABI's instance-major layout [B][nmax][k] (qmhip_mpc_download): transposed on the DEVICE into one
// staging buffer per call, so that the host sees ONE contiguous copy per array instead of B x nmax strided pieces.  Thread per 8-byte word (k doubles, or one int widened)
struct QmGatherArgs { int B, nmax, k; const double* src_d; const int* src_i; double* dst_d; int* dst_i; };
__global__ void qm_gather_kernel(QmGatherArgs a) {
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x, total = (size_t)a.B * a.nmax * a.k;
  if (g >= total) return;
  const int q = (int)(g % a.k); const size_t bi = g / a.k; const int i = (int)(bi % a.nmax), b = (int)(bi / a.nmax);      // destination index (b, i, q): coalesced writes
  const size_t s = ((size_t)i * a.B + b) * a.k + q;
  if (a.src_d) a.dst_d[g] = a.src_d[s]; else a.dst_i[g] = a.src_i[s];
}
