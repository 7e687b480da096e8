// k_riccati.h — K3: discrete-time Riccati backward sweep + forward rollout of the projected QP.
//
// One 256-thread workgroup per MPC instance; stages are sequential, the dense 30x30 / 30xm products of each
// stage run on the f64 matrix cores.  With every equality constraint projected out and no inequality rows
// the QP sub-problem the reference hands to HPIPM is solved exactly by one Riccati factorise+solve
// ([upstream ocs2_sqp SqpSolver::getOCPSolution -> hpipm]):
//   Hux = P + Bᵀ S A, Huu = R + Bᵀ S B, hu = r + Bᵀ(s + S b);  L Lᵀ = Huu;  W = L⁻¹ Hux, y = L⁻¹ hu
//   S' = Q + Aᵀ S A − Wᵀ W (symmetrised),  s' = q + Aᵀ(s + S b) − Wᵀ y,  K = −L⁻ᵀ W, k = −L⁻ᵀ y
// Event nodes (PreEvent -> PostEvent, identity jump, nu = 0): S' = S, s' = s + S b, b = x_i − x_{i+1}.
// Forward: dx_0 = x0 − x_0;  ut = K dx + k;  dx+ = Ap dx + Bp ut + bp;  du = Pe + Px dx + Pu ut
// and the Armijo descent metric  sum q·dx + r·ut  (projected gradients).
#pragma once
#include "qm_dev_common.h"

struct QmRiccatiArgs {
  int B, nmax;
  const int* n_nodes; const int* node_ev;      // [B], [nmax][B]
  const double* x0;                            // [B][30]
  const double* x;                             // [nmax][B][30] (current iterate; event defects, dx0)
  double* stage;                               // [B][nmax][SR_SIZE]  (K, kff written here)
  double* dx; double* du;                      // [nmax][B][30]
  double* step_info;                           // [B][4]: armijo, |dx|², |du|², chol status
};

#define RC_T(n) ((n) * QM_TILE)
#define RC_VEC (7 * QM_TILE)
#define RC_V_S    (RC_VEC + 0)     /* s */
#define RC_V_B    (RC_VEC + 32)    /* bp */
#define RC_V_SPSB (RC_VEC + 64)    /* s + S b */
#define RC_V_HU   (RC_VEC + 96)    /* hu -> y */
#define RC_V_Q    (RC_VEC + 128)
#define RC_V_DX   (RC_VEC + 160)
#define RC_V_UT   (RC_VEC + 192)
#define RC_V_DXN  (RC_VEC + 224)
#define RC_V_RED  (RC_VEC + 256)
#define RC_LDS_DOUBLES (RC_VEC + 272)
#define RC_LDS_BYTES (RC_LDS_DOUBLES * 8)

__device__ __forceinline__ double rc_block_sum(double v, double* red) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  double s = 0.0; for (int i = 0; i < nw; ++i) s += red[i];
  return s;
}

__global__ void __launch_bounds__(QM_BLOCK) qm_riccati_kernel(QmRiccatiArgs a) {
  extern __shared__ double qm_smem[];
  double* S = qm_smem;
  const int tid = threadIdx.x, b = blockIdx.x;
  const int n = a.n_nodes[b];
  double* St = S + RC_T(0); double* At = S + RC_T(1); double* Bt = S + RC_T(2); double* SA = S + RC_T(3); double* SB = S + RC_T(4); double* W = S + RC_T(5); double* H = S + RC_T(6);
  double* sv = S + RC_V_S; double* bv = S + RC_V_B; double* spsb = S + RC_V_SPSB; double* hu = S + RC_V_HU; double* qv = S + RC_V_Q;
  tile_zero(S, RC_LDS_DOUBLES);
  __syncthreads();
  int chol_fail = 0;
  // terminal value function
  {
    const double* rec = a.stage + ((size_t)b * a.nmax + (n - 1)) * SR_SIZE;
    tile_load(St, rec + SR_QP, 30, 30, 30);
    if (tid < 30) sv[tid] = rec[SR_QPV + tid];
  }
  __syncthreads();
  for (int k = n - 2; k >= 0; --k) {
    double* rec = a.stage + ((size_t)b * a.nmax + k) * SR_SIZE;
    if (a.node_ev[k * a.B + b] == QM_EV_PRE) {
      if (tid < 30) bv[tid] = a.x[(k * a.B + b) * 30 + tid] - a.x[((k + 1) * a.B + b) * 30 + tid];
      __syncthreads();
      double add = 0.0; if (tid < 30) add = tile_row_dot(St, tid, bv, 30);
      __syncthreads();
      if (tid < 30) sv[tid] += add;
      __syncthreads();
      continue;
    }
    const int m = (int)rec[SR_SCAL]; const int mtm = (m + 15) / 16; const int ksm = (m + 3) / 4;
    // stage data -> LDS
    tile_zero(Bt); tile_zero(W); tile_zero(H);
    __syncthreads();
    tile_load(At, rec + SR_AP, 30, 30, 30);
    tile_load(Bt, rec + SR_BP, 30, m, QM_MMAX);
    if (tid < 30) { bv[tid] = rec[SR_BPV + tid]; qv[tid] = rec[SR_QPV + tid]; }
    __syncthreads();
    if (tid < 30) spsb[tid] = sv[tid] + tile_row_dot(St, tid, bv, 30);
    wg_gemm<false, false>(St, At, 2, 2, 0, 8, [&](int r, int c, double v) { SA[r * QM_LD + c] = v; });
    wg_gemm<false, false>(St, Bt, 2, mtm, 0, 8, [&](int r, int c, double v) { SB[r * QM_LD + c] = v; });
    __syncthreads();
    wg_gemm<true, false>(Bt, SA, mtm, 2, 0, 8, [&](int r, int c, double v) { if (r < m && c < 30) W[r * QM_LD + c] = rec[SR_PP + r * 30 + c] + v; });
    wg_gemm<true, false>(Bt, SB, mtm, mtm, 0, 8, [&](int r, int c, double v) { if (r < m && c < m) H[r * QM_LD + c] = rec[SR_RP + r * QM_MMAX + c] + v; });
    if (tid < m) hu[tid] = rec[SR_RPV + tid] + tile_col_dot(Bt, tid, spsb, 30);
    __syncthreads();
    // wave 0: Cholesky of Huu (symmetrised), W <- L⁻¹ Hux, y <- L⁻¹ hu ; other waves start Aᵀ S A
    if (tid < 64) {
      const int l = tid;
      for (int idx = l; idx < m * m; idx += 64) { const int r = idx / m, c = idx - r * m; if (r > c) { const double v = 0.5 * (H[r * QM_LD + c] + H[c * QM_LD + r]); H[r * QM_LD + c] = v; } }
      qm_wave_sync();
      for (int j = 0; j < m; ++j) {
        const double djj = H[j * QM_LD + j];
        if (!(djj > 0.0)) chol_fail = 1;
        const double d = sqrt(djj);
        qm_wave_sync();
        if (l > j && l < m) H[l * QM_LD + j] /= d;
        if (l == j) H[j * QM_LD + j] = d;
        qm_wave_sync();
        const int rem = m - 1 - j;                       // trailing (i,c), j < c <= i < m
        for (int idx = l; idx < rem * (rem + 1) / 2; idx += 64) {
          int i = 0, acc = 0; while (acc + i + 1 <= idx) { acc += i + 1; ++i; }
          const int c = idx - acc; const int ii = j + 1 + i, cc = j + 1 + c;
          H[ii * QM_LD + cc] -= H[ii * QM_LD + j] * H[cc * QM_LD + j];
        }
        qm_wave_sync();
      }
      if (l <= 30) {                                       // forward substitution, lane per column (col 30 = hu)
        for (int r = 0; r < m; ++r) {
          double v = (l < 30) ? W[r * QM_LD + l] : hu[r];
          for (int q = 0; q < r; ++q) v -= H[r * QM_LD + q] * ((l < 30) ? W[q * QM_LD + l] : hu[q]);
          v /= H[r * QM_LD + r];
          if (l < 30) W[r * QM_LD + l] = v; else hu[r] = v;
        }
      }
    }
    __syncthreads();
    // S' = Q + Aᵀ SA − Wᵀ W ; s' = q + Aᵀ spsb − Wᵀ y
    wg_gemm<true, false>(At, SA, 2, 2, 0, 8, [&](int r, int c, double v) { if (r < 30 && c < 30) St[r * QM_LD + c] = rec[SR_QP + r * 30 + c] + v; });
    double snew = 0.0; if (tid >= 64 && tid < 94) { const int r = tid - 64; snew = qv[r] + tile_col_dot(At, r, spsb, 30) - tile_col_dot(W, r, hu, m); }
    __syncthreads();
    wg_gemm<true, false>(W, W, 2, 2, 0, ksm, [&](int r, int c, double v) { if (r < 30 && c < 30) St[r * QM_LD + c] -= v; });
    if (tid >= 64 && tid < 94) sv[tid - 64] = snew;
    __syncthreads();
    for (int idx = tid; idx < 900; idx += blockDim.x) { const int r = idx / 30, c = idx - r * 30; if (r > c) { const double v = 0.5 * (St[r * QM_LD + c] + St[c * QM_LD + r]); St[r * QM_LD + c] = v; SA[r * QM_LD + c] = v; } }
    __syncthreads();
    for (int idx = tid; idx < 900; idx += blockDim.x) { const int r = idx / 30, c = idx - r * 30; if (r < c) St[r * QM_LD + c] = SA[c * QM_LD + r]; }
    // K = −L⁻ᵀ W, kff = −L⁻ᵀ y (back substitution, lane per column) -> HBM
    if (tid <= 30) {
      const int l = tid;
      for (int r = m - 1; r >= 0; --r) {
        double v = (l < 30) ? W[r * QM_LD + l] : hu[r];
        for (int q = r + 1; q < m; ++q) v -= H[q * QM_LD + r] * ((l < 30) ? W[q * QM_LD + l] : hu[q]);
        v /= H[r * QM_LD + r];
        if (l < 30) W[r * QM_LD + l] = v; else hu[r] = v;
      }
      for (int r = 0; r < m; ++r) { if (l < 30) rec[SR_K + r * 30 + l] = -W[r * QM_LD + l]; else rec[SR_KFF + r] = -hu[r]; }
    }
    __syncthreads();
  }
  // ---- forward rollout ----
  double* dxv = S + RC_V_DX; double* ut = S + RC_V_UT; double* dxn = S + RC_V_DXN;
  double* Kt = SA; double* Pxt = SB; double* Put = W;     // reuse tiles
  if (tid < 30) dxv[tid] = a.x0[(size_t)b * 30 + tid] - a.x[(0 * a.B + b) * 30 + tid];
  __syncthreads();
  double armijo = 0.0, dx2 = 0.0, du2 = 0.0;
  for (int k = 0; k < n - 1; ++k) {
    double* rec = a.stage + ((size_t)b * a.nmax + k) * SR_SIZE;
    const int nb = k * a.B + b;
    if (tid < 30) { a.dx[nb * 30 + tid] = dxv[tid]; dx2 += dxv[tid] * dxv[tid]; }
    if (a.node_ev[nb] == QM_EV_PRE) {
      if (tid < 30) { a.du[nb * 30 + tid] = 0.0; dxv[tid] += a.x[nb * 30 + tid] - a.x[((k + 1) * a.B + b) * 30 + tid]; }
      __syncthreads();
      continue;
    }
    const int m = (int)rec[SR_SCAL];
    tile_zero(Bt); tile_zero(Kt); tile_zero(Put);
    __syncthreads();
    tile_load(At, rec + SR_AP, 30, 30, 30); tile_load(Bt, rec + SR_BP, 30, m, QM_MMAX); tile_load(Kt, rec + SR_K, m, 30, 30);
    tile_load(Pxt, rec + SR_PX, 30, 30, 30); tile_load(Put, rec + SR_PU, 30, m, QM_MMAX);
    __syncthreads();
    if (tid < m) ut[tid] = rec[SR_KFF + tid] + tile_row_dot(Kt, tid, dxv, 30);
    if (tid >= 64 && tid < 94) armijo += rec[SR_QPV + tid - 64] * dxv[tid - 64];
    __syncthreads();
    if (tid < 30) dxn[tid] = rec[SR_BPV + tid] + tile_row_dot(At, tid, dxv, 30) + tile_row_dot(Bt, tid, ut, m);
    if (tid >= 64 && tid < 94) { const int r = tid - 64; const double v = rec[SR_PE + r] + tile_row_dot(Pxt, r, dxv, 30) + tile_row_dot(Put, r, ut, m); a.du[nb * 30 + r] = v; du2 += v * v; }
    if (tid >= 128 && tid < 128 + m) armijo += rec[SR_RPV + tid - 128] * ut[tid - 128];
    __syncthreads();
    if (tid < 30) dxv[tid] = dxn[tid];
    __syncthreads();
  }
  {
    const int nb = (n - 1) * a.B + b; const double* rec = a.stage + ((size_t)b * a.nmax + (n - 1)) * SR_SIZE;
    if (tid < 30) { a.dx[nb * 30 + tid] = dxv[tid]; a.du[nb * 30 + tid] = 0.0; dx2 += dxv[tid] * dxv[tid]; armijo += rec[SR_QPV + tid] * dxv[tid]; }
  }
  const double arm = rc_block_sum(armijo, S + RC_V_RED);
  const double sx = rc_block_sum(dx2, S + RC_V_RED);
  const double su = rc_block_sum(du2, S + RC_V_RED);
  if (tid == 0) { a.step_info[b * 4] = arm; a.step_info[b * 4 + 1] = sx; a.step_info[b * 4 + 2] = su; }
  if (tid == 0) a.step_info[b * 4 + 3] = (double)chol_fail;
}
