// k_riccati.h — K3: discrete-time Riccati backward sweep + forward rollout of the projected QP.
//
// One 256-thread workgroup per MPC instance (50 KB LDS -> 3 workgroups / CU); stages are sequential, the dense
// 30x30 / 30xm products of each stage run on the f64 matrix cores.  With every equality constraint projected out and no
// inequality rows the QP sub-problem the reference hands to HPIPM is solved exactly by one Riccati factorise+solve
// ([upstream ocs2_sqp SqpSolver::getOCPSolution -> hpipm]):
//   Hux = P + Bᵀ S A, Huu = R + Bᵀ S B, hu = r + Bᵀ(s + S b);  L Lᵀ = Huu;  W = L⁻¹ Hux, y = L⁻¹ hu
//   S' = Q + Aᵀ S A − Wᵀ W (symmetrised),  s' = q + Aᵀ(s + S b) − Wᵀ y,  K = −L⁻ᵀ W, k = −L⁻ᵀ y
// Event nodes (PreEvent -> PostEvent, identity jump, nu = 0): S' = S, s' = s + S b, b = x_i − x_{i+1}.
// The backward sweep also leaves the CLOSED-LOOP stage maps in the record, so the forward rollout is two mat-vecs:
//   Ahat = Ap + Bp K, bhat = bp + Bp k :  dx+ = Ahat dx + bhat          Khat = Px + Pu K, khat = Pe + Pu k :  du = Khat dx + khat
//   ghat = qp + Kᵀ rp, c0 = rp·k       :  Armijo metric  sum (qp·dx + rp·ut) = sum (ghat·dx + c0)
// Layout notes: the m-wide operands (Bp, S Bp, W, Huu, Pu) are kept TRANSPOSED as 20-row tiles; the Cholesky and the
// triangular solves run on wave 0 with each lane's column in registers while waves 1-3 already form Q + Aᵀ S A.
#pragma once
#include "qm_dev_common.h"

struct QmRiccatiArgs {
  int B, nmax;
  const int* n_nodes; const int* node_ev;      // [B], [nmax][B]
  const double* x0;                            // [B][30]
  const double* x;                             // [nmax][B][30] (current iterate; event defects, dx0)
  double* stage;                               // [B][nmax][SR_SIZE]  (K, kff and the closed-loop maps are written here)
  double* dx; double* du;                      // [nmax][B][30]
  double* step_info;                           // [B][4]: armijo, |dx|², |du|², chol status
  int skip;                                    // profiling only (bit mask of phases to skip; results are then meaningless)
};

// closed-loop maps overwrite record fields that are dead after the backward visit of the stage
#define SR_AHAT SR_AP
#define SR_KHAT SR_PX
#define SR_BHAT SR_BPV
#define SR_KHATV SR_PE
#define SR_GHAT SR_QPV
#define SR_C0 (SR_SCAL + 1)

#define RC_R20 (20 * QM_LD)
#define RC_S   0
#define RC_A   QM_TILE
#define RC_SA  (2 * QM_TILE)
#define RC_BT  (3 * QM_TILE)                 /* Bpᵀ  [m][30]  (later Puᵀ) */
#define RC_SBT (RC_BT + RC_R20)              /* (S Bp)ᵀ [m][30] */
#define RC_W   (RC_SBT + RC_R20)             /* Hux -> W -> X = L⁻ᵀ W  [m][30] */
#define RC_H   (RC_W + RC_R20)               /* Huu -> L  [m][m] */
#define RC_VEC (RC_H + RC_R20)
#define RC_V_S    (RC_VEC + 0)     /* s */
#define RC_V_B    (RC_VEC + 32)    /* bp */
#define RC_V_SPSB (RC_VEC + 64)    /* s + S b */
#define RC_V_HU   (RC_VEC + 96)    /* hu -> y -> L⁻ᵀ y */
#define RC_V_Q    (RC_VEC + 128)
#define RC_V_RP   (RC_VEC + 160)   /* rp */
#define RC_V_DX   (RC_VEC + 192)
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
// coalesced load of a rows x 30 global matrix into a tile: thread (r = t>>3, g = t&7) moves 4 consecutive columns
__device__ __forceinline__ void rc_load30(double* T, const double* src, int rows) {
  const int r = threadIdx.x >> 3, c0 = (threadIdx.x & 7) * 4;
  if (r < rows) for (int c = c0; c < c0 + 4 && c < 30; ++c) T[r * QM_LD + c] = src[r * 30 + c];
}
// transposed load: src is rows x m with leading dim QM_MMAX -> T[c][r]
__device__ __forceinline__ void rc_load_T(double* T, const double* src, int rows, int m) {
  for (int idx = threadIdx.x; idx < rows * m; idx += blockDim.x) { const int r = idx / m, c = idx - r * m; T[c * QM_LD + r] = src[r * QM_MMAX + c]; }
}

__global__ void __launch_bounds__(QM_BLOCK) qm_riccati_kernel(QmRiccatiArgs a) {
  extern __shared__ double qm_smem[];
  double* S = qm_smem;
  const int tid = threadIdx.x, b = blockIdx.x, wave = tid >> 6;
  const int n = a.n_nodes[b];
  double* St = S + RC_S; double* At = S + RC_A; double* SA = S + RC_SA; double* Bt = S + RC_BT; double* SBt = S + RC_SBT; double* W = S + RC_W; double* H = S + RC_H;
  double* sv = S + RC_V_S; double* bv = S + RC_V_B; double* spsb = S + RC_V_SPSB; double* hu = S + RC_V_HU; double* qv = S + RC_V_Q; double* rpv = S + RC_V_RP;
  tile_zero(S, RC_LDS_DOUBLES);
  __syncthreads();
  int chol_fail = 0;
  {   // terminal value function
    const double* rec = a.stage + ((size_t)b * a.nmax + (n - 1)) * SR_SIZE;
    rc_load30(St, rec + SR_QP, 30);
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
    const int m = (int)rec[SR_SCAL]; const int mtm = (m + 15) / 16;
    // ---- stage data -> LDS (the m-wide tiles are cleared: m may change between stages) ----
    if (a.skip & 32) continue;
    for (int idx = tid; idx < 4 * RC_R20; idx += blockDim.x) Bt[idx] = 0.0;
    __syncthreads();
    rc_load30(At, rec + SR_AP, 30);
    rc_load_T(Bt, rec + SR_BP, 30, m);
    rc_load30(W, rec + SR_PP, m);                        // Hux starts as Pp
    for (int idx = tid; idx < m * m; idx += blockDim.x) { const int r = idx / m, c = idx - r * m; H[r * QM_LD + c] = rec[SR_RP + r * QM_MMAX + c]; }
    if (tid < 30) { bv[tid] = rec[SR_BPV + tid]; qv[tid] = rec[SR_QPV + tid]; }
    if (tid >= 32 && tid < 32 + m) rpv[tid - 32] = rec[SR_RPV + tid - 32];
    __syncthreads();
    if (tid < 30) spsb[tid] = sv[tid] + tile_row_dot(St, tid, bv, 30);
    if (!(a.skip & 8)) wg_gemm<false, false>(St, At, 2, 2, 0, 8, [&](int r, int c, double v) { SA[r * QM_LD + c] = v; });
    if (!(a.skip & 8)) wg_gemm<false, false>(Bt, St, mtm, 2, 0, 8, [&](int r, int c, double v) { if (r < m) SBt[r * QM_LD + c] = v; });      // (S B)ᵀ = Bᵀ S
    __syncthreads();
    if (!(a.skip & 16)) wg_gemm<false, false>(Bt, SA, mtm, 2, 0, 8, [&](int r, int c, double v) { if (r < m && c < 30) W[r * QM_LD + c] += v; });   // Hux = Pp + Bᵀ S A
    if (!(a.skip & 16)) wg_gemm<false, true>(Bt, SBt, mtm, mtm, 0, 8, [&](int r, int c, double v) { if (r < m && c < m) H[r * QM_LD + c] += v; });  // Huu = Rp + Bᵀ S B
    if (tid < m) hu[tid] = rpv[tid] + tile_row_dot(Bt, tid, spsb, 30);
    __syncthreads();
    if (a.skip & 1) { } else if (wave == 0) {
      // ---- wave 0: Cholesky of Huu (symmetrised) in LDS, then W <- L⁻¹ Hux, y <- L⁻¹ hu with the column in registers ----
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
      if (l <= 30) {                                       // lane = column of Hux (lane 30: hu); L entries are wave-uniform LDS broadcasts
        double w[QM_MMAX];
#pragma unroll
        for (int r = 0; r < QM_MMAX; ++r) w[r] = (r < m) ? ((l < 30) ? W[r * QM_LD + l] : hu[r]) : 0.0;
#pragma unroll
        for (int r = 0; r < QM_MMAX; ++r) if (r < m) {
          double v = w[r];
#pragma unroll
          for (int q = 0; q < r; ++q) v -= H[r * QM_LD + q] * w[q];
          w[r] = v / H[r * QM_LD + r];
        }
#pragma unroll
        for (int r = 0; r < QM_MMAX; ++r) if (r < m) { if (l < 30) W[r * QM_LD + l] = w[r]; else hu[r] = w[r]; }
      }
    } else {
      // ---- waves 1-3 meanwhile: S' <- Q + Aᵀ (S A)   (S itself is dead once S A and (S B)ᵀ exist) ----
      const int l = tid & 63, li = l & 15, lk = l >> 4;
      for (int t = wave - 1; t < 4; t += 3) {
        const int I = t >> 1, J = t & 1; qm_d4 acc = {0.0, 0.0, 0.0, 0.0};
        double q4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int row = 16 * I + lk + 4 * r, col = 16 * J + li; q4[r] = (row < 30 && col < 30) ? rec[SR_QP + row * 30 + col] : 0.0; }
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) { const double av = At[(4 * kk + lk) * QM_LD + 16 * I + li], bvv = SA[(4 * kk + lk) * QM_LD + 16 * J + li]; acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bvv, acc, 0, 0, 0); }
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int row = 16 * I + lk + 4 * r, col = 16 * J + li; if (row < 30 && col < 30) St[row * QM_LD + col] = q4[r] + acc[r]; }
      }
    }
    __syncthreads();
    // S' -= Wᵀ W ; s' = q + Aᵀ spsb − Wᵀ y
    double snew = 0.0; if (tid >= 64 && tid < 94) { const int r = tid - 64; snew = qv[r] + tile_col_dot(At, r, spsb, 30) - tile_col_dot(W, r, hu, m); }
    wg_gemm<true, false>(W, W, 2, 2, 0, 5, [&](int r, int c, double v) { if (r < 30 && c < 30) St[r * QM_LD + c] -= v; });
    __syncthreads();
    if (tid >= 64 && tid < 94) sv[tid - 64] = snew;
    // symmetrise S' (lower triangle authoritative) — SA is free to serve as scratch
    for (int idx = tid; idx < 900; idx += blockDim.x) { const int r = idx / 30, c = idx - r * 30; if (r > c) { const double v = 0.5 * (St[r * QM_LD + c] + St[c * QM_LD + r]); SA[r * QM_LD + c] = v; } }
    // X = L⁻ᵀ W, y <- L⁻ᵀ y (back substitution, column in registers) ; K = −X, k = −y
    if (wave == 0 && tid <= 30) {
      const int l = tid; double w[QM_MMAX];
#pragma unroll
      for (int r = 0; r < QM_MMAX; ++r) w[r] = (r < m) ? ((l < 30) ? W[r * QM_LD + l] : hu[r]) : 0.0;
#pragma unroll
      for (int r = QM_MMAX - 1; r >= 0; --r) if (r < m) {
        double v = w[r];
#pragma unroll
        for (int q = r + 1; q < QM_MMAX; ++q) if (q < m) v -= H[q * QM_LD + r] * w[q];
        w[r] = v / H[r * QM_LD + r];
      }
#pragma unroll
      for (int r = 0; r < QM_MMAX; ++r) if (r < m) { if (l < 30) { W[r * QM_LD + l] = w[r]; rec[SR_K + r * 30 + l] = -w[r]; } else { hu[r] = w[r]; rec[SR_KFF + r] = -w[r]; } }
    }
    __syncthreads();
    for (int idx = tid; idx < 900; idx += blockDim.x) { const int r = idx / 30, c = idx - r * 30; if (r > c) { const double v = SA[r * QM_LD + c]; St[r * QM_LD + c] = v; St[c * QM_LD + r] = v; } }
    // ---- closed-loop maps of this stage (K = −X in W, k = −y in hu) ----
    if (a.skip & 2) { __syncthreads(); continue; }
    // Ahat = Ap − Bp X : op(A)[i][k] = Bp[i][k] = Bt[k][i] (TA), op(B)[k][j] = X[k][j]
    wg_gemm<true, false>(Bt, W, 2, 2, 0, 5, [&](int r, int c, double v) { if (r < 30 && c < 30) rec[SR_AHAT + r * 30 + c] = At[r * QM_LD + c] - v; });
    if (tid < 30) rec[SR_BHAT + tid] = bv[tid] - tile_col_dot(Bt, tid, hu, m);
    if (tid >= 64 && tid < 94) { const int c = tid - 64; rec[SR_GHAT + c] = qv[c] - tile_col_dot(W, c, rpv, m); }
    if (tid == 128) { double c0 = 0.0; for (int r = 0; r < m; ++r) c0 -= rpv[r] * hu[r]; rec[SR_C0] = c0; }
    __syncthreads();                                       // Bt dead -> reuse for Puᵀ
    for (int idx = tid; idx < RC_R20; idx += blockDim.x) Bt[idx] = 0.0;
    __syncthreads();
    rc_load_T(Bt, rec + SR_PU, 30, m);
    __syncthreads();
    // Khat = Px − Pu X ; khat = Pe − Pu y
    wg_gemm<true, false>(Bt, W, 2, 2, 0, 5, [&](int r, int c, double v) { if (r < 30 && c < 30) rec[SR_KHAT + r * 30 + c] = rec[SR_PX + r * 30 + c] - v; });
    if (tid < 30) rec[SR_KHATV + tid] = rec[SR_PE + tid] - tile_col_dot(Bt, tid, hu, m);
    __syncthreads();
  }
  // ---- forward rollout on the closed-loop maps ----
  double* dxv = S + RC_V_DX; double* dxn = S + RC_V_DXN; double* Ah = At; double* Kh = SA;
  if (tid < 30) dxv[tid] = a.x0[(size_t)b * 30 + tid] - a.x[(0 * a.B + b) * 30 + tid];
  __syncthreads();
  double armijo = 0.0, dx2 = 0.0, du2 = 0.0;
  for (int k = 0; k < n - 1; ++k) {
    if (a.skip & 4) break;
    const double* rec = a.stage + ((size_t)b * a.nmax + k) * SR_SIZE;
    const int nb = k * a.B + b;
    if (tid < 30) { a.dx[nb * 30 + tid] = dxv[tid]; dx2 += dxv[tid] * dxv[tid]; }
    if (a.node_ev[nb] == QM_EV_PRE) {
      if (tid < 30) { a.du[nb * 30 + tid] = 0.0; dxv[tid] += a.x[nb * 30 + tid] - a.x[((k + 1) * a.B + b) * 30 + tid]; }
      __syncthreads();
      continue;
    }
    rc_load30(Ah, rec + SR_AHAT, 30); rc_load30(Kh, rec + SR_KHAT, 30);
    __syncthreads();
    if (tid < 30) dxn[tid] = rec[SR_BHAT + tid] + tile_row_dot(Ah, tid, dxv, 30);
    if (tid >= 64 && tid < 94) { const int r = tid - 64; const double v = rec[SR_KHATV + r] + tile_row_dot(Kh, r, dxv, 30); a.du[nb * 30 + r] = v; du2 += v * v; armijo += rec[SR_GHAT + r] * dxv[r]; }
    if (tid == 128) armijo += rec[SR_C0];
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
  if (tid == 0) { a.step_info[b * 4] = arm; a.step_info[b * 4 + 1] = sx; a.step_info[b * 4 + 2] = su; a.step_info[b * 4 + 3] = (double)chol_fail; }
}
