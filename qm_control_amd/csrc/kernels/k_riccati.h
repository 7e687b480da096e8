// k_riccati.h — K3: discrete-time Riccati backward sweep + forward rollout of the projected QP.
//
// ONE WAVEFRONT per MPC instance (64-thread workgroups, ~10 KB LDS): the sweep is a serial chain of small dense products, so
// the whole stage lives in the registers of one wave as f64-MFMA fragments and no workgroup barrier is ever needed.  With every
// equality constraint projected out and no inequality rows the QP sub-problem the reference hands to HPIPM is solved exactly by
// one Riccati factorise+solve ([upstream ocs2_sqp SqpSolver::getOCPSolution -> hpipm]):
//   Hux = P + Bᵀ S A, Huu = R + Bᵀ S B, hu = r + Bᵀ(s + S b);  L Lᵀ = Huu;  W = L⁻¹ Hux, y = L⁻¹ hu
//   S' = Q + Aᵀ S A − Wᵀ W (symmetrised),  s' = q + Aᵀ(s + S b) − Wᵀ y
// Event nodes (PreEvent -> PostEvent, identity jump, nu = 0): S' = S, s' = s + S b, b = x_i − x_{i+1}.
//
// Fragment algebra.  v_mfma_f64_16x16x4_f64 takes A[i = l&15][k = l>>4], B[k = l>>4][j = l&15] and returns
// D[row = (l>>4) + 4r][col = l&15].  A matrix held as D-fragments ("D-layout": tile (I,J), register r <-> element
// (16I + (l>>4) + 4r, 16J + (l&15))) is therefore directly the B operand of k-step kk = 4K + r, and — read as an A operand —
// it supplies its TRANSPOSE.  Every product of the recursion has the form P = Zᵀ Y (S is symmetric):
//   S A = Sᵀ A,  S B = Sᵀ B,  Hux = Bᵀ (S A),  Huu = Bᵀ (S B),  Aᵀ (S A),  Wᵀ W
// so results chain from MFMA to MFMA without any layout conversion.  The vectors ride in the padding column 30 of the 32-wide
// tiles: A|b, (S A | S b + s), (P | r), (Q | q), (W | y) — the mat-vecs cost nothing extra.
// Only two steps leave the registers (wave-local LDS round trips): the Cholesky + forward substitution of [Huu | Hux hu]
// (lane = column, column in registers, pivots broadcast with v_readlane) and the symmetrisation of S'.
//
// The backward sweep leaves L (in SR_RP), W (in SR_PP) and y (in SR_KFF) in the stage record; the forward rollout
//   ut = −L⁻ᵀ (W dx + y),  dx+ = Ap dx + Bp ut + bp,  du = Px dx + Pu ut + Pe,  Armijo metric += qp·dx + rp·ut
// streams the records once more with one matrix row per lane.
#pragma once
#include "qm_dev_common.h"

struct QmRiccatiArgs {
  int B, nmax;
  const int* n_nodes; const int* node_ev;      // [B], [nmax][B]
  const double* x0;                            // [B][30]
  const double* x;                             // [nmax][B][30] (current iterate; event defects, dx0)
  double* stage;                               // [B][nmax][SR_SIZE]  (L, W, y are written here)
  double* dx; double* du;                      // [nmax][B][30]
  double* step_info;                           // [B][4]: armijo, |dx|², |du|², chol status
  int skip;                                    // profiling only (bit mask: 1 Cholesky/solve, 2 matrix products, 4 forward, 8 symmetrise; results are then meaningless)
};

#define RW_BLOCK 64
#define RW_TLD 34                 /* transposition buffer [32][34] */
#define RW_CLD 66                 /* Cholesky staging [18][66]: lanes 0..17 Huu columns, lanes 32..62 [Hux | hu] columns */
#define RW_LDS_DOUBLES 1200
#define RW_LDS_BYTES (RW_LDS_DOUBLES * 8)

// P += (neg ? −1 : 1) · Zᵀ Y over k-steps [0, ksteps); Z: [KT][IT] tiles, Y: [KT][JT] tiles, P: [IT][JT] tiles (all D-layout)
template <int KT, int IT, int JT>
__device__ __forceinline__ void rw_gemm_tn(const qm_d4 (&Z)[KT][IT], const qm_d4 (&Y)[KT][JT], qm_d4 (&P)[IT][JT], int ksteps, bool neg) {
#pragma unroll
  for (int kk = 0; kk < 4 * KT; ++kk) if (kk < ksteps) {
#pragma unroll
    for (int I = 0; I < IT; ++I) {
      const double av = neg ? -Z[kk >> 2][I][kk & 3] : Z[kk >> 2][I][kk & 3];
#pragma unroll
      for (int J = 0; J < JT; ++J) P[I][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, Y[kk >> 2][J][kk & 3], P[I][J], 0, 0, 0);
    }
  }
}
template <int IT, int JT>
__device__ __forceinline__ void rw_zero(qm_d4 (&T)[IT][JT]) {
#pragma unroll
  for (int I = 0; I < IT; ++I)
#pragma unroll
    for (int J = 0; J < JT; ++J) T[I][J] = qm_d4{0.0, 0.0, 0.0, 0.0};
}
// D-layout load of a rows x cols row-major matrix (leading dim ld); optional vector in column 30 (rows < rows)
template <int IT, int JT>
__device__ __forceinline__ void rw_load(qm_d4 (&T)[IT][JT], const double* src, int ld, int rows, int cols, const double* col30) {
  const int g = (threadIdx.x & 63) >> 4, c = threadIdx.x & 15;
#pragma unroll
  for (int I = 0; I < IT; ++I)
#pragma unroll
    for (int J = 0; J < JT; ++J)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * I + g + 4 * r, col = 16 * J + c; double v = 0.0;
        if (row < rows) { if (col < cols) v = src[row * ld + col]; else if (J == 1 && col == 30 && col30) v = col30[row]; }
        T[I][J][r] = v;
      }
}

// one regular stage of the backward sweep; MT = number of 16-row tiles covering the m reduced inputs
template <int MT>
__device__ __forceinline__ void rw_stage(double* rec, int m, double* buf, qm_d4 (&S)[2][2], qm_d4 (&sv)[2], int skip, int& chol_fail) {
  const int l = threadIdx.x & 63, g = l >> 4, c = l & 15;
  qm_d4 A[2][2], Bm[2][MT], Hux[MT][2], Huu[MT][MT], Sn[2][2];
  rw_load<2, 2>(A, rec + SR_AP, 30, 30, 30, rec + SR_BPV);          // [Ap | bp]
  rw_load<2, MT>(Bm, rec + SR_BP, QM_MMAX, 30, m, nullptr);
  rw_load<MT, 2>(Hux, rec + SR_PP, 30, m, 30, rec + SR_RPV);        // [Pp | rp]
  rw_load<MT, MT>(Huu, rec + SR_RP, QM_MMAX, m, m, nullptr);
  rw_load<2, 2>(Sn, rec + SR_QP, 30, 30, 30, rec + SR_QPV);         // [Qp | qp]
  qm_d4 SA[2][2], SB[2][MT];
  rw_zero<2, 2>(SA); rw_zero<2, MT>(SB);
  if (!(skip & 2)) {
    rw_gemm_tn<2, 2, 2>(S, A, SA, 8, false);                         // [S A | S b]
#pragma unroll
    for (int I = 0; I < 2; ++I) SA[I][1] += sv[I];                    // column 30 += s
    rw_gemm_tn<2, 2, MT>(S, Bm, SB, 8, false);                       // S B
    rw_gemm_tn<2, MT, 2>(Bm, SA, Hux, 8, false);                     // [Hux | hu]
    rw_gemm_tn<2, MT, MT>(Bm, SB, Huu, 8, false);                    // Huu
    rw_gemm_tn<2, 2, 2>(A, SA, Sn, 8, false);                        // [Q + Aᵀ S A | q + Aᵀ (S b + s)]  (row 30 is garbage, masked below)
  }
  // ---- Cholesky of Huu and forward substitution of [Hux | hu]: lane = column, column in registers ----
  qm_d4 W[MT][2];
  if (!(skip & 1)) {
    qm_wave_sync();
#pragma unroll
    for (int I = 0; I < MT; ++I)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * I + g + 4 * r;
        if (row < QM_MMAX) {
#pragma unroll
          for (int J = 0; J < MT; ++J) buf[row * RW_CLD + 16 * J + c] = Huu[I][J][r];
#pragma unroll
          for (int J = 0; J < 2; ++J) buf[row * RW_CLD + 32 + 16 * J + c] = Hux[I][J][r];
        }
      }
    qm_wave_sync();
    const bool live = (l < 16 * MT) || (l >= 32);
    double col[QM_MMAX];
#pragma unroll
    for (int i = 0; i < QM_MMAX; ++i) col[i] = (i < 16 * MT && live) ? buf[i * RW_CLD + l] : 0.0;
    if (l < QM_MMAX && l < 16 * MT) {                                // symmetrise Huu
#pragma unroll
      for (int i = 0; i < QM_MMAX; ++i) if (i < 16 * MT) col[i] = 0.5 * (col[i] + buf[l * RW_CLD + i]);
    }
#pragma unroll
    for (int j = 0; j < QM_MMAX; ++j) if (j < m) {
      const double djj = qm_bcast(col[j], j);
      if (!(djj > 0.0)) chol_fail = 1;
      const double d = sqrt(djj), inv = 1.0 / d;
      const double lcj = col[j] * inv;                               // L[c][j] on the Huu lanes (symmetry), (L⁻¹ rhs)[j] on the rhs lanes
#pragma unroll
      for (int i = j + 1; i < QM_MMAX; ++i) {
        const double lij = qm_bcast(col[i], j) * inv;
        col[i] = (l > j) ? col[i] - lij * lcj : ((l == j) ? lij : col[i]);
      }
      col[j] = (l > j) ? lcj : ((l == j) ? d : col[j]);
    }
    // lane j < m now holds column j of L (rows >= j); lanes 32.. hold the columns of [W | y]
    qm_wave_sync();
#pragma unroll
    for (int i = 0; i < QM_MMAX; ++i) if (i < m) {
      if (l < m && i >= l) rec[SR_RP + i * QM_MMAX + l] = col[i];
      if (l >= 32 && l < 62) rec[SR_PP + i * 30 + (l - 32)] = col[i];
      if (l == 62) rec[SR_KFF + i] = col[i];
      if (l >= 32) buf[i * RW_CLD + l] = col[i];
    }
    qm_wave_sync();
#pragma unroll
    for (int I = 0; I < MT; ++I)
#pragma unroll
      for (int J = 0; J < 2; ++J)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int row = 16 * I + g + 4 * r; W[I][J][r] = (row < m) ? buf[row * RW_CLD + 32 + 16 * J + c] : 0.0; }
  } else rw_zero<MT, 2>(W);
  if (!(skip & 2)) rw_gemm_tn<MT, 2, 2>(W, W, Sn, (m + 3) >> 2, true);   // −[Wᵀ W | Wᵀ y]
  // ---- S' <- sym(Sn[0:30, 0:30]), s' <- Sn[0:30, 30] ----
#pragma unroll
  for (int I = 0; I < 2; ++I)
#pragma unroll
    for (int r = 0; r < 4; ++r) { const int row = 16 * I + g + 4 * r; sv[I][r] = (c == 14 && row < 30) ? Sn[I][1][r] : 0.0; }
  if (skip & 8) {
#pragma unroll
    for (int I = 0; I < 2; ++I)
#pragma unroll
      for (int J = 0; J < 2; ++J)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int row = 16 * I + g + 4 * r, cc = 16 * J + c; S[I][J][r] = (row < 30 && cc < 30) ? Sn[I][J][r] : 0.0; }
    return;
  }
  qm_wave_sync();
#pragma unroll
  for (int I = 0; I < 2; ++I)
#pragma unroll
    for (int J = 0; J < 2; ++J)
#pragma unroll
      for (int r = 0; r < 4; ++r) buf[(16 * I + g + 4 * r) * RW_TLD + 16 * J + c] = Sn[I][J][r];
  qm_wave_sync();
#pragma unroll
  for (int I = 0; I < 2; ++I)
#pragma unroll
    for (int J = 0; J < 2; ++J)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * I + g + 4 * r, cc = 16 * J + c;
        S[I][J][r] = (row < 30 && cc < 30) ? 0.5 * (Sn[I][J][r] + buf[cc * RW_TLD + row]) : 0.0;
      }
}

__global__ void __launch_bounds__(RW_BLOCK) qm_riccati_kernel(QmRiccatiArgs a) {
  extern __shared__ double qm_smem[];
  double* buf = qm_smem;
  const int l = threadIdx.x & 63, g = l >> 4, c = l & 15, b = blockIdx.x;
  if (b >= a.B) return;
  const int n = a.n_nodes[b];
  int chol_fail = 0;
  qm_d4 S[2][2], sv[2];
  {   // terminal value function
    const double* rec = a.stage + ((size_t)b * a.nmax + (n - 1)) * SR_SIZE;
    rw_load<2, 2>(S, rec + SR_QP, 30, 30, 30, nullptr);
#pragma unroll
    for (int I = 0; I < 2; ++I)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const int row = 16 * I + g + 4 * r; sv[I][r] = (c == 14 && row < 30) ? rec[SR_QPV + row] : 0.0; }
  }
  for (int k = n - 2; k >= 0; --k) {
    double* rec = a.stage + ((size_t)b * a.nmax + k) * SR_SIZE;
    if (a.node_ev[k * a.B + b] == QM_EV_PRE) {
      // s += S (x_k − x_{k+1}): the defect rides in column 30 of a one-tile-wide right-hand side
      qm_d4 Y[2][1], P[2][1];
#pragma unroll
      for (int I = 0; I < 2; ++I) {
        P[I][0] = qm_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int row = 16 * I + g + 4 * r; Y[I][0][r] = (c == 14 && row < 30) ? a.x[(k * a.B + b) * 30 + row] - a.x[((k + 1) * a.B + b) * 30 + row] : 0.0; }
      }
      rw_gemm_tn<2, 2, 1>(S, Y, P, 8, false);
#pragma unroll
      for (int I = 0; I < 2; ++I) sv[I] += P[I][0];
      continue;
    }
    const int m = (int)rec[SR_SCAL];
    if (m <= 16) rw_stage<1>(rec, m, buf, S, sv, a.skip, chol_fail);
    else rw_stage<2>(rec, m, buf, S, sv, a.skip, chol_fail);
  }
  // L, W, y were stored by other lanes than the ones that read them back below
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
  // ---- forward rollout: lanes 0..29 own the rows of [Ap Bp bp], lanes 32..61 the rows of [Px Pu Pe]; lanes 0..m-1 also own
  //      row i of W and column i of L; lane c carries dx[c] ----
  double dxl = (l < 30) ? a.x0[(size_t)b * 30 + l] - a.x[(0 * a.B + b) * 30 + l] : 0.0;
  double armijo = 0.0, dx2 = 0.0, du2 = 0.0;
  const int half = l >> 5, r = l & 31;
  for (int k = 0; k < n - 1; ++k) {
    if (a.skip & 4) break;
    const double* rec = a.stage + ((size_t)b * a.nmax + k) * SR_SIZE;
    const int nb = k * a.B + b;
    if (l < 30) { a.dx[nb * 30 + l] = dxl; dx2 += dxl * dxl; }
    if (a.node_ev[nb] == QM_EV_PRE) {
      if (l < 30) { a.du[nb * 30 + l] = 0.0; dxl += a.x[nb * 30 + l] - a.x[((k + 1) * a.B + b) * 30 + l]; }
      continue;
    }
    const int m = (int)rec[SR_SCAL];
    double ra[30], rb[QM_MMAX], wr[30], lc[QM_MMAX];
    const double* srcA = rec + (half ? SR_PX : SR_AP) + r * 30; const double* srcB = rec + (half ? SR_PU : SR_BP) + r * QM_MMAX;
#pragma unroll
    for (int q = 0; q < 30; ++q) ra[q] = (r < 30) ? srcA[q] : 0.0;
#pragma unroll
    for (int q = 0; q < QM_MMAX; ++q) rb[q] = (r < 30 && q < m) ? srcB[q] : 0.0;
#pragma unroll
    for (int q = 0; q < 30; ++q) wr[q] = (l < m) ? rec[SR_PP + l * 30 + q] : 0.0;
#pragma unroll
    for (int q = 0; q < QM_MMAX; ++q) lc[q] = (l < m && q >= l && q < m) ? rec[SR_RP + q * QM_MMAX + l] : 0.0;
    double acc = (r < 30) ? rec[(half ? SR_PE : SR_BPV) + r] : 0.0;
    double t = (l < m) ? rec[SR_KFF + l] : 0.0;
    const double qv = (l < 30) ? rec[SR_QPV + l] : 0.0, rp = (l < m) ? rec[SR_RPV + l] : 0.0;
#pragma unroll
    for (int q = 0; q < 30; ++q) { const double dq = qm_bcast(dxl, q); t += wr[q] * dq; acc += ra[q] * dq; }
    // Lᵀ v = t (lane i keeps v_i), ut = −v
    double v = 0.0;
#pragma unroll
    for (int q = QM_MMAX - 1; q >= 0; --q) if (q < m) {
      const double vq = qm_bcast(t / lc[q], q);          // lane q: t_q / L[q][q]
      if (l == q) v = vq;
      t -= lc[q] * vq;                                    // lanes i < q: L[q][i] v_q   (lanes >= q: their t is dead)
    }
    const double ut = -v;
    armijo += qv * dxl + rp * ut;
#pragma unroll
    for (int q = 0; q < QM_MMAX; ++q) if (q < m) acc += rb[q] * qm_bcast(ut, q);
    if (half && r < 30) { a.du[nb * 30 + r] = acc; du2 += acc * acc; }
    dxl = (l < 30) ? acc : 0.0;
  }
  {
    const int nb = (n - 1) * a.B + b; const double* rec = a.stage + ((size_t)b * a.nmax + (n - 1)) * SR_SIZE;
    if (l < 30) { a.dx[nb * 30 + l] = dxl; a.du[nb * 30 + l] = 0.0; dx2 += dxl * dxl; armijo += rec[SR_QPV + l] * dxl; }
  }
  double arm = armijo, sx = dx2, su = du2;
  for (int off = 32; off > 0; off >>= 1) { arm += __shfl_xor(arm, off, 64); sx += __shfl_xor(sx, off, 64); su += __shfl_xor(su, off, 64); }
  if (l == 0) { a.step_info[b * 4] = arm; a.step_info[b * 4 + 1] = sx; a.step_info[b * 4 + 2] = su; a.step_info[b * 4 + 3] = (double)chol_fail; }
}
