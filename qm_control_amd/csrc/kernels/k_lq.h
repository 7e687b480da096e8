// k_lq.h — K1: per-shooting-node linear-quadratic approximation + equality-constraint projection.
//
// One 256-thread workgroup (4 wavefronts) per (instance b, node i).  Restates, MI355X-first, what
// [upstream ocs2_sqp multiple_shooting::setupIntermediateNode + projectTranscription] do per node for the
// OCP of qm_interface/src/QMInterface.cpp:79-142 (SURVEY.md §8 a2–a8, a11; Appendix B.6 steps 2–3):
//   K1a qm_lq_kin_kernel (one THREAD per node): all scalar kinematics — both Heun/RK2 stages, flow values, EE pose
//       error — written as a 4 KB "kin record" per node (lanes = instances: no idle lanes, no barriers)
//   K1b qm_lq_kernel (one WORKGROUP per node, 46 KB LDS -> 3 workgroups / CU):
//   phase I   analytic Jacobian columns of the flow map, one lane per column.  df/dx and df/du only have 12 (+4 identity)
//             non-trivial rows (SRBD), so they live in 16-row half tiles; RK2 sensitivity composition
//             A_d = I + dt/2 (A1 + A2 + dt A2 A1), B_d likewise, on f64 MFMA over the non-zero k range only
//   phase II  equality rows (zero force / zero foot velocity / swing normal velocity) and their closed-form block
//             projection du = Pe + Px dx + Pu ut (D is block structured by construction: each row touches one foot's
//             force triple or one leg's joint-velocity triple); projected dynamics Ap, Bp, bp streamed to HBM
//   phase III cost quadratic model (tracking + arm soft box + friction-cone barrier + EE pose), x dt, and the projected
//             cost Qp, Pp, Rp, qp, rp (f64 MFMA, k restricted to the 12 rows Px / R Px occupy) streamed to HBM
// The terminal node only carries the final EE soft constraint (QMInterface.cpp:104).
#pragma once
#include "qm_dev_kin.h"

struct QmLqArgs {
  const double* mb; const double* st;
  int B, nmax;
  const int* n_nodes;        // [B]
  const double* node_ts;     // [nmax][B] interval start time of node i
  const double* node_dt;     // [nmax][B] interval duration (0 for event / terminal nodes)
  const int* node_ev;        // [nmax][B] QM_EV_*
  const int* node_mode;      // [nmax][B] contact mode at the interval start
  const double* zvel;        // [nmax][B][4] swing z-velocity reference per contact
  const double* zpos;        // [nmax][B][4]
  const double* xref;        // [nmax][B][30]
  const double* eeref;       // [nmax][B][7]  pos(3) quat xyzw(4)
  const double* x;           // [nmax][B][30]
  const double* u;           // [nmax][B][30]
  double* stage;             // [B][nmax][SR_SIZE]
  double* perf;              // [nmax][B][PF_SIZE]
  double* dbg;               // optional [B][nmax][LQ_DBG_SIZE] unprojected LQ data (parity tests); may be null
  double* kin;               // [nmax][B][KR_SIZE] kin records (K1a -> K1b)
};

// debug record (unprojected LQ): A(900) B(900) b(30) Q(900) R(900) q(30) r(30) C(16x30) D(16x30) e(16) c nc
#define LQ_DBG_A 0
#define LQ_DBG_B 900
#define LQ_DBG_b 1800
#define LQ_DBG_Q 1830
#define LQ_DBG_R 2730
#define LQ_DBG_q 3630
#define LQ_DBG_r 3660
#define LQ_DBG_C 3690
#define LQ_DBG_D 4170
#define LQ_DBG_e 4650
#define LQ_DBG_c 4666
#define LQ_DBG_nc 4667
#define LQ_DBG_SIZE 4668

// kin record (doubles) per node
#define KR_K1   0
#define KR_K2   KW_SIZE
#define KR_F1   (2 * KW_SIZE)
#define KR_F2   (KR_F1 + 30)
#define KR_X2   (KR_F2 + 30)
#define KR_EEG  (KR_X2 + 30)          /* g(6) */
#define KR_QEE  (KR_EEG + 6)          /* qee(4) */
#define KR_SIZE (KR_QEE + 4 + 2)

// LDS carve (doubles).  Tile pool of 4896 doubles re-used by the three phases (row-major, leading dim QM_LD):
//   phase I : A1h[0] B1h[544] A2h[1088] B2h[1632] Th[2176] (16-row halves)  Ad[2720] Bd[3808] (32-row tiles)
//   phase II: Px[0] (rows 12..27) PuT[544] (m<=18 rows, transposed Pu)  C[1156] D[1700] (16 rows)   Ad Bd
//   phase III: Px PuT  RPx[1156] (rows 12..27)  RPuT[1700] (m rows, (R Pu)^T)  Q[2720] R[3808]
#define LQ_HALF (16 * QM_LD)
#define LQ_R18  (18 * QM_LD)
#define LQ_P_A1 0
#define LQ_P_B1 544
#define LQ_P_A2 1088
#define LQ_P_B2 1632
#define LQ_P_T  2176
#define LQ_P_AD 2720
#define LQ_P_BD 3808
#define LQ_P_PX 0
#define LQ_P_PUT 544
#define LQ_P_C  1156
#define LQ_P_D  1700
#define LQ_P_RPX 1156
#define LQ_P_RPUT 1700
#define LQ_P_Q  2720
#define LQ_P_R  3808
#define LQ_POOL 4896
#define LQ_VEC LQ_POOL
#define LQ_V_X    (LQ_VEC + 0)      /* x(32) */
#define LQ_V_U    (LQ_VEC + 32)
#define LQ_V_XN   (LQ_VEC + 64)
#define LQ_V_X2   (LQ_VEC + 96)
#define LQ_V_F1   (LQ_VEC + 128)
#define LQ_V_F2   (LQ_VEC + 160)
#define LQ_V_B    (LQ_VEC + 192)    /* b */
#define LQ_V_Q    (LQ_VEC + 224)    /* q */
#define LQ_V_R    (LQ_VEC + 256)    /* r */
#define LQ_V_PE   (LQ_VEC + 288)
#define LQ_V_RR   (LQ_VEC + 320)    /* r + R Pe */
#define LQ_V_E    (LQ_VEC + 352)    /* e(16) */
#define LQ_V_DU   (LQ_VEC + 368)    /* u - unom */
#define LQ_V_RED  (LQ_VEC + 400)    /* reduction scratch (8) */
#define LQ_V_G    (LQ_VEC + 408)    /* per contact: Ginv or g data (4 x 12) */
#define LQ_V_EE   (LQ_VEC + 456)    /* g(6), mu(6), qee(4), ref(7) */
#define LQ_V_K1   (LQ_VEC + 480)
#define LQ_V_K2   (LQ_VEC + 480 + KW_SIZE)
#define LQ_V_JEE  (LQ_VEC + 480 + 2 * KW_SIZE)    /* EE Jacobian transposed [30][6] */
#define LQ_LDS_DOUBLES (LQ_VEC + 480 + 2 * KW_SIZE + 180)
#define LQ_LDS_BYTES (LQ_LDS_DOUBLES * 8)

__device__ __forceinline__ double block_sum(double v, double* red) {   // sum over the workgroup; result to all
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  double s = 0.0; for (int i = 0; i < nw; ++i) s += red[i];
  return s;
}

// ---- K1a: scalar kinematics, one thread per (node, instance) ----
__global__ void qm_lq_kin_kernel(QmLqArgs a) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = g / a.B, b = g - i * a.B;
  if (i >= a.nmax) return;
  const int nn = a.n_nodes[b];
  if (i >= nn) return;
  const int nb = i * a.B + b; const bool terminal = (i == nn - 1);
  if (!terminal && a.node_ev[nb] == QM_EV_PRE) return;
  double* rec = a.kin + (size_t)nb * KR_SIZE;
  double x[30], u[30], K[KW_SIZE];
  for (int q = 0; q < 30; ++q) x[q] = a.x[nb * 30 + q];
  const double* ee = a.eeref + nb * 7;
  if (terminal) {
    kin_base(a.mb, x, K); kin_arm(a.mb, x, K);
    for (int q = 0; q < KW_SIZE; ++q) rec[KR_K1 + q] = K[q];
    double gq[6], qee[4]; ee_error(K, ee, ee + 3, qee, gq);
    for (int q = 0; q < 6; ++q) rec[KR_EEG + q] = gq[q]; for (int q = 0; q < 4; ++q) rec[KR_QEE + q] = qee[q];
    return;
  }
  for (int q = 0; q < 30; ++q) u[q] = a.u[nb * 30 + q];
  const double dt = a.node_dt[nb];
  kin_base(a.mb, x, K); for (int c = 0; c < 4; ++c) kin_leg(a.mb, c, x, u, K); kin_arm(a.mb, x, K);
  for (int q = 0; q < KW_SIZE; ++q) rec[KR_K1 + q] = K[q];
  { double gq[6], qee[4]; ee_error(K, ee, ee + 3, qee, gq); for (int q = 0; q < 6; ++q) rec[KR_EEG + q] = gq[q]; for (int q = 0; q < 4; ++q) rec[KR_QEE + q] = qee[q]; }
  double f1[30], x2[30], f2[30];
  flow_from_kin(a.mb, x, u, K, f1);
  for (int q = 0; q < 30; ++q) { x2[q] = x[q] + dt * f1[q]; rec[KR_F1 + q] = f1[q]; rec[KR_X2 + q] = x2[q]; }
  kin_base(a.mb, x2, K); for (int c = 0; c < 4; ++c) kin_leg(a.mb, c, x2, u, K);
  flow_from_kin(a.mb, x2, u, K, f2);
  for (int q = 0; q < KW_ARM; ++q) rec[KR_K2 + q] = K[q];
  for (int q = 0; q < 30; ++q) rec[KR_F2 + q] = f2[q];
}

// ---- K1b: one workgroup per node ----
__global__ void __launch_bounds__(QM_BLOCK) qm_lq_kernel(QmLqArgs a) {
  extern __shared__ double qm_smem[];
  double* S = qm_smem;
  const int tid = threadIdx.x;
  const int b = blockIdx.x / a.nmax, i = blockIdx.x - b * a.nmax;
  const int nn = a.n_nodes[b];
  if (i >= nn) return;
  const int nb = i * a.B + b;                       // node-major index
  const int ev = a.node_ev[nb];
  const bool terminal = (i == nn - 1);
  if (!terminal && ev == QM_EV_PRE) return;         // event nodes carry no LQ data (identity jump, handled by K3)
  const double* mb = a.mb; const double* st = a.st;
  double* rec = a.stage + ((size_t)b * a.nmax + i) * SR_SIZE;
  double* dbg = a.dbg ? a.dbg + ((size_t)b * a.nmax + i) * LQ_DBG_SIZE : nullptr;
  const double* kr = a.kin + (size_t)nb * KR_SIZE;
  const double dt = a.node_dt[nb];
  const int mode = a.node_mode[nb];

  // ---- P0: clear LDS, stage inputs and the kin record ----
  tile_zero(S, LQ_LDS_DOUBLES);
  __syncthreads();
  if (tid < 30) { S[LQ_V_X + tid] = a.x[nb * 30 + tid]; S[LQ_V_U + tid] = terminal ? 0.0 : a.u[nb * 30 + tid]; S[LQ_V_XN + tid] = terminal ? 0.0 : a.x[((i + 1) * a.B + b) * 30 + tid]; }
  if (tid >= 32 && tid < 39) S[LQ_V_EE + 16 + (tid - 32)] = a.eeref[nb * 7 + (tid - 32)];
  if (tid >= 40 && tid < 46) S[LQ_V_EE + (tid - 40)] = kr[KR_EEG + (tid - 40)];
  if (tid >= 48 && tid < 52) S[LQ_V_EE + 12 + (tid - 48)] = kr[KR_QEE + (tid - 48)];
  for (int idx = tid; idx < KW_SIZE; idx += blockDim.x) { S[LQ_V_K1 + idx] = kr[KR_K1 + idx]; if (!terminal && idx < KW_ARM) S[LQ_V_K2 + idx] = kr[KR_K2 + idx]; }
  if (!terminal && tid >= 64 && tid < 94) { const int q = tid - 64; S[LQ_V_F1 + q] = kr[KR_F1 + q]; S[LQ_V_F2 + q] = kr[KR_F2 + q]; S[LQ_V_X2 + q] = kr[KR_X2 + q]; }
  __syncthreads();
  double* X = S + LQ_V_X; double* U = S + LQ_V_U;
  double* K1 = S + LQ_V_K1; double* K2 = S + LQ_V_K2;
  double* EE = S + LQ_V_EE; double* JEE = S + LQ_V_JEE;

  if (terminal) {
    // final EE soft constraint only: Q_N = J^T mu J, q_N = J^T mu g, c_N = 1/2 g mu g
    if (tid < 6) EE[6 + tid] = (tid < 3 ? st[ST_MU_EEF_POS] : st[ST_MU_EEF_ORI]);
    if (tid >= 32 && tid < 62) { const int c = tid - 32; double col[6]; ee_jac_col(X, K1, EE + 12, EE + 19, c, col); for (int r = 0; r < 6; ++r) JEE[c * 6 + r] = col[r]; }
    __syncthreads();
    for (int idx = tid; idx < 900; idx += blockDim.x) { const int r = idx / 30, c = idx - r * 30; double s = 0.0; for (int k = 0; k < 6; ++k) s += JEE[r * 6 + k] * EE[6 + k] * JEE[c * 6 + k]; rec[SR_QP + idx] = s; }
    if (tid < 30) { double s = 0.0; for (int k = 0; k < 6; ++k) s += JEE[tid * 6 + k] * EE[6 + k] * EE[k]; rec[SR_QPV + tid] = s; }
    if (tid == 0) { double c = 0.0; for (int k = 0; k < 6; ++k) c += 0.5 * EE[6 + k] * EE[k] * EE[k]; rec[SR_SCAL] = 0.0; rec[SR_SCAL + 1] = c; a.perf[nb * PF_SIZE] = c; a.perf[nb * PF_SIZE + 1] = 0.0; a.perf[nb * PF_SIZE + 2] = 0.0; }
    return;
  }

  // ---- phase I: Jacobian columns (lanes 0..59 stage 1, lanes 64..123 stage 2) into 16-row half tiles ----
  double* A1 = S + LQ_P_A1; double* B1 = S + LQ_P_B1; double* A2 = S + LQ_P_A2; double* B2 = S + LQ_P_B2; double* Th = S + LQ_P_T;
  double* Ad = S + LQ_P_AD; double* Bd = S + LQ_P_BD;
  {
    const int which = tid >> 6, c = tid & 63;
    if (which < 2 && c < 60) {
      double col[12]; flow_jac_col(mb, which ? S + LQ_V_X2 : X, U, which ? K2 : K1, c, col);
      double* M = (c < 30) ? (which ? A2 : A1) : (which ? B2 : B1); const int cc = (c < 30) ? c : c - 30;
      for (int r = 0; r < 12; ++r) M[r * QM_LD + cc] = col[r];
      if (c >= 42 && c < 46) M[(c - 30) * QM_LD + cc] = 1.0;      // d qdot_j / d u_j rows 12..15 (rows >= 16 are handled analytically)
    }
  }
  __syncthreads();
  // A2 A1: A1 only has rows < 16 -> k slabs 0..3
  wg_gemm<false, false>(A2, A1, 1, 2, 0, 4, [&](int r, int c, double v) { Th[r * QM_LD + c] = v; });
  __syncthreads();
  for (int idx = tid; idx < 900; idx += blockDim.x) { const int r = idx / 30, c = idx - r * 30; const int o = r * QM_LD + c; Ad[o] = ((r < 16) ? 0.5 * dt * A1[o] + 0.5 * dt * (A2[o] + dt * Th[o]) : 0.0) + (r == c ? 1.0 : 0.0); }
  __syncthreads();
  // A2 B1 = A2[:, :16] B1h + A2[:, 16:30] (rows >= 16 of B1 are unit rows e_k)
  wg_gemm<false, false>(A2, B1, 1, 2, 0, 4, [&](int r, int c, double v) { Th[r * QM_LD + c] = v + ((c >= 16 && c < 30) ? A2[r * QM_LD + c] : 0.0); });
  __syncthreads();
  for (int idx = tid; idx < 900; idx += blockDim.x) { const int r = idx / 30, c = idx - r * 30; const int o = r * QM_LD + c; Bd[o] = (r < 16) ? 0.5 * dt * B1[o] + 0.5 * dt * (B2[o] + dt * Th[o]) : ((r == c) ? dt : 0.0); }
  if (tid < 30) S[LQ_V_B + tid] = X[tid] + 0.5 * dt * S[LQ_V_F1 + tid] + 0.5 * dt * S[LQ_V_F2 + tid] - S[LQ_V_XN + tid];
  __syncthreads();
  if (dbg) { tile_store(Ad, dbg + LQ_DBG_A, 30, 30, 30); tile_store(Bd, dbg + LQ_DBG_B, 30, 30, 30); if (tid < 30) dbg[LQ_DBG_b + tid] = S[LQ_V_B + tid]; }

  // ---- phase II: equality rows + closed-form block projection ----
  // rows ordered per contact i = LF,RF,LH,RH: swing -> [F_i = 0 (3)] , stance -> [v_i = 0 (3)] , swing -> [v_iz = zvel_ref (1)]
  double* Ct = S + LQ_P_C; double* Dt = S + LQ_P_D; double* Pxs = S + LQ_P_PX; double* PuT = S + LQ_P_PUT;
  double* Px = Pxs - 12 * QM_LD;                         // virtual base: Px[r] valid for rows 12..27
  __syncthreads();
  tile_zero(S, LQ_P_AD);                                 // clears the phase-I halves (Px, PuT, C, D regions)
  __syncthreads();
  int row0[4]; int nc = 0; for (int k = 0; k < 4; ++k) { row0[k] = nc; nc += mode_flag(mode, k) ? 3 : 4; }
  const double gain = st[ST_POS_ERR_GAIN];
  {
    const int k = tid >> 6, c = tid & 63;                 // one wave per contact, one lane per column of [x | u]
    if (c < 60) {
      const bool stance = mode_flag(mode, k);
      double dv[3], dpz; foot_vel_jac_col(mb, X, U, K1, k, c, dv, &dpz);
      double* M = (c < 30) ? Ct : Dt; const int cc = (c < 30) ? c : c - 30;
      if (stance) { for (int r = 0; r < 3; ++r) M[(row0[k] + r) * QM_LD + cc] = dv[r] + ((r == 2 && gain != 0.0) ? gain * dpz : 0.0); }
      else {
        M[(row0[k] + 3) * QM_LD + cc] = dv[2] + (gain != 0.0 ? gain * dpz : 0.0);
        if (c >= 30 && c - 30 >= 3 * k && c - 30 < 3 * k + 3) M[(row0[k] + (c - 30 - 3 * k)) * QM_LD + cc] = 1.0;
      }
    }
    if (c == 60) {
      const bool stance = mode_flag(mode, k); double v[3]; foot_velocity(X, K1, k, v); const double pz = kin_foot(K1, k)[2];
      if (stance) { for (int r = 0; r < 3; ++r) S[LQ_V_E + row0[k] + r] = v[r] + ((r == 2 && gain != 0.0) ? gain * pz : 0.0); }
      else {
        for (int r = 0; r < 3; ++r) S[LQ_V_E + row0[k] + r] = U[3 * k + r];
        double bb = -a.zvel[nb * 4 + k]; if (gain != 0.0) bb -= gain * a.zpos[nb * 4 + k];
        S[LQ_V_E + row0[k] + 3] = bb + v[2] + (gain != 0.0 ? gain * pz : 0.0);
      }
    }
  }
  __syncthreads();
  if (dbg) { tile_store(Ct, dbg + LQ_DBG_C, 16, 30, 30); tile_store(Dt, dbg + LQ_DBG_D, 16, 30, 30); if (tid < 16) dbg[LQ_DBG_e + tid] = S[LQ_V_E + tid]; if (tid == 0) dbg[LQ_DBG_nc] = nc; }
  // per contact: stance -> Ginv (3x3) of the joint-velocity block; swing -> g/(g.g) and a 3x2 orthonormal complement of g
  double* G = S + LQ_V_G;
  if (tid < 4) {
    const int k = tid, ch = contact_to_chain(k), jc = 12 + 3 * ch; double* g = G + 12 * k;
    if (mode_flag(mode, k)) { double M3[9]; for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) M3[3 * r + c] = Dt[(row0[k] + r) * QM_LD + jc + c]; m3_inv(M3, g); }
    else {
      const double gv[3] = {Dt[(row0[k] + 3) * QM_LD + jc], Dt[(row0[k] + 3) * QM_LD + jc + 1], Dt[(row0[k] + 3) * QM_LD + jc + 2]};
      const double n2 = gv[0] * gv[0] + gv[1] * gv[1] + gv[2] * gv[2], nrm = sqrt(n2);
      for (int r = 0; r < 3; ++r) g[r] = gv[r] / n2;
      // Householder H = I - 2 v v^T/(v^T v), v = g - alpha e1, alpha = -sign(g0)|g| : H e1 || g, columns 2,3 of H span g-perp
      const double alpha = gv[0] > 0.0 ? -nrm : nrm; const double v[3] = {gv[0] - alpha, gv[1], gv[2]}; const double vv = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
      for (int r = 0; r < 3; ++r) { g[3 + r] = ((r == 1) ? 1.0 : 0.0) - 2.0 * v[r] * v[1] / vv; g[6 + r] = ((r == 2) ? 1.0 : 0.0) - 2.0 * v[r] * v[2] / vv; }
    }
  }
  __syncthreads();
  double* Pe = S + LQ_V_PE;
  for (int idx = tid; idx < 12 * 31; idx += blockDim.x) {
    const int r = idx / 31, c = idx - r * 31;             // input row 12 + r ; c == 30 -> Pe entry
    const int ch = r / 3, jj = r - 3 * ch, k = chain_to_contact(ch); const double* g = G + 12 * k;
    double s = 0.0;
    if (mode_flag(mode, k)) { for (int q = 0; q < 3; ++q) s -= g[3 * jj + q] * ((c < 30) ? Ct[(row0[k] + q) * QM_LD + c] : S[LQ_V_E + row0[k] + q]); }
    else { s = -g[jj] * ((c < 30) ? Ct[(row0[k] + 3) * QM_LD + c] : S[LQ_V_E + row0[k] + 3]); }
    if (c < 30) Px[(12 + r) * QM_LD + c] = s; else Pe[12 + r] = s;
  }
  if (tid < 12) { const int k = tid / 3; Pe[tid] = mode_flag(mode, k) ? 0.0 : -U[tid]; }
  if (tid >= 24 && tid < 30) Pe[tid] = 0.0;
  // Pu^T (m x 30): rows = projected inputs: stance forces, swing-leg null spaces, arm
  int m = 0;
  {
    int col = 0;
    for (int k = 0; k < 4; ++k) if (mode_flag(mode, k)) { if (tid < 3) PuT[(col + tid) * QM_LD + 3 * k + tid] = 1.0; col += 3; }
    for (int k = 0; k < 4; ++k) if (!mode_flag(mode, k)) { const int jc = 12 + 3 * contact_to_chain(k); const double* g = G + 12 * k; if (tid < 6) { const int r = tid % 3, cc = tid / 3; PuT[(col + cc) * QM_LD + jc + r] = g[3 + 3 * cc + r]; } col += 2; }
    if (tid < 6) PuT[(col + tid) * QM_LD + 24 + tid] = 1.0; col += 6;
    m = col;
  }
  const double eq2 = block_sum((tid < nc) ? S[LQ_V_E + tid] * S[LQ_V_E + tid] : 0.0, S + LQ_V_RED);
  const double b2 = block_sum((tid < 30) ? S[LQ_V_B + tid] * S[LQ_V_B + tid] : 0.0, S + LQ_V_RED);
  __syncthreads();
  const int mt_m = (m + 15) / 16;
  // projected dynamics  Ap = Ad + Bd Px (Px rows 12..23 -> k slabs 3..5) ; Bp = Bd Pu ; bp = b + Bd Pe
  wg_gemm<false, false>(Bd, Px, 2, 2, 3, 6, [&](int r, int c, double v) { if (r < 30 && c < 30) rec[SR_AP + r * 30 + c] = Ad[r * QM_LD + c] + v; });
  wg_gemm<false, true>(Bd, PuT, 2, mt_m, 0, 8, [&](int r, int c, double v) { if (r < 30 && c < m) rec[SR_BP + r * QM_MMAX + c] = v; });
  if (tid < 30) rec[SR_BPV + tid] = S[LQ_V_B + tid] + tile_row_dot(Bd, tid, Pe, 30);
  __syncthreads();                                        // Ad, Bd, C, D dead from here

  // ---- phase III: cost quadratic model (x dt): Q, R ----
  double* Qt = S + LQ_P_Q; double* Rt = S + LQ_P_R; double* RPxs = S + LQ_P_RPX; double* RPuT = S + LQ_P_RPUT; double* RPx = RPxs - 12 * QM_LD;
  for (int i2 = tid; i2 < LQ_POOL - LQ_P_RPX; i2 += blockDim.x) S[LQ_P_RPX + i2] = 0.0;       // RPx, RPuT, Q, R regions
  __syncthreads();
  tile_load(Rt, st + ST_R, 30, 30, 30);
  double cost = 0.0;                                      // per-thread partial of the cost value
  if (tid < 30) {
    const double dx = X[tid] - a.xref[nb * 30 + tid]; const double qd = st[ST_Q + tid];
    S[LQ_V_Q + tid] = qd * dx; Qt[tid * QM_LD + tid] = qd; cost += 0.5 * qd * dx * dx;
    int nst = 0; for (int k = 0; k < 4; ++k) nst += mode_flag(mode, k);
    double unom = 0.0; if (tid < 12 && (tid % 3) == 2 && mode_flag(mode, tid / 3) && nst > 0) unom = mb[MB_ROBOTMASS] * 9.81 / nst;
    S[LQ_V_DU + tid] = U[tid] - unom;
  }
  if (tid >= 32 && tid < 62) { const int c = tid - 32; double col[6]; ee_jac_col(X, K1, EE + 12, EE + 19, c, col); for (int r = 0; r < 6; ++r) JEE[c * 6 + r] = col[r]; }
  if (tid >= 64 && tid < 70) EE[6 + (tid - 64)] = ((tid - 64) < 3 ? st[ST_MU_EE_POS] : st[ST_MU_EE_ORI]);
  __syncthreads();
  if (tid < 30) { const double s = tile_row_dot(Rt, tid, S + LQ_V_DU, 30); S[LQ_V_R + tid] = s; cost += 0.5 * S[LQ_V_DU + tid] * s; }
  __syncthreads();
  // arm soft box (a6), friction cone barrier (a7): few lanes, disjoint entries
  if (tid < 6) {
    const double mu = st[ST_JPOS_MU], de = st[ST_JPOS_DELTA]; const double lo = mb[MB_QLO + 12 + tid], hi = mb[MB_QHI + 12 + tid], z = X[24 + tid];
    cost += barrier_val(mu, de, z - lo) + barrier_val(mu, de, hi - z) - (barrier_val(mu, de, -lo) + barrier_val(mu, de, hi));
    S[LQ_V_Q + 24 + tid] += barrier_d1(mu, de, z - lo) - barrier_d1(mu, de, hi - z);
    Qt[(24 + tid) * QM_LD + 24 + tid] += barrier_d2(mu, de, z - lo) + barrier_d2(mu, de, hi - z);
  } else if (tid >= 8 && tid < 14) {
    const int k = tid - 8; const double mu = st[ST_JVEL_MU], de = st[ST_JVEL_DELTA]; const double lo = st[ST_JVEL_LO + k], hi = st[ST_JVEL_HI + k], w = U[24 + k];
    cost += barrier_val(mu, de, w - lo) + barrier_val(mu, de, hi - w) - (barrier_val(mu, de, -lo) + barrier_val(mu, de, hi));
    S[LQ_V_R + 24 + k] += barrier_d1(mu, de, w - lo) - barrier_d1(mu, de, hi - w);
    Rt[(24 + k) * QM_LD + 24 + k] += barrier_d2(mu, de, w - lo) + barrier_d2(mu, de, hi - w);
  } else if (tid >= 64 && tid < 68) {   // friction cone, one lane per contact (disjoint 3x3 blocks); Hessian shift summed below
    const int k = tid - 64; double ds = 0.0;
    if (mode_flag(mode, k)) {
      const double mu = st[ST_FRIC_MU], de = st[ST_FRIC_DELTA], muf = st[ST_FRIC_COEF], reg = st[ST_FRIC_REG], shift = st[ST_FRIC_SHIFT];
      const double Fx = U[3 * k], Fy = U[3 * k + 1], Fz = U[3 * k + 2]; const double T2 = Fx * Fx + Fy * Fy + reg, Tn = sqrt(T2), T3 = Tn * Tn * Tn;
      const double h = muf * Fz - Tn; cost += barrier_val(mu, de, h);
      const double p1 = barrier_d1(mu, de, h), p2 = barrier_d2(mu, de, h);
      const double dh[3] = {-Fx / Tn, -Fy / Tn, muf};
      const double ddh[9] = {-(Fy * Fy + reg) / T3, Fx * Fy / T3, 0.0, Fx * Fy / T3, -(Fx * Fx + reg) / T3, 0.0, 0.0, 0.0, 0.0};
      for (int r = 0; r < 3; ++r) { S[LQ_V_R + 3 * k + r] += p1 * dh[r]; for (int c = 0; c < 3; ++c) Rt[(3 * k + r) * QM_LD + 3 * k + c] += p2 * dh[r] * dh[c] + p1 * ddh[3 * r + c]; }
      ds = p1 * (-shift);
    }
    S[LQ_V_RED + 4 + k] = ds;
  } else if (tid == 96) { for (int k = 0; k < 6; ++k) cost += 0.5 * EE[6 + k] * EE[k] * EE[k]; }
  __syncthreads();
  { const double dsum = S[LQ_V_RED + 4] + S[LQ_V_RED + 5] + S[LQ_V_RED + 6] + S[LQ_V_RED + 7]; if (tid < 30) { Rt[tid * QM_LD + tid] += dsum; Qt[tid * QM_LD + tid] += dsum; } }
  __syncthreads();
  // EE pose soft constraint (a5): Q += J^T mu J, q += J^T mu g ; then scale by dt
  for (int idx = tid; idx < 900; idx += blockDim.x) { const int r = idx / 30, c = idx - r * 30; double s = 0.0; for (int k = 0; k < 6; ++k) s += JEE[r * 6 + k] * EE[6 + k] * JEE[c * 6 + k]; Qt[r * QM_LD + c] = (Qt[r * QM_LD + c] + s) * dt; Rt[r * QM_LD + c] *= dt; }
  if (tid < 30) { double s = 0.0; for (int k = 0; k < 6; ++k) s += JEE[tid * 6 + k] * EE[6 + k] * EE[k]; S[LQ_V_Q + tid] = (S[LQ_V_Q + tid] + s) * dt; S[LQ_V_R + tid] *= dt; }
  const double ctot = block_sum(cost, S + LQ_V_RED) * dt;
  if (tid == 0) { a.perf[nb * PF_SIZE] = ctot; a.perf[nb * PF_SIZE + 1] = dt * b2; a.perf[nb * PF_SIZE + 2] = dt * eq2; }
  if (dbg) { tile_store(Qt, dbg + LQ_DBG_Q, 30, 30, 30); tile_store(Rt, dbg + LQ_DBG_R, 30, 30, 30); if (tid < 30) { dbg[LQ_DBG_q + tid] = S[LQ_V_Q + tid]; dbg[LQ_DBG_r + tid] = S[LQ_V_R + tid]; } if (tid == 0) dbg[LQ_DBG_c] = ctot; }
  // ---- projected cost ----
  if (tid >= 64 && tid < 94) { const int r = tid - 64; S[LQ_V_RR + r] = S[LQ_V_R + r] + tile_row_dot(Rt, r, Pe, 30); }
  // R Px: only rows 12..23 of R[:, 12:24] are non-zero -> 16-row result (rows 12..27), k slabs 3..5
  wg_gemm<false, false>(Rt + 12 * QM_LD, Px, 1, 2, 3, 6, [&](int r, int c, double v) { RPxs[r * QM_LD + c] = v; });
  // (R Pu)^T = Pu^T R  (m x 30)
  wg_gemm<false, false>(PuT, Rt, mt_m, 2, 0, 8, [&](int r, int c, double v) { if (r < m) RPuT[r * QM_LD + c] = v; });
  __syncthreads();
  wg_gemm<true, false>(Px, RPx, 2, 2, 3, 6, [&](int r, int c, double v) { if (r < 30 && c < 30) rec[SR_QP + r * 30 + c] = Qt[r * QM_LD + c] + v; });
  wg_gemm<false, false>(PuT, RPx, mt_m, 2, 3, 6, [&](int r, int c, double v) { if (r < m && c < 30) rec[SR_PP + r * 30 + c] = v; });
  wg_gemm<false, true>(PuT, RPuT, mt_m, mt_m, 0, 8, [&](int r, int c, double v) { if (r < m && c < m) rec[SR_RP + r * QM_MMAX + c] = v; });
  if (tid < 30) { double s = 0.0; for (int r = 12; r < 24; ++r) s += Px[r * QM_LD + tid] * S[LQ_V_RR + r]; rec[SR_QPV + tid] = S[LQ_V_Q + tid] + s; }
  if (tid >= 64 && tid < 64 + m) rec[SR_RPV + tid - 64] = tile_row_dot(PuT, tid - 64, S + LQ_V_RR, 30);
  if (tid >= 128 && tid < 158) rec[SR_PE + tid - 128] = Pe[tid - 128];
  for (int idx = tid; idx < 900; idx += blockDim.x) { const int r = idx / 30, c = idx - r * 30; rec[SR_PX + idx] = (r >= 12 && r < 24) ? Px[r * QM_LD + c] : 0.0; }
  for (int idx = tid; idx < 30 * m; idx += blockDim.x) { const int r = idx / m, c = idx - r * m; rec[SR_PU + r * QM_MMAX + c] = PuT[c * QM_LD + r]; }
  const double rpe = block_sum((tid < 30) ? (S[LQ_V_R + tid] + 0.5 * (S[LQ_V_RR + tid] - S[LQ_V_R + tid])) * Pe[tid] : 0.0, S + LQ_V_RED);
  if (tid == 0) { rec[SR_SCAL] = (double)m; rec[SR_SCAL + 1] = ctot + rpe; }
}
