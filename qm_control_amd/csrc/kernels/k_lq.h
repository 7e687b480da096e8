// k_lq.h — K1: per-shooting-node linear-quadratic approximation + equality-constraint projection.
//
// One 256-thread workgroup (4 wavefronts) per (instance b, node i).  Restates, MI355X-first, what
// [upstream ocs2_sqp multiple_shooting::setupIntermediateNode + projectTranscription] do per node for the
// OCP of qm_interface/src/QMInterface.cpp:79-142 (SURVEY.md §8 a2–a8, a11; Appendix B.6 steps 2–3):
//   phase A  kinematics of 5 chains on 5 lanes (LDS workspace), twice (Heun/RK2 stages)
//   phase B  analytic Jacobian columns of the flow map, one lane per column, into LDS tiles
//   phase C  RK2 sensitivity composition   A_d = I + dt/2 (A1 + A2 + dt A2 A1),  B_d likewise  (f64 MFMA)
//   phase D  cost quadratic model (tracking + arm soft box + friction-cone barrier + EE pose), × dt
//   phase E  equality rows (zero force / zero foot velocity / swing normal velocity) and their
//            closed-form block projection  du = Pe + Px dx + Pu ut   (D is block structured by construction:
//            each row touches one foot's force triple or one leg's joint-velocity triple)
//   phase F  projected stage  Ap, Bp, Qp, Pp, Rp, ...  (f64 MFMA) streamed to the HBM stage record
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

// LDS carve (doubles): 7 tiles + vector area
#define LQ_T(n) ((n) * QM_TILE)
#define LQ_VEC (7 * QM_TILE)
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
#define LQ_V_EE   (LQ_VEC + 456)    /* g(6), mu g(6), qee(4), ref(7) */
#define LQ_V_K1   (LQ_VEC + 480)
#define LQ_V_K2   (LQ_VEC + 480 + KW_SIZE)
#define LQ_V_COLMAP (LQ_VEC + 480 + 2 * KW_SIZE)   /* 32 ints as doubles: Pu column -> (row, kind) */
#define LQ_LDS_DOUBLES (LQ_VEC + 480 + 2 * KW_SIZE + 40)
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

__device__ __forceinline__ void lq_kinematics(const double* mb, const double* x, const double* u, double* K) {
  if (threadIdx.x == 0) kin_base(mb, x, K);
  __syncthreads();
  if (threadIdx.x < 4) kin_leg(mb, threadIdx.x, x, u, K);
  else if (threadIdx.x == 4) kin_arm(mb, x, K);
  __syncthreads();
}

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
  const double dt = a.node_dt[nb];
  const int mode = a.node_mode[nb];

  // ---- P0: stage inputs in LDS, clear tiles ----
  tile_zero(S, LQ_LDS_DOUBLES);
  __syncthreads();
  if (tid < 30) { S[LQ_V_X + tid] = a.x[nb * 30 + tid]; S[LQ_V_U + tid] = terminal ? 0.0 : a.u[nb * 30 + tid]; S[LQ_V_XN + tid] = terminal ? 0.0 : a.x[((i + 1) * a.B + b) * 30 + tid]; }
  if (tid >= 32 && tid < 39) S[LQ_V_EE + 16 + (tid - 32)] = a.eeref[nb * 7 + (tid - 32)];
  __syncthreads();
  double* X = S + LQ_V_X; double* U = S + LQ_V_U;
  double* K1 = S + LQ_V_K1; double* K2 = S + LQ_V_K2;

  if (terminal) {
    // final EE soft constraint only: Q_N = Jᵀ mu J, q_N = Jᵀ mu g, c_N = ½ g mu g
    lq_kinematics(mb, X, nullptr, K1);
    double* EE = S + LQ_V_EE;
    if (tid == 0) { ee_error(K1, EE + 16, EE + 19, EE + 12, EE); for (int r = 0; r < 6; ++r) EE[6 + r] = (r < 3 ? st[ST_MU_EEF_POS] : st[ST_MU_EEF_ORI]); }
    __syncthreads();
    double* J = S + LQ_T(0);                           // J stored transposed: J[c][r] (30 x 6)
    if (tid < 30) { double col[6]; ee_jac_col(X, K1, EE + 12, EE + 19, tid, col); for (int r = 0; r < 6; ++r) J[tid * QM_LD + r] = col[r]; }
    __syncthreads();
    for (int idx = tid; idx < 900; idx += blockDim.x) { const int r = idx / 30, c = idx - r * 30; double s = 0.0; for (int k = 0; k < 6; ++k) s += J[r * QM_LD + k] * EE[6 + k] * J[c * QM_LD + k]; rec[SR_QP + idx] = s; }
    if (tid < 30) { double s = 0.0; for (int k = 0; k < 6; ++k) s += J[tid * QM_LD + k] * EE[6 + k] * EE[k]; rec[SR_QPV + tid] = s; }
    if (tid == 0) { double c = 0.0; for (int k = 0; k < 6; ++k) c += 0.5 * EE[6 + k] * EE[k] * EE[k]; rec[SR_SCAL] = 0.0; rec[SR_SCAL + 1] = c; a.perf[nb * PF_SIZE] = c; a.perf[nb * PF_SIZE + 1] = 0.0; a.perf[nb * PF_SIZE + 2] = 0.0; }
    return;
  }

  // ---- P1/P2: Heun stages ----
  lq_kinematics(mb, X, U, K1);
  if (tid == 0) flow_from_kin(mb, X, U, K1, S + LQ_V_F1);
  __syncthreads();
  if (tid < 30) S[LQ_V_X2 + tid] = X[tid] + dt * S[LQ_V_F1 + tid];
  __syncthreads();
  lq_kinematics(mb, S + LQ_V_X2, U, K2);
  if (tid == 0) flow_from_kin(mb, S + LQ_V_X2, U, K2, S + LQ_V_F2);
  // ---- P3: Jacobian columns: lanes 0..59 stage 1, lanes 64..123 stage 2 ----
  double* A1 = S + LQ_T(0); double* B1 = S + LQ_T(1); double* A2 = S + LQ_T(2); double* B2 = S + LQ_T(3); double* T4 = S + LQ_T(4);
  {
    const int which = tid >> 6, c = tid & 63;
    if (which < 2 && c < 60) {
      double col[12]; flow_jac_col(mb, which ? S + LQ_V_X2 : X, U, which ? K2 : K1, c, col);
      double* M = (c < 30) ? (which ? A2 : A1) : (which ? B2 : B1); const int cc = (c < 30) ? c : c - 30;
      for (int r = 0; r < 12; ++r) M[r * QM_LD + cc] = col[r];
      if (c >= 42) M[(c - 30) * QM_LD + cc] = 1.0;      // d qdot_j / d u_j
    }
  }
  __syncthreads();
  // ---- P4: RK2 sensitivities ----
  wg_gemm<false, false>(A2, A1, 2, 2, 0, 8, [&](int r, int c, double v) { T4[r * QM_LD + c] = v; });
  __syncthreads();
  for (int idx = tid; idx < 900; idx += blockDim.x) { const int r = idx / 30, c = idx - r * 30; const int o = r * QM_LD + c; A1[o] = 0.5 * dt * A1[o] + 0.5 * dt * (A2[o] + dt * T4[o]) + (r == c ? 1.0 : 0.0); }
  __syncthreads();
  wg_gemm<false, false>(A2, B1, 2, 2, 0, 8, [&](int r, int c, double v) { T4[r * QM_LD + c] = v; });
  __syncthreads();
  for (int idx = tid; idx < 900; idx += blockDim.x) { const int r = idx / 30, c = idx - r * 30; const int o = r * QM_LD + c; B1[o] = 0.5 * dt * B1[o] + 0.5 * dt * (B2[o] + dt * T4[o]); }
  if (tid < 30) S[LQ_V_B + tid] = X[tid] + 0.5 * dt * S[LQ_V_F1 + tid] + 0.5 * dt * S[LQ_V_F2 + tid] - S[LQ_V_XN + tid];
  __syncthreads();
  double* Ad = A1; double* Bd = B1;                       // tiles 0,1 ; tiles 2,3,4 free
  if (dbg) { tile_store(Ad, dbg + LQ_DBG_A, 30, 30, 30); tile_store(Bd, dbg + LQ_DBG_B, 30, 30, 30); if (tid < 30) dbg[LQ_DBG_b + tid] = S[LQ_V_B + tid]; }

  // ---- P5: cost quadratic model (× dt) : Q -> tile 2, R -> tile 3 ----
  double* Qt = S + LQ_T(2); double* Rt = S + LQ_T(3);
  tile_zero(Qt); tile_zero(Rt); tile_zero(T4);
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
  }
  __syncthreads();
  if (tid == 0) {   // friction cone: serial over stance feet (touches the whole diagonal through the Hessian shift)
    const double mu = st[ST_FRIC_MU], de = st[ST_FRIC_DELTA], muf = st[ST_FRIC_COEF], reg = st[ST_FRIC_REG], shift = st[ST_FRIC_SHIFT];
    double dsum = 0.0;
    for (int k = 0; k < 4; ++k) if (mode_flag(mode, k)) {
      const double Fx = U[3 * k], Fy = U[3 * k + 1], Fz = U[3 * k + 2]; const double T2 = Fx * Fx + Fy * Fy + reg, Tn = sqrt(T2), T3 = Tn * Tn * Tn;
      const double h = muf * Fz - Tn; cost += barrier_val(mu, de, h);
      const double p1 = barrier_d1(mu, de, h), p2 = barrier_d2(mu, de, h);
      const double dh[3] = {-Fx / Tn, -Fy / Tn, muf};
      const double ddh[9] = {-(Fy * Fy + reg) / T3, Fx * Fy / T3, 0.0, Fx * Fy / T3, -(Fx * Fx + reg) / T3, 0.0, 0.0, 0.0, 0.0};
      for (int r = 0; r < 3; ++r) { S[LQ_V_R + 3 * k + r] += p1 * dh[r]; for (int c = 0; c < 3; ++c) Rt[(3 * k + r) * QM_LD + 3 * k + c] += p2 * dh[r] * dh[c] + p1 * ddh[3 * r + c]; }
      dsum += p1 * (-shift);
    }
    S[LQ_V_RED + 7] = dsum;
  }
  __syncthreads();
  if (tid < 30) { Rt[tid * QM_LD + tid] += S[LQ_V_RED + 7]; Qt[tid * QM_LD + tid] += S[LQ_V_RED + 7]; }
  // EE pose soft constraint (a5): g, J (as Jᵀ in tile 4: [c][r]), Q += Jᵀ mu J, q += Jᵀ mu g
  double* EE = S + LQ_V_EE;
  if (tid == 32) { ee_error(K1, EE + 16, EE + 19, EE + 12, EE); for (int r = 0; r < 6; ++r) EE[6 + r] = (r < 3 ? st[ST_MU_EE_POS] : st[ST_MU_EE_ORI]); }
  __syncthreads();
  if (tid < 30) { double col[6]; ee_jac_col(X, K1, EE + 12, EE + 19, tid, col); for (int r = 0; r < 6; ++r) T4[tid * QM_LD + r] = col[r]; }
  if (tid == 32) { for (int k = 0; k < 6; ++k) cost += 0.5 * EE[6 + k] * EE[k] * EE[k]; }
  __syncthreads();
  for (int idx = tid; idx < 900; idx += blockDim.x) { const int r = idx / 30, c = idx - r * 30; double s = 0.0; for (int k = 0; k < 6; ++k) s += T4[r * QM_LD + k] * EE[6 + k] * T4[c * QM_LD + k]; Qt[r * QM_LD + c] += s; }
  if (tid < 30) { double s = 0.0; for (int k = 0; k < 6; ++k) s += T4[tid * QM_LD + k] * EE[6 + k] * EE[k]; S[LQ_V_Q + tid] += s; }
  __syncthreads();
  // scale by dt
  for (int idx = tid; idx < 900; idx += blockDim.x) { const int r = idx / 30, c = idx - r * 30; Qt[r * QM_LD + c] *= dt; Rt[r * QM_LD + c] *= dt; }
  if (tid < 30) { S[LQ_V_Q + tid] *= dt; S[LQ_V_R + tid] *= dt; }
  const double ctot = block_sum(cost, S + LQ_V_RED) * dt;
  if (dbg) { tile_store(Qt, dbg + LQ_DBG_Q, 30, 30, 30); tile_store(Rt, dbg + LQ_DBG_R, 30, 30, 30); if (tid < 30) { dbg[LQ_DBG_q + tid] = S[LQ_V_Q + tid]; dbg[LQ_DBG_r + tid] = S[LQ_V_R + tid]; } if (tid == 0) dbg[LQ_DBG_c] = ctot; }

  // ---- P6: equality rows + closed-form block projection ----
  // rows ordered per contact i = LF,RF,LH,RH: swing -> [F_i = 0 (3)] , stance -> [v_i = 0 (3)] , swing -> [v_iz = zvel_ref (1)]
  double* Ct = S + LQ_T(4); double* Dt = S + LQ_T(5); double* Px = S + LQ_T(6);
  __syncthreads();
  tile_zero(Ct); tile_zero(Dt); tile_zero(Px);
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
  // per contact: stance -> Ginv (3x3) of the joint-velocity block; swing -> g/(g·g) and a 3x2 orthonormal complement of g
  double* G = S + LQ_V_G;
  if (tid < 4) {
    const int k = tid, ch = contact_to_chain(k), jc = 12 + 3 * ch; double* g = G + 12 * k;
    if (mode_flag(mode, k)) { double M3[9]; for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) M3[3 * r + c] = Dt[(row0[k] + r) * QM_LD + jc + c]; m3_inv(M3, g); }
    else {
      const double gv[3] = {Dt[(row0[k] + 3) * QM_LD + jc], Dt[(row0[k] + 3) * QM_LD + jc + 1], Dt[(row0[k] + 3) * QM_LD + jc + 2]};
      const double n2 = gv[0] * gv[0] + gv[1] * gv[1] + gv[2] * gv[2], nrm = sqrt(n2);
      for (int r = 0; r < 3; ++r) g[r] = gv[r] / n2;
      // Householder H = I − 2 v vᵀ/(vᵀv), v = g − alpha e1, alpha = −sign(g0)|g| : H e1 ∥ g, columns 2,3 of H span g^⊥
      const double alpha = gv[0] > 0.0 ? -nrm : nrm; const double v[3] = {gv[0] - alpha, gv[1], gv[2]}; const double vv = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
      for (int r = 0; r < 3; ++r) { g[3 + r] = ((r == 1) ? 1.0 : 0.0) - 2.0 * v[r] * v[1] / vv; g[6 + r] = ((r == 2) ? 1.0 : 0.0) - 2.0 * v[r] * v[2] / vv; }
    }
  }
  __syncthreads();
  // Px rows (inputs 12..23), Pe, and the Pu column map
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
  const double eq2 = block_sum((tid < nc) ? S[LQ_V_E + tid] * S[LQ_V_E + tid] : 0.0, S + LQ_V_RED);
  const double b2 = block_sum((tid < 30) ? S[LQ_V_B + tid] * S[LQ_V_B + tid] : 0.0, S + LQ_V_RED);
  if (tid == 0) { a.perf[nb * PF_SIZE] = ctot; a.perf[nb * PF_SIZE + 1] = dt * b2; a.perf[nb * PF_SIZE + 2] = dt * eq2; }
  // Pu (30 x m) into tile 4 (C no longer needed), columns: stance forces, swing-leg null spaces, arm
  __syncthreads();
  double* PuT = S + LQ_T(4);
  tile_zero(PuT);
  __syncthreads();
  int m = 0;
  {
    int col = 0;
    for (int k = 0; k < 4; ++k) if (mode_flag(mode, k)) { if (tid < 3) PuT[(3 * k + tid) * QM_LD + col + tid] = 1.0; col += 3; }
    for (int k = 0; k < 4; ++k) if (!mode_flag(mode, k)) { const int jc = 12 + 3 * contact_to_chain(k); const double* g = G + 12 * k; if (tid < 6) { const int r = tid % 3, cc = tid / 3; PuT[(jc + r) * QM_LD + col + cc] = g[3 + 3 * cc + r]; } col += 2; }
    if (tid < 6) PuT[(24 + tid) * QM_LD + col + tid] = 1.0; col += 6;
    m = col;
  }
  __syncthreads();
  const int mt_m = (m + 15) / 16;
  // ---- P7: projected dynamics  Ap = Ad + Bd Px ; Bp = Bd Pu ; bp = b + Bd Pe ----
  wg_gemm<false, false>(Bd, Px, 2, 2, 3, 6, [&](int r, int c, double v) { if (r < 30 && c < 30) rec[SR_AP + r * 30 + c] = Ad[r * QM_LD + c] + v; });
  wg_gemm<false, false>(Bd, PuT, 2, mt_m, 0, 8, [&](int r, int c, double v) { if (r < 30 && c < m) rec[SR_BP + r * QM_MMAX + c] = v; });
  if (tid < 30) rec[SR_BPV + tid] = S[LQ_V_B + tid] + tile_row_dot(Bd, tid, Pe, 30);
  // ---- P8: projected cost ----
  if (tid >= 64 && tid < 94) { const int r = tid - 64; S[LQ_V_RR + r] = S[LQ_V_R + r] + tile_row_dot(Rt, r, Pe, 30); }
  __syncthreads();                                        // all reads of Ad/Bd done -> tiles 0,1 reusable
  double* RPx = S + LQ_T(0); double* RPu = S + LQ_T(1);
  wg_gemm<false, false>(Rt, Px, 2, 2, 3, 6, [&](int r, int c, double v) { RPx[r * QM_LD + c] = v; });
  wg_gemm<false, false>(Rt, PuT, 2, mt_m, 0, 8, [&](int r, int c, double v) { RPu[r * QM_LD + c] = v; });
  __syncthreads();
  wg_gemm<true, false>(Px, RPx, 2, 2, 3, 6, [&](int r, int c, double v) { if (r < 30 && c < 30) rec[SR_QP + r * 30 + c] = Qt[r * QM_LD + c] + v; });
  wg_gemm<true, false>(PuT, RPx, mt_m, 2, 0, 8, [&](int r, int c, double v) { if (r < m && c < 30) rec[SR_PP + r * 30 + c] = v; });
  wg_gemm<true, false>(PuT, RPu, mt_m, mt_m, 0, 8, [&](int r, int c, double v) { if (r < m && c < m) rec[SR_RP + r * QM_MMAX + c] = v; });
  if (tid < 30) rec[SR_QPV + tid] = S[LQ_V_Q + tid] + tile_col_dot(Px, tid, S + LQ_V_RR, 30);
  if (tid >= 64 && tid < 64 + m) rec[SR_RPV + tid - 64] = tile_col_dot(PuT, tid - 64, S + LQ_V_RR, 30);
  if (tid >= 128 && tid < 158) rec[SR_PE + tid - 128] = Pe[tid - 128];
  tile_store(Px, rec + SR_PX, 30, 30, 30);
  for (int idx = tid; idx < 30 * m; idx += blockDim.x) { const int r = idx / m, c = idx - r * m; rec[SR_PU + r * QM_MMAX + c] = PuT[r * QM_LD + c]; }
  const double rpe = block_sum((tid < 30) ? (S[LQ_V_R + tid] + 0.5 * (S[LQ_V_RR + tid] - S[LQ_V_R + tid])) * Pe[tid] : 0.0, S + LQ_V_RED);
  if (tid == 0) { rec[SR_SCAL] = (double)m; rec[SR_SCAL + 1] = ctot + rpe; }
}
