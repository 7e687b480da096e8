// k_lq.h — K1: per-shooting-node linear-quadratic approximation + equality-constraint projection.
//
// One WAVEFRONT per (instance b, node i).  Restates, MI355X-first, what
// [upstream ocs2_sqp multiple_shooting::setupIntermediateNode + projectTranscription] do per node for the
// OCP of qm_interface/src/QMInterface.cpp:79-142 (SURVEY.md §8 a2–a8, a11; Appendix B.6 steps 2–3):
//   K1a qm_lq_kin_kernel (one THREAD per node): all scalar kinematics — both Heun/RK2 stages, flow values, EE pose
//       error — written as a 4 KB "kin record" per node (lanes = instances: no idle lanes, no barriers)
//   K1b qm_lq_kernel / qm_lq_m18_kernel (one WAVEFRONT per node, 64-thread workgroups, 13.1 KB LDS, 168 registers: three waves per SIMD, twelve per CU): every matrix lives in the wave's registers as
//       f64-MFMA D-fragments (qm_dev_common.h), all products are P = Zᵀ Y chains, the vectors ride in column 30 of the 32-wide
//       tiles, and LDS is only the hand-over point between the lane-per-column analytic Jacobians and the fragments.  No
//       workgroup barrier anywhere: the 100k nodes of a batch are independent waves.
//   Order of the phases (round 4 — chosen so that nothing large is live across the two lane-per-column Jacobian evaluations and the kernel fits 168 registers without
//   scratch, i.e. THREE waves per SIMD; rounds 1–3 ran I before II with A_d, B_dᵀ live across II at 216 registers):
//   P0        inputs and the kin record -> LDS; what the prologue's loads are needed for (defect b, tracking terms of the cost, u − u_nom) is formed at once
//   phase II  equality rows (zero force / zero foot velocity / swing normal velocity) and their closed-form block
//             projection du = Pe + Px dx + Pu ut (D is block structured by construction: each row touches one foot's
//             force triple or one leg's joint-velocity triple): per-contact blocks G, the twelve non-zero rows of Px, Pe, the column descriptors of Pu
//   phase I   analytic Jacobian columns of the flow map, one lane per column (divergent class bodies: 126 registers against the 166 of the interleaved form).  df/dx and
//             df/du only have 12 (+4 identity) non-trivial rows (SRBD); RK2 sensitivity composition A_d = I + dt/2 (A1 + A2 + dt A2 A1) and B_d likewise, B_d turned into
//             B_dᵀ through the tile (that is the operand form B_d Px and B_d Pu need), k restricted to the 16 live rows
//   then      projected dynamics Ap, Bp, bp streamed to HBM — the last use of A_d, B_dᵀ
//   phase III cost quadratic model (tracking + arm soft box + friction-cone barrier + EE pose), x dt, and the projected
//             cost Qp, Pp, Rp, qp, rp (f64 MFMA, k restricted to the 12 rows Px / R Px occupy) streamed to HBM
//   The body is instantiated per tile count MT of the reduced inputs and runs as TWO product kernels (qm_lq_kernel: m <= 16, qm_lq_m18_kernel: m = 17, 18): one body with a
//   branch between the two projections at its end made the register allocator carry both paths' worst cases (216 registers).
// The terminal node only carries the final EE soft constraint (QMInterface.cpp:104).
#pragma once
#include "qm_dev_kin.h"
#ifndef QM_LQ_RB_ONLY
#define QM_LQ_RB_ONLY 0      /* instruction counting only (tools/isa_hist.py): 1 compiles K1b without the dense R0 path, i.e. the instruction stream a wave executes with the shipped task file */
#endif

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
  int prof;                  // profiling only: thread 0 leaves phase cycle stamps in the (unused) SR_K field of the record
  int ncap;                  // K1b: nodes per instance covered by the launch (the batch's largest node count, <= nmax; with node slices: the slice length)
  int i0;                    // first node of the launch (node slices: K1a and K1b take the horizon in [i0, i0 + ncap) pieces so that a piece's kin records are consumed while they are still cached)
  // interior-point instances only (k_ipm.h): slack / dual of the node's QM_NH inequality rows [nmax][B][QM_NH], barrier parameter per instance info[b * 8]
  const double* ipm_s; const double* ipm_l; const double* ipm_info;
  int single_mt;             // K1b product kernels: the host knows that NO node of the launch has more than 16 reduced inputs (K0 publishes it with the node capacity): qm_lq_kernel then takes every node without
                             // first loading its contact mode to decide whether the node is its own — a memory round trip of its own in front of everything else (qm_lq_m18_kernel is not launched)
  int rb;                    // K1b: the input weight R0 of the settings table is block diagonal (diag(12) + four 3 x 3 leg blocks + diag(6): qm_r_is_block_diagonal, k_ls.h — the host checks the table entry by entry)
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
#define LQ_DBG_PU 4668       /* [30][18] the null-space basis Pu of this design (K3 rebuilds Pu ut from the contact mode and the swing blocks: the product never stores it) */
#define LQ_DBG_SIZE 5208

// kin record (doubles) per node, in the order K1a produces it (its stores stream through the record front to back).  Round 5: the second stage's state x2 is no longer
// stored (nobody read it), the flow values keep their twelve non-trivial rows only (rows 12..29 are the input's joint velocities, which K1b holds anyway) and the
// second stage's workspace stops in front of the arm block (the arm's pose is a stage-1 quantity): 504 -> 384 doubles = 48 cache lines per node
#define KR_K1   0
#define KR_EEG  KW_SIZE               /* g(6)  */
#define KR_QEE  (KR_EEG + 6)          /* qee(4) */
#define KR_F1   (KR_QEE + 4)          /* rows 0..11 of f(x, u) */
#define KR_K2   (KR_F1 + 12)          /* base + leg blocks of the second Heun stage (KW_ARM doubles) */
#define KR_F2   (KR_K2 + KW_ARM)      /* rows 0..11 of f(x + dt f1, u) */
#define KR_USED (KR_F2 + 12)
#define KR_SIZE 384
static_assert(KR_USED + 7 <= KR_SIZE && KR_SIZE % 8 == 0, "kin record: whole 64-byte lines, room for the last piece's padding");

// LDS carve (doubles) of one wave
#define LW_BLOCK 64
#define LW_TLD 33                   /* odd: the transposed fragment reads (lanes run down a column, stride LW_TLD doubles) then hit 16 different bank pairs; 34 made them 2-way conflicts (K1b − 0.5 %) */
#define LW_T     0                    /* [32][LW_TLD] hand-over tile (columns from lanes -> fragments) */
#define LW_V     (32 * LW_TLD)
#define LW_V_X   (LW_V + 0)
#define LW_V_U   (LW_V + 32)
#define LW_V_B   (LW_V + 64)          /* b (slots 30, 31 zero) */
#define LW_V_RV  LW_V_B               /* r.  ALIASES b: the defect is dead once the projected dynamics are stored, r is formed after that (cost model); its slots 30, 31 are b's zeros */
#define LW_V_E   (LW_V + 96)          /* e(16) */
#define LW_V_PE  (LW_V + 112)         /* Pe(32) */
#define LW_V_G   (LW_V + 144)         /* per contact: Ginv or g data (4 x 12) */
#define LW_V_EE  (LW_V + 192)         /* g(6) mu(6) qee(4) ref(7) */
#define LW_V_QV  (LW_V + 216)         /* q */
#define LW_V_RR  (LW_V + 248)         /* r + R Pe (before the projection: the lanes' tracking cost) */
#define LW_V_SIZE 280
/* what only the cost model (phase III) reads and writes lives in the hand-over tile, which nobody touches between Bp (lw_bp) and the projection (lw_project) — the twelfth wave per CU */
#define LW3_QD   (LW_T + 0)           /* diagonal additions of Q (32) */
#define LW3_RD   (LW_T + 32)          /* diagonal additions of R (32) */
#define LW3_DU   (LW_T + 64)          /* u − unom (32; slots 30, 31 zero) */
#define LW3_FR   (LW_T + 96)          /* friction cone terms per contact: p2 dh dhᵀ + p1 ddh (9), zeros (3), ds (1) -> 4 x 16 */
#define LW3_JT   (LW_T + 160)         /* [6][32] rows of the end-effector Jacobian */
#define LW_K1    (LW_V + LW_V_SIZE)
#define LW_K2    (LW_K1 + KW_SIZE)     /* base + leg blocks of the second Heun stage: KW_ARM doubles (the arm block belongs to the node's own state only) */
#define LW_K2SZ  148
#define LW_PD    LW_K2                  /* Pu column descriptors: first source row i0 as double [32], weights [32][3].  ALIAS the stage-2 workspace: written behind phase I, its last reader */
#define LW_LDS_DOUBLES (LW_K2 + LW_K2SZ)
#define LQ_LDS_BYTES (LW_LDS_DOUBLES * 8)
static_assert(KW_ARM <= LW_K2SZ && 128 <= LW_K2SZ && LW_K2SZ % 2 == 0 && LW3_JT + 6 * 32 <= 32 * LW_TLD && LW_V_SIZE % 2 == 0, "aliases of the stage-2 kin workspace and of the hand-over tile");
static_assert(12 * ((LQ_LDS_BYTES + 255) / 256 * 256) <= 160 * 1024, "twelve waves per CU (160 KB of LDS; three per SIMD is the register file's limit): round 6 — with the LDS accesses un-paired the kernel gains from every wave (10 -> 11 -> 12: 0.931 -> 0.871 -> 0.831 ms, measured with an aliased layout before this one was built)");
#define LQ_KIN_TILE (64 * 31)               /* K1a: [64][31] rows — the wave's inputs x (transposed on the way in), then the input u of each thread for the whole kernel ... */
#define LQ_KIN_LDS_BYTES ((LQ_KIN_TILE + 64 * 9) * 8)      /* ... + the [64][9] hand-over tile of the record stores: 20 KB per wave, seven waves per CU (the benchmark launch has 6.45 per CU) */

// value v[row] placed in column 30 (tile J = 1, lane column 14) of a two-tile-high fragment column.  (One lane-conditional region; the forms without one — an unconditional read
// pinned by an empty asm + a select, or an addition through a selected address — measured slower resp. cost more instructions here, profiles/r06_ab_lq_regions.log)
__device__ __forceinline__ void lw_set_col30(qm_d4 (&F)[2][2], const double* v) {
  const int g = (threadIdx.x & 63) >> 4, c = threadIdx.x & 15;
#pragma unroll
  for (int I = 0; I < 2; ++I)
#pragma unroll
    for (int r = 0; r < 4; ++r) { const int row = 16 * I + g + 4 * r; if (c == 14) F[I][1][r] = (row < 30) ? v[row] : 0.0; }
}
// column 30 of a fragment column -> v[row], rows < rows.  ONE lane-conditional region for all registers: RMIN <= rows <= RMAX are compile-time bounds of `rows`, a register
// whose rows all lie below RMIN is stored without a test, one whose rows all lie at or above RMAX is not stored at all
template <int IT, int RMIN, int RMAX>
__device__ __forceinline__ void lw_get_col30(const qm_d4 (&F)[IT][2], double* v, int rows) {
  const int g = (threadIdx.x & 63) >> 4, c = threadIdx.x & 15;
  if (c == 14) {
#pragma unroll
    for (int I = 0; I < IT; ++I)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * I + g + 4 * r;
        if (16 * I + 4 * r + 3 < RMIN) v[row] = F[I][1][r];
        else if (16 * I + 4 * r < RMAX) { if (row < rows) v[row] = F[I][1][r]; }
      }
  }
}

// zero fill of N doubles of LDS (N even, 16-byte aligned) with UNROLLED 16-byte stores at immediate offsets: a rolled `for (idx = l; idx < N; idx += 64)` loop of 8-byte
// stores spends six instructions per store on its counter, compare and branch (K1b cleared its tile twice and its vector area once per node that way: ≈ 250 of a wave's
// ≈ 5000 dynamic instructions)
template <int N>
__device__ __forceinline__ void lw_zero(double* base, int l) {
  static_assert(N % 2 == 0, "pairs of doubles");
  double2* b2 = (double2*)base;
#pragma unroll
  for (int t = 0; t < (N / 2) / 64; ++t) b2[l + 64 * t] = double2{0.0, 0.0};
  if ((N / 2) % 64 != 0) { if (l < (N / 2) % 64) b2[l + 64 * ((N / 2) / 64)] = double2{0.0, 0.0}; }
}
// projected cost + record stores; MT = tiles covering the m reduced inputs
// Bp = Bd Pu (rows 0..11 of the record; a joint row is dt Pu[j], K3 rebuilds it): Bp[row][j] = Σ_k w_k(j) Bdᵀ[i0(j) + k][row] — the tile holds Bdᵀ (rows = inputs).
// Formed right behind the projected dynamics, so that the discrete-time Jacobians are dead before the cost model is assembled (three waves per SIMD: 168 registers)
template <int MT>
__device__ __forceinline__ void lw_bp(double* S, double* rec, int m, const qm_d4 (&Bdt)[2][2]) {
  const int l = threadIdx.x & 63, g = l >> 4, c = l & 15;
  double* T = S + LW_T; const double* PD = S + LW_PD;
  qm_wave_sync();
#pragma unroll
  for (int I = 0; I < 2; ++I)
#pragma unroll
    for (int J = 0; J < 2; ++J)
#pragma unroll
      for (int r = 0; r < 4; ++r) T[(16 * I + g + 4 * r) * LW_TLD + 16 * J + c] = Bdt[I][J][r];
  qm_wave_sync();
  qm_d4 Bp[1][MT];                                             // rows 0..11 only
#pragma unroll
  for (int J = 0; J < MT; ++J) {
    const int j = 16 * J + c; const int ci0 = (int)PD[j]; const double w0 = PD[32 + 3 * j], w1 = PD[33 + 3 * j], w2 = PD[34 + 3 * j];
#pragma unroll
    for (int r = 0; r < 4; ++r) { const int row = g + 4 * r; const double* src = T + ci0 * LW_TLD + row; Bp[0][J][r] = w0 * src[0] + w1 * src[LW_TLD] + w2 * src[2 * LW_TLD]; }
  }
  qm_frag_store<1, MT, true>(Bp, rec + SR_BP, QM_MMAX, 12, m);
  qm_wave_sync();
}
template <int MT>
__device__ __forceinline__ void lw_project(double* S, double* rec, double* dbg_pu, int m, const qm_d4 (&PxA)[2][2], const qm_d4 (&PuF)[2][2], const qm_d4 (&Rm)[2][2], qm_d4 (&Qa)[2][2], double& rpe) {
  const int l = threadIdx.x & 63;
  qm_d4 Pu[2][MT];
#pragma unroll
  for (int I = 0; I < 2; ++I)
#pragma unroll
    for (int J = 0; J < MT; ++J) Pu[I][J] = PuF[I][J];
  if (dbg_pu) qm_frag_store<2, MT>(Pu, dbg_pu, QM_MMAX, 30, m);
  // Every column of Pu is a unit vector or three consecutive entries (a swing leg's null-space column): products with Pu are gathers of
  // columns / rows of the other factor, not matrix products.  The other factor goes through the LDS tile once and each lane picks the entries
  // its output elements need (descriptors in LW_PD): ≈ 40 LDS operations per product instead of 8..16 MFMAs of 64 cycles each.
  double* T = S + LW_T; const double* PD = S + LW_PD;
  const int g = l >> 4, c = l & 15;
  // descriptors are (re)read from LDS next to their use: holding them across the whole function costs more registers than this kernel has
  int ci0[MT]; double cw[MT][3];                               // columns j = 16 J + c this lane holds in (X Pu)
  auto load_col_desc = [&]() {
#pragma unroll
    for (int J = 0; J < MT; ++J) { const int j = 16 * J + c; ci0[J] = (int)PD[j]; cw[J][0] = PD[32 + 3 * j]; cw[J][1] = PD[33 + 3 * j]; cw[J][2] = PD[34 + 3 * j]; }
  };
  load_col_desc();
  auto tile_put = [&](const qm_d4 (&F)[2][2]) {                // fragments (zero padded) -> T[row][col]
#pragma unroll
    for (int I = 0; I < 2; ++I)
#pragma unroll
      for (int J = 0; J < 2; ++J)
#pragma unroll
        for (int r = 0; r < 4; ++r) T[(16 * I + g + 4 * r) * LW_TLD + 16 * J + c] = F[I][J][r];
  };
  // [R Px | R Pe + r]: Px rows 12..23 (k-steps 3..5); Pe also has rows 0..11 (column 30 only -> tile column 1, k-steps 0..2)
  qm_d4 RPx[2][2]; qm_frag_zero<2, 2>(RPx);
  qm_gemm_tn<2, 2, 2>(Rm, PxA, RPx, 3, 6, false);
  { qm_d4 Y1[2][1], P1[2][1];
#pragma unroll
    for (int I = 0; I < 2; ++I) { Y1[I][0] = PxA[I][1]; P1[I][0] = RPx[I][1]; }
#pragma unroll
    for (int r = 0; r < 3; ++r) Y1[0][0][r] = *((c == 14) ? S + LW_V_PE + g + 4 * r : S + LW_V_X + 30);      // (Pe in lane column 14, a zero elsewhere: read through a selected address, slot 30 of the state vector holds zero)
    qm_gemm_tn<2, 2, 1>(Rm, Y1, P1, 0, 3, false);
#pragma unroll
    for (int I = 0; I < 2; ++I) RPx[I][1] = P1[I][0]; }
  { const int g = l >> 4, c = l & 15; const double m14 = (c == 14) ? 1.0 : 0.0;      // (slots 30, 31 of r are zero: every lane reads its rows, column 14 adds them — no execution-mask region per register)
#pragma unroll
    for (int I = 0; I < 2; ++I)
#pragma unroll
      for (int r = 0; r < 4; ++r) RPx[I][1][r] = fma(m14, S[LW_V_RV + 16 * I + g + 4 * r], RPx[I][1][r]); }
  qm_wave_sync();
  lw_get_col30<2, 30, 30>(RPx, S + LW_V_RR, 30);
  qm_wave_sync();
  rpe = qm_wave_sum((l < 30) ? (S[LW_V_RV + l] + 0.5 * (S[LW_V_RR + l] - S[LW_V_RV + l])) * S[LW_V_PE + l] : 0.0);
  // [Qp | qp] = [Q | q] + Pxᵀ [R Px | rr]
  qm_gemm_tn<2, 2, 2>(PxA, RPx, Qa, 3, 6, false);
  double* const frag = rec + SR_FRAG + l;                              // fragment-order operands of K3's backward sweep: register r of a tile = one contiguous 512-byte row
  { // [Qp | qp] as it stands in the registers (column 30 carries qp; rows 30, 31 and column 31 are padding K3 never looks at).  Qp is symmetric and K3 forms only the
    // upper tiles of the value function: the lower-left tile (rows 16.., columns < 16) is not stored
#pragma unroll
    for (int I = 0; I < 2; ++I)
#pragma unroll
      for (int J = I; J < 2; ++J)
#pragma unroll
        for (int r = 0; r < 4; ++r) QM_STREAM_ST(frag + SR_F_QP + (4 * (I + J) + r) * 64, Qa[I][J][r]); }
  qm_wave_sync();
  lw_get_col30<2, 30, 30>(Qa, S + LW_V_QV, 30);
  // [Pp | rp] = Puᵀ [R Px | rr]: Pp[j][col] = Σ_k w_k(j) [R Px | rr][i0(j) + k][col]
  qm_wave_sync(); tile_put(RPx); qm_wave_sync();
  { qm_d4 Pp[MT][2];
#pragma unroll
    for (int I = 0; I < MT; ++I)
#pragma unroll
      for (int J = 0; J < 2; ++J)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int j = 16 * I + g + 4 * r; const double* src = T + (int)PD[j] * LW_TLD + 16 * J + c; Pp[I][J][r] = PD[32 + 3 * j] * src[0] + PD[33 + 3 * j] * src[LW_TLD] + PD[34 + 3 * j] * src[2 * LW_TLD]; }
    // [Pp | rp]: rows >= m are zero in the record (K3's Wᵀ W runs over whole k-steps); of the second tile row only rows 16..19 (register 0) can be live (m <= 18)
#pragma unroll
    for (int J = 0; J < 2; ++J) {
#pragma unroll
      for (int r = 0; r < 4; ++r) QM_STREAM_ST(frag + SR_F_PP + (4 * J + r) * 64, (g + 4 * r < m) ? Pp[0][J][r] : 0.0);
      if (MT == 2) QM_STREAM_ST(frag + SR_F_PP1 + J * 64, (16 + g < m) ? Pp[MT - 1][J][0] : 0.0);
    }
    lw_get_col30<MT, 14, 18>(Pp, S + LW_V_RV, m); }
  // Rp = Puᵀ (R Pu): first R Pu[i][j] = Σ_k w_k(j) R[i][i0(j) + k], then the rows of that by the same descriptors
  qm_wave_sync(); tile_put(Rm); qm_wave_sync();
  load_col_desc();
  { qm_d4 RPu[2][2];
#pragma unroll
    for (int I = 0; I < 2; ++I)
#pragma unroll
      for (int J = 0; J < 2; ++J)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          double v = 0.0;
          if (J < MT) { const double* src = T + (16 * I + g + 4 * r) * LW_TLD + ci0[J < MT ? J : 0]; v = cw[J < MT ? J : 0][0] * src[0] + cw[J < MT ? J : 0][1] * src[1] + cw[J < MT ? J : 0][2] * src[2]; }
          RPu[I][J][r] = v;
        }
    qm_wave_sync(); tile_put(RPu); qm_wave_sync();
    qm_d4 Rp[MT][MT];
#pragma unroll
    for (int I = 0; I < MT; ++I)
#pragma unroll
      for (int J = 0; J < MT; ++J)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int j = 16 * I + g + 4 * r; const double* src = T + (int)PD[j] * LW_TLD + 16 * J + c; Rp[I][J][r] = PD[32 + 3 * j] * src[0] + PD[33 + 3 * j] * src[LW_TLD] + PD[34 + 3 * j] * src[2 * LW_TLD]; }
    // Rp: zero outside [0, m) x [0, m) (K3 puts the unit diagonal of the padding rows itself); the lower-left tile is never read (symmetric)
#pragma unroll
    for (int r = 0; r < 4; ++r) QM_STREAM_ST(frag + SR_F_RP + r * 64, (g + 4 * r < m && c < m) ? Rp[0][0][r] : 0.0);
    if (MT == 2) {
#pragma unroll
      for (int r = 0; r < 4; ++r) QM_STREAM_ST(frag + SR_F_RP01 + r * 64, (g + 4 * r < m && 16 + c < m) ? Rp[0][MT - 1][r] : 0.0);
      QM_STREAM_ST(frag + SR_F_RP11, (16 + g < m && 16 + c < m) ? Rp[MT - 1][MT - 1][0] : 0.0);
    } }
  qm_wave_sync();
  if (l < 30) rec[SR_QPV + l] = S[LW_V_QV + l];
  if (l < m) rec[SR_RPV + l] = S[LW_V_RV + l];
}

// ---- K1a: scalar kinematics, one thread per (node, instance) ----
// Thread g = i B + b owns node i of instance b; the 64 threads of a wave own 64 CONSECUTIVE rows of the node-major arrays [nmax][B][.]: one contiguous block of the
// iterate (64 x 30 doubles) and one contiguous block of kin records (64 x KR_SIZE doubles).  Rounds 1-4 let every thread load its own row and store its own record
// with 16-byte accesses — each such instruction touches 64 different cache lines (lane stride 240 B / 4 KB): 272 stores x 64 = 17 k partial-line transactions per
// wave through the CU's one address / tag pipeline, 8 waves per CU, ≈ 140 k cycles of a 310 k-cycle kernel whose arithmetic needs ≈ 30 k per wave.  Round 5: the
// wave moves its blocks TOGETHER — lanes run over consecutive doubles of the block (full 64-byte lines), thread-private values change hands through LDS:
//   kin_rows_in    block of 64 x 30 inputs -> a [64][31] LDS tile (row r = thread r's vector)
//   kin_emit<N>    N values per thread (record doubles [off, off + N)) -> [64][9] tile -> lanes 8 r' + j store double j of an 8-double piece of record 8 k + r'
// There is no divergence left: every thread of the launch runs the full sequence on whatever its rows hold — PreEvent nodes, nodes behind an instance's last one and
// the terminal node's leg / second-stage parts produce records (or parts) nobody reads (K1b takes the terminal node's base, arm and end-effector error only); the
// kin buffer carries 64 records of slack behind the last row so the last wave of a launch needs no bounds either.
struct KinOut { double* tile; double* wave_rec; int l; };
__device__ __forceinline__ void kin_rows_in(double* tile, const double* src, size_t row0, size_t nrows, int l) {
  // 960 pairs of doubles (30 is even: a pair never straddles two rows), 16 bytes per lane: 1 KB = 16 full lines per instruction; three rounds of five (a rolled outer loop keeps
  // the fifteen row / column / address sets of the two calls from living across the kernel)
#pragma nounroll
  for (int t0 = 0; t0 < 15; t0 += 5) {
    double2 v[5]; int off[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) { const int e = (t0 + k) * 64 + l; const int r = e / 15, c = 2 * (e - r * 15); size_t row = row0 + r; if (row >= nrows) row = nrows - 1; off[k] = r * 31 + c; v[k] = *(const double2*)(src + row * 30 + c); }
#pragma unroll
    for (int k = 0; k < 5; ++k) { tile[off[k]] = v[k].x; tile[off[k] + 1] = v[k].y; }
  }
}
template <int N> __device__ __forceinline__ void kin_emit(const KinOut& o, int off, const double* v) {
#pragma unroll
  for (int p = 0; p < (N + 7) / 8; ++p) {
#pragma unroll
    for (int j = 0; j < 8; ++j) o.tile[o.l * 9 + j] = (8 * p + j < N) ? v[8 * p + j < N ? 8 * p + j : 0] : 0.0;
    qm_wave_sync();
    const int j = o.l & 7, rr = o.l >> 3;
    // every piece is stored WHOLE (no exec masks: their save / restore pairs were this kernel's scalar-register spills): the zeros behind a block's last value land on the
    // first doubles of the block that follows in the record — which this wave writes later, in program order — or on the record's padding
#pragma unroll
    for (int k = 0; k < 8; ++k) { const int r = 8 * k + rr; o.wave_rec[(size_t)r * KR_SIZE + off + 8 * p + j] = o.tile[r * 9 + j]; }
    qm_wave_sync();
  }
}
#ifndef QM_LQ_ONLY_K1B      /* (qmhip_lq.hip, the translation unit of the K1b instances: K1a stays in the main one) */
__global__ void QM_UNPAIRED_LDS __launch_bounds__(64, 2) qm_lq_kin_kernel(QmLqArgs a) {
  const int l = threadIdx.x & 63;
  const size_t g0 = (size_t)a.i0 * a.B + (size_t)blockIdx.x * 64, nrows = (size_t)a.nmax * a.B;      // first row of this wave's block (blockDim.x == 64)
  size_t g = g0 + l; if (g >= nrows) g = nrows - 1;                             // (rows behind the arrays' end: the last wave of a launch that covers all nmax nodes)
  const double* mb = qm_table(a.mb);
  extern __shared__ double qm_smem[];                  // LQ_KIN_LDS_BYTES
  double* rows = qm_smem; KinOut o; o.tile = qm_smem + LQ_KIN_TILE; o.wave_rec = a.kin + g0 * KR_SIZE; o.l = l;
  // x and the kinematics workspace K live in registers; the input u — read-only here — stays in its LDS row (31-double pitch: conflict free):
  // x + u + K + two flow values do not fit 512 registers, and what does not fit would otherwise be spilled to scratch memory
  double x[30], K[KW_SIZE]; double* u = rows + l * 31;
  kin_rows_in(rows, a.x, g0, nrows, l); qm_wave_sync();
  _Pragma("unroll") for (int q = 0; q < 30; ++q) x[q] = u[q];
  qm_wave_sync();
  kin_rows_in(rows, a.u, g0, nrows, l); qm_wave_sync();
  const double* ee = a.eeref + g * 7; const double dt = a.node_dt[g];
  // Every block of the workspace goes to the record as soon as it is complete and only what the flow map needs (the feet, the base block) stays live: the
  // kernel then fits 256 registers, i.e. TWO waves per SIMD — every wavefront of the benchmark launch is resident at once and a wave's dependent chains overlap
  // with its neighbour's
  kin_base<true>(mb, x, K);
  kin_emit<KW_LEG>(o, KR_K1, K);
  _Pragma("unroll") for (int c = 0; c < 4; ++c) { kin_leg<true>(mb, c, x, u, K); kin_emit<KW_LEGSZ>(o, KR_K1 + KW_LEG + KW_LEGSZ * c, K + KW_LEG + KW_LEGSZ * c); __builtin_amdgcn_sched_barrier(0); }
  kin_arm<true>(mb, x, K);
  kin_emit<KW_SIZE - KW_ARM>(o, KR_K1 + KW_ARM, K + KW_ARM);
  double f[12 + 10];                                    // [0, 10): end-effector error g(6), qee(4); then the twelve non-trivial rows of the flow value (record order)
  ee_error(K, ee, ee + 3, f + 6, f);
  flow_head_from_kin(mb, x, u, K, f + 10);
  kin_emit<22>(o, KR_EEG, f);
  double x2[30];                                        // the joint part of the flow value is the input's joint velocities (u, in LDS)
  _Pragma("unroll") for (int q = 0; q < 30; ++q) x2[q] = x[q] + dt * ((q < 12) ? f[10 + (q < 12 ? q : 0)] : u[q]);
  const double* mb2 = qm_table(a.mb);                   // the second stage reads the model table through its own (opaque) pointer: the first stage's table entries are then
                                                        // loaded again from the scalar cache instead of being carried across the stage in scalar registers (16 of them were spilled)
  kin_base<true>(mb2, x2, K);
  kin_emit<KW_LEG>(o, KR_K2, K);
  _Pragma("unroll") for (int c = 0; c < 4; ++c) { kin_leg<true>(mb2, c, x2, u, K); kin_emit<KW_LEGSZ>(o, KR_K2 + KW_LEG + KW_LEGSZ * c, K + KW_LEG + KW_LEGSZ * c); __builtin_amdgcn_sched_barrier(0); }
  flow_head_from_kin(mb2, x2, u, K, f);
  kin_emit<12>(o, KR_F2, f);
}
#endif

// ---- K1b: one wavefront per node ----
// DBG: the instance that also writes the debug records (a.dbg) and the phase cycle stamps (a.prof) — parity tests and profiling; the product instance has neither branch
template <bool DBG, int MT, bool IPM = false>
__device__ __forceinline__ void qm_lq_body(QmLqArgs a) {
  if (!DBG) { a.dbg = nullptr; a.prof = 0; }
  extern __shared__ double qm_smem[];
  double* S = qm_smem;
  const int l = threadIdx.x & 63, g = l >> 4, c = l & 15;
  const int b = blockIdx.x / a.ncap, i = a.i0 + blockIdx.x - b * a.ncap;
  const int nb = i * a.B + b;                       // node-major index
  // instrumented instance, profiling on: wave entry in shader-clock cycles and in ticks of the constant 100 MHz reference clock (tools/lq_residency_probe.py)
  const long long c0_ = (DBG && a.prof) ? (long long)__builtin_readcyclecounter() : 0; const long long r0_ = (DBG && a.prof) ? (long long)__builtin_amdgcn_s_memrealtime() : 0;
  const double* mb = qm_table(a.mb); const double* st = qm_table(a.st);
#ifdef QM_LQ_FOLD_RECORDS      /* bandwidth experiment only (profiles/r06_ab_lq_write_bound.log): every wave writes its stage record into one of QM_LQ_FOLD_RECORDS slots — the stores stay in cache, the results are meaningless */
  double* rec = a.stage + ((size_t)(blockIdx.x % QM_LQ_FOLD_RECORDS)) * SR_SIZE;
#else
  double* rec = a.stage + ((size_t)b * a.nmax + i) * SR_SIZE;
#endif
  double* dbg = (DBG && a.dbg) ? a.dbg + ((size_t)b * a.nmax + i) * LQ_DBG_SIZE : nullptr;
  const double* kr = a.kin + (size_t)nb * KR_SIZE;
  // every input address depends on (b, i) only: issue all loads before looking at the node's status (one memory round trip).  The node's status words are wave-uniform and
  // nobody writes them in this kernel: read through the CONSTANT address space they are scalar loads (round 6) — as plain loads the compiler made them vector loads + a
  // v_readfirstlane, whose wait (the vector memory counter) stood in front of every other load of the prologue: a second round trip
  typedef const int __attribute__((address_space(4)))* qm_citab;
  const int nn = ((qm_citab)a.n_nodes)[b], ev = ((qm_citab)a.node_ev)[nb], mode = ((qm_citab)a.node_mode)[nb];
  const double dt = ((qm_ctab)a.node_dt)[nb];
  const int nxt = (i + 1 < a.nmax) ? ((i + 1) * a.B + b) : nb;
  double in_x = 0.0, in_u = 0.0, xn = 0.0, f1 = 0.0, f2 = 0.0, in_ee = 0.0, in_xref = 0.0, in_qd = 0.0;
  // the two kinematics workspaces of the kin record come in as PAIRS of doubles (16-byte global loads, 16-byte LDS stores: half the instructions of both kinds)
  static_assert(KR_K1 % 2 == 0 && KR_K2 % 2 == 0 && KR_SIZE % 2 == 0 && KW_SIZE % 2 == 0 && LW_K1 % 2 == 0 && KW_SIZE / 2 <= 128, "kin workspaces: 16-byte aligned, two pairs per lane");
  double2 in_k[2], in_k2[2];
  if (l < 30) { in_xref = a.xref[nb * 30 + l]; in_qd = st[ST_Q + l]; in_x = a.x[nb * 30 + l]; in_u = a.u[nb * 30 + l]; xn = a.x[nxt * 30 + l]; }
  if (l < 30) { const int lf = (l < 12) ? l : 0; f1 = kr[KR_F1 + lf]; f2 = kr[KR_F2 + lf]; }      // rows 0..11; rows 12..29 of the flow map are the input's joint velocities (selected below)
  if (l >= 32 && l < 39) in_ee = a.eeref[nb * 7 + (l - 32)];
  if (l >= 40 && l < 46) in_ee = kr[KR_EEG + (l - 40)];
  if (l >= 48 && l < 52) in_ee = kr[KR_QEE + (l - 48)];
  double in_zv = 0.0, in_zp = 0.0;                     // swing-height references of contact l − 60 (the constraint values' lanes): loaded with everything else — in the constraint phase they were a memory round trip of their own, waited for on the spot
  if (l >= 60) { in_zv = a.zvel[nb * 4 + (l - 60)]; in_zp = a.zpos[nb * 4 + (l - 60)]; }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int p = l + 64 * t;                                          // doubles 2 p, 2 p + 1
    in_k[t] = (p < KW_SIZE / 2) ? ((const double2*)(kr + KR_K1))[p] : double2{0.0, 0.0};
    in_k2[t] = (2 * p < KW_ARM) ? ((const double2*)(kr + KR_K2))[p] : double2{0.0, 0.0};      // (the stage-2 workspace ends in front of the arm block, at an odd index: the pair's second half is dropped below)
    if (2 * p + 1 >= KW_ARM) in_k2[t].y = 0.0;
  }
  QM_SCALARS_READY(nn, ev, mode, dt);      // the four scalar loads are waited for HERE, together (left alone the compiler sinks each one to its first use: three waits in a row)
  if (i >= nn) return;
  const bool terminal = (i == nn - 1);
  if (!terminal && ev == QM_EV_PRE) {                 // event nodes carry no LQ data (identity jump, handled by K3): only clear their merit terms
    const double jd = (l < 30) ? in_x - xn : 0.0; const double s2 = qm_wave_sum(jd * jd);   // jump defect x_i − x_{i+1} (identity jump map), unweighted
    if (l < 3) a.perf[nb * PF_SIZE + l] = (l == 1) ? s2 : 0.0;   // (the node index of an event moves between receding-horizon solves)
    return;
  }
  long long tp_[10]; int np_ = 0;
// phase boundary: a scheduling barrier in every instance (without one the scheduler merges the phases of the product instance into one region, lengthens the
// fragments' live ranges and spills 100 bytes per lane), a cycle stamp in the instrumented one
#define LQT() { __builtin_amdgcn_sched_barrier(0); if (DBG && a.prof) tp_[np_] = (long long)__builtin_readcyclecounter(); ++np_; }
  LQT()
  // ---- P0: inputs and the kin record -> LDS ----
  double* T = S + LW_T; double* X = S + LW_V_X; double* U = S + LW_V_U; double* K1 = S + LW_K1; double* K2 = S + LW_K2; double* EE = S + LW_V_EE;
  static_assert(LW_V % 2 == 0 && (LW_K1 - LW_V) % 2 == 0 && (32 * LW_TLD) % 2 == 0, "16-byte zero fills");
  lw_zero<LW_K1 - LW_V>(S + LW_V, l);
  qm_wave_sync();
  if (terminal) { in_u = 0.0; xn = 0.0; f1 = 0.0; f2 = 0.0; }
  if (l < 30) { X[l] = in_x; U[l] = in_u; }
  if (l >= 32 && l < 39) EE[16 + (l - 32)] = in_ee;
  if (l >= 40 && l < 46) EE[l - 40] = in_ee;
  if (l >= 48 && l < 52) EE[12 + (l - 48)] = in_ee;
#pragma unroll
  for (int t = 0; t < 2; ++t) { const int p = l + 64 * t; if (p < KW_SIZE / 2) ((double2*)K1)[p] = in_k[t]; if (p < LW_K2SZ / 2) ((double2*)K2)[p] = terminal ? double2{0.0, 0.0} : in_k2[t]; }
  qm_wave_sync();

  if (terminal) {
    // final EE soft constraint only: Q_N = Jᵀ mu J, q_N = Jᵀ mu g, c_N = 1/2 g mu g
    if (l < 6) EE[6 + l] = (l < 3 ? st[ST_MU_EEF_POS] : st[ST_MU_EEF_ORI]);
    if (l < 30) { double col[6]; ee_jac_col(X, K1, EE + 12, EE + 19, l, col); for (int r = 0; r < 6; ++r) T[l * 6 + r] = col[r]; }
    qm_wave_sync();
    for (int idx = l; idx < 900; idx += 64) { const int r = idx / 30, cc = idx - r * 30; double s = 0.0; for (int k = 0; k < 6; ++k) s += T[r * 6 + k] * EE[6 + k] * T[cc * 6 + k]; rec[SR_QP + idx] = s; }
    if (l < 30) { double s = 0.0; for (int k = 0; k < 6; ++k) s += T[l * 6 + k] * EE[6 + k] * EE[k]; rec[SR_QPV + l] = s; }
    if (l == 0) { double cN = 0.0; for (int k = 0; k < 6; ++k) cN += 0.5 * EE[6 + k] * EE[k] * EE[k]; rec[SR_SCAL] = 0.0; rec[SR_SCAL + 1] = cN; a.perf[nb * PF_SIZE] = cN; a.perf[nb * PF_SIZE + 1] = 0.0; a.perf[nb * PF_SIZE + 2] = 0.0; }
    return;
  }
  // what the prologue's loads are needed for is formed HERE and only the results stay live (three waves per SIMD: 168 registers): the defect b, the tracking terms
  // of the cost (state weights are diagonal) and u − u_nom
  double* QD = S + LW3_QD; double* RD = S + LW3_RD; double* FR = S + LW3_FR;      // (phase III only: in the hand-over tile)
  const double bl = (l < 30) ? X[l] + 0.5 * dt * ((l < 12) ? f1 : U[l]) + 0.5 * dt * ((l < 12) ? f2 : U[l]) - xn : 0.0;
  if (l < 32) S[LW_V_B + l] = bl;
  if (l < 30) {
    const double dx = X[l] - in_xref; const double qd = in_qd;
    S[LW_V_QV + l] = qd * dx; S[LW_V_RR + l] = 0.5 * qd * dx * dx;      // the lane's tracking cost waits in the (idle) r + R Pe slot: one register pair less across the Jacobians
  }      // (the diagonal of Q, the zero diagonal additions of R and u − unom are formed at the start of phase III: nothing reads them before, and there they fit into the idle tile)
  qm_wave_sync();
  LQT()
  // ---- phase II: equality rows + closed-form block projection ----
  // rows ordered per contact i = LF,RF,LH,RH: swing -> [F_i = 0 (3)] , stance -> [v_i = 0 (3)] , swing -> [v_iz = zvel_ref (1)]
  // tile rows 0..15 = C (state part), rows 16..31 = D (input part)
  lw_zero<(DBG ? 32 : 16) * LW_TLD>(T, l);      // (the C half only: of the D half nothing but the joint-velocity blocks the lanes below write is ever read — the debug instance dumps all of it)
  qm_wave_sync();
  int row0[4]; int nc = 0; for (int k = 0; k < 4; ++k) { row0[k] = nc; nc += mode_flag(mode, k) ? 3 : 4; }
  const double gain = st[ST_POS_ERR_GAIN];
  {
    // one (column, contact) task per lane for the non-trivial classes of the foot-velocity Jacobian: lanes 0-11 columns 3..5 x 4 contacts,
    // 12-23 columns 9..11 x 4 contacts, 24-35 the leg's own joint angles 12..23, 36-47 its own joint velocities 42..53
    int tc = 3, tk = 0;
    if (l < 12) { tc = 3 + (l >> 2); tk = l & 3; } else if (l < 24) { tc = 9 + ((l - 12) >> 2); tk = l & 3; }
    else if (l < 36) { tc = 12 + (l - 24); tk = chain_to_contact((l - 24) / 3); } else if (l < 48) { tc = 42 + (l - 36); tk = chain_to_contact((l - 36) / 3); }
    int rk = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) if (j < tk) rk += mode_flag(mode, j) ? 3 : 4;
    if (l >= 60) tk = l - 60;                              // lanes 60..63: one contact each, the constraint values
    double dv[3], dpz, vk[3], pzk; foot_vel_jac_col_wave(mb, X, U, K1, tk, tc, dv, &dpz, vk, &pzk);   // branch-free over the column classes
    if (l < 48) {
      double* M = (tc < 30) ? T : T + 16 * LW_TLD; const int cc = (tc < 30) ? tc : tc - 30;
      if (mode_flag(mode, tk)) { for (int r = 0; r < 3; ++r) M[(rk + r) * LW_TLD + cc] = dv[r] + ((r == 2 && gain != 0.0) ? gain * dpz : 0.0); }
      else M[(rk + 3) * LW_TLD + cc] = dv[2] + (gain != 0.0 ? gain * dpz : 0.0);
    } else if (l < 60) {
      // structurally constant entries: columns 0..2 (d v / d h_lin = I), column 8 (d p_z / d z = 1), swing force rows F_k = 0
      const int t = l - 48, k = t & 3, j = t >> 2;           // j = 0..2: column j resp. force component j ; j = 3: column 8
      int r0k = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) if (q < k) r0k += mode_flag(mode, q) ? 3 : 4;
      const bool stance = mode_flag(mode, k);
      if (j < 3) {
        if (stance) T[(r0k + j) * LW_TLD + j] = 1.0; else { if (j == 2) T[(r0k + 3) * LW_TLD + 2] = 1.0; T[(16 + r0k + j) * LW_TLD + 3 * k + j] = 1.0; }
      } else if (gain != 0.0) T[(r0k + (stance ? 2 : 3)) * LW_TLD + 8] = gain;
    } else {
      const int k = tk; const bool stance = mode_flag(mode, k); const int rr = row0[k];
      const double gz = (gain != 0.0) ? gain * pzk : 0.0;
      double bb = -in_zv; if (gain != 0.0) bb -= gain * in_zp;
      const double e0 = stance ? vk[0] : U[3 * k], e1 = stance ? vk[1] : U[3 * k + 1], e2 = stance ? vk[2] + gz : U[3 * k + 2], e3 = bb + vk[2] + gz;
      // e also rides in column 30 of the C rows: [C | e]
      S[LW_V_E + rr] = e0; T[rr * LW_TLD + 30] = e0; S[LW_V_E + rr + 1] = e1; T[(rr + 1) * LW_TLD + 30] = e1; S[LW_V_E + rr + 2] = e2; T[(rr + 2) * LW_TLD + 30] = e2;
      if (!stance) { S[LW_V_E + rr + 3] = e3; T[(rr + 3) * LW_TLD + 30] = e3; }
    }
  }
  qm_wave_sync();
  const double* Ct = T; const double* Dt = T + 16 * LW_TLD;
  if (DBG && dbg) { for (int idx = l; idx < 480; idx += 64) { const int r = idx / 30, cc = idx - r * 30; dbg[LQ_DBG_C + idx] = Ct[r * LW_TLD + cc]; dbg[LQ_DBG_D + idx] = Dt[r * LW_TLD + cc]; } if (l < 16) dbg[LQ_DBG_e + l] = S[LW_V_E + l]; if (l == 0) dbg[LQ_DBG_nc] = nc; }
  LQT()
  // per contact: stance -> Ginv (3x3) of the joint-velocity block; swing -> g/(g.g) and a 3x2 orthonormal complement of g
  double* G = S + LW_V_G;
  if (l < 4) {
    const int k = l, ch = contact_to_chain(k), jc = 12 + 3 * ch; double* gg = G + 12 * k;
    if (mode_flag(mode, k)) { double M3[9]; for (int r = 0; r < 3; ++r) for (int q = 0; q < 3; ++q) M3[3 * r + q] = Dt[(row0[k] + r) * LW_TLD + jc + q]; m3_inv(M3, gg); }
    else {
      const double gv[3] = {Dt[(row0[k] + 3) * LW_TLD + jc], Dt[(row0[k] + 3) * LW_TLD + jc + 1], Dt[(row0[k] + 3) * LW_TLD + jc + 2]};
      const double n2 = gv[0] * gv[0] + gv[1] * gv[1] + gv[2] * gv[2], nrm = sqrt(n2), in2 = qm_frcp(n2);
      for (int r = 0; r < 3; ++r) gg[r] = gv[r] * in2;
      // Householder H = I - 2 v vᵀ/(vᵀ v), v = g - alpha e1, alpha = -sign(g0)|g| : H e1 || g, columns 2,3 of H span g-perp
      const double alpha = gv[0] > 0.0 ? -nrm : nrm; const double v[3] = {gv[0] - alpha, gv[1], gv[2]}; const double vv = v[0] * v[0] + v[1] * v[1] + v[2] * v[2], ivv = qm_frcp(vv);
      for (int r = 0; r < 3; ++r) { gg[3 + r] = ((r == 1) ? 1.0 : 0.0) - 2.0 * v[r] * v[1] * ivv; gg[6 + r] = ((r == 2) ? 1.0 : 0.0) - 2.0 * v[r] * v[2] * ivv; }
    }
  }
  qm_wave_sync();
  // [Px | Pe] straight into fragments: input rows 12..23 of Px, all rows of Pe (column 30).  One code path for stance and swing legs:
  // row 12 + 3 ch + jj of Px is −(c0 Crow(i0) + c1 Crow(i1) + c2 Crow(i2)) with (c, i) = (Ginv row jj, the leg's 3 velocity rows) for a
  // stance leg and (g_jj / g·g, 0, 0; the normal-velocity row) for a swing leg; [C | e] carries e in column 30.
  qm_d4 PxA[2][2]; qm_frag_zero<2, 2>(PxA);
  // (one row slot at a time, its coefficients shared by the two column tiles, a scheduling barrier between the slots: all 36 LDS reads in flight at once were the
  //  register peak of this phase)
  auto px_rows = [&](int row, double& o0, double& o1) {
    const int rr = row - 12, ch = rr / 3, jj = rr - 3 * ch, k = chain_to_contact(ch); const double* gg = G + 12 * k;
    int r0k = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) if (q < k) r0k += mode_flag(mode, q) ? 3 : 4;
    const bool stance = mode_flag(mode, k);
    double c0 = gg[stance ? 3 * jj : jj], c1 = gg[3 * jj + 1], c2 = gg[3 * jj + 2]; QM_LOADED(c0); QM_LOADED(c1); QM_LOADED(c2);      // (all three slots exist for either leg state)
    c1 = stance ? c1 : 0.0; c2 = stance ? c2 : 0.0;
    const int i0 = stance ? r0k : r0k + 3, i1 = stance ? r0k + 1 : r0k + 3, i2 = stance ? r0k + 2 : r0k + 3;
    o0 = -(c0 * Ct[i0 * LW_TLD + c] + c1 * Ct[i1 * LW_TLD + c] + c2 * Ct[i2 * LW_TLD + c]);
    o1 = -(c0 * Ct[i0 * LW_TLD + 16 + c] + c1 * Ct[i1 * LW_TLD + 16 + c] + c2 * Ct[i2 * LW_TLD + 16 + c]);      // column 31 of [C | e] is zero
  };
  { double o0, o1;
    px_rows(12 + g, o0, o1); PxA[0][0][3] = o0; PxA[0][1][3] = o1; __builtin_amdgcn_sched_barrier(0);
    px_rows(16 + g, o0, o1); PxA[1][0][0] = o0; PxA[1][1][0] = o1; __builtin_amdgcn_sched_barrier(0);
    px_rows(20 + g, o0, o1); PxA[1][0][1] = o0; PxA[1][1][1] = o1; __builtin_amdgcn_sched_barrier(0); }
  if (c == 14) {                                              // Pe rows 0..11: −F of a swing foot
#pragma unroll
    for (int r = 0; r < 3; ++r) { const int row = g + 4 * r; PxA[0][1][r] = mode_flag(mode, row / 3) ? 0.0 : -U[row]; }
  }
  qm_wave_sync();
  lw_get_col30<2, 30, 30>(PxA, S + LW_V_PE, 30);
  PxA[0][1][0] = 0.0; PxA[0][1][1] = 0.0; PxA[0][1][2] = 0.0;      // Pe rows 0..11 now live in LW_V_PE only: re-read where they are used (six registers less across the Jacobians)
  const double eq2 = qm_wave_sum((l < nc) ? S[LW_V_E + l] * S[LW_V_E + l] : 0.0);
  const double b2 = qm_wave_sum(bl * bl);
  // Pu (30 x m) straight into fragments: columns = stance forces (identity triples), swing-leg null spaces (3 x 2 blocks), arm (identity)
  int nst = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) nst += mode_flag(mode, k);
  const int m = 3 * nst + 2 * (4 - nst) + 6;
  // which block does column j of Pu belong to?  type 0: stance contact kk, component t; 1: swing contact kk, null-space column t; 2: arm joint t; 3: none
  auto pu_col_class = [&](int j, int& type, int& kk, int& t) {
    type = 3; kk = 0; t = 0;
    if (j < 3 * nst) { type = 0; t = j % 3; int cnt = j / 3;
#pragma unroll
      for (int k = 3; k >= 0; --k) { int before = 0; for (int q = 0; q < 4; ++q) if (q < k && mode_flag(mode, q)) ++before; if (mode_flag(mode, k) && before == cnt) kk = k; } }
    else if (j < 3 * nst + 2 * (4 - nst)) { type = 1; const int jj2 = j - 3 * nst; t = jj2 & 1; const int cnt = jj2 >> 1;
#pragma unroll
      for (int k = 3; k >= 0; --k) { int before = 0; for (int q = 0; q < 4; ++q) if (q < k && !mode_flag(mode, q)) ++before; if (!mode_flag(mode, k) && before == cnt) kk = k; } }
    else if (j < m) { type = 2; t = j - (3 * nst + 2 * (4 - nst)); }
  };
  qm_d4 PuF[2][2];
#pragma unroll
  for (int J = 0; J < 2; ++J) {
    int type, kk, t; pu_col_class(16 * J + c, type, kk, t);
    const int jc = 12 + 3 * contact_to_chain(kk); const double* gg = G + 12 * kk;
#pragma unroll
    for (int I = 0; I < 2; ++I)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * I + g + 4 * r; double v = 0.0;
        if (type == 0) v = (row == 3 * kk + t) ? 1.0 : 0.0;
        else if (type == 1) v = (row >= jc && row < jc + 3) ? gg[3 + 3 * t + (row - jc)] : 0.0;
        else if (type == 2) v = (row == 24 + t) ? 1.0 : 0.0;
        PuF[I][J][r] = v;
      }
  }
  LQT()
  // ---- phase I: Jacobian columns -> fragments.  Tile rows 0..15 take [df/dx], rows 16..31 take [df/du] (rows 0..15 of each) ----
  qm_d4 Ad[2][2], Bdt[2][2];
  {
    // Live set kept small (three waves per SIMD): stage 1 as A1, B1; of stage 2 first only A2ᵀ[0:16, 0:16] for the two products, then A2 and B2 themselves; B_d is formed
    // untransposed (16 x 32) and turned into B_dᵀ through the tile once — instead of holding both stages in both layouts (six fragments) at the same time
    qm_d4 A1[1][2], B1[1][2];
    // one (column, Heun stage) task per lane for the four non-trivial column classes of the flow Jacobian — a wave executes each
    // divergent class body once: lanes 0-5 dθ-rate columns 3..5, 6-11 zyx columns 9..11, 12-35 leg joints 12..23, 36-59 forces 30..41
    int fc = 0, fs = l & 1;
    fc = (l < 6) ? 3 + (l >> 1) : ((l < 12) ? 9 + ((l - 6) >> 1) : ((l < 36) ? 12 + ((l - 12) >> 1) : 30 + ((l - 36) >> 1)));
    double colf[12];
    int kofs = fs ? (int)(K2 - K1) : 0; QM_LANE_OPAQUE(kofs);                  // the lane's stage: one base, immediate offsets
    flow_jac_col(mb, X, U, K1 + kofs, fc, colf);
    const int cc = (fc < 30) ? fc : fc - 30, r0 = (fc < 30) ? 0 : 16;
    lw_zero<32 * LW_TLD>(T, l);
    qm_wave_sync();
    if (l < 60 && fs == 0) { for (int r = 0; r < 12; ++r) T[(r0 + r) * LW_TLD + cc] = colf[r]; }
    if (l < 3) T[(6 + l) * LW_TLD + l] = 1.0;                                   // d rdot / d h_lin (both stages)
    if (l >= 4 && l < 8) T[(16 + 8 + l) * LW_TLD + 8 + l] = 1.0;                // d qdot_j / d u_j rows 12..15 (rows >= 16 are handled analytically)
    qm_wave_sync();
    qm_frag_load_tile<1, 2, false>(A1, T, LW_TLD); qm_frag_load_tile<1, 2, false>(B1, T + 16 * LW_TLD, LW_TLD);   // columns 30, 31 of the tile are zero
    qm_wave_sync();
    if (l < 60 && fs == 1) { for (int r = 0; r < 12; ++r) T[(r0 + r) * LW_TLD + cc] = colf[r]; }      // same sparsity pattern: no re-zeroing needed
    qm_wave_sync();
    LQT()
    qm_d4 TA[1][2], TB[1][2]; qm_frag_zero<1, 2>(TA); qm_frag_zero<1, 2>(TB);
    { qm_d4 Z[1][1]; qm_frag_load_tile<1, 1, true>(Z, T, LW_TLD);               // A2ᵀ[0:16, 0:16]
      qm_gemm_tn<1, 1, 2>(Z, A1, TA, 0, 4, false);                              // A2 A1 (rows < 16)
      qm_gemm_tn<1, 1, 2>(Z, B1, TB, 0, 4, false); }                            // A2[:, 0:16] B1[0:16, :]; rows >= 16 of B1 are unit rows: + A2[:, 16:32] below
    {
      qm_d4 A2[1][2]; qm_frag_load_tile<1, 2, false>(A2, T, LW_TLD);
      TB[0][1] += A2[0][1];
#pragma unroll
      for (int J = 0; J < 2; ++J)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = g + 4 * r, col = 16 * J + c;
          Ad[0][J][r] = (col < 30) ? 0.5 * dt * A1[0][J][r] + 0.5 * dt * (A2[0][J][r] + dt * TA[0][J][r]) + ((row == col) ? 1.0 : 0.0) : 0.0;
          Ad[1][J][r] = (16 + row == col && col < 30) ? 1.0 : 0.0;
        }
    }
    {
      qm_d4 B2[1][2]; qm_frag_load_tile<1, 2, false>(B2, T + 16 * LW_TLD, LW_TLD);
      qm_d4 Bd[1][2];                                                            // rows 0..15 of B_d (state rows), all 32 input columns
#pragma unroll
      for (int J = 0; J < 2; ++J)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int col = 16 * J + c; Bd[0][J][r] = (col < 30) ? 0.5 * dt * B1[0][J][r] + 0.5 * dt * (B2[0][J][r] + dt * TB[0][J][r]) : 0.0; }
      qm_wave_sync();
#pragma unroll
      for (int J = 0; J < 2; ++J)
#pragma unroll
        for (int r = 0; r < 4; ++r) T[(g + 4 * r) * LW_TLD + 16 * J + c] = Bd[0][J][r];
      qm_wave_sync();
      qm_d4 Bt[2][1]; qm_frag_load_tile<2, 1, true>(Bt, T, LW_TLD);             // B_dᵀ[:, 0:16]
#pragma unroll
      for (int I = 0; I < 2; ++I)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int row = 16 * I + g + 4 * r; Bdt[I][0][r] = Bt[I][0][r]; Bdt[I][1][r] = (row == 16 + c && row < 30) ? dt : 0.0; }
    }
  }
  // Pin the results of phase I: the discrete-time Jacobians are complete HERE.  Without a use at this point the compiler sinks their arithmetic towards the
  // products of phase II, keeps the six stage Jacobian fragments alive next to them and spills 100 bytes per lane (the debug stores below used to be that use).
#pragma unroll
  for (int I = 0; I < 2; ++I)
#pragma unroll
    for (int J = 0; J < 2; ++J) { QM_PIN4(Ad[I][J]); QM_PIN4(Bdt[I][J]); }
  if (DBG && dbg) {
    qm_frag_store<2, 2>(Ad, dbg + LQ_DBG_A, 30, 30, 30);
#pragma unroll
    for (int I = 0; I < 2; ++I)
#pragma unroll
      for (int J = 0; J < 2; ++J)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int row = 16 * I + g + 4 * r, col = 16 * J + c; if (row < 30 && col < 30) dbg[LQ_DBG_B + col * 30 + row] = Bdt[I][J][r]; }
    if (l < 30) dbg[LQ_DBG_b + l] = bl;
  }
  LQT()
  // projected dynamics  [Ap | bp] = [Ad | b] + Bd [Px | Pe]  (Z = Bdᵀ; Px rows 12..23 -> k-steps 3..5, Pe rows 0..11 -> column tile 1, k-steps 0..2)
  {
    qm_d4 ApA[2][2];
#pragma unroll
    for (int I = 0; I < 2; ++I)
#pragma unroll
      for (int J = 0; J < 2; ++J) ApA[I][J] = Ad[I][J];
    lw_set_col30(ApA, S + LW_V_B);
    qm_gemm_tn<2, 2, 2>(Bdt, PxA, ApA, 3, 6, false);
    { qm_d4 Y1[2][1], P1[2][1];
#pragma unroll
      for (int I = 0; I < 2; ++I) { Y1[I][0] = PxA[I][1]; P1[I][0] = ApA[I][1]; }
#pragma unroll
      for (int r = 0; r < 3; ++r) Y1[0][0][r] = *((c == 14) ? S + LW_V_PE + g + 4 * r : S + LW_V_X + 30);      // (Pe in lane column 14, a zero elsewhere: read through a selected address, slot 30 of the state vector holds zero)
      qm_gemm_tn<2, 2, 1>(Bdt, Y1, P1, 0, 3, false);
#pragma unroll
      for (int I = 0; I < 2; ++I) ApA[I][1] = P1[I][0]; }
    qm_frag_store<2, 2, true>(ApA, rec + SR_AP, 30, 12, 30);               // rows 0..11 only: a joint row is e_j + dt Px[j], K3 rebuilds it (k_riccati.h)
    const int gg2 = l >> 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) { const int row = gg2 + 4 * r; if (c == 14) rec[SR_BPV + row] = ApA[0][1][r]; }
    // rows 16..29 are joint rows: bp_j = b_j + dt Pe_j, the one non-zero term of the product (the second tile row of [Ap | bp] is never formed)
    if (l >= 16 && l < 30) rec[SR_BPV + l] = fma(dt, S[LW_V_PE + l], S[LW_V_B + l]);
  }
  // column descriptors of Pu (every column is a unit vector or three consecutive entries): source row i0 and weights w0..w2.  Written HERE, behind phase I: they live where the
  // stage-2 kin workspace was (LW_PD = LW_K2 — the eleventh wave per CU)
  qm_wave_sync();
  if (g == 0) {
#pragma unroll
    for (int J = 0; J < 2; ++J) {
      const int j = 16 * J + c; int type, kk, t; pu_col_class(j, type, kk, t);
      const int jc = 12 + 3 * contact_to_chain(kk); const double* gg = G + 12 * kk;
      const int i0 = (type == 0) ? 3 * kk + t : ((type == 1) ? jc : ((type == 2) ? 24 + t : 0));
      S[LW_PD + j] = (double)i0;
      { const double m1 = (type == 1) ? 1.0 : 0.0, c0 = (type == 0 || type == 2) ? 1.0 : 0.0;      // (the three block entries are read by every lane of the group — in bounds for any t — and enter through a 0 / 1 factor)
        S[LW_PD + 32 + 3 * j] = fma(m1, gg[3 + 3 * t], c0); S[LW_PD + 32 + 3 * j + 1] = m1 * gg[4 + 3 * t]; S[LW_PD + 32 + 3 * j + 2] = m1 * gg[5 + 3 * t]; }
    }
  }
  lw_bp<MT>(S, rec, m, Bdt);      // Bp = Bd Pu: the last use of the discrete-time Jacobians
  LQT()
  // ---- phase III: cost quadratic model (x dt) ----
  double cost = (l < 30) ? S[LW_V_RR + l] : 0.0;
  // per-lane table entries of this phase, requested TOGETHER with R0 (one memory round trip instead of three: each was loaded where it is used and waited for on the spot):
  // the three entries of the lane's row of R0 for the structured product below, the box limits of the lane's arm joint for the barriers
  const int lr = (l < 30) ? l : 29; const bool rblk = lr >= 12 && lr < 24; const int lb = lr / 3, b0 = rblk ? 3 * lb : (lr < 27 ? lr : 27);
  const double r0a = st[ST_R + 30 * lr + b0], r0b = st[ST_R + 30 * lr + b0 + 1], r0c = st[ST_R + 30 * lr + b0 + 2];
  const bool boxv = l < 12, boxc = (l >= 32 && l < 44); const int kb = boxv ? l : (boxc ? l - 32 : 0); const bool pos = kb < 6; const int kj = pos ? kb : kb - 6;
  const double box_lo = pos ? mb[MB_QLO + 12 + kj] : st[ST_JVEL_LO + kj], box_hi = pos ? mb[MB_QHI + 12 + kj] : st[ST_JVEL_HI + kj];
  const double qd3 = st[ST_Q + lr];                                       // the lane's state weight again (the prologue's copy went into q and the tracking cost)
  if (l < 32) {      // phase III's own vectors, in the idle hand-over tile (the last reader of the tile — lw_bp — ended with a wave sync)
    double unom = 0.0; if (l < 12 && (l % 3) == 2 && mode_flag(mode, l / 3) && nst > 0) unom = mb[MB_ROBOTMASS] * 9.81 / nst;
    QD[l] = (l < 30) ? qd3 : 0.0; RD[l] = 0.0; S[LW3_DU + l] = (l < 30) ? U[l] - unom : 0.0;
  }
  qm_d4 Rm[2][2]; qm_frag_load<2, 2, false>(Rm, st + ST_R, 30, 30, 30);      // input cost weights (L2 / scalar-cache resident table)
  { const double mup = st[ST_MU_EE_POS], muo = st[ST_MU_EE_ORI];      // two scalar table reads + a select of the VALUES (a select between the two addresses is a vector load with its own wait)
    if (l >= 32 && l < 38) EE[6 + (l - 32)] = ((l - 32) < 3) ? mup : muo; }
  qm_wave_sync();
  if (QM_LQ_RB_ONLY || a.rb) {
    // r = R0 (u − unom) with the block-diagonal R0 of the shipped task file (wave-uniform test): a row has at most three non-zero entries, lane = row.  Bit-identical to the
    // dense product below: there every lane holds ONE exactly rounded product (the other tile's entry of its column is an exact zero) and the row_shr tree (steps 1, 2, 4, 8)
    // adds the three of a leg block as  p2 + (p1 + p0)  for the columns 12-14 (lanes 12, 13, 14) and 18-20 (lanes 2, 3, 4 of the second tile) resp.  (p2 + p1) + p0  for
    // 15-17 (lane 15 of the first tile | lanes 0, 1 of the second: the pair meets first) and 21-23 (lanes 5, 6, 7: 6 and 7 meet first); adding the exact zeros of the other
    // lanes changes nothing.  A diagonal row reads two exact zeros beside its entry (columns clamped to 27..29 for the last rows).
#pragma clang fp contract(off)
    if (l < 30) {
      const double p0 = r0a * S[LW3_DU + b0], p1 = r0b * S[LW3_DU + b0 + 1], p2 = r0c * S[LW3_DU + b0 + 2];
      S[LW_V_RV + l] = (lb & 1) ? (p2 + p1) + p0 : p2 + (p1 + p0);          // leg blocks: lb = 4 .. 7
    }
  } else { // r = R0 (u − unom): a mat-vec on the register fragments — two partial products per lane and register, then a 16-lane DPP row sum
    // (as an MFMA it would spend 16 issues of 64 cycles on a single useful column; f64 MFMA and f64 VALU run at the same rate on gfx950)
    const double d0 = S[LW3_DU + c], d1 = S[LW3_DU + 16 + c];
#pragma unroll
    for (int I = 0; I < 2; ++I)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double p = Rm[I][0][r] * d0 + Rm[I][1][r] * d1;
        p += qm_dpp0<0x111>(p); p += qm_dpp0<0x112>(p); p += qm_dpp0<0x114>(p); p += qm_dpp0<0x118>(p);   // row_shr 1, 2, 4, 8: lane 15 of each row holds the sum
        const int row = 16 * I + g + 4 * r;
        if (c == 15 && row < 30) S[LW_V_RV + row] = p;
      }
  }
  qm_wave_sync();
  if (l < 30) cost += 0.5 * S[LW3_DU + l] * S[LW_V_RV + l];
  qm_wave_sync();
  double ipm_res = 0.0;                                                  // interior-point instance: this lane's (h − s)² terms
  // arm soft box (a6), joint-velocity box, friction cone barrier (a7).  All three are "relaxed barrier of h" evaluations: ONE branch-free
  // body serves lanes 0..11 (variable part of the boxes: lane < 6 position of arm joint l, else velocity), lanes 32..43 (the boxes' constant
  // offsets b(−lo) + b(hi): same code, other arguments) and lanes 16..19 (friction cone of contact l − 16); idle lanes evaluate h = 1.
  {
    const bool fric = (l >= 16 && l < 20); const int k = kj;
    const int kf = fric ? l - 16 : 0; const bool fon = fric && mode_flag(mode, kf);
    const double muf = st[ST_FRIC_COEF], reg = st[ST_FRIC_REG];
    const double Fx = U[3 * kf], Fy = U[3 * kf + 1], Fz = U[3 * kf + 2]; const double T2 = Fx * Fx + Fy * Fy + reg, Tn = sqrt(T2);
    const double mu_f = st[ST_FRIC_MU], mu_p = st[ST_JPOS_MU], mu_v = st[ST_JVEL_MU], de_f = st[ST_FRIC_DELTA], de_p = st[ST_JPOS_DELTA], de_v = st[ST_JVEL_DELTA];      // scalar table reads, values selected per lane
    const double mu = fric ? mu_f : (pos ? mu_p : mu_v), de = fric ? de_f : (pos ? de_p : de_v);
    const double lo = box_lo, hi = box_hi, z = pos ? X[24 + k] : U[24 + k];
    const double h1 = boxv ? z - lo : (boxc ? -lo : (fon ? muf * Fz - Tn : 1.0)), h2 = boxv ? hi - z : (boxc ? hi : 1.0);
    double v1, v2, p1, p2, q1, q2;
    if (!IPM) { v1 = barrier_val(mu, de, h1); v2 = barrier_val(mu, de, h2); barrier_d12(mu, de, h1, p1, p2); barrier_d12(mu, de, h2, q1, q2); }
    else {
      // interior-point instance (k_ipm.h): rows r1 / r2 of this lane (arm joint k: position rows 2k, 2k + 1, velocity rows 12 + 2k, 13 + 2k; contact: row 24 + kf) carry a slack s and a
      // dual lam.  CONDENSING puts  lam / s  where the soft cost has its second barrier derivative and  (lam h − mu_b) / s − lam  where it has the first ([upstream
      // ipm::condenseIneqConstraints]); no constraint curvature, no Hessian shift.  Node merit −mu_b ln s, node constraint term (h − s)² (both summed over the wave below).
      const bool act = boxv || fon; const int r1 = boxv ? (pos ? 2 * k : 12 + 2 * k) : (24 + kf);
      const double mub = a.ipm_info[b * 8];
      const double s1 = act ? a.ipm_s[(size_t)nb * QM_NH + r1] : 1.0, l1 = act ? a.ipm_l[(size_t)nb * QM_NH + r1] : 0.0;
      const double s2 = boxv ? a.ipm_s[(size_t)nb * QM_NH + r1 + 1] : 1.0, l2 = boxv ? a.ipm_l[(size_t)nb * QM_NH + r1 + 1] : 0.0;
      // (the condensed rows are NOT weighted by the interval length, the cost blocks they join are multiplied by dt below: hence the 1 / dt)
      const double i1 = 1.0 / s1, i2 = 1.0 / s2, idt = (dt != 0.0) ? 1.0 / dt : 0.0;      // (an interval of exactly zero length — a node exactly weakEpsilon in front of a gait event — carries no cost and no condensed rows: inf * 0 would poison R, r and the merit)
      p2 = l1 * i1 * idt; p1 = ((l1 * h1 - mub) * i1 - l1) * idt; q2 = l2 * i2 * idt; q1 = ((l2 * h2 - mub) * i2 - l2) * idt;
      if (!act) { p1 = 0.0; p2 = 0.0; } if (!boxv) { q1 = 0.0; q2 = 0.0; }
      v1 = act ? -mub * log(s1) * idt : 0.0; v2 = boxv ? -mub * log(s2) * idt : 0.0;
      ipm_res = (act ? (h1 - s1) * (h1 - s1) : 0.0) + (boxv ? (h2 - s2) * (h2 - s2) : 0.0);
    }
    cost += IPM ? (v1 + v2) : (boxv ? v1 + v2 : (boxc ? -(v1 + v2) : (fon ? v1 : 0.0)));
    if (boxv) { const double g1 = p1 - q1, g2 = p2 + q2; double* pv = (pos ? S + LW_V_QV : S + LW_V_RV) + 24 + k; double* pd = (pos ? QD : RD) + 24 + k; *pv += g1; *pd += g2; }      // (one region, the target vector selected by address)
    else if (fric) {                                                     // one lane per contact (disjoint 3x3 blocks)
      double* fr = FR + 16 * kf; double ds = 0.0;
      double fv[9] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};      // (the block is written ONCE: cleared first and then overwritten it cost this lane class 13 more LDS stores)
      if (fon) {
        const double shift = IPM ? 0.0 : st[ST_FRIC_SHIFT], iTn = IPM ? 1.0 / Tn : qm_frcp(Tn), iT3 = iTn * iTn * iTn;
        const double dh[3] = {-Fx * iTn, -Fy * iTn, muf};
        const double ddh[9] = {-(Fy * Fy + reg) * iT3, Fx * Fy * iT3, 0.0, Fx * Fy * iT3, -(Fx * Fx + reg) * iT3, 0.0, 0.0, 0.0, 0.0};
        for (int r = 0; r < 3; ++r) { S[LW_V_RV + 3 * kf + r] += p1 * dh[r]; for (int q = 0; q < 3; ++q) fv[3 * r + q] = p2 * dh[r] * dh[q] + (IPM ? 0.0 : p1 * ddh[3 * r + q]); }
        ds = p1 * (-shift);
      }
      for (int q = 0; q < 9; ++q) fr[q] = fv[q];
      fr[9] = 0.0; fr[10] = 0.0; fr[11] = 0.0; fr[12] = ds;
    } else if (l == 24) { for (int q = 0; q < 6; ++q) cost += 0.5 * EE[6 + q] * EE[q] * EE[q]; }
  }
  qm_wave_sync();
  const double dsum = FR[12] + FR[28] + FR[44] + FR[60];
  if (l < 30) { QD[l] += dsum; RD[l] += dsum; }      // the diagonals as they are added below, once per row instead of once per fragment register
  qm_wave_sync();
  // PREDICATION BY ADDRESS.  "Lane-conditional value from LDS" is written below as an unconditional read through a SELECTED ADDRESS — the value's slot, or a slot that holds
  // zero — followed by a plain addition (x + 0.0 is x): one v_cndmask_b32 on the address per register.  Written as `cond ? LDS[i] : 0.0` the compiler sinks the read into a
  // lane-conditional region (s_and_saveexec / s_cbranch_execz / s_or + the wait inside it) per register; 0 / 1 masks multiplied in cost a register pair each (two such sites
  // together spill).  Same bits; R assembly + Q assembly + the EE rows: K1b − 3 % (profiles/r06_ab_lq_regions.log)
  const double* const ZS = X + 30;                   // a double that holds zero (slots 30, 31 of the state vector are never written)
  LQT()
  // R = (R0 + diag + friction blocks + shift) dt
#pragma unroll
  for (int I = 0; I < 2; ++I)
#pragma unroll
    for (int J = 0; J < 2; ++J)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rl = g + 4 * r; double v = Rm[I][J][r];
        if (I == J) { const double* pd = (rl == c && (I == 0 || rl < 14)) ? RD + 16 * I + rl : ZS; v += *pd; }
        if (I == 0 && J == 0 && r < 3) { const int rb = rl / 3; const double* pf = (c / 3 == rb) ? FR + 16 * rb + 3 * (rl - 3 * rb) + (c % 3) : ZS; v += *pf; }
        Rm[I][J][r] = v * dt;
      }
  if (l < 30) S[LW_V_RV + l] *= dt;
  const double ctot = qm_wave_sum(cost) * dt;
  const double ineq2 = IPM ? qm_wave_sum(ipm_res) : 0.0;
  if (l == 0) { a.perf[nb * PF_SIZE] = ctot; a.perf[nb * PF_SIZE + 1] = dt * b2; a.perf[nb * PF_SIZE + 2] = dt * (eq2 + ineq2); }
  // keep Pu safe in registers?  It stays in the tile: the EE term uses its own small staging area (K2 is dead by now)
  double* JT = S + LW3_JT;                                              // [6][32] J rows, in the idle hand-over tile
  qm_wave_sync();
  if (l < 30) { double col[6]; ee_jac_col(X, K1, EE + 12, EE + 19, l, col); for (int r = 0; r < 6; ++r) JT[r * 32 + l] = col[r]; }
  if (l >= 30 && l < 32) for (int r = 0; r < 6; ++r) JT[r * 32 + l] = 0.0;
  qm_wave_sync();
  // [Q | q] = ([diag + shift | q] + Jᵀ mu [J | g]) dt
  qm_d4 Qa[2][2];
  {
    qm_d4 Jz[1][2], Jy[1][2];
#pragma unroll
    for (int J = 0; J < 2; ++J)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double jv = 0.0, jy = 0.0;
        if (r < 2) {
          const int row = g + 4 * r, col = 16 * J + c; const bool live = (r == 0) || g < 2; const double* const ZJ = X + 30;
          const double mu = EE[6 + row];
          jv = *(live ? JT + row * 32 + col : ZJ);
          if (J == 1) { const double ge = *((c == 14 && live) ? EE + row : ZJ); jy = mu * (jv + ge); } else jy = mu * jv;
        }
        Jz[0][J][r] = jv; Jy[0][J][r] = jy;
      }
#pragma unroll
    for (int I = 0; I < 2; ++I)
#pragma unroll
      for (int J = 0; J < 2; ++J)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int rl = g + 4 * r, row = 16 * I + rl; double v = 0.0;
          if (I == J) { const double* pd = (rl == c && (I == 0 || rl < 14)) ? QD + row : ZS; v = *pd; }
          if (J == 1) { const double* pq = (c == 14) ? S + LW_V_QV + row : ZS; v += *pq; }
          Qa[I][J][r] = v;
        }
    qm_gemm_tn<1, 2, 2>(Jz, Jy, Qa, 0, 2, false);
#pragma unroll
    for (int I = 0; I < 2; ++I)
#pragma unroll
      for (int J = 0; J < 2; ++J) Qa[I][J] *= dt;
  }
  if (DBG && dbg) {
    qm_frag_store<2, 2>(Qa, dbg + LQ_DBG_Q, 30, 30, 30); qm_frag_store<2, 2>(Rm, dbg + LQ_DBG_R, 30, 30, 30);
#pragma unroll
    for (int I = 0; I < 2; ++I)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const int row = 16 * I + g + 4 * r; if (c == 14 && row < 30) dbg[LQ_DBG_q + row] = Qa[I][1][r]; }
    if (l < 30) dbg[LQ_DBG_r + l] = S[LW_V_RV + l]; if (l == 0) dbg[LQ_DBG_c] = ctot;
  }
  LQT()
  // ---- projected cost + stores ----
  // Px: only rows 12..23 (leg joint velocities) are non-zero and only they are stored / read back by K3 (register (I = 0, r = 3), (1, 0), (1, 1))
#pragma unroll
  for (int J = 0; J < 2; ++J) {
    const int col = 16 * J + c;
    if (col < 30) { QM_STREAM_ST(rec + SR_PX + (12 + g) * 30 + col, PxA[0][J][3]); QM_STREAM_ST(rec + SR_PX + (16 + g) * 30 + col, PxA[1][J][0]); QM_STREAM_ST(rec + SR_PX + (20 + g) * 30 + col, PxA[1][J][1]); }
  }
  double rpe = 0.0;
  lw_project<MT>(S, rec, (DBG && dbg != nullptr) ? dbg + LQ_DBG_PU : nullptr, m, PxA, PuF, Rm, Qa, rpe);
  // what K3's forward rollout needs to apply Pu without reading it: the swing legs' null-space blocks and the contact mode
  if (l < 24) rec[SR_SWG + l] = G[12 * (l / 6) + 3 + (l % 6)];
  if (l == 24) rec[SR_MODEF] = (double)mode;
  if (l == 25) rec[SR_MODEF + 1] = dt;                 // the joint rows of the projected dynamics are not read back by the rollout: x_j+ = x_j + dt u_j
  if (l == 26 || l == 27) rec[SR_MODEF + l - 24] = (l == 26) ? 1.0 : 0.0;      // the constants K3's Bp = dt Pu rebuild reads behind (swing blocks, mode, dt): copied into LDS with them
  if (l < 30) rec[SR_PE + l] = S[LW_V_PE + l];
  if (l == 0) { rec[SR_SCAL] = (double)m; rec[SR_SCAL + 1] = ctot + rpe; }
  LQT()
  if (DBG && a.prof && l == 0) for (int k = 0; k + 1 < np_ && k < 9; ++k) rec[SR_K + k] = (double)(tp_[k + 1] - tp_[k]);
  // [9] wave lifetime in cycles (entry -> all stores acknowledged), [10] the same in 10 ns ticks, [11], [12] entry / exit tick, [13], [14] HW_ID / XCC_ID of the slot,
  // [15] entry -> first stamp, [16] last stamp -> vmcnt 0: clock under load, resident waves per SIMD, slot gaps, prologue and store-drain shares
  if (DBG && a.prof && l == 0) { __builtin_amdgcn_s_waitcnt(0); const long long c1_ = (long long)__builtin_readcyclecounter(); const long long r1_ = (long long)__builtin_amdgcn_s_memrealtime();
    rec[SR_K + 9] = (double)(c1_ - c0_); rec[SR_K + 10] = (double)(r1_ - r0_); rec[SR_K + 11] = (double)r0_; rec[SR_K + 12] = (double)r1_;
    rec[SR_K + 13] = (double)__builtin_amdgcn_s_getreg(63492); rec[SR_K + 14] = (double)__builtin_amdgcn_s_getreg(63508); rec[SR_K + 15] = (double)(tp_[0] - c0_); rec[SR_K + 16] = (double)(c1_ - tp_[np_ - 1]); }
#undef LQT
}
__device__ __forceinline__ int qm_lq_node_mt(const QmLqArgs& a) {
  const int b = blockIdx.x / a.ncap, i = a.i0 + blockIdx.x - b * a.ncap;
  const int mode = a.node_mode[i * a.B + b];
  int nst = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) nst += mode_flag(mode, k);
  return (3 * nst + 2 * (4 - nst) + 6 <= 16) ? 1 : 2;
}
#ifndef QM_LQ_WAVES
#define QM_LQ_WAVES 3      /* waves per SIMD the two product instances are compiled for */
#endif
#ifdef QM_LQ_KERNELS_EXTERN      /* the main translation unit (qmhip.hip): the four instances of the K1b body are compiled in qmhip_lq.hip with their own scheduling strategy
                                    (qm_control_amd/build_flags.py: max-ilp pays for this kernel — 2.4 waves per SIMD — and costs the lone-wave kernels 5 %) */
__global__ void qm_lq_kernel(QmLqArgs a); __global__ void qm_lq_m18_kernel(QmLqArgs a); __global__ void qm_lq_ipm_kernel(QmLqArgs a); __global__ void qm_lq_dbg_kernel(QmLqArgs a);
#else
__global__ void QM_UNPAIRED_LDS __launch_bounds__(LW_BLOCK, QM_LQ_WAVES) qm_lq_kernel(QmLqArgs a) { if (a.single_mt || qm_lq_node_mt(a) == 1) qm_lq_body<false, 1>(a); }
__global__ void QM_UNPAIRED_LDS __launch_bounds__(LW_BLOCK, QM_LQ_WAVES) qm_lq_m18_kernel(QmLqArgs a) { if (qm_lq_node_mt(a) == 2) qm_lq_body<false, 2>(a); }
__global__ void QM_UNPAIRED_LDS __launch_bounds__(LW_BLOCK, 2) qm_lq_ipm_kernel(QmLqArgs a) { if (qm_lq_node_mt(a) == 1) qm_lq_body<false, 1, true>(a); else qm_lq_body<false, 2, true>(a); }      // interior-point instance (solver 3, k_ipm.h)
__global__ void QM_UNPAIRED_LDS __launch_bounds__(LW_BLOCK, 2) qm_lq_dbg_kernel(QmLqArgs a) { if (qm_lq_node_mt(a) == 1) qm_lq_body<true, 1>(a); else qm_lq_body<true, 2>(a); }
#endif
