// k_riccati.h — K3: discrete-time Riccati backward sweep + forward rollout of the projected QP.
//
// ONE WAVEFRONT per MPC instance (64-thread workgroups, 38 KB LDS -> four per CU): the sweep is a serial chain of small dense products, so
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
// The Cholesky factorisation of Huu and the forward substitution of [Hux | hu] stay in fragment layout as well: four pivots per block,
// two MFMAs per tile and block with all four k-slots live (see rw_stage).  Only the symmetrisation of S' takes a wave-local LDS
// round trip.  (On gfx950 f64 MFMA and f64 VALU share one rate and do not overlap: the MFMA buys the data movement, not flops.)
// The operands of the NEXT stage are copied global -> LDS asynchronously (global_load_lds_dwordx4, 1 KB per wave instruction, no
// VGPRs) while the current stage computes; a stage starts by pulling its fragments out of that buffer.
//
// The backward sweep leaves the gain K = −L⁻ᵀ W (in SR_PP) and the offset k = −L⁻ᵀ y (in SR_KFF) in the stage record; the forward rollout
//   ut = K dx + k,  dx+ = Ap dx + Bp ut + bp,  du = Px dx + Pu ut + Pe,  Armijo metric += qp·dx + rp·ut
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
  double* step_info;                           // [B][4]: armijo, |dx|², |du|², pivot flags (as a double): bit 0 = a stage of NON-POSITIVE duration had non-positive pivots of Huu (zeroed, see rw_stage: the warning QM_MPC_WARN_PIVOT), bit 1 = a stage of positive duration had one, or a pivot was not a number (hard failure -4)
  // baseline performance of the current iterate (sum of K1b's node terms) + arming of the line search, done by the instance's wave before the sweep
  // (what a separate one-wave-per-instance launch did: qm_perf_sum_kernel with with_alpha == 0); perf == nullptr: skipped
  const double* perf; double* base_sum; double* alpha; int* done; double* out_perf; int* open_cnt; int* tickets;
  int skip;                                    // profiling only (bit mask: 1 Cholesky/solve, 2 matrix products, 4 forward, 8 symmetrise, 16 all regular backward stages, 64 lean operand prefetch (rw_prefetch), 256 no gain stores, 512 the rollout's fetch without its last two chunks, 128 nothing (the instrumented instance as it is): results are then
                                               // meaningless; 32: results intact, per-phase cycle counts are written to SR_K of each instance's first stage record)
};

#define RW_BLOCK 64
#define RW_TLD 34                 /* transposition buffer [32][34] (33 measured: no difference here) */
/* forward staging (aliases the backward buffers): two flat copies (double buffer) of the fields of a stage record the rollout reads, in the order of the fetch
   list below, written by global_load_lds (no VGPR staging, 1 KB per wave instruction) one stage ahead of their use */
#define RF_F0   0                 /* buffer of the even regular stages */
#define RF_F1   1952              /* buffer of the odd ones */
#define RF_ZERO 3904              /* [32] zeros: the row the lanes without a Px / Bp row read */
#define RF_PUD  3936              /* [30][18] Pu as a dense matrix: unit columns per contact mode, the swing legs' 3x2 blocks refreshed per stage */
/* offsets inside one buffer */
#define RFO_A   0                 /* [12][30] Ap rows 0..11 */
#define RFO_B   360               /* [12][18] Bp rows 0..11 */
#define RFO_W   576               /* [18][30] K = −L⁻ᵀ W */
#define RFO_PX  1116              /* [12][30] Px rows 12..23 (the only non-zero ones: leg joint velocities) */
#define RFO_V   1476              /* bp(30) qp(30) rp(18) Pe(30) | k(18) | swing blocks [4][6] (126..149) mode (150) dt (151) */
/* backward prefetch buffer (global_load_lds): the fields of the NEXT regular stage's record the backward sweep reads, landing while the current stage computes;
   lives behind the 1200-double Cholesky / transposition buffer.  Six segments, each padded to whole 1 KB wave instructions so that an instruction's source
   offset is a compile-time constant.  The joint rows of the projected dynamics are NOT in the record: a joint row of the Heun-discretised flow map is exactly
   x_j+ = x_j + dt u_j, so  Ap[j] = e_j + dt Px[j],  Bp[j] = dt Pu[j]  (j >= 12) are rebuilt from Px, the contact mode and the swing legs' 3x2 blocks */
#define RP_REC   1200
#define RPO_A    0                 /* [12][30] Ap rows 0..11                     <- SR_AP            360 */
#define RPO_B    384               /* [12][18] Bp rows 0..11                     <- SR_BP            216 */
#define RPO_Q    640               /* [Qp | qp], [Pp | rp], Rp in fragment order <- SR_FRAG          1536 (+ 448 when m > 16) */
#define RPO_PX   2688              /* [12][30] Px rows 12..23                    <- SR_PX + 360      360 */
#define RPO_VEC  3072              /* bp(30) qp(30) rp(18)                       <- SR_BPV            78 */
#define RPO_SWG  3200              /* swing blocks [4][6], mode, dt              <- SR_SWG            26 */
#define RP_END   (RP_REC + 3328)
#define RF_LIST  4528             /* int list[RW_MAXNODES]: m | event tag << 8 per node (behind both the prefetch buffer and the forward staging) */
#define RW_MAXNODES 512
#define RW_LDS_DOUBLES (RF_LIST + RW_MAXNODES / 2)
#define RW_LDS_BYTES (RW_LDS_DOUBLES * 8)
#define RF_NLOAD 13               /* ceil((360 + 216 + 540 + 360 + 108 + 18 + 24 + 2) / 128): sixteen-byte units, 64 per wave instruction */

template <int KT, int IT, int JT>
__device__ __forceinline__ void rw_gemm_tn(const qm_d4 (&Z)[KT][IT], const qm_d4 (&Y)[KT][JT], qm_d4 (&P)[IT][JT], int ksteps, bool neg) { qm_gemm_tn<KT, IT, JT>(Z, Y, P, 0, ksteps, neg); }
template <int IT, int JT>
__device__ __forceinline__ void rw_zero(qm_d4 (&T)[IT][JT]) { qm_frag_zero<IT, JT>(T); }
// P += (−) Zᵀ Y for a SYMMETRIC 2x2-tile result: only the tiles (0,0), (0,1), (1,1) are formed — the lower-left tile is the mirror image
// of (0,1) and is restored by the symmetrisation at the end of the stage (12 MFMAs less per stage)
template <int KT>
__device__ __forceinline__ void rw_gemm_tn_upper(const qm_d4 (&Z)[KT][2], const qm_d4 (&Y)[KT][2], qm_d4 (&P)[2][2], int ksteps, bool neg) {
  qm_d4 Z0[KT][1], Z1[KT][1], Y1[KT][1], P0[1][2], P1[1][1];
#pragma unroll
  for (int K = 0; K < KT; ++K) { Z0[K][0] = Z[K][0]; Z1[K][0] = Z[K][1]; Y1[K][0] = Y[K][1]; }
  P0[0][0] = P[0][0]; P0[0][1] = P[0][1]; P1[0][0] = P[1][1];
  qm_gemm_tn<KT, 1, 2>(Z0, Y, P0, 0, ksteps, neg);
  qm_gemm_tn<KT, 1, 1>(Z1, Y1, P1, 0, ksteps, neg);
  P[0][0] = P0[0][0]; P[0][1] = P0[0][1]; P[1][1] = P1[0][0];
}
// D-layout load of a rows x cols row-major matrix (leading dim ld); optional vector in column 30 (rows < rows)
// (measured: exec-masked conditional loads are faster here than unconditional clamped loads + selects)
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

// asynchronous copy of the backward operands of one stage record into the LDS prefetch buffer: 24 wave instructions of 1 KB.  Every instruction copies a FULL
// 1 KB chunk: a segment's last chunk runs past the segment's end into the fields that follow it in the record (all sources end inside the record) and lands in
// the segment's padding in LDS — no per-chunk lane mask.  The lane's byte offset 16 l is one 32-bit register for all of them, the chunk's address a wave-uniform
// base (scalar registers): global_load_lds with scalar base + vector offset instead of a 64-bit vector address per instruction.
// (round 6: four chunks share one global base and one LDS base — the instruction's immediate offset moves both sides, qm_dma16_at)
template <int SRC, int DST, int LEN>
__device__ __forceinline__ void rw_prefetch_seg(const double* rec, qm_lds_ptr lds, unsigned lane_bytes) {
#pragma unroll
  for (int t0 = 0; t0 * 128 < LEN; t0 += 4) {
    const char* g = (const char*)(rec + SRC + 128 * t0) + lane_bytes; const qm_lds_ptr l3 = lds + 8 * (RP_REC + DST + 128 * t0);
    qm_dma16_at<0>(g, l3);
    if ((t0 + 1) * 128 < LEN) qm_dma16_at<1024>(g, l3);
    if ((t0 + 2) * 128 < LEN) qm_dma16_at<2048>(g, l3);
    if ((t0 + 3) * 128 < LEN) qm_dma16_at<3072>(g, l3);
  }
}
// lean (profiling only, skip bit 64 of the instrumented instance): three of the twelve fragment chunks are NOT copied — the 360 doubles per stage that packed triangles of the
// symmetric tiles Qp(0,0), Qp(1,1), Rp would save (round-5 review item 5).  The arithmetic then runs on stale operands (results meaningless); what is measured is the TIME
// of a backward sweep that moves 7 % fewer bytes and pays nothing for unpacking them: the upper bound of what the packing could gain.
__device__ __forceinline__ void rw_prefetch(const double* rec, double* lds_generic, int m, bool lean = false) {
  const qm_lds_ptr lds = qm_lds(lds_generic);
  const unsigned lane_bytes = 16u * (threadIdx.x & 63);
  rw_prefetch_seg<SR_AP, RPO_A, 360>(rec, lds, lane_bytes);
  rw_prefetch_seg<SR_BP, RPO_B, 216>(rec, lds, lane_bytes);
  if (lean) rw_prefetch_seg<SR_FRAG, RPO_Q, SR_F_PP1 - 384>(rec, lds, lane_bytes); else
  rw_prefetch_seg<SR_FRAG, RPO_Q, SR_F_PP1>(rec, lds, lane_bytes);
  if (m > 16) rw_prefetch_seg<SR_FRAG + SR_F_PP1, RPO_Q + SR_F_PP1, SR_F_SIZE - SR_F_PP1>(rec, lds, lane_bytes);      // wave-uniform: the second tile row of the reduced inputs
  rw_prefetch_seg<SR_PX + 360, RPO_PX, 360>(rec, lds, lane_bytes);
  rw_prefetch_seg<SR_BPV, RPO_VEC, 78>(rec, lds, lane_bytes);
  rw_prefetch_seg<SR_SWG, RPO_SWG, 26>(rec, lds, lane_bytes);      // brings SR_MODEF (mode, dt) and the constants 1.0, 0.0 K1b keeps behind them
}
static_assert(SR_AP + 384 <= SR_SIZE && SR_BP + 256 <= SR_SIZE && SR_FRAG + SR_F_SIZE <= SR_SIZE && SR_F_PP1 % 128 == 0 && SR_F_SIZE % 128 == 0 && SR_PX + 360 + 384 <= SR_SIZE && SR_BPV + 128 <= SR_SIZE && SR_SWG + 128 <= SR_SIZE, "a full 1 KB chunk must end inside the record");
static_assert(RPO_A + 384 <= RPO_B && RPO_B + 256 <= RPO_Q && RPO_Q + SR_F_SIZE <= RPO_PX && RPO_PX + 384 <= RPO_VEC && RPO_VEC + 128 <= RPO_SWG && RP_REC + RPO_SWG + 128 <= RP_END, "segments are padded to whole chunks");
// [Ap | bp] as D-layout fragments: rows 0..11 from the record, rows 12..23 = e_j + dt Px[j], rows 24..29 = e_j (arm joints), column 30 = bp.
// Unconditional loads + selects: a read at column 30 / 31 of a 30-wide row lands in the next row (inside the buffer) and is replaced afterwards — the exec-mask
// bookkeeping of conditional loads (save / branch / restore per element) costs more here than the selects
__device__ __forceinline__ void rw_load_A(qm_d4 (&A)[2][2], const double* PA, const double* PX, const double* bp, double dt) {
  const int g = (threadIdx.x & 63) >> 4, c = threadIdx.x & 15;
#pragma unroll
  for (int I = 0; I < 2; ++I)
#pragma unroll
    for (int J = 0; J < 2; ++J)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * I + g + 4 * r, col = 16 * J + c; double v;     // (I, r) decides the row class at compile time: rows 0..11 are I = 0, r < 3
        if (I == 0 && r < 3) v = PA[row * 30 + col];
        else if (I == 0 || r < 2) v = fma(dt, PX[(row - 12) * 30 + col], (row == col) ? 1.0 : 0.0);
        else v = (row == col) ? 1.0 : 0.0;
        if (J == 1) { const double bv = bp[row]; v = (c == 14) ? bv : ((c == 15) ? 0.0 : v); }      // bp is followed by qp in the buffer: rows 30, 31 read in bounds
        if (I == 1 && r == 3) v = (g < 2) ? v : 0.0;                                               // rows 30, 31
        A[I][J][r] = v;
      }
}
// per contact mode: where the entries of  Bp[j] = dt Pu[j]  (j >= 12) of this lane's five row slots come from.  Leg joint rows: the two null-space columns of the
// joint's leg if it swings (code = index into the swing blocks), nothing if it stands; arm joint rows: a unit entry in the arm's column.  The swing blocks in LDS are
// followed by mode, dt and the constants 1.0 (index 26) and 0.0 (index 27), so every entry is dt * SWG[code] without a select.  Five 5-bit codes per column tile in one int
struct RwPuCodes { int mode; int pk[2]; };
__device__ __forceinline__ void rw_pu_codes(RwPuCodes& pc, int md) {
  const int g = (threadIdx.x & 63) >> 4, c = threadIdx.x & 15;
  pc.mode = md;
  int nst = 0;
#pragma unroll
  for (int kq = 0; kq < 4; ++kq) nst += mode_flag(md, kq);
#pragma unroll
  for (int J = 0; J < 2; ++J) {
    int pk = 0;
#pragma unroll
    for (int sl = 0; sl < 5; ++sl) {
      const int row = 12 + g + 4 * sl, col = 16 * J + c; int code = 27;        // slots: (I = 0, r = 3), (I = 1, r = 0 .. 3) -> rows 12 + g + 4 sl
      if (row < 24) {
        const int chain = (row - 12) / 3, r3 = (row - 12) % 3, kk = chain_to_contact(chain);
        int before_sw = 0;
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) if (kq < kk) before_sw += !mode_flag(md, kq);
        const int t = col - (3 * nst + 2 * before_sw);
        if (!mode_flag(md, kk) && (t == 0 || t == 1)) code = 6 * kk + 3 * t + r3;
      } else if (row < 30) { if (col == 3 * nst + 2 * (4 - nst) + (row - 24)) code = 26; }
      pk |= code << (5 * sl);
    }
    pc.pk[J] = pk;
  }
}
// Bp (30 x m) as D-layout fragments: rows 0..11 from the record, rows 12..29 = dt Pu
template <int MT>
__device__ __forceinline__ void rw_load_B(qm_d4 (&Bm)[2][MT], const double* PB, const double* SWG, const RwPuCodes& pc, double dt, int m) {
  const int g = (threadIdx.x & 63) >> 4, c = threadIdx.x & 15;
#pragma unroll
  for (int I = 0; I < 2; ++I)
#pragma unroll
    for (int J = 0; J < MT; ++J)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * I + g + 4 * r, col = 16 * J + c; double v = 0.0;
        if (I == 0 && r < 3) { const double e = PB[row * QM_MMAX + col]; v = (col < m) ? e : 0.0; }      // unconditional read (in bounds), then the select
        else v = dt * SWG[(pc.pk[J] >> (5 * (4 * I + r - 3))) & 31];
        Bm[I][J][r] = v;
      }
}
// one regular stage of the backward sweep; MT = number of 16-row tiles covering the m reduced inputs
// PROF: the instrumented instance (qm_riccati_prof_kernel) — phase skip bits and in-kernel cycle counters; the product instance carries neither (the eight 64-bit
// accumulators and the skip tests cost scalar registers — spilled to vector-register lanes around every phase boundary — and branches on the hot path)
template <int MT, bool PROF>
__device__ __forceinline__ void rw_stage(double* rec, int m, const double* nrec, int mnext, double* buf, qm_d4 (&S)[2][2], qm_d4 (&sv)[2], int skip_arg, int& chol_fail, long long (&tacc)[8], RwPuCodes& pc, int md) {
  const int l = threadIdx.x & 63, g = l >> 4, c = l & 15;
  const int skip = PROF ? skip_arg : 0;
  const bool prof = PROF && (skip & 32) != 0; long long tq_ = prof ? (long long)__builtin_readcyclecounter() : 0;
#define RWT(i) { if (PROF && prof) { const long long t_ = (long long)__builtin_readcyclecounter(); tacc[i] += t_ - tq_; tq_ = t_; } }
  qm_d4 A[2][2], Bm[2][MT], Hux[MT][2], Huu[MT][MT], Sn[2][2];
  int failbit;                                                    // what a non-positive pivot of this stage means: 1 benign (duration <= 0), 2 hard failure
  {
    // this stage's operands were copied into LDS (asynchronously, global_load_lds) while the previous stage computed
    const double* P = buf + RP_REC; const double* PV = P + RPO_VEC;
    if (md != pc.mode) rw_pu_codes(pc, md);                       // wave-uniform (the mode comes from the node list): a few times per sweep, ahead of the wait
    qm_dma_wait();
    const double dt = P[RPO_SWG + 25]; failbit = (dt <= 0.0) ? 1 : 2;
    rw_load_A(A, P + RPO_A, P + RPO_PX, PV, dt);                    // [Ap | bp]
    rw_load_B<MT>(Bm, P + RPO_B, P + RPO_SWG, pc, dt, m);
    // [Pp | rp], Rp, [Qp | qp]: K1b left them in fragment order (SR_FRAG) — one LDS load per register, no masks, no per-element addresses
    { const double* F = P + RPO_Q + l;
#pragma unroll
      for (int J = 0; J < 2; ++J)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          Hux[0][J][r] = F[SR_F_PP + (4 * J + r) * 64];
          if (MT == 2) Hux[MT - 1][J][r] = (r == 0) ? F[SR_F_PP1 + J * 64] : 0.0;
        }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        Huu[0][0][r] = F[SR_F_RP + r * 64];
        if (MT == 2) { Huu[0][MT - 1][r] = F[SR_F_RP01 + r * 64]; Huu[MT - 1][0][r] = 0.0; Huu[MT - 1][MT - 1][r] = (r == 0) ? F[SR_F_RP11] : 0.0; }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) { Sn[0][0][r] = F[SR_F_QP + r * 64]; Sn[0][1][r] = F[SR_F_QP + (4 + r) * 64]; Sn[1][1][r] = F[SR_F_QP + (8 + r) * 64]; Sn[1][0][r] = 0.0; } }
    qm_lds_drain();
    if (nrec) rw_prefetch(nrec, buf, mnext, PROF && (skip & 64));                        // next regular stage: flies during this stage's products and Cholesky
  }
  RWT(0)
  qm_d4 SA[2][2], SB[2][MT];
  rw_zero<2, 2>(SA); rw_zero<2, MT>(SB);
  if (!(skip & 2)) {
    // [S A | S b].  Rows 24..29 of A are unit rows (arm joints: x_j+ = x_j + dt u_j, no Px), rows 30, 31 padding: in the left column tile (columns < 16) the last two
    // k-steps (rows 24..31 of A) multiply zeros and are skipped; the right one keeps them — column 30 carries b
    { qm_d4 A0[2][1], A1[2][1], P0[2][1], P1[2][1];
#pragma unroll
      for (int K = 0; K < 2; ++K) { A0[K][0] = A[K][0]; A1[K][0] = A[K][1]; P0[K][0] = SA[K][0]; P1[K][0] = SA[K][1]; }
      rw_gemm_tn<2, 2, 1>(S, A0, P0, 6, false);
      rw_gemm_tn<2, 2, 1>(S, A1, P1, 8, false);
#pragma unroll
      for (int K = 0; K < 2; ++K) { SA[K][0] = P0[K][0]; SA[K][1] = P1[K][0]; } }
#pragma unroll
    for (int I = 0; I < 2; ++I) SA[I][1] += sv[I];                    // column 30 += s
    rw_gemm_tn<2, 2, MT>(S, Bm, SB, 8, false);                       // S B
    rw_gemm_tn<2, MT, 2>(Bm, SA, Hux, 8, false);                     // [Hux | hu]
    rw_gemm_tn<2, MT, MT>(Bm, SB, Huu, 8, false);                    // Huu
    // [Q + Aᵀ S A | q + Aᵀ (S b + s)], upper tiles (row 30 is garbage, masked below).  The unit rows 24..29 of A contribute row k of (S A) to row k of the result —
    // the same fragment position: six k-steps on the matrix core and one lane-local addition instead of eight k-steps (the additions come last either way: same rounding)
    rw_gemm_tn_upper<2>(A, SA, Sn, 6, false);
    Sn[1][1][2] += SA[1][1][2];                                        // rows 24..27
    Sn[1][1][3] += (g < 2) ? SA[1][1][3] : 0.0;                        // rows 28, 29
  }
  RWT(1)
  // ---- Cholesky of Huu and forward substitution of [Hux | hu] IN FRAGMENT LAYOUT (right-looking).
  // Row j of a D-layout matrix is register (j&15)>>2 of lane group g = j&3, i.e. it already is k-slot j&3 of an MFMA B operand, and
  // — Huu being symmetric — the same register read as an A operand supplies column j: the updates of the trailing rows of
  // [Huu | Hux hu] are MFMAs on the fragments as they are, no LDS staging and no layout change, and W = L⁻¹[Hux | hu] comes out as
  // fragments, ready for Wᵀ W.  Only the upper triangle of Huu is ever read; a row keeps its value at elimination time: L_jj · (row j of Lᵀ resp. W).
  qm_d4 W[MT][2];
  if (!(skip & 1)) {
    double dsel[MT][4];                                              // pivot d_row of this lane's rows (1 on padding rows)
#pragma unroll
    for (int I = 0; I < MT; ++I)
#pragma unroll
      for (int r = 0; r < 4; ++r) dsel[I][r] = 1.0;
    RWT(2)
    // ---- BLOCKED elimination, four pivots at a time.  Rows 4b .. 4b+3 of a D-layout tile are register b of the four lane groups, i.e. exactly the four k-slots
    // of an MFMA B operand.  Per block: the 4 x 4 diagonal block is read with v_readlane and factored as L~ Δ L~ᵀ in wave-uniform scalars (one reciprocal chain per
    // pivot); then ONE MFMA per tile replaces the block rows by L~⁻¹ (block rows) (A = L~⁻¹ − I on the block's rows) and ONE more subtracts
    // Σ_k (R'[k][row] / d_k) R'[k] from the trailing rows (A = −R'ᵀ Δ⁻¹ masked to the rows behind the block) — MFMAs with all four k-slots live instead of the
    // single-slot rank-1 MFMAs of four consecutive pivots, and one dependent chain per block instead of four.  Row j ends up as L_jj · (row j of Lᵀ resp. W).
    // The same row operations run on an identity tile E, which therefore ends as L~⁻¹: scaled by Δ^(-1/2) it is L⁻¹, what the forward rollout multiplies with
    // (a triangular SOLVE there is an 18-step dependent chain on a lone wave; a product with L⁻ᵀ is not).
    // Padding rows (>= m) get a unit diagonal: their pivots are 1, their rows stay zero.
    qm_d4 E[MT][MT];
#pragma unroll
    for (int I = 0; I < MT; ++I)
#pragma unroll
      for (int J = 0; J < MT; ++J)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool diag = (I == J) && (c == g + 4 * r);
          E[I][J][r] = diag ? 1.0 : 0.0;
          if (diag && 16 * I + g + 4 * r >= m) Huu[I][I][r] = 1.0;
        }
    auto recip = [](double d) { double x = __builtin_amdgcn_rcp(d); const double e = fma(-d, x, 1.0); return fma(fma(e, e, e), x, x); };   // 2^-24 estimate + one third-order step
    constexpr int NBLK = (MT == 1) ? 4 : 5;                              // m <= 16: rows 0..15;  m = 17, 18: one more block in the second tile row
    if (MT == 1) {
      // ---- ONE tile row (m <= 16), SOFTWARE PIPELINED over the blocks (round 6).  A wave issues in order, so the seven MFMAs of a block written back to back (as the general
      // loop below is compiled) are followed by the next block's pivot chain — read the 4 x 4 diagonal block with v_readlane, four dependent reciprocals — with the matrix
      // pipe idle.  Only the two MFMAs on Huu are on the path to the next block's pivots; the six on [Hux | hu] and on the identity tile are not.  So: the two Huu MFMAs
      // first, then the next block's pivot chain in five pieces with one of the six other MFMAs in front of each piece (scheduling barriers keep the order; the operand
      // selects are flat — a nested select compiles into lane-conditional branches, which end the scheduling region and let the MFMAs clump again).  Same operations on
      // the same operands in the same per-tile order: same bits.  Measured − 1.3 % of the kernel (profiles/r06_ab_riccati_chol_pipeline.log), a quarter of what the cycle
      // budget of the two chains promised: the FP64 vector instructions of the pivot chain and the f64 MFMA share the FP64 datapath on gfx950 (equal peak rates), so only
      // the chain's lane reads, selects and moves run beside an MFMA.
      qm_d4& Hd = Huu[0][0];
      double d0, d1, d2, d3, rd0, rd1, rd2, rd3, M10, M20, M21, M30, M31, M32;
      double D00, D01, D02, D03, D11, D12, D13, D22, D23, D33, l10, l20, l30, t12, t13, l21, l31, t23, l32;
#define RW_SB() __builtin_amdgcn_sched_barrier(0)
#define RW_PIV1(rb_) { const int q0_ = 4 * (rb_); D00 = qm_bcast(Hd[rb_], q0_); D01 = qm_bcast(Hd[rb_], q0_ + 1); D02 = qm_bcast(Hd[rb_], q0_ + 2); D03 = qm_bcast(Hd[rb_], q0_ + 3); \
        D11 = qm_bcast(Hd[rb_], 16 + q0_ + 1); D12 = qm_bcast(Hd[rb_], 16 + q0_ + 2); D13 = qm_bcast(Hd[rb_], 16 + q0_ + 3); \
        D22 = qm_bcast(Hd[rb_], 32 + q0_ + 2); D23 = qm_bcast(Hd[rb_], 32 + q0_ + 3); D33 = qm_bcast(Hd[rb_], 48 + q0_ + 3); \
        d0 = D00; rd0 = (d0 > 0.0) ? recip(d0) : 0.0; }
#define RW_PIV2() { l10 = D01 * rd0; l20 = D02 * rd0; l30 = D03 * rd0; d1 = fma(-l10, D01, D11); rd1 = (d1 > 0.0) ? recip(d1) : 0.0; }
#define RW_PIV3() { t12 = fma(-l20, D01, D12); t13 = fma(-l30, D01, D13); l21 = t12 * rd1; l31 = t13 * rd1; d2 = fma(-l21, t12, fma(-l20, D02, D22)); rd2 = (d2 > 0.0) ? recip(d2) : 0.0; }
#define RW_PIV4() { t23 = fma(-l31, t12, fma(-l30, D02, D23)); l32 = t23 * rd2; d3 = fma(-l32, t23, fma(-l31, t13, fma(-l30, D03, D33))); rd3 = (d3 > 0.0) ? recip(d3) : 0.0; }
#define RW_PIV5() { chol_fail |= ((d0 > 0.0) & (d1 > 0.0) & (d2 > 0.0) & (d3 > 0.0)) ? 0 : failbit; \
        M10 = -l10; M21 = -l21; M32 = -l32; M20 = fma(l21, l10, -l20); M31 = fma(l32, l21, -l31); M30 = -(l30 + l31 * M10 + l32 * M20); }
      RW_PIV1(0) RW_PIV2() RW_PIV3() RW_PIV4() RW_PIV5()
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
        const int c0 = 4 * rb, ri = c - c0;
        double a1 = 0.0;                                                  // A[i = c][k = g] = (L~⁻¹ − I)[ri][g]: flat selects — the nested form compiles into lane-conditional branches, which end the scheduling region
        a1 = (ri == 1 && g == 0) ? M10 : a1; a1 = (ri == 2 && g == 0) ? M20 : a1; a1 = (ri == 2 && g == 1) ? M21 : a1;
        a1 = (ri == 3 && g == 0) ? M30 : a1; a1 = (ri == 3 && g == 1) ? M31 : a1; a1 = (ri == 3 && g == 2) ? M32 : a1;
        const double dg = (g == 0) ? d0 : ((g == 1) ? d1 : ((g == 2) ? d2 : d3)), rdg = (g == 0) ? rd0 : ((g == 1) ? rd1 : ((g == 2) ? rd2 : rd3));
        dsel[0][rb] = dg;
        Hd = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, Hd[rb], Hd, 0, 0, 0);
        double a2 = 0.0;
        if (rb < 3) { a2 = (c > c0 + 3) ? -rdg * Hd[rb] : 0.0; Hd = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, Hd[rb], Hd, 0, 0, 0); }
        RW_SB();
        // the six MFMAs off the pivots' path, one in front of each piece of the next block's pivot chain
        Hux[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, Hux[0][0][rb], Hux[0][0], 0, 0, 0); RW_SB();
        if (rb < 3) { RW_PIV1(rb + 1) RW_SB(); }
        Hux[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, Hux[0][1][rb], Hux[0][1], 0, 0, 0); RW_SB();
        if (rb < 3) { RW_PIV2() RW_SB(); }
        E[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, E[0][0][rb], E[0][0], 0, 0, 0); RW_SB();
        if (rb < 3) {
          RW_PIV3() RW_SB();
          Hux[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, Hux[0][0][rb], Hux[0][0], 0, 0, 0); RW_SB();
          RW_PIV4() RW_SB();
          Hux[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, Hux[0][1][rb], Hux[0][1], 0, 0, 0); RW_SB();
          RW_PIV5() RW_SB();
          E[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, E[0][0][rb], E[0][0], 0, 0, 0); RW_SB();
        }
      }
#undef RW_PIV1
#undef RW_PIV2
#undef RW_PIV3
#undef RW_PIV4
#undef RW_PIV5
#undef RW_SB
    } else
#pragma unroll
    for (int b = 0; b < NBLK; ++b) {
      const int I = b >> 2, rb = b & 3, c0 = 4 * rb;
      const qm_d4& Hd = Huu[I][I];
      const double D00 = qm_bcast(Hd[rb], c0), D01 = qm_bcast(Hd[rb], c0 + 1), D02 = qm_bcast(Hd[rb], c0 + 2), D03 = qm_bcast(Hd[rb], c0 + 3);
      const double D11 = qm_bcast(Hd[rb], 16 + c0 + 1), D12 = qm_bcast(Hd[rb], 16 + c0 + 2), D13 = qm_bcast(Hd[rb], 16 + c0 + 3);
      const double D22 = qm_bcast(Hd[rb], 32 + c0 + 2), D23 = qm_bcast(Hd[rb], 32 + c0 + 3), D33 = qm_bcast(Hd[rb], 48 + c0 + 3);
      // A pivot that is NOT positive (the negative-duration interval in front of a gait event: Huu ≈ duration · R) gets a ZERO reciprocal — [upstream, recalled] BLASFEO's
      // dpotrf kernels under HPIPM's Riccati factorisation store a zero diagonal and a zero reciprocal there instead of failing: its multipliers l_ij, its row of the
      // trailing update (a2 below) and its rows of W and L⁻¹ (invr below) all vanish, i.e. that reduced input gets K_j = 0, k_j = 0 and the others are solved as if it were
      // not there.  The flag travels to the instance's status as the warning QM_MPC_WARN_PIVOT (ST_RICCATI_STRICT: as the failure -4).
      const double d0 = D00, rd0 = (d0 > 0.0) ? recip(d0) : 0.0;
      const double l10 = D01 * rd0, l20 = D02 * rd0, l30 = D03 * rd0;
      const double d1 = fma(-l10, D01, D11), rd1 = (d1 > 0.0) ? recip(d1) : 0.0;
      const double t12 = fma(-l20, D01, D12), t13 = fma(-l30, D01, D13);            // D12 − l20 l10 d0, D13 − l30 l10 d0
      const double l21 = t12 * rd1, l31 = t13 * rd1;
      const double d2 = fma(-l21, t12, fma(-l20, D02, D22)), rd2 = (d2 > 0.0) ? recip(d2) : 0.0;
      const double t23 = fma(-l31, t12, fma(-l30, D02, D23));                        // D23 − l30 l20 d0 − l31 l21 d1
      const double l32 = t23 * rd2;
      const double d3 = fma(-l32, t23, fma(-l31, t13, fma(-l30, D03, D33))), rd3 = (d3 > 0.0) ? recip(d3) : 0.0;
      if (!(d0 > 0.0) || !(d1 > 0.0) || !(d2 > 0.0) || !(d3 > 0.0)) chol_fail |= failbit;      // bit 0: on a stage of non-positive duration (benign, see above); bit 1: anywhere else or not a number -> hard failure
      // L~⁻¹ of the block (unit lower): its strictly lower entries
      const double M10 = -l10, M21 = -l21, M32 = -l32, M20 = fma(l21, l10, -l20), M31 = fma(l32, l21, -l31), M30 = -(l30 + l31 * M10 + l32 * M20);
      const int ri = c - c0;                                                           // A[i = c][k = g]: row i of the tile against block row k
      const double a1 = (ri == 1) ? ((g == 0) ? M10 : 0.0) : ((ri == 2) ? ((g == 0) ? M20 : ((g == 1) ? M21 : 0.0)) : ((ri == 3) ? ((g == 0) ? M30 : ((g == 1) ? M31 : ((g == 2) ? M32 : 0.0))) : 0.0));
#pragma unroll
      for (int J = I; J < MT; ++J) Huu[I][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, Huu[I][J][rb], Huu[I][J], 0, 0, 0);
#pragma unroll
      for (int J = 0; J < 2; ++J) Hux[I][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, Hux[I][J][rb], Hux[I][J], 0, 0, 0);
#pragma unroll
      for (int J = 0; J <= I; ++J) E[I][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, E[I][J][rb], E[I][J], 0, 0, 0);
      const double dg = (g == 0) ? d0 : ((g == 1) ? d1 : ((g == 2) ? d2 : d3)), rdg = (g == 0) ? rd0 : ((g == 1) ? rd1 : ((g == 2) ? rd2 : rd3));
      dsel[I][rb] = dg;
      // trailing rows: behind the block in its own tile row, and every later tile row
#pragma unroll
      for (int I2 = I; I2 < MT; ++I2) {
        if (I2 == I && rb == 3) continue;                                              // the tile row's last block has nothing behind it there
        const double a2 = (I2 == I) ? ((c > c0 + 3) ? -rdg * Huu[I][I][rb] : 0.0) : -rdg * Huu[I][I2][rb];       // −R'[k][row] / d_k
#pragma unroll
        for (int J = I2; J < MT; ++J) Huu[I2][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, Huu[I][J][rb], Huu[I2][J], 0, 0, 0);
#pragma unroll
        for (int J = 0; J < 2; ++J) Hux[I2][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, Hux[I][J][rb], Hux[I2][J], 0, 0, 0);
#pragma unroll
        for (int J = 0; J <= I; ++J) E[I2][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, E[I][J][rb], E[I2][J], 0, 0, 0);
      }
    }
    double invr[MT][4];                                              // 1/L_jj = d_j^(-1/2) for this lane's rows: four independent chains
#pragma unroll
    for (int I = 0; I < MT; ++I)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double d = dsel[I][r]; double inv = __builtin_amdgcn_rsq(d);
        inv = fma(0.5 * inv, fma(-d * inv, inv, 1.0), inv); inv = fma(0.5 * inv, fma(-d * inv, inv, 1.0), inv);
        invr[I][r] = (d > 0.0) ? inv : 0.0;                            // a zeroed pivot: its rows of W and of L⁻¹ are zero
      }
    RWT(3)
    // The forward rollout needs ũ = −L⁻ᵀ (W δx + y): its gain  K = −L⁻ᵀ W  and offset  k = −L⁻ᵀ y  (column 30) are formed HERE, on the matrix core, from the fragments
    // at hand — G = L⁻¹ = Δ^(-1/2) L~⁻¹ is the scaled identity tile, (Gᵀ W) one more P = Zᵀ Y product — and go to the stage record in place of W and y: the rollout
    // then has ONE matrix–vector product on its dependent chain instead of two and neither fetches nor multiplies a triangular factor
    qm_d4 Gs[MT][MT];
#pragma unroll
    for (int I = 0; I < MT; ++I)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double sc = invr[I][r];
#pragma unroll
        for (int J = 0; J < 2; ++J) W[I][J][r] = Hux[I][J][r] * sc;
#pragma unroll
        for (int J = 0; J < MT; ++J) Gs[I][J][r] = (J <= I) ? E[I][J][r] * sc : 0.0;
      }
    { qm_d4 Kf[MT][2]; rw_zero<MT, 2>(Kf);
      rw_gemm_tn<MT, MT, 2>(Gs, W, Kf, (m + 3) >> 2, true);
      if (MT == 1) {
        // One tile row of reduced inputs (m = 14, 15, 16): EIGHT UNCONDITIONAL stores from two per-lane bases.  Left tile: K[row][c], rows g + 4 r at a stride of 4 x 30 doubles.
        // Right tile: lane columns < 14 -> K[row][16 + c]; lane column 14 -> the offset k[row] (stride 4); lane column 15 (the tile's padding column) -> a slot of the
        // record's profiling area nobody reads.  Rows m .. 15 exist in the record (the gain has 18 rows, the offset 18 entries) and are never read: their (finite) values
        // go out with the rest.  (Round 6: one lane-conditional region and one 64-bit address per element cost this lone wave ≈ 650 cycles per stage, 3 % of the sweep —
        // measured with the stores left out, profiles/r06_ab_riccati_dma.log)
        if (!(PROF && (skip & 256))) {
          double* p0 = rec + SR_PP + g * 30 + c;
          double* p1 = (c < 14) ? rec + SR_PP + g * 30 + 16 + c : ((c == 14) ? rec + SR_KFF + g : rec + SR_K + 16 + g);
          const int st1 = (c < 14) ? 120 : 4;
#pragma unroll
          for (int r = 0; r < 4; ++r) { p0[120 * r] = Kf[0][0][r]; p1[st1 * r] = Kf[0][1][r]; }
        }
      } else
#pragma unroll
      for (int I = 0; I < MT; ++I)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * I + g + 4 * r;
#pragma unroll
          for (int J = 0; J < 2; ++J) {
            const int col = 16 * J + c;
            if (row < m && !(PROF && (skip & 256))) { if (col < 30) rec[SR_PP + row * 30 + col] = Kf[I][J][r]; else if (col == 30) rec[SR_KFF + row] = Kf[I][J][r]; }      // (skip bit 256, instrumented instance only: the gain is NOT stored — what the stores' acknowledgements cost the next stage's operand wait; results meaningless)
          }
        } }
  } else rw_zero<MT, 2>(W);
  RWT(4)
  if (!(skip & 2)) rw_gemm_tn_upper<MT>(W, W, Sn, (m + 3) >> 2, true);   // −[Wᵀ W | Wᵀ y], upper tiles
  RWT(5)
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
  // the diagonal tiles are averaged with their transposes, the lower-left tile is the transpose of the upper-right one (never computed)
  qm_wave_sync();
#pragma unroll
  for (int I = 0; I < 2; ++I)
#pragma unroll
    for (int J = I; J < 2; ++J)
#pragma unroll
      for (int r = 0; r < 4; ++r) buf[(16 * I + g + 4 * r) * RW_TLD + 16 * J + c] = Sn[I][J][r];
  qm_wave_sync();
#pragma unroll
  for (int I = 0; I < 2; ++I)
#pragma unroll
    for (int J = 0; J < 2; ++J)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * I + g + 4 * r, cc = 16 * J + c; const double tr = buf[cc * RW_TLD + row];
        const double v = (I == J) ? 0.5 * (Sn[I][J][r] + tr) : ((I < J) ? Sn[I][J][r] : tr);
        S[I][J][r] = (row < 30 && cc < 30) ? v : 0.0;
      }
  RWT(6)
#undef RWT
}

// flat fetch of everything the forward rollout needs from one stage record: element e of the concatenation
// [Ap rows 0..11 | Bp rows 0..11 | K | Px rows 12..23 | bp qp rp Pe | k | swing blocks | mode dt] lives at record offset rf_src(e).
// Only the momentum / base-pose rows of the projected dynamics are read: a joint row of the Heun-discretised flow map is exactly
// x_j+ = x_j + dt u_j (its rows of A_d, B_d are unit rows resp. dt times unit rows), so dx_j+ = dx_j + dt (du_j − Pe_j) + bp_j comes from the
// input step du the same stage computes anyway — 31 % fewer bytes for a phase that runs at HBM speed.  (Px has no other non-zero rows:
// contact forces and arm joint velocities are free or constant inputs, only the leg joint velocities depend on dx through the constraints.
// Pu is not read at all: its columns are unit vectors — stance force components, arm joint velocities — and one 3x2 block per swing leg.)
__device__ __forceinline__ int rf_src(int e) {
  return (e < 360) ? e : ((e < 576) ? e + (SR_BP - 360) : ((e < 1116) ? e + (SR_PP - 576) : ((e < 1476) ? e + (SR_PX + 360 - 1116) : ((e < 1584) ? e + (SR_BPV - 1476) :
         ((e < 1602) ? e + (SR_KFF - 1584) : ((e < 1626) ? e + (SR_SWG - 1602) : e + (SR_MODEF - 1626)))))));
}
#define RF_TOTAL 1628

template <bool PROF>
__device__ __forceinline__ void qm_riccati_body(QmRiccatiArgs a) {
  if (!PROF) a.skip = 0;
  extern __shared__ double qm_smem[];
  double* buf = qm_smem;
  int* nlist = (int*)(qm_smem + RF_LIST);
#define mlist(k) (nlist[k] & 255)
#define evlist(k) ((nlist[k] >> 8) & 255)
#define modelist(k) (nlist[k] >> 16)
  const int l = threadIdx.x & 63, g = l >> 4, c = l & 15, b = blockIdx.x;
  if (b >= a.B) return;
  const int n = a.n_nodes[b];
  if (a.perf) {
    double pc = 0.0, pd = 0.0, pe = 0.0;
    for (int i = l; i < n; i += 64) { const double* pf = a.perf + (size_t)(i * a.B + b) * PF_SIZE; pc += pf[0]; pd += pf[1]; pe += pf[2]; }
    if (l < 30) { const double dd = a.x0[(size_t)b * 30 + l] - a.x[b * 30 + l]; pd += dd * dd; }
    pc = qm_wave_sum(pc); pd = qm_wave_sum(pd); pe = qm_wave_sum(pe);
    if (l == 0) {
      const double ps[4] = {pc, pc, pd, pe};
      for (int q = 0; q < 4; ++q) { a.base_sum[b * 4 + q] = ps[q]; a.out_perf[b * 10 + q] = ps[q]; a.out_perf[b * 10 + 4 + q] = ps[q]; }
      a.out_perf[b * 10 + 8] = 0.0; a.alpha[b] = 1.0; a.done[b] = 0;
    }
    if (b == 0 && l < 16) { a.open_cnt[l] = 0; a.tickets[l] = 0; }
  }
  for (int k = l; k < n; k += 64) { const int ev = a.node_ev[k * a.B + b]; const double* rk = a.stage + ((size_t)b * a.nmax + k) * SR_SIZE; const bool reg = (ev != QM_EV_PRE) && k < n - 1;
    const int mk = reg ? (int)rk[SR_SCAL] : 0, mdk = reg ? (int)rk[SR_MODEF] : 0; nlist[k] = (mk & 255) | (ev << 8) | (mdk << 16); }      // m, event tag, contact mode of the interval
  qm_wave_sync();
  int chol_fail = 0;
  RwPuCodes pc; pc.mode = -1; pc.pk[0] = pc.pk[1] = 0;
  // (the constants 1.0, 0.0 behind the swing blocks — indices 26, 27 of the swing segment — arrive with every stage's copy: K1b writes them into the record)
  long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; const long long tstart = PROF ? (long long)__builtin_readcyclecounter() : 0;
  qm_d4 S[2][2], sv[2];
  {   // terminal value function
    const double* rec = a.stage + ((size_t)b * a.nmax + (n - 1)) * SR_SIZE;
    rw_load<2, 2>(S, rec + SR_QP, 30, 30, 30, nullptr);
#pragma unroll
    for (int I = 0; I < 2; ++I)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const int row = 16 * I + g + 4 * r; sv[I][r] = (c == 14 && row < 30) ? rec[SR_QPV + row] : 0.0; }
  }
  { int k0 = n - 2; while (k0 >= 0 && evlist(k0) == QM_EV_PRE) --k0;
    if (k0 >= 0 && !(a.skip & 16)) rw_prefetch(a.stage + ((size_t)b * a.nmax + k0) * SR_SIZE, buf, mlist(k0)); }
  for (int k = n - 2; k >= 0; --k) {
    double* rec = a.stage + ((size_t)b * a.nmax + k) * SR_SIZE;
    if (evlist(k) == QM_EV_PRE) {
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
    if (a.skip & 16) continue;
    int kn = k - 1; while (kn >= 0 && evlist(kn) == QM_EV_PRE) --kn;      // next regular stage: its operands are prefetched into LDS
    const double* nrec = (kn >= 0) ? a.stage + ((size_t)b * a.nmax + kn) * SR_SIZE : nullptr;
    const int m = mlist(k), mnext = (kn >= 0) ? mlist(kn) : 0;
    if (m <= 16) rw_stage<1, PROF>(rec, m, nrec, mnext, buf, S, sv, a.skip, chol_fail, tacc, pc, modelist(k));
#ifndef QM_RW_ONLY_MT1
    else rw_stage<2, PROF>(rec, m, nrec, mnext, buf, S, sv, a.skip, chol_fail, tacc, pc, modelist(k));
#endif
  }
  const long long tback = PROF ? (long long)__builtin_readcyclecounter() : 0;
  // L, W, y were stored by other lanes than the ones that read them back below
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");   // same wave, same CU: ordering only, no L2 write-back
  // ---- forward rollout.  The fields of a stage record the rollout reads are copied global -> LDS by the DMA path (global_load_lds, 16 B per lane, no VGPRs) one
  //      regular stage AHEAD into the other half of a double buffer, and consumed one matrix row per lane: lanes 0..29 rows of [Ap Bp bp], lanes 32..61 rows of
  //      [Px Pu Pe] (only rows 12..23 of Px exist), lanes 0..m-1 also row i of W and column i of L⁻¹; lane c carries dx[c].
  //      A stage's dx / du go to HBM one stage LATE, right behind the wait for the stage's operands and in front of the next copy: the wait (vmcnt(0)) then only
  //      ever sees a copy issued a whole stage earlier and two stores older than that, never a store's round trip ----
  double dxl = (l < 30) ? a.x0[(size_t)b * 30 + l] - a.x[(0 * a.B + b) * 30 + l] : 0.0;
  double armijo = 0.0, dx2 = 0.0, du2 = 0.0;
  const int half = l >> 5, r = l & 31;
  if (l < 32) buf[RF_ZERO + l] = 0.0;                       // the zero row (no stage writes it; a wave sync precedes its first use)
  int pu_mode = -1, pu_col = 0, pu_kind = 0, pu_off = 126;     // (Pu ut) source of this lane's du row, cached per contact mode
  // the lane's record offsets do not depend on the stage: computed once (the select chain of rf_src costs ≈ 20 integer instructions per element)
  int fsrc[RF_NLOAD];
#pragma unroll
  for (int t = 0; t < RF_NLOAD; ++t) { const int e = 2 * (t * 64 + l); fsrc[t] = (e < RF_TOTAL) ? rf_src(e) : -1; }
  // (round 6: the LDS side of four consecutive chunks is one base + the instruction's immediate offset 0 / 1 / 2 / 3 KB; the offset moves the global side too, so the lane's
  //  source offset of chunk t is kept MINUS (t mod 4) KB.  Only the last chunk has lanes without an element.)
  static_assert(RF_NLOAD == 13 && (RF_TOTAL + 1) / 2 > 12 * 64, "forward fetch: twelve full chunks and a partial thirteenth");
#pragma unroll
  for (int t = 0; t < RF_NLOAD; ++t) if (fsrc[t] >= 0) fsrc[t] -= 128 * (t & 3);
  const qm_lds_ptr buf3 = qm_lds(buf);
  auto fetch = [&](int k, int which) {
    const double* rec = a.stage + ((size_t)b * a.nmax + k) * SR_SIZE; const qm_lds_ptr F = buf3 + 8 * (which ? RF_F1 : RF_F0);
#pragma unroll
    for (int t0 = 0; t0 < 12; t0 += 4) {
      const qm_lds_ptr l3 = F + 1024 * t0;
      qm_dma16_at<0>((const char*)(rec + fsrc[t0]), l3); qm_dma16_at<1024>((const char*)(rec + fsrc[t0 + 1]), l3);
      qm_dma16_at<2048>((const char*)(rec + fsrc[t0 + 2]), l3); if (!(PROF && (a.skip & 512) && t0 == 8)) qm_dma16_at<3072>((const char*)(rec + fsrc[t0 + 3]), l3);
    }
    if (fsrc[12] >= 0 && !(PROF && (a.skip & 512))) qm_dma16_at<0>((const char*)(rec + fsrc[12]), F + 1024 * 12);      // (skip bit 512, instrumented instance only: the rollout's fetch without its last two chunks — is the rollout bound by its bytes?)
  };
  int cur = 0;
  { int k0 = 0; while (k0 < n - 1 && evlist(k0) == QM_EV_PRE) ++k0; if (k0 < n - 1 && !(a.skip & 4)) fetch(k0, 0); }
  long long tfw[6] = {0, 0, 0, 0, 0, 0}; long long tfl = PROF ? (long long)__builtin_readcyclecounter() : 0;
#define RFT(i) { if (PROF && (a.skip & 32)) { const long long now_ = (long long)__builtin_readcyclecounter(); tfw[i] += now_ - tfl; tfl = now_; } }
  double du_pend = 0.0; int nb_pend = -1;                   // du of the last regular stage, not yet stored
  for (int k = 0; k < n - 1; ++k) {
    if (a.skip & 4) break;
    const int nb = k * a.B + b;
    if (evlist(k) == QM_EV_PRE) {
      if (l < 30) { a.dx[nb * 30 + l] = dxl; dx2 += dxl * dxl; a.du[nb * 30 + l] = 0.0; dxl += a.x[nb * 30 + l] - a.x[((k + 1) * a.B + b) * 30 + l]; }
      continue;
    }
    const int m = mlist(k);
    qm_dma_wait();                                            // this stage's operands have landed
    RFT(0)
    if (l < 30) { a.dx[nb * 30 + l] = dxl; dx2 += dxl * dxl; }
    if (nb_pend >= 0 && half && r < 30) a.du[nb_pend * 30 + r] = du_pend;
    RFT(1)
    { int kn = k + 1; while (kn < n - 1 && evlist(kn) == QM_EV_PRE) ++kn; if (kn < n - 1) fetch(kn, cur ^ 1); }
    RFT(2)
    const double* F = buf + (cur ? RF_F1 : RF_F0); cur ^= 1;
    const int rr = (r < 30) ? r : 29, lw = (l < m) ? l : 0;          // idle lanes read a valid row and drop the result
    const int ra = (rr < 12) ? rr : 0;                                 // lanes 12..29 (joint rows) do not use their products: any valid row
    // lanes 12..29 compute THEIR OWN copy of du row l (the same row lane l + 32 holds for the store): a joint row's dx+ then needs no cross-lane traffic
    const bool durow = half ? (r < 30) : (l >= 12 && l < 30);
    const int pr = (rr >= 12 && rr < 24) ? rr - 12 : -1;                                  // row of the Px block, if the du row has one
    const double* rowA = (!half && l < 12) ? F + RFO_A + l * 30 : ((durow && pr >= 0) ? F + RFO_PX + pr * 30 : buf + RF_ZERO);
    const double* rowB = (!half && l < 12) ? F + RFO_B + l * QM_MMAX : (durow ? buf + RF_PUD + rr * QM_MMAX : buf + RF_ZERO);
    const double* rowW = F + RFO_W + lw * 30; const double* vecs = F + RFO_V;
    double acc = vecs[(half || l >= 12) ? 78 + rr : rr];                                    // Pe of a du row, bp of a dynamics row
    double t = vecs[108 + lw];
    const double qv = (l < 30) ? vecs[30 + l] : 0.0, rp = (l < m) ? vecs[60 + l] : 0.0;
    const double mdv = vecs[150]; double c1 = vecs[pu_off], c2 = vecs[pu_off + 3];          // the swing block entries of this lane's du row (if its leg swings in the mode seen last)
    { double ap[3] = {0.0, 0.0, 0.0}, tp[3] = {0.0, 0.0, 0.0};          // three partial sums each: a 30-long dependent FMA chain is what a lone wave waits on
#pragma unroll
      for (int q = 0; q < 30; ++q) { const double dq = qm_bcast(dxl, q); ap[q % 3] += rowA[q] * dq; tp[q % 3] += rowW[q] * dq; }
      acc += (ap[0] + ap[1]) + ap[2]; t += (tp[0] + tp[1]) + tp[2]; }
    RFT(3)
    // Pu as a dense matrix in LDS, one row per du row (written by the row's lane in the upper half): a stance force component or an arm joint velocity IS one
    // entry of ut (a unit entry, constant per contact mode), a swing leg's joint velocity combines the two null-space coordinates of its leg (two entries per
    // stage), swing forces get nothing (Pe carries −F).  Pu ut then is one more row-times-ut product, with nothing of it on the chain behind ut
    { const int md = (int)mdv;
      if (md != pu_mode) {                                   // the lane's columns only change with the contact mode (wave-uniform test, a few times per sweep)
        pu_mode = md;
        int nst = 0;
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) nst += mode_flag(md, kq);
        const int row = half ? rr : 0;
        const int kk = (row < 12) ? row / 3 : ((row < 24) ? chain_to_contact((row - 12) / 3) : 0), r3 = (row < 12) ? row % 3 : ((row < 24) ? (row - 12) % 3 : row - 24);
        int before_st = 0, before_sw = 0;
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) if (kq < kk) { before_st += mode_flag(md, kq); before_sw += !mode_flag(md, kq); }
        const bool st = mode_flag(md, kk);
        pu_col = (row < 12) ? 3 * before_st + r3 : ((row < 24) ? 3 * nst + 2 * before_sw : 3 * nst + 2 * (4 - nst) + r3);
        pu_kind = !(half && r < 30) ? 0 : ((row < 12) ? (st ? 1 : 0) : ((row < 24) ? (st ? 0 : 2) : 1));      // 0: nothing, 1: one entry of ut, 2: a swing leg's 3x2 block
        pu_off = 126 + 6 * kk + r3;
        if (half && r < 30) {
#pragma unroll
          for (int q = 0; q < QM_MMAX; ++q) buf[RF_PUD + rr * QM_MMAX + q] = (pu_kind == 1 && q == pu_col) ? 1.0 : 0.0;
        }
        c1 = vecs[pu_off]; c2 = vecs[pu_off + 3];
      }
      if (pu_kind == 2) { buf[RF_PUD + rr * QM_MMAX + pu_col] = c1; buf[RF_PUD + rr * QM_MMAX + pu_col + 1] = c2; }
      qm_wave_sync(); }
    const double ut = (l < m) ? t : 0.0;                       // ut = K dx + k: the gain and the offset come ready from the backward sweep
    RFT(4)
    armijo += qv * dxl + rp * ut;
    { double bp[3] = {0.0, 0.0, 0.0};
#pragma unroll
      for (int q = 0; q < QM_MMAX; ++q) bp[q % 3] += rowB[q] * qm_bcast(ut, q);          // ut == 0 on lanes >= m: no bound needed
      acc += (bp[0] + bp[1]) + bp[2]; }
    if (half && r < 30) du2 += acc * acc;
    du_pend = acc; nb_pend = nb;
    if (l >= 12) acc = vecs[rr] + dxl + vecs[151] * (acc - vecs[78 + rr]);      // joint rows: dx_j+ = dx_j + dt (du_j − Pe_j) + bp_j  (lanes >= 32: dropped)
    dxl = (l < 30) ? acc : 0.0;
    qm_lds_drain();                                           // every read of this buffer has returned before the copy after next overwrites it
    RFT(5)
  }
  if (nb_pend >= 0 && half && r < 30) a.du[nb_pend * 30 + r] = du_pend;
  {
    const int nb = (n - 1) * a.B + b; const double* rec = a.stage + ((size_t)b * a.nmax + (n - 1)) * SR_SIZE;
    if (l < 30) { a.dx[nb * 30 + l] = dxl; a.du[nb * 30 + l] = 0.0; dx2 += dxl * dxl; armijo += rec[SR_QPV + l] * dxl; }
  }
  double arm = armijo, sx = dx2, su = du2;
  for (int off = 32; off > 0; off >>= 1) { arm += __shfl_xor(arm, off, 64); sx += __shfl_xor(sx, off, 64); su += __shfl_xor(su, off, 64); }
  if (PROF && (a.skip & 32) && l == 0) {              // profiling: cycles per backward phase, whole sweeps
    double* r0 = a.stage + (size_t)b * a.nmax * SR_SIZE + SR_K;
    for (int i = 0; i < 7; ++i) r0[i] = (double)tacc[i];
    r0[7] = (double)(tback - tstart); r0[8] = (double)((long long)__builtin_readcyclecounter() - tback);
    for (int i = 0; i < 6; ++i) r0[9 + i] = (double)tfw[i];
  }
  if (l == 0) { a.step_info[b * 4] = arm; a.step_info[b * 4 + 1] = sx; a.step_info[b * 4 + 2] = su; a.step_info[b * 4 + 3] = (double)chol_fail; }
}
__global__ void QM_UNPAIRED_LDS __launch_bounds__(RW_BLOCK) qm_riccati_kernel(QmRiccatiArgs a) { qm_riccati_body<false>(a); }
__global__ void QM_UNPAIRED_LDS __launch_bounds__(RW_BLOCK) qm_riccati_prof_kernel(QmRiccatiArgs a) { qm_riccati_body<true>(a); }      // profiling / parity switches only (a.skip != 0)
#undef mlist
#undef evlist
#undef modelist
