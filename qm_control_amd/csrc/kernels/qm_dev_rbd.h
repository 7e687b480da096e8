// qm_dev_rbd.h — rigid-body algorithms of the fixed tree in WORLD-frame quantities (device, f64).
//
// Gives the WBC what the reference takes from Pinocchio (qm_wbc/src/WbcBase.cpp:134-226): joint-space
// inertia (crba), non-linear effects (rnea with zero acceleration), LOCAL_WORLD_ALIGNED frame Jacobians,
// the products  Jdot·v  (classical bias accelerations — the reference only ever uses dJ through dJ*v),
// frame velocities, and centroidal momentum / momentum-rate sums (ccrba / dccrba).
// One pass per serial chain: forward kinematics with velocity and bias-acceleration propagation, then a
// tip-to-root accumulation of composite inertias (for M) and of bias wrenches (for nle).
//   generalized velocity v = [pdot_world, zyx rates, qd_j]  (Pinocchio composite root Translation+SphericalZYX)
#pragma once
#include "qm_dev_common.h"

struct RbdBase {          // root body state
  double R[9], p[3], E[9], w[3], al[3];      // rotation, origin, Euler map, angular velocity, bias angular acceleration (Edot thetadot)
  double vlin[3];                             // pdot
};
// sums over all bodies (world frame, moments about the WORLD origin), a flat array (a struct of arrays was not promoted to registers by the compiler):
// [0] mass, [1..3] m c, [4..6] linear momentum m v_c, [7..9] angular momentum about O, [10..12] bias force sum m a_c (no gravity), [13..15] bias moment about O
#define RBD_SUMS 16
typedef double RbdSums;
struct RbdTip { double p[3], R[9], v[3], w[3], a[3], al[3]; };   // frame pose, velocity, bias accelerations

__device__ __forceinline__ void rbd_base(const double* q, const double* v, RbdBase& B) {
  rot_zyx<true>(q[3], q[4], q[5], B.R); euler_E<true>(q[3], q[4], B.E);      // qm_sincos: the library's sin / cos pair costs ≈ 10 x the instructions on these one-wave chains
  for (int i = 0; i < 3; ++i) { B.p[i] = q[i]; B.vlin[i] = v[i]; }
  const double thd[3] = {v[3], v[4], v[5]};
  m3_mulv(B.E, thd, B.w);
  // Edot thetadot = thd0 (z × E thd) + thd1 (E1 × E2) thd2
  const double z[3] = {0.0, 0.0, 1.0}; double t0[3]; v3_cross(z, B.w, t0);
  const double c1[3] = {B.E[1], B.E[4], B.E[7]}, c2[3] = {B.E[2], B.E[5], B.E[8]}; double t1[3]; v3_cross(c1, c2, t1);
  for (int i = 0; i < 3; ++i) B.al[i] = thd[0] * t0[i] + thd[1] * thd[2] * t1[i];
}
// spatial motion vector (w, vO) of dof d about the world origin: base translation / Euler / revolute joint
// (one-hot arithmetic instead of E[k] / vO[d]: d may be a lane index; a runtime-indexed member array puts the whole RbdBase into scratch memory, and the
//  compiler folds a chain of selects of loads back into exactly that indexed load.  x * 1.0 + y * 0.0 + z * 0.0 == x for the finite entries of E.)
__device__ __forceinline__ void rbd_S_base(const RbdBase& B, int d, double* w, double* vO) {
  if (d < 3) { w[0] = w[1] = w[2] = 0.0; vO[0] = (d == 0) ? 1.0 : 0.0; vO[1] = (d == 1) ? 1.0 : 0.0; vO[2] = (d == 2) ? 1.0 : 0.0; }
  else { const int k = d - 3; const double o0 = (k == 0) ? 1.0 : 0.0, o1 = (k == 1) ? 1.0 : 0.0, o2 = (k == 2) ? 1.0 : 0.0;
    w[0] = B.E[0] * o0 + B.E[1] * o1 + B.E[2] * o2; w[1] = B.E[3] * o0 + B.E[4] * o1 + B.E[5] * o2; w[2] = B.E[6] * o0 + B.E[7] * o1 + B.E[8] * o2;
    v3_cross(B.p, w, vO); }
}
__device__ __forceinline__ void add_body(const double m, const double* c, const double* Iw, const double* vc, const double* w, const double* ac, const double* al,
                                         double& cm, double* ch, double* cI, double* F, double* NO, RbdSums* S) {
  // composite (mass, first moment, inertia about O); bias wrench about O; momentum sums
  cm += m; for (int i = 0; i < 3; ++i) ch[i] += m * c[i];
  const double cc = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) cI[3 * i + j] += Iw[3 * i + j] + m * ((i == j ? cc : 0.0) - c[i] * c[j]);
  double Iw_w[3], Iw_al[3], t[3]; m3_mulv(Iw, w, Iw_w); m3_mulv(Iw, al, Iw_al); v3_cross(w, Iw_w, t);
  const double f[3] = {m * ac[0], m * ac[1], m * (ac[2] + 9.81)};      // nle includes gravity: m (a_c + g z)
  double cf[3]; v3_cross(c, f, cf);
  for (int i = 0; i < 3; ++i) { F[i] += f[i]; NO[i] += cf[i] + Iw_al[i] + t[i]; }
  if (S) {
    S[0] += m; double mv[3] = {m * vc[0], m * vc[1], m * vc[2]}, cmv[3]; v3_cross(c, mv, cmv);
    const double fb[3] = {m * ac[0], m * ac[1], m * ac[2]}; double cfb[3]; v3_cross(c, fb, cfb);
#pragma unroll
    for (int i = 0; i < 3; ++i) { S[1 + i] += m * c[i]; S[4 + i] += mv[i]; S[7 + i] += cmv[i] + Iw_w[i]; S[10 + i] += fb[i]; S[13 + i] += cfb[i] + Iw_al[i] + t[i]; }
  }
}
__device__ __forceinline__ void body_state(const double* mb, int body, const double* R, const double* o, const double* vo, const double* w, const double* ao, const double* al,
                                           double* c, double* Iw, double* vc, double* ac) {
  double r[3]; m3_mulv(R, mb + MB_COM + 3 * body, r);
  double T[9], Rt[9]; m3_mul(R, mb + MB_INERTIA + 9 * body, T); for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Rt[3 * i + j] = R[3 * j + i]; m3_mul(T, Rt, Iw);
  double wr[3], wwr[3], alr[3]; v3_cross(w, r, wr); v3_cross(w, wr, wwr); v3_cross(al, r, alr);
  for (int i = 0; i < 3; ++i) { c[i] = o[i] + r[i]; vc[i] = vo[i] + wr[i]; ac[i] = ao[i] + alr[i] + wwr[i]; }
}

// Full pass over one chain (joints j0..j0+nj-1, tip frame `frame`).
//  M, nle (may be null): fills entries of the chain's dofs against themselves and the base dofs (both triangles)
//  cm/ch/cI/F/NO: composite + bias wrench of the whole chain added to the caller's accumulators (for the base block)
//  Jt (may be null): 6x24 tip Jacobian [lin; ang] columns of this chain's joints (base columns are the caller's job)
// tip-Jacobian sink that keeps the first nrows rows (3: linear rows of a foot, 6: the arm): index i * QM_NQ + col -> rows[i][col], a dummy otherwise
struct RbdJsink { double* rows; int nrows; double dummy; __device__ __forceinline__ double& operator[](int idx) { return (idx < nrows * QM_NQ) ? rows[idx] : dummy; } };

// Storage of the per-joint subtree composites {m, h(3), I(9), F(3), N(3)} = RBD_COMP doubles per joint.  A wave-per-instance kernel that runs several
// chains on neighbouring lanes keeps them in a lane-interleaved LDS workspace (element e of joint jj of lane-slot s at base[(jj * RBD_COMP + e) * stride + s]):
// 19 x NJ doubles per lane are 228 registers at NJ = 6, which — next to the chain's own state — does not fit 512 registers without private-segment spills.
#define RBD_COMP 19
template <int NJ> struct RbdCompRegs {
  double d[NJ][RBD_COMP];
  __device__ __forceinline__ void put(int jj, const double* c) {
#pragma unroll
    for (int e = 0; e < RBD_COMP; ++e) d[jj][e] = c[e]; }
  __device__ __forceinline__ void get(int jj, double* c) const {
#pragma unroll
    for (int e = 0; e < RBD_COMP; ++e) c[e] = d[jj][e]; }
};
struct RbdCompLds {
  double* base; int stride;
  // (fully unrolled: a runtime-indexed private array would be placed in scratch memory)
  __device__ __forceinline__ void put(int jj, const double* c) {
#pragma unroll
    for (int e = 0; e < RBD_COMP; ++e) base[(jj * RBD_COMP + e) * stride] = c[e]; }
  __device__ __forceinline__ void get(int jj, double* c) const {
#pragma unroll
    for (int e = 0; e < RBD_COMP; ++e) c[e] = base[(jj * RBD_COMP + e) * stride]; }
};

template <int NJ, class PM, class PJ, class CS>
__device__ __forceinline__ void rbd_chain(const double* mb, int j0, int frame, const double* q, const double* v, const RbdBase& B,
                                          PM M /*[24][24]*/, double* nle /*[24]*/, bool wantM, double& cm, double* ch, double* cI, double* F, double* NO, RbdSums* S,
                                          RbdTip& tip, PJ Jt /*[6][24]*/, bool wantJ, int nj, CS comp) {
  // nj <= NJ joints are live (lane-dependent): chains of different length run through the SAME instruction stream side by side
  // (a wave executes divergent template instances one after the other); the dead joints are masked out, their arrays stay zero
  double a[NJ][3], o[NJ][3];                        // world axes / joint origins
  // per-joint subtree composites (accumulated tip->root) live in `comp`; one joint's worth is in registers at a time: c[0] m, c[1..3] h, c[4..12] I, c[13..15] F, c[16..18] N
  double Rp[9], op[3], vp[3], wp[3], ap[3], alp[3];
  for (int i = 0; i < 9; ++i) Rp[i] = B.R[i];
  for (int i = 0; i < 3; ++i) { op[i] = B.p[i]; vp[i] = B.vlin[i]; wp[i] = B.w[i]; ap[i] = 0.0; alp[i] = B.al[i]; }
#pragma unroll
  for (int jj = 0; jj < NJ; ++jj) {                 // fully unrolled: the per-joint arrays must stay in registers, not in the private segment
    const bool live = jj < nj; const int j = live ? j0 + jj : j0; const double qd = v[6 + j];
    double r[3]; m3_mulv(Rp, mb + MB_JP + 3 * j, r);
    double wr[3], wwr[3], alr[3]; v3_cross(wp, r, wr); v3_cross(wp, wr, wwr); v3_cross(alp, r, alr);
    double vo[3], ao[3];
    for (int i = 0; i < 3; ++i) { o[jj][i] = op[i] + r[i]; vo[i] = vp[i] + wr[i]; ao[i] = ap[i] + alr[i] + wwr[i]; }
    double cj[RBD_COMP];
#pragma unroll
    for (int e = 0; e < RBD_COMP; ++e) cj[e] = 0.0;
    if (!live) { for (int i = 0; i < 3; ++i) { a[jj][i] = 0.0; o[jj][i] = 0.0; } comp.put(jj, cj); continue; }
    double Rj[9], Rq[9], Rc[9]; m3_mul(Rp, mb + MB_JR + 9 * j, Rj); m3_mulv(Rj, mb + MB_AXIS + 3 * j, a[jj]);
    rot_axis_angle<true>(mb + MB_AXIS + 3 * j, q[6 + j], Rq); m3_mul(Rj, Rq, Rc);
    double wa[3]; v3_cross(wp, a[jj], wa);
    double wc[3], alc[3]; for (int i = 0; i < 3; ++i) { wc[i] = wp[i] + a[jj][i] * qd; alc[i] = alp[i] + wa[i] * qd; }
    double c[3], Iw[9], vc[3], ac[3]; body_state(mb, j + 1, Rc, o[jj], vo, wc, ao, alc, c, Iw, vc, ac);
    add_body(mb[MB_MASS + j + 1], c, Iw, vc, wc, ac, alc, cj[0], cj + 1, cj + 4, cj + 13, cj + 16, S);
    comp.put(jj, cj);
    for (int i = 0; i < 9; ++i) Rp[i] = Rc[i];
    for (int i = 0; i < 3; ++i) { op[i] = o[jj][i]; vp[i] = vo[i]; wp[i] = wc[i]; ap[i] = ao[i]; alp[i] = alc[i]; }
  }
  // tip frame
  {
    double r[3]; m3_mulv(Rp, mb + MB_FP + 3 * frame, r); m3_mul(Rp, mb + MB_FR + 9 * frame, tip.R);
    double wr[3], wwr[3], alr[3]; v3_cross(wp, r, wr); v3_cross(wp, wr, wwr); v3_cross(alp, r, alr);
    for (int i = 0; i < 3; ++i) { tip.p[i] = op[i] + r[i]; tip.v[i] = vp[i] + wr[i]; tip.w[i] = wp[i]; tip.a[i] = ap[i] + alr[i] + wwr[i]; tip.al[i] = alp[i]; }
  }
  if (wantJ) {
#pragma unroll
  for (int jj = 0; jj < NJ; ++jj) if (jj < nj) {
    const double d[3] = {tip.p[0] - o[jj][0], tip.p[1] - o[jj][1], tip.p[2] - o[jj][2]}; double l[3]; v3_cross(a[jj], d, l);
    for (int i = 0; i < 3; ++i) { Jt[i * QM_NQ + 6 + j0 + jj] = l[i]; Jt[(3 + i) * QM_NQ + 6 + j0 + jj] = a[jj][i]; }
  }
  }
  // tip -> root accumulation
  // running composite from the tip: after joint jj has been added it is the composite of the subtree hanging from joint jj — exactly what column jj of M and
  // entry jj of nle need (same summation order as an array accumulated tip -> root)
  double run[RBD_COMP];
#pragma unroll
  for (int e = 0; e < RBD_COMP; ++e) run[e] = 0.0;
#pragma unroll
  for (int jj = NJ - 1; jj >= 0; --jj) {
    { double cj[RBD_COMP]; comp.get(jj, cj);
#pragma unroll
      for (int e = 0; e < RBD_COMP; ++e) run[e] = (jj == NJ - 1) ? cj[e] : cj[e] + run[e]; }
    const double bmj = run[0]; const double* bhj = run + 1; const double* bIj = run + 4; const double* bFj = run + 13; const double* bNj = run + 16;
    if (wantM && jj < nj) {
      const int dj = 6 + j0 + jj;
      // S_j = (a_j, o_j × a_j);  momentum of the subtree composite: f = m vO + w × h ; nO = I_O w + h × vO
      double vO[3]; v3_cross(o[jj], a[jj], vO);
      double wh[3], hv[3], Iw_[3]; v3_cross(a[jj], bhj, wh); v3_cross(bhj, vO, hv); m3_mulv(bIj, a[jj], Iw_);
      const double f[3] = {bmj * vO[0] + wh[0], bmj * vO[1] + wh[1], bmj * vO[2] + wh[2]};
      const double nO[3] = {Iw_[0] + hv[0], Iw_[1] + hv[1], Iw_[2] + hv[2]};
#pragma unroll
      for (int ii = 0; ii < NJ; ++ii) if (ii <= jj) {   // chain ancestors (incl. itself)
        double vOi[3]; v3_cross(o[ii], a[ii], vOi);
        const double val = a[ii][0] * nO[0] + a[ii][1] * nO[1] + a[ii][2] * nO[2] + vOi[0] * f[0] + vOi[1] * f[1] + vOi[2] * f[2];
        M[(6 + j0 + ii) * QM_NQ + dj] = val; M[dj * QM_NQ + 6 + j0 + ii] = val;
      }
#pragma unroll
      for (int d = 0; d < 6; ++d) {         // base dofs
        double w[3], vOb[3]; rbd_S_base(B, d, w, vOb);
        const double val = w[0] * nO[0] + w[1] * nO[1] + w[2] * nO[2] + vOb[0] * f[0] + vOb[1] * f[1] + vOb[2] * f[2];
        M[d * QM_NQ + dj] = val; M[dj * QM_NQ + d] = val;
      }
      nle[dj] = a[jj][0] * bNj[0] + a[jj][1] * bNj[1] + a[jj][2] * bNj[2] + vO[0] * bFj[0] + vO[1] * bFj[1] + vO[2] * bFj[2];
    }
  }
  cm += run[0]; for (int i = 0; i < 3; ++i) { ch[i] += run[1 + i]; F[i] += run[13 + i]; NO[i] += run[16 + i]; } for (int i = 0; i < 9; ++i) cI[i] += run[4 + i];
}
// register-resident composites (thread-per-instance callers, short chains)
template <int NJ, class PM, class PJ>
__device__ __forceinline__ void rbd_chain(const double* mb, int j0, int frame, const double* q, const double* v, const RbdBase& B,
                                          PM M, double* nle, bool wantM, double& cm, double* ch, double* cI, double* F, double* NO, RbdSums* S,
                                          RbdTip& tip, PJ Jt, bool wantJ, int nj = NJ) {
  RbdCompRegs<NJ> regs;
  rbd_chain<NJ, PM, PJ, RbdCompRegs<NJ>&>(mb, j0, frame, q, v, B, M, nle, wantM, cm, ch, cI, F, NO, S, tip, Jt, wantJ, nj, regs);
}

// Whole-tree pass.  Outputs (any pointer may be null):
//   M[24][24], nle[24]; feet tips [4] (contact order LF,RF,LH,RH), arm tip; Jfeet [12][24] linear rows; Jarm [6][24]; sums
template <class PM, class PJ>
__device__ __forceinline__ void rbd_tree(const double* mb, const double* q, const double* v, RbdBase& B, PM M, double* nle, bool wantM, RbdTip* feet, RbdTip* arm,
                                         PJ Jfeet, PJ Jarm, bool wantJ, RbdSums* S) {
  rbd_base(q, v, B);
  if (wantM) for (int i = 0; i < QM_NQ * QM_NQ; ++i) M[i] = 0.0;
  if (wantJ) { for (int i = 0; i < 12 * QM_NQ; ++i) Jfeet[i] = 0.0; for (int i = 0; i < 6 * QM_NQ; ++i) Jarm[i] = 0.0; }
  if (S) { for (int i = 0; i < RBD_SUMS; ++i) S[i] = 0.0; }
  double cm = 0.0, ch[3] = {0, 0, 0}, cI[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, F[3] = {0, 0, 0}, NO[3] = {0, 0, 0};
  { // root body
    const double zero3[3] = {0.0, 0.0, 0.0}; double c[3], Iw[9], vc[3], ac[3];
    body_state(mb, 0, B.R, B.p, B.vlin, B.w, zero3, B.al, c, Iw, vc, ac);
    add_body(mb[MB_MASS], c, Iw, vc, B.w, ac, B.al, cm, ch, cI, F, NO, S);
  }
  double Jt[6 * QM_NQ];
  for (int chain = 0; chain < 4; ++chain) {
    const int contact = chain_to_contact(chain); RbdTip tip;
    if (wantJ) for (int i = 0; i < 6 * QM_NQ; ++i) Jt[i] = 0.0;
    rbd_chain<3, PM, double*>(mb, 3 * chain, contact, q, v, B, M, nle, wantM, cm, ch, cI, F, NO, S, tip, Jt, wantJ);
    if (feet) feet[contact] = tip;
    if (wantJ) {
      for (int r = 0; r < 3; ++r) { for (int cidx = 6 + 3 * chain; cidx < 9 + 3 * chain; ++cidx) Jfeet[(3 * contact + r) * QM_NQ + cidx] = Jt[r * QM_NQ + cidx]; Jfeet[(3 * contact + r) * QM_NQ + r] = 1.0; }
      for (int k = 0; k < 3; ++k) { const double e[3] = {B.E[k], B.E[3 + k], B.E[6 + k]}, d[3] = {tip.p[0] - B.p[0], tip.p[1] - B.p[1], tip.p[2] - B.p[2]}; double l[3]; v3_cross(e, d, l); for (int r = 0; r < 3; ++r) Jfeet[(3 * contact + r) * QM_NQ + 3 + k] = l[r]; }
    }
  }
  {
    RbdTip tip; rbd_chain<6, PM, PJ>(mb, 12, 4, q, v, B, M, nle, wantM, cm, ch, cI, F, NO, S, tip, Jarm, wantJ);
    if (arm) *arm = tip;
    if (wantJ) {
      for (int r = 0; r < 3; ++r) Jarm[r * QM_NQ + r] = 1.0;
      for (int k = 0; k < 3; ++k) { const double e[3] = {B.E[k], B.E[3 + k], B.E[6 + k]}, d[3] = {tip.p[0] - B.p[0], tip.p[1] - B.p[1], tip.p[2] - B.p[2]}; double l[3]; v3_cross(e, d, l); for (int r = 0; r < 3; ++r) { Jarm[r * QM_NQ + 3 + k] = l[r]; Jarm[(3 + r) * QM_NQ + 3 + k] = e[r]; } }
    }
  }
  if (wantM) {   // base block from the whole-tree composite; base rows of nle from the whole-tree bias wrench
    for (int d = 0; d < 6; ++d) {
      double w[3], vO[3]; rbd_S_base(B, d, w, vO);
      double wh[3], hv[3], Iw_[3]; v3_cross(w, ch, wh); v3_cross(ch, vO, hv); m3_mulv(cI, w, Iw_);
      const double f[3] = {cm * vO[0] + wh[0], cm * vO[1] + wh[1], cm * vO[2] + wh[2]}, nO[3] = {Iw_[0] + hv[0], Iw_[1] + hv[1], Iw_[2] + hv[2]};
      for (int e = 0; e < 6; ++e) { double w2[3], vO2[3]; rbd_S_base(B, e, w2, vO2); M[e * QM_NQ + d] = w2[0] * nO[0] + w2[1] * nO[1] + w2[2] * nO[2] + vO2[0] * f[0] + vO2[1] * f[1] + vO2[2] * f[2]; }
      nle[d] = w[0] * NO[0] + w[1] * NO[1] + w[2] * NO[2] + vO[0] * F[0] + vO[1] * F[1] + vO[2] * F[2];
    }
  }
}
