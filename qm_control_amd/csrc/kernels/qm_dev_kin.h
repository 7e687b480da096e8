// qm_dev_kin.h — kinematics / centroidal quantities of the fixed-topology 24-DoF quadruped-manipulator
// (base + 4 leg chains of 3 revolute joints + one arm chain of 6), hand-derived, f64.
//
// Replaces on device what the reference gets from CppAD-generated Pinocchio code
// (qm_interface/src/dynamics/QMDynamicsAD.cpp:22-33, qm_interface/src/QMInterface.cpp:363-379):
// the SRBD flow map, foot positions/velocities, arm end-effector pose and ALL their first derivatives.
// Derivatives are analytic (cross-product identities of rotating frames), not a port of generated code.
//
// A "kin workspace" K (KW_SIZE doubles, in LDS or registers/scratch) holds the shared intermediates.
#pragma once
#include "qm_dev_common.h"

#define KW_RB   0     /* [9] base rotation R(θ)                                   */
#define KW_E    9     /* [9] ω_world = E θdot                                      */
#define KW_EINV 18    /* [9]                                                        */
#define KW_IINV 27    /* [9] (R I_nom Rᵀ)⁻¹ = R I_nom⁻¹ Rᵀ                          */
#define KW_RW   36    /* [3] R r_nom  (com = p_base − R r_nom)                       */
#define KW_COM  39    /* [3]                                                        */
#define KW_OM   42    /* [3] ω = (R I Rᵀ)⁻¹ m h_ang                                 */
#define KW_THD  45    /* [3] θdot = E⁻¹ ω                                           */
#define KW_L    48    /* [3] ℓ = m h_ang                                            */
#define KW_LEG  51    /* 4 x { a[3][3] world axes, o[3][3] joint origins, p[3] foot, s[3] Σ qd_j a_j×(p−o_j) } */
#define KW_LEGSZ 24
#define KW_ARM  147   /* a[6][3], o[6][3], p[3], R[9]                               */
#define KW_SIZE 196

// base / centroidal part.  x = [h_lin, h_ang, p_b, zyx, q_j]
template <bool FAST = false>
__device__ __forceinline__ void kin_base(const double* mb, const double* x, double* K) {
  rot_zyx<FAST>(x[9], x[10], x[11], K + KW_RB);
  euler_E<FAST>(x[9], x[10], K + KW_E);
  m3_inv(K + KW_E, K + KW_EINV);
  double Ii[9], T[9], Rt[9];
  m3_inv(mb + MB_INOM, Ii);
  m3_mul(K + KW_RB, Ii, T);
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Rt[3 * i + j] = K[KW_RB + 3 * j + i];
  m3_mul(T, Rt, K + KW_IINV);
  m3_mulv(K + KW_RB, mb + MB_RNOM, K + KW_RW);
  const double m = mb[MB_ROBOTMASS];
  for (int i = 0; i < 3; ++i) { K[KW_COM + i] = x[6 + i] - K[KW_RW + i]; K[KW_L + i] = m * x[3 + i]; }
  m3_mulv(K + KW_IINV, K + KW_L, K + KW_OM);
  m3_mulv(K + KW_EINV, K + KW_OM, K + KW_THD);
}
// one serial chain hanging off the base: joints j0..j0+nj-1, tip frame `frame`.
// a,o: world axes / joint origins [nj][3]; p: tip position; Rend (optional): tip rotation;
// s (optional, needs u): joint-induced tip velocity Σ qd_j a_j × (p − o_j)
// HAS_S / HAS_REND are compile-time (not `if (s)` / `if (Rend)`): the outputs usually point into a thread-private workspace, and comparing such an address
// with null keeps the whole workspace out of registers (address 0 is a valid private address on this target, the test cannot be folded)
template <bool HAS_S, bool HAS_REND, bool FAST = false>
__device__ __forceinline__ void kin_chain(const double* mb, int j0, int nj, int frame, const double* x, const double* u, const double* K,
                                          double* a, double* o, double* p, double* s, double* Rend) {
  double Rp[9], pp[3], Rj[9], Rq[9], Rn[9];
  for (int i = 0; i < 9; ++i) Rp[i] = K[KW_RB + i];
  for (int i = 0; i < 3; ++i) pp[i] = x[6 + i];
  for (int jj = 0; jj < nj; ++jj) {
    const int j = j0 + jj;
    double t[3]; m3_mulv(Rp, mb + MB_JP + 3 * j, t);
    for (int i = 0; i < 3; ++i) { pp[i] += t[i]; o[3 * jj + i] = pp[i]; }
    m3_mul(Rp, mb + MB_JR + 9 * j, Rj);
    m3_mulv(Rj, mb + MB_AXIS + 3 * j, a + 3 * jj);
    rot_axis_angle<FAST>(mb + MB_AXIS + 3 * j, x[12 + j], Rq);
    m3_mul(Rj, Rq, Rn);
    for (int i = 0; i < 9; ++i) Rp[i] = Rn[i];
  }
  double t[3]; m3_mulv(Rp, mb + MB_FP + 3 * frame, t);
  for (int i = 0; i < 3; ++i) p[i] = pp[i] + t[i];
  if (HAS_REND) m3_mul(Rp, mb + MB_FR + 9 * frame, Rend);
  if (HAS_S) {
    s[0] = s[1] = s[2] = 0.0;
    for (int jj = 0; jj < nj; ++jj) {
      double d[3] = {p[0] - o[3 * jj], p[1] - o[3 * jj + 1], p[2] - o[3 * jj + 2]}, c[3];
      v3_cross(a + 3 * jj, d, c);
      const double qd = u[12 + j0 + jj];
      s[0] += qd * c[0]; s[1] += qd * c[1]; s[2] += qd * c[2];
    }
  }
}
template <bool FAST = false>
__device__ __forceinline__ void kin_leg(const double* mb, int chain, const double* x, const double* u, double* K) {
  double* L = K + KW_LEG + KW_LEGSZ * chain;
  // u must not be null (no `u ? … : nullptr` here: comparing the address of a thread-private array with null keeps the whole array out of registers,
  // address 0 being a valid private address on this target)
  kin_chain<true, false, FAST>(mb, 3 * chain, 3, chain_to_contact(chain), x, u, K, L, L + 9, L + 18, L + 21, nullptr);
}
template <bool FAST = false>
__device__ __forceinline__ void kin_arm(const double* mb, const double* x, double* K) {
  double* A = K + KW_ARM;
  kin_chain<false, true, FAST>(mb, 12, 6, 4, x, nullptr, K, A, A + 18, A + 36, nullptr, A + 39);
}
__device__ __forceinline__ const double* kin_foot(const double* K, int contact) { return K + KW_LEG + KW_LEGSZ * contact_to_chain(contact) + 18; }

// a3: flow map value from a filled workspace (legs + base).  flow_head_from_kin: the twelve non-trivial rows only (momentum rates, base pose rates); rows 12..29 are
// the input's joint velocities u[12..29]
__device__ __forceinline__ void flow_head_from_kin(const double* mb, const double* x, const double* u, const double* K, double* f) {
  const double m = mb[MB_ROBOTMASS], im = 1.0 / m;
  double lin[3] = {0.0, 0.0, -9.81 * m}, ang[3] = {0.0, 0.0, 0.0};
  for (int i = 0; i < 4; ++i) {
    const double* p = kin_foot(K, i);
    const double d[3] = {p[0] - K[KW_COM], p[1] - K[KW_COM + 1], p[2] - K[KW_COM + 2]};
    double c[3]; v3_cross(d, u + 3 * i, c);
    for (int k = 0; k < 3; ++k) { lin[k] += u[3 * i + k]; ang[k] += c[k]; }
  }
  double wr[3]; v3_cross(K + KW_OM, K + KW_RW, wr);
  for (int k = 0; k < 3; ++k) { f[k] = lin[k] * im; f[3 + k] = ang[k] * im; f[6 + k] = x[k] + wr[k]; f[9 + k] = K[KW_THD + k]; }
}
__device__ __forceinline__ void flow_from_kin(const double* mb, const double* x, const double* u, const double* K, double* f) {
  const double m = mb[MB_ROBOTMASS], im = 1.0 / m;
  double lin[3] = {0.0, 0.0, -9.81 * m}, ang[3] = {0.0, 0.0, 0.0};
  for (int i = 0; i < 4; ++i) {
    const double* p = kin_foot(K, i);
    const double d[3] = {p[0] - K[KW_COM], p[1] - K[KW_COM + 1], p[2] - K[KW_COM + 2]};
    double c[3]; v3_cross(d, u + 3 * i, c);
    for (int k = 0; k < 3; ++k) { lin[k] += u[3 * i + k]; ang[k] += c[k]; }
  }
  double wr[3]; v3_cross(K + KW_OM, K + KW_RW, wr);
  for (int k = 0; k < 3; ++k) { f[k] = lin[k] * im; f[3 + k] = ang[k] * im; f[6 + k] = x[k] + wr[k]; f[9 + k] = K[KW_THD + k]; }
  _Pragma("unroll") for (int j = 0; j < QM_NJ; ++j) f[12 + j] = u[12 + j];      // (static indices: f / u stay in registers)
}
// One Heun stage of one node on thread-private data with the legs in a ROLLED loop (small code, few live registers, so that
// several waves fit a SIMD): x and u may point to global memory.  Kb[KW_LEG] receives the base block, leg(c, L) is called with
// every leg block L[KW_LEGSZ] = {a, o, p, s}, f[30] is the flow-map value (same terms as flow_from_kin).
template <class LegFn> __device__ __forceinline__ void kin_stage(const double* mb, const double* x, const double* u, double* Kb, double* f, LegFn leg) {
  kin_base(mb, x, Kb);
  const double m = mb[MB_ROBOTMASS], im = 1.0 / m;
  double lin[3] = {0.0, 0.0, -9.81 * m}, ang[3] = {0.0, 0.0, 0.0};
#pragma nounroll
  for (int c = 0; c < 4; ++c) {
    double L[KW_LEGSZ];
    const int contact = chain_to_contact(c);
    kin_chain<true, false>(mb, 3 * c, 3, contact, x, u, Kb, L, L + 9, L + 18, L + 21, nullptr);
    leg(c, L);
    const double F[3] = {u[3 * contact], u[3 * contact + 1], u[3 * contact + 2]};
    const double d[3] = {L[18] - Kb[KW_COM], L[19] - Kb[KW_COM + 1], L[20] - Kb[KW_COM + 2]};
    double t[3]; v3_cross(d, F, t);
    for (int k = 0; k < 3; ++k) { lin[k] += F[k]; ang[k] += t[k]; }
  }
  double wr[3]; v3_cross(Kb + KW_OM, Kb + KW_RW, wr);
  for (int k = 0; k < 3; ++k) { f[k] = lin[k] * im; f[3 + k] = ang[k] * im; f[6 + k] = x[k] + wr[k]; f[9 + k] = Kb[KW_THD + k]; }
  for (int j = 0; j < QM_NJ; ++j) f[12 + j] = u[12 + j];
}
// arm block A[KW_SIZE - KW_ARM] = {a[6][3], o[6][3], p, R} from the base block
__device__ __forceinline__ void kin_arm_block(const double* mb, const double* x, const double* Kb, double* A) {
  kin_chain<false, true>(mb, 12, 6, 4, x, nullptr, Kb, A, A + 18, A + 36, nullptr, A + 39);
}
// foot velocity v_i = h_lin + ω × d_i + s_i   (LOCAL_WORLD_ALIGNED linear velocity of the foot frame under the SRBD map)
__device__ __forceinline__ void foot_velocity(const double* x, const double* K, int contact, double* v) {
  const double* L = K + KW_LEG + KW_LEGSZ * contact_to_chain(contact);
  const double d[3] = {L[18] - K[KW_COM], L[19] - K[KW_COM + 1], L[20] - K[KW_COM + 2]};
  double c[3]; v3_cross(K + KW_OM, d, c);
  for (int k = 0; k < 3; ++k) v[k] = x[k] + c[k] + L[21 + k];
}

// dω/dθ_k = e_k × ω − 𝓘⁻¹(e_k × ℓ),  e_k = E[:,k]
__device__ __forceinline__ void d_omega_dtheta(const double* K, int k, double* ek, double* dw) {
  ek[0] = K[KW_E + k]; ek[1] = K[KW_E + 3 + k]; ek[2] = K[KW_E + 6 + k];
  double t1[3], t2[3], t3[3];
  v3_cross(ek, K + KW_OM, t1); v3_cross(ek, K + KW_L, t2); m3_mulv(K + KW_IINV, t2, t3);
  for (int i = 0; i < 3; ++i) dw[i] = t1[i] - t3[i];
}
// (dE/dθ_k) θdot
__device__ __forceinline__ void dE_thd(const double* K, int k, double* r) {
  if (k == 0) { const double z[3] = {0.0, 0.0, 1.0}; v3_cross(z, K + KW_OM, r); }
  else if (k == 1) { const double c1[3] = {K[KW_E + 1], K[KW_E + 4], K[KW_E + 7]}, c2[3] = {K[KW_E + 2], K[KW_E + 5], K[KW_E + 8]}; double t[3]; v3_cross(c1, c2, t); for (int i = 0; i < 3; ++i) r[i] = t[i] * K[KW_THD + 2]; }
  else { r[0] = r[1] = r[2] = 0.0; }
}

// column c (0..59) of [∂f/∂x | ∂f/∂u]: rows 0..11 returned in col12 (rows 12..29 are the identity on qd_j)
__device__ __forceinline__ void flow_jac_col(const double* mb, const double* x, const double* u, const double* K, int c, double* col12) {
  const double m = mb[MB_ROBOTMASS], im = 1.0 / m;
  for (int i = 0; i < 12; ++i) col12[i] = 0.0;
  if (c < 3) { col12[6 + c] = 1.0; }
  else if (c < 6) {
    const int k = c - 3; const double w[3] = {K[KW_IINV + k] * m, K[KW_IINV + 3 + k] * m, K[KW_IINV + 6 + k] * m};
    v3_cross(w, K + KW_RW, col12 + 6); m3_mulv(K + KW_EINV, w, col12 + 9);
  } else if (c < 9) { }
  else if (c < 12) {
    const int k = c - 9; double ek[3], dw[3]; d_omega_dtheta(K, k, ek, dw);
    for (int i = 0; i < 4; ++i) {
      const double* p = kin_foot(K, i); const double d[3] = {p[0] - K[KW_COM], p[1] - K[KW_COM + 1], p[2] - K[KW_COM + 2]};
      double t[3], t2[3]; v3_cross(ek, d, t); v3_cross(t, u + 3 * i, t2);
      for (int r = 0; r < 3; ++r) col12[3 + r] += t2[r] * im;
    }
    double t1[3], t2[3], t3[3]; v3_cross(dw, K + KW_RW, t1); v3_cross(ek, K + KW_RW, t2); v3_cross(K + KW_OM, t2, t3);
    for (int r = 0; r < 3; ++r) col12[6 + r] = t1[r] + t3[r];
    double de[3]; dE_thd(K, k, de); const double rhs[3] = {dw[0] - de[0], dw[1] - de[1], dw[2] - de[2]};
    m3_mulv(K + KW_EINV, rhs, col12 + 9);
  } else if (c < 24) {
    const int j = c - 12, chain = j / 3, jj = j - 3 * chain, contact = chain_to_contact(chain);
    const double* L = K + KW_LEG + KW_LEGSZ * chain;
    const double d[3] = {L[18] - L[9 + 3 * jj], L[19] - L[10 + 3 * jj], L[20] - L[11 + 3 * jj]};
    double t[3], t2[3]; v3_cross(L + 3 * jj, d, t); v3_cross(t, u + 3 * contact, t2);
    for (int r = 0; r < 3; ++r) col12[3 + r] = t2[r] * im;
  } else if (c < 30) { }
  else if (c < 42) {
    const int i = (c - 30) / 3, k = (c - 30) - 3 * i;
    col12[k] = im;
    const double* p = kin_foot(K, i); const double d[3] = {p[0] - K[KW_COM], p[1] - K[KW_COM + 1], p[2] - K[KW_COM + 2]};
    const double ek[3] = {(k == 0) ? 1.0 : 0.0, (k == 1) ? 1.0 : 0.0, (k == 2) ? 1.0 : 0.0}; double t[3]; v3_cross(d, ek, t);      // (selects, not ek[k] = 1: a private array indexed by a run-time value lives in the private segment)
    for (int r = 0; r < 3; ++r) col12[3 + r] = t[r] * im;
  }
}

// flow_jac_col for a wave that holds one column per lane: the four non-trivial column classes (3..5, 9..11, 12..23, 30..41) are all
// evaluated by every lane with clamped indices and the lane's class selects the result — one basic block of four independent chains
// (interleaved by the scheduler) instead of four divergent bodies executed one after the other.  Same arithmetic per class.
__device__ __forceinline__ void flow_jac_col_wave(const double* mb, const double* u, const double* K, int c, double* col12) {
  const double m = mb[MB_ROBOTMASS], im = 1.0 / m;
  const int cls = (c < 6) ? 0 : (c < 12) ? 1 : (c < 24) ? 2 : 3;
  const int k3 = (cls == 0) ? c - 3 : (cls == 1) ? c - 9 : 0;
  // class 0: d/d(h_ang)_k
  double a6[3], a9[3];
  { const double w[3] = {K[KW_IINV + k3] * m, K[KW_IINV + 3 + k3] * m, K[KW_IINV + 6 + k3] * m}; v3_cross(w, K + KW_RW, a6); m3_mulv(K + KW_EINV, w, a9); }
  // class 1: d/d(zyx)_k
  double b3[3] = {0.0, 0.0, 0.0}, b6[3], b9[3];
  { double ek[3], dw[3]; d_omega_dtheta(K, k3, ek, dw);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const double* p = kin_foot(K, i); const double d[3] = {p[0] - K[KW_COM], p[1] - K[KW_COM + 1], p[2] - K[KW_COM + 2]};
      double t[3], t2[3]; v3_cross(ek, d, t); v3_cross(t, u + 3 * i, t2);
      b3[0] += t2[0] * im; b3[1] += t2[1] * im; b3[2] += t2[2] * im;
    }
    double t1[3], t2[3], t3[3]; v3_cross(dw, K + KW_RW, t1); v3_cross(ek, K + KW_RW, t2); v3_cross(K + KW_OM, t2, t3);
    for (int r = 0; r < 3; ++r) b6[r] = t1[r] + t3[r];
    // (dE/dθ_k) θdot: k = 0: e_z × ω, k = 1: (E[:,1] × E[:,2]) θdot_2, k = 2: 0
    const double c1[3] = {K[KW_E + 1], K[KW_E + 4], K[KW_E + 7]}, c2[3] = {K[KW_E + 2], K[KW_E + 5], K[KW_E + 8]}; double cx[3]; v3_cross(c1, c2, cx);
    const double th2 = K[KW_THD + 2];
    const double de[3] = {(k3 == 0) ? -K[KW_OM + 1] : (k3 == 1) ? cx[0] * th2 : 0.0, (k3 == 0) ? K[KW_OM] : (k3 == 1) ? cx[1] * th2 : 0.0, (k3 == 1) ? cx[2] * th2 : 0.0};
    const double rhs[3] = {dw[0] - de[0], dw[1] - de[1], dw[2] - de[2]};
    m3_mulv(K + KW_EINV, rhs, b9); }
  // class 2: d/d(q_j), leg joints
  double c3[3];
  { const int j = (cls == 2) ? c - 12 : 0, chain = j / 3, jj = j - 3 * chain, contact = chain_to_contact(chain);
    const double* L = K + KW_LEG + KW_LEGSZ * chain;
    const double d[3] = {L[18] - L[9 + 3 * jj], L[19] - L[10 + 3 * jj], L[20] - L[11 + 3 * jj]};
    double t[3], t2[3]; v3_cross(L + 3 * jj, d, t); v3_cross(t, u + 3 * contact, t2);
    for (int r = 0; r < 3; ++r) c3[r] = t2[r] * im; }
  // class 3: d/d(F_i)_k
  double d0[3], d3[3];
  { const int q = (cls == 3) ? c - 30 : 0, i = q / 3, k = q - 3 * i;
    const double* p = kin_foot(K, i); const double d[3] = {p[0] - K[KW_COM], p[1] - K[KW_COM + 1], p[2] - K[KW_COM + 2]};
    for (int r = 0; r < 3; ++r) d0[r] = (r == k) ? im : 0.0;
    d3[0] = ((k == 1) ? -d[2] : (k == 2) ? d[1] : 0.0) * im; d3[1] = ((k == 0) ? d[2] : (k == 2) ? -d[0] : 0.0) * im; d3[2] = ((k == 0) ? -d[1] : (k == 1) ? d[0] : 0.0) * im; }
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    col12[r] = (cls == 3) ? d0[r] : 0.0;
    col12[3 + r] = (cls == 1) ? b3[r] : (cls == 2) ? c3[r] : (cls == 3) ? d3[r] : 0.0;
    col12[6 + r] = (cls == 0) ? a6[r] : (cls == 1) ? b6[r] : 0.0;
    col12[9 + r] = (cls == 0) ? a9[r] : (cls == 1) ? b9[r] : 0.0;
  }
}

// a8: derivative of foot `contact`'s velocity (and z-position, for positionErrorGain) wrt column c of [x | u] -> dv[3], dpz
__device__ __forceinline__ void foot_vel_jac_col(const double* mb, const double* x, const double* u, const double* K, int contact, int c, double* dv, double* dpz) {
  const double m = mb[MB_ROBOTMASS];
  const int chain = contact_to_chain(contact);
  const double* L = K + KW_LEG + KW_LEGSZ * chain;
  const double* p = L + 18;
  const double d[3] = {p[0] - K[KW_COM], p[1] - K[KW_COM + 1], p[2] - K[KW_COM + 2]};
  dv[0] = dv[1] = dv[2] = 0.0; *dpz = 0.0;
  if (c < 3) { dv[c] = 1.0; }
  else if (c < 6) { const int k = c - 3; const double w[3] = {K[KW_IINV + k] * m, K[KW_IINV + 3 + k] * m, K[KW_IINV + 6 + k] * m}; v3_cross(w, d, dv); }
  else if (c < 9) { if (c == 8) *dpz = 1.0; }
  else if (c < 12) {
    const int k = c - 9; double ek[3], dw[3]; d_omega_dtheta(K, k, ek, dw);
    double t1[3], t2[3], t3[3], t4[3]; v3_cross(dw, d, t1); v3_cross(ek, d, t2); v3_cross(K + KW_OM, t2, t3); v3_cross(ek, L + 21, t4);
    for (int r = 0; r < 3; ++r) dv[r] = t1[r] + t3[r] + t4[r];
    const double pb[3] = {p[0] - x[6], p[1] - x[7], p[2] - x[8]}; double t5[3]; v3_cross(ek, pb, t5); *dpz = t5[2];
  } else if (c < 24) {
    const int j = c - 12; if (j / 3 != chain) return;
    const int l = j - 3 * chain; const double* al = L + 3 * l;
    const double dl[3] = {p[0] - L[9 + 3 * l], p[1] - L[10 + 3 * l], p[2] - L[11 + 3 * l]};
    double dp[3]; v3_cross(al, dl, dp);                 // ∂p/∂q_l
    *dpz = dp[2];
    double acc[3]; v3_cross(K + KW_OM, dp, acc);       // ω × ∂d/∂q_l
    for (int jj = 0; jj < 3; ++jj) {
      const double qd = u[12 + 3 * chain + jj]; const double* aj = L + 3 * jj;
      const double dj[3] = {p[0] - L[9 + 3 * jj], p[1] - L[10 + 3 * jj], p[2] - L[11 + 3 * jj]};
      double t[3];
      if (jj > l) { double t1[3], t2[3], t3[3], t4[3]; v3_cross(al, aj, t1); v3_cross(t1, dj, t2); v3_cross(al, dj, t3); v3_cross(aj, t3, t4); for (int r = 0; r < 3; ++r) t[r] = t2[r] + t4[r]; }
      else { v3_cross(aj, dp, t); }
      for (int r = 0; r < 3; ++r) acc[r] += qd * t[r];
    }
    for (int r = 0; r < 3; ++r) dv[r] = acc[r];
  } else if (c < 42) { }
  else if (c < 54) {
    const int j = c - 42; if (j / 3 != chain) return;
    const int jj = j - 3 * chain; const double dj[3] = {p[0] - L[9 + 3 * jj], p[1] - L[10 + 3 * jj], p[2] - L[11 + 3 * jj]};
    v3_cross(L + 3 * jj, dj, dv);
  }
}

// foot_vel_jac_col for a wave that holds one (column, contact) task per lane: the four non-trivial classes (columns 3..5, 9..11, the
// leg's own joint angles 12..23 and joint velocities 42..53) evaluated by every lane with clamped indices, selected by the lane's
// class (see flow_jac_col_wave).  Also returns the foot velocity v and height pz of the contact (the constraint values need them).
__device__ __forceinline__ void foot_vel_jac_col_wave(const double* mb, const double* x, const double* u, const double* K, int contact, int c,
                                                       double* dv, double* dpz, double* v, double* pz) {
  const double m = mb[MB_ROBOTMASS];
  const int chain = contact_to_chain(contact);
  const double* L = K + KW_LEG + KW_LEGSZ * chain;
  const double p[3] = {L[18], L[19], L[20]};
  const double d[3] = {p[0] - K[KW_COM], p[1] - K[KW_COM + 1], p[2] - K[KW_COM + 2]};
  const double om[3] = {K[KW_OM], K[KW_OM + 1], K[KW_OM + 2]};
  { double cv[3]; v3_cross(om, d, cv); for (int r = 0; r < 3; ++r) v[r] = x[r] + cv[r] + L[21 + r]; *pz = p[2]; }
  const int cls = (c < 6) ? 0 : (c < 12) ? 1 : (c < 24) ? 2 : 3;
  const int k3 = (cls == 0) ? c - 3 : (cls == 1) ? c - 9 : 0;
  // class 0: d/d(h_ang)_k
  double a[3];
  { const double w[3] = {K[KW_IINV + k3] * m, K[KW_IINV + 3 + k3] * m, K[KW_IINV + 6 + k3] * m}; v3_cross(w, d, a); }
  // class 1: d/d(zyx)_k
  double b[3], bz;
  { double ek[3], dw[3]; d_omega_dtheta(K, k3, ek, dw);
    double t1[3], t2[3], t3[3], t4[3]; v3_cross(dw, d, t1); v3_cross(ek, d, t2); v3_cross(om, t2, t3); v3_cross(ek, L + 21, t4);
    for (int r = 0; r < 3; ++r) b[r] = t1[r] + t3[r] + t4[r];
    const double pb[3] = {p[0] - x[6], p[1] - x[7], p[2] - x[8]}; double t5[3]; v3_cross(ek, pb, t5); bz = t5[2]; }
  // classes 2, 3: the leg's own joint l (angle resp. velocity)
  const int l = (cls == 2) ? c - 12 - 3 * chain : (cls == 3) ? c - 42 - 3 * chain : 0;
  const double al[3] = {L[3 * l], L[3 * l + 1], L[3 * l + 2]};
  const double dl[3] = {p[0] - L[9 + 3 * l], p[1] - L[10 + 3 * l], p[2] - L[11 + 3 * l]};
  double dp[3]; v3_cross(al, dl, dp);                  // ∂p/∂q_l  (= class 3's answer)
  double acc[3]; v3_cross(om, dp, acc);                // ω × ∂d/∂q_l
#pragma unroll
  for (int jj = 0; jj < 3; ++jj) {
    const double qd = u[12 + 3 * chain + jj]; const double* aj = L + 3 * jj;
    const double dj[3] = {p[0] - L[9 + 3 * jj], p[1] - L[10 + 3 * jj], p[2] - L[11 + 3 * jj]};
    double t1[3], t2[3], t3[3], t4[3], t0[3];
    v3_cross(aj, dp, t0);
    if (jj == 0) { for (int r = 0; r < 3; ++r) acc[r] += qd * t0[r]; continue; }      // l >= 0: joint 0 is never behind the lane's joint — the four products of the other branch are not formed at all
    v3_cross(al, aj, t1); v3_cross(t1, dj, t2); v3_cross(al, dj, t3); v3_cross(aj, t3, t4);
    for (int r = 0; r < 3; ++r) acc[r] += qd * ((jj > l) ? t2[r] + t4[r] : t0[r]);
  }
  for (int r = 0; r < 3; ++r) dv[r] = (cls == 0) ? a[r] : (cls == 1) ? b[r] : (cls == 2) ? acc[r] : dp[r];
  *dpz = (cls == 1) ? bz : (cls == 2) ? dp[2] : 0.0;
}

// rotation matrix -> quaternion xyzw ([upstream] ocs2 matrixToQuaternion branches)
__device__ __forceinline__ void mat_to_quat(const double* R, double* q) {
  double t;
  if (R[8] < 0.0) {
    if (R[0] > R[4]) { t = 1.0 + R[0] - R[4] - R[8]; q[0] = t; q[1] = R[3] + R[1]; q[2] = R[2] + R[6]; q[3] = R[7] - R[5]; }
    else             { t = 1.0 - R[0] + R[4] - R[8]; q[0] = R[3] + R[1]; q[1] = t; q[2] = R[7] + R[5]; q[3] = R[2] - R[6]; }
  } else {
    if (R[0] < -R[4]) { t = 1.0 - R[0] - R[4] + R[8]; q[0] = R[2] + R[6]; q[1] = R[7] + R[5]; q[2] = t; q[3] = R[3] - R[1]; }
    else              { t = 1.0 + R[0] + R[4] + R[8]; q[0] = R[7] - R[5]; q[1] = R[2] - R[6]; q[2] = R[3] - R[1]; q[3] = t; }
  }
  const double s = 0.5 / sqrt(t);
  for (int i = 0; i < 4; ++i) q[i] *= s;
}
// a5: EE pose error g(6) (EndEffectorConstraint.cpp:36-80) from the arm part of the workspace
__device__ __forceinline__ void ee_error_arm(const double* A, const double* pref, const double* qref, double* qee, double* g) {
  for (int i = 0; i < 3; ++i) g[i] = A[36 + i] - pref[i];
  mat_to_quat(A + 39, qee);
  double c[3]; v3_cross(qee, qref, c);
  for (int i = 0; i < 3; ++i) g[3 + i] = qee[3] * qref[i] - qref[3] * qee[i] + c[i];
}
__device__ __forceinline__ void ee_error(const double* K, const double* pref, const double* qref, double* qee, double* g) { ee_error_arm(K + KW_ARM, pref, qref, qee, g); }
// column of the 6x30 EE error Jacobian wrt x_c; returns false for structurally zero columns.
// c in 6..8: base position, 9..11: zyx, 24..29: arm joints
__device__ __forceinline__ bool ee_jac_col(const double* x, const double* K, const double* qee, const double* qref, int c, double* col6) {
  const double* A = K + KW_ARM; double phi[3];
  for (int i = 0; i < 6; ++i) col6[i] = 0.0;
  if (c >= 6 && c < 9) { col6[c - 6] = 1.0; return true; }
  if (c >= 9 && c < 12) {
    const int k = c - 9; phi[0] = K[KW_E + k]; phi[1] = K[KW_E + 3 + k]; phi[2] = K[KW_E + 6 + k];
    const double d[3] = {A[36] - x[6], A[37] - x[7], A[38] - x[8]}; v3_cross(phi, d, col6);
  } else if (c >= 24 && c < 30) {
    const int jj = c - 24; phi[0] = A[3 * jj]; phi[1] = A[3 * jj + 1]; phi[2] = A[3 * jj + 2];
    const double d[3] = {A[36] - A[18 + 3 * jj], A[37] - A[19 + 3 * jj], A[38] - A[20 + 3 * jj]}; v3_cross(phi, d, col6);
  } else return false;
  // δq for a world-frame rotation δφ: δq_v = ½(q_w φ + φ × q_v), δq_w = −½ φ·q_v ; g = q_w r_v − r_w q_v + q_v × r_v
  double cx[3]; v3_cross(phi, qee, cx);
  const double dqv[3] = {0.5 * (qee[3] * phi[0] + cx[0]), 0.5 * (qee[3] * phi[1] + cx[1]), 0.5 * (qee[3] * phi[2] + cx[2])};
  const double dqw = -0.5 * (phi[0] * qee[0] + phi[1] * qee[1] + phi[2] * qee[2]);
  double c2[3]; v3_cross(dqv, qref, c2);
  for (int i = 0; i < 3; ++i) col6[3 + i] = dqw * qref[i] - qref[3] * dqv[i] + c2[i];
  return true;
}
