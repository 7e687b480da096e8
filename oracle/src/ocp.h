// oracle/src/ocp.h — TEST INFRASTRUCTURE (CPU oracle). The optimal control problem the reference
// assembles in qm_interface/src/QMInterface.cpp:79-142, restated term by term (SURVEY.md §8 a2–a10).
// PARITY UNPINNED (no reference goldens exist; upstream OCS2/Pinocchio absent — see oracle/README.md).
#pragma once
#include "geom.h"
#include "la.h"

// ------------------------------------------------------------------------------------------------
// SRBD centroidal map  [upstream ocs2_centroidal_model updateCentroidalDynamics, type 1]
// call sites: qm_interface/src/QMInterface.cpp:369-372, qm_wbc/src/WbcBase.cpp:205
// ------------------------------------------------------------------------------------------------
template <class T> struct Srbd { M3<T> Rb, E, A12, A22, A22inv; V3<T> rw, com; };
template <class T> inline void srbd(const Model& M, const T* q, Srbd<T>& c) {
  c.Rb = rotZyx(q[3], q[4], q[5]);
  c.E = eulerZyxE(q[3], q[4]);
  c.rw = c.Rb * v3d<T>(M.mb + MB_RNOM);                       // comToBasePositionInWorld
  c.com = v3<T>(q[0], q[1], q[2]) - c.rw;
  c.A12 = scale(skew(c.rw) * c.E, M.robotMass());
  c.A22 = (c.Rb * m3d<T>(M.mb + MB_INOM)) * (transpose(c.Rb) * c.E);
  c.A22inv = inverse(c.A22);
}
// v_base(6) = A_b⁻¹ (m h)   [upstream computeFloatingBaseCentroidalMomentumMatrixInverse +
// CentroidalModelPinocchioMapping::getPinocchioJointVelocity, SRBD branch]
template <class T> inline void baseVelocity(const Model& M, const Srbd<T>& c, const T* h, T* v6) {
  const double m = M.robotMass();
  V3<T> ml = v3<T>(h[0] * m, h[1] * m, h[2] * m), ma = v3<T>(h[3] * m, h[4] * m, h[5] * m);
  V3<T> thd = c.A22inv * ma;
  V3<T> corr = (c.A12 * c.A22inv) * ma;
  for (int i = 0; i < 3; ++i) { v6[i] = ml[i] * (1.0 / m) - corr[i] * (1.0 / m); v6[3 + i] = thd[i]; }
}

// LOCAL_WORLD_ALIGNED velocity of frame f given Pinocchio velocity v(24) = [pdot, zyx rates, qd_j]
template <class T> inline void frameVelocity(const Model& M, const Kin<T>& k, const T* q, const T* v, int f, V3<T>& lin, V3<T>& ang) {
  M3<T> E = eulerZyxE(q[3], q[4]);
  ang = E * v3<T>(v[3], v[4], v[5]);
  const V3<T> p = k.fp[f];
  lin = v3<T>(v[0], v[1], v[2]) + cross(ang, p - k.p[0]);
  std::vector<int> chain; for (int b = M.fparent(f); b > 0; b = M.parent(b - 1)) chain.push_back(b);
  for (int idx = (int)chain.size() - 1; idx >= 0; --idx) {
    const int b = chain[idx], j = b - 1;
    V3<T> aw = k.R[b] * v3d<T>(M.axis(j));
    lin = lin + cross(aw, p - k.p[b]) * v[6 + j];
    ang = ang + aw * v[6 + j];
  }
}

// ------------------------------------------------------------------------------------------------
// a3: flow map  xdot = f(x,u)   (qm_interface/src/dynamics/QMDynamicsAD.cpp:22-33 ->
//      [upstream PinocchioCentroidalDynamicsAD::getValueCppAd, getNormalizedCentroidalMomentumRate])
// ------------------------------------------------------------------------------------------------
template <class T> inline void flowMap(const Model& M, const T* x, const T* u, T* dx) {
  const T* q = x + 6;
  Kin<T> k; forwardKinematics(M, q, k);
  Srbd<T> c; srbd(M, q, c);
  const double m = M.robotMass();
  V3<T> lin = v3<T>(T(0.0), T(0.0), T(-9.81 * m)), ang = v3<T>(T(0.0), T(0.0), T(0.0));
  for (int i = 0; i < 4; ++i) {
    V3<T> F = v3<T>(u[3 * i], u[3 * i + 1], u[3 * i + 2]);
    lin = lin + F;
    ang = ang + cross(k.fp[i] - c.com, F);
  }
  for (int i = 0; i < 3; ++i) { dx[i] = lin[i] * (1.0 / m); dx[3 + i] = ang[i] * (1.0 / m); }
  baseVelocity(M, c, x, dx + 6);
  for (int j = 0; j < QM_NJ; ++j) dx[12 + j] = u[12 + j];
}

// foot i position & LWA linear velocity as functions of (x,u)  (QMInterface.cpp:363-379 kinematics)
template <class T> inline void footPosVel(const Model& M, const T* x, const T* u, int i, V3<T>& pos, V3<T>& vel) {
  const T* q = x + 6;
  Kin<T> k; forwardKinematics(M, q, k);
  Srbd<T> c; srbd(M, q, c);
  T v[QM_NQ];
  baseVelocity(M, c, x, v);
  for (int j = 0; j < QM_NJ; ++j) v[6 + j] = u[12 + j];
  V3<T> ang; frameVelocity(M, k, q, v, i, vel, ang);
  pos = k.fp[i];
}

// rotation matrix -> quaternion (x,y,z,w)   [upstream ocs2_robotic_tools matrixToQuaternion]
template <class T> inline void matToQuat(const M3<T>& R, T* q /*xyzw*/) {
  T t;
  if (R(2, 2) < 0.0) {
    if (R(0, 0) > R(1, 1)) { t = 1.0 + R(0, 0) - R(1, 1) - R(2, 2); q[0] = t; q[1] = R(1, 0) + R(0, 1); q[2] = R(0, 2) + R(2, 0); q[3] = R(2, 1) - R(1, 2); }
    else                   { t = 1.0 - R(0, 0) + R(1, 1) - R(2, 2); q[0] = R(1, 0) + R(0, 1); q[1] = t; q[2] = R(2, 1) + R(1, 2); q[3] = R(0, 2) - R(2, 0); }
  } else {
    if (R(0, 0) < -R(1, 1)) { t = 1.0 - R(0, 0) - R(1, 1) + R(2, 2); q[0] = R(0, 2) + R(2, 0); q[1] = R(2, 1) + R(1, 2); q[2] = t; q[3] = R(1, 0) - R(0, 1); }
    else                    { t = 1.0 + R(0, 0) + R(1, 1) + R(2, 2); q[0] = R(2, 1) - R(1, 2); q[1] = R(0, 2) - R(2, 0); q[2] = R(1, 0) - R(0, 1); q[3] = t; }
  }
  T s = 0.5 / sqrt(t);
  for (int i = 0; i < 4; ++i) q[i] = q[i] * s;
}

// a5: EE pose error g(x) = [p_ee − p_ref ; quaternionDistance(q_ee, q_ref)]
// (qm_interface/src/constraint/EndEffectorConstraint.cpp:36-80; quaternionDistance [upstream]:
//  q.w * qRef.vec − qRef.w * q.vec + q.vec × qRef.vec)
template <class T> inline void eePoseError(const Model& M, const T* x, const double* pref, const double* qref /*xyzw*/, T* g) {
  Kin<T> k; forwardKinematics(M, x + 6, k);
  for (int i = 0; i < 3; ++i) g[i] = k.fp[4][i] - pref[i];
  T q[4]; matToQuat(k.fR[4], q);
  V3<T> qv = v3<T>(q[0], q[1], q[2]); V3<T> rv = v3d<T>(qref);
  V3<T> cr = cross(qv, rv);
  for (int i = 0; i < 3; ++i) g[3 + i] = rv[i] * q[3] - qv[i] * qref[3] + cr[i];
}

// ------------------------------------------------------------------------------------------------
// penalties  [upstream ocs2_core RelaxedBarrierPenalty / QuadraticPenalty], SURVEY.md B.5
// ------------------------------------------------------------------------------------------------
struct Barrier {
  double mu, delta;
  double value(double h) const { if (h > delta) return -mu * std::log(h); const double t = (h - 2.0 * delta) / delta; return mu * (-std::log(delta) + 0.5 * t * t - 0.5); }
  double d1(double h) const { if (h > delta) return -mu / h; return mu * (h - 2.0 * delta) / (delta * delta); }
  double d2(double h) const { if (h > delta) return mu / (h * h); return mu / (delta * delta); }
};

// ------------------------------------------------------------------------------------------------
// mode schedule, contact flags, swing-z planner, reference trajectory
// ------------------------------------------------------------------------------------------------
inline int findIndexInTimeArray(const Vec& t, double time) { return (int)(std::lower_bound(t.begin(), t.end(), time) - t.begin()); }   // [upstream lookup::findIndexInTimeArray]
inline void modeToFlags(int mode, bool f[4]) { f[0] = (mode >> 3) & 1; f[1] = (mode >> 2) & 1; f[2] = (mode >> 1) & 1; f[3] = mode & 1; }  // LF,RF,LH,RH

struct ModeSchedule {
  Vec ev; std::vector<int> modes;   // modes.size() == ev.size() + 1
  int modeAt(double t) const { return modes[findIndexInTimeArray(ev, t)]; }
};

struct CubicSpline {   // [upstream ocs2_legged_robot CubicSpline]
  double t0, t1, dt, c0, c1, c2, c3;
  void set(double ta, double pa, double va, double tb, double pb, double vb) {
    t0 = ta; t1 = tb; dt = tb - ta; const double dp = pb - pa, dv = vb - va;
    c0 = pa; c1 = va * dt; c2 = -(3.0 * va + dv) * dt + 3.0 * dp; c3 = (2.0 * va + dv) * dt - 2.0 * dp;
  }
  double tn(double t) const { return (t - t0) / dt; }
  double position(double t) const { const double s = tn(t); return c3 * s * s * s + c2 * s * s + c1 * s + c0; }
  double velocity(double t) const { const double s = tn(t); return (3.0 * c3 * s * s + 2.0 * c2 * s + c1) / dt; }
};
struct SplineCpg {     // [upstream SplineCpg]: lift-off -> apex (mid time) -> touch-down
  double mid; CubicSpline left, right;
  void set(double ts, double ps, double vs, double midHeight, double te, double pe, double ve) {
    mid = (ts + te) / 2.0; left.set(ts, ps, vs, mid, midHeight, 0.0); right.set(mid, midHeight, 0.0, te, pe, ve);
  }
  double position(double t) const { return t < mid ? left.position(t) : right.position(t); }
  double velocity(double t) const { return t < mid ? left.velocity(t) : right.velocity(t); }
};

// a10: [upstream SwingTrajectoryPlanner::update] with terrain height 0, config task.info:23-30
struct SwingPlanner {
  Vec ev; std::vector<SplineCpg> traj[4];
  int status = 0;   // <0: a swing phase is not enclosed by stance phases inside the schedule window
  void update(const Model& M, const ModeSchedule& ms) {
    const int P = (int)ms.modes.size(); ev = ms.ev; status = 0;
    for (int leg = 0; leg < 4; ++leg) {
      std::vector<char> c(P); for (int p = 0; p < P; ++p) { bool f[4]; modeToFlags(ms.modes[p], f); c[p] = f[leg]; }
      traj[leg].assign(P, SplineCpg());
      for (int p = 0; p < P; ++p) {
        if (!c[p]) {
          int s = -1; for (int ip = p - 1; ip >= 0; --ip) if (c[ip]) { s = ip; break; }
          int e = P - 1; for (int ip = p + 1; ip < P; ++ip) if (c[ip]) { e = ip - 1; break; }
          if (s < 0 || e >= P - 1) { status = -1; traj[leg][p].set(0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0); continue; }
          const double ts = ms.ev[s], te = ms.ev[e];
          const double scaling = std::min(1.0, (te - ts) / M.st[ST_SWING_TIME_SCALE]);
          traj[leg][p].set(ts, 0.0, scaling * M.st[ST_LIFTOFF_VEL], 0.0 + scaling * M.st[ST_SWING_HEIGHT], te, 0.0, scaling * M.st[ST_TOUCHDOWN_VEL]);
        } else {
          traj[leg][p].set(0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0);
        }
      }
    }
  }
  double zVel(int leg, double t) const { return traj[leg][findIndexInTimeArray(ev, t)].velocity(t); }
  double zPos(int leg, double t) const { return traj[leg][findIndexInTimeArray(ev, t)].position(t); }
};

// [upstream LinearInterpolation::timeSegment]: value = alpha * v[i] + (1-alpha) * v[i+1]
inline void timeSegment(double t, const Vec& ta, int& index, double& alpha) {
  const int n = (int)ta.size();
  if (n <= 1) { index = 0; alpha = 1.0; return; }
  int interval;
  { int part = findIndexInTimeArray(ta, t); interval = (part == 0 && t == ta.front()) ? 0 : part - 1; }   // findIntervalInTimeArray
  const int last = n - 1;
  if (interval >= 0) {
    if (interval < last) {
      const double len = ta[interval + 1] - ta[interval], till = ta[interval + 1] - t;
      index = interval;
      if (len > 2.0 * 1e-6) alpha = till / len;               // weakEpsilon
      else alpha = (till > 0.5 * len) ? 1.0 : 0.0;
    } else { index = std::max(last - 1, 0); alpha = 0.0; }
  } else { index = 0; alpha = 1.0; }
}

struct Target {   // TargetTrajectories: 37-dim knots (QmTargetTrajectoriesPublisher_node.cpp:44-68)
  Vec t; std::vector<Vec> x;
  Vec desiredState(double time) const {
    if (x.size() == 1) return x[0];
    int i; double a; timeSegment(time, t, i, a);
    Vec r(QM_NREF); for (int k = 0; k < QM_NREF; ++k) r[k] = a * x[i][k] + (1.0 - a) * x[i + 1][k];
    return r;
  }
  // EndEffectorConstraint::interpolateEndEffectorPose (EndEffectorConstraint.cpp:82-113): lerp + Eigen slerp
  void eePose(double time, double* pos, double* quat /*xyzw*/) const {
    if (x.size() > 1) {
      int i; double a; timeSegment(time, t, i, a);
      const double* l = x[i].data() + 30; const double* r = x[i + 1].data() + 30;
      for (int k = 0; k < 3; ++k) pos[k] = a * l[k] + (1.0 - a) * r[k];
      const double* ql = l + 3; const double* qr = r + 3; const double tt = 1.0 - a;
      const double one = 1.0 - 2.220446049250313e-16;
      const double d = ql[0] * qr[0] + ql[1] * qr[1] + ql[2] * qr[2] + ql[3] * qr[3];
      const double ad = std::fabs(d); double s0, s1;
      if (ad >= one) { s0 = 1.0 - tt; s1 = tt; }
      else { const double th = std::acos(ad), sth = std::sin(th); s0 = std::sin((1.0 - tt) * th) / sth; s1 = std::sin(tt * th) / sth; }
      if (d < 0) s1 = -s1;
      for (int k = 0; k < 4; ++k) quat[k] = s0 * ql[k] + s1 * qr[k];
    } else {
      for (int k = 0; k < 3; ++k) pos[k] = x[0][30 + k];
      for (int k = 0; k < 4; ++k) quat[k] = x[0][33 + k];
    }
  }
};

inline Vec weightCompensatingInput(const Model& M, const bool flags[4]) {   // [upstream ocs2_legged_robot utils]
  Vec u(QM_NU, 0.0); int n = 0; for (int i = 0; i < 4; ++i) n += flags[i];
  if (n > 0) { const double fz = M.robotMass() * 9.81 / n; for (int i = 0; i < 4; ++i) if (flags[i]) u[3 * i + 2] = fz; }
  return u;
}

// ------------------------------------------------------------------------------------------------
// per-node evaluations (double)
// ------------------------------------------------------------------------------------------------
struct Problem {
  const Model* M; ModeSchedule ms; SwingPlanner swing; Target target;
};

inline void flowMapValue(const Model& M, const Vec& x, const Vec& u, Vec& f) { f.assign(QM_NX, 0.0); flowMap<double>(M, x.data(), u.data(), f.data()); }

typedef Dual<60> D60;
inline void seedXU(const Vec& x, const Vec& u, D60* xs, D60* us) {
  for (int i = 0; i < QM_NX; ++i) xs[i] = D60::seed(x[i], i);
  for (int i = 0; i < QM_NU; ++i) us[i] = D60::seed(u[i], 30 + i);
}
// SEEDED forward mode (round 5).  A dual number carries one derivative slot per seeded INDEPENDENT variable; a variable that is not seeded is a constant of the evaluation,
// i.e. its Jacobian column is taken to be structurally zero (or is known in closed form and filled in by the caller).  Forward mode propagates every slot independently of the
// others, so the entries of the seeded columns are bit-identical to those of the full 60-slot evaluation — only the work shrinks (rounds 1-4 pushed 60 slots through every
// kinematics pass: 83 of the 88 ms of a single-instance MPC iteration).  qm_ad_full_seeding = true restores the full evaluation (tests/test_oracle.py compares the two entry by entry).
//   flow map:           h (6), zyx (3), leg joints (12), contact forces (12) -> 33 slots; the base position and the arm joints do not enter (com and feet move together; the arm's
//                       frames are not used), the joint-velocity columns are the identity on rows 12..29
//   feet pos / vel:     h (6), base pose (6), leg joints (12), leg joint velocities (12) -> 36 slots, ONE kinematics pass for the four feet (rounds 1-4: one per foot)
//   end-effector error: base pose (6), arm joints (6) -> 12 slots
static bool qm_ad_full_seeding = false;
struct SeedPlan { int xs[QM_NX], us[QM_NU]; };       // slot of x_i / u_i, −1: constant
inline SeedPlan seedPlanFlow() { SeedPlan p; for (int i = 0; i < 30; ++i) { p.xs[i] = -1; p.us[i] = -1; }
  for (int i = 0; i < 6; ++i) p.xs[i] = i; for (int i = 0; i < 3; ++i) p.xs[9 + i] = 6 + i; for (int i = 0; i < 12; ++i) { p.xs[12 + i] = 9 + i; p.us[i] = 21 + i; } return p; }
inline SeedPlan seedPlanFeet() { SeedPlan p; for (int i = 0; i < 30; ++i) { p.xs[i] = -1; p.us[i] = -1; }
  for (int i = 0; i < 24; ++i) p.xs[i] = i; for (int i = 0; i < 12; ++i) p.us[12 + i] = 24 + i; return p; }
template <int N> inline void seedPlanned(const SeedPlan& sp, const Vec& x, const Vec& u, Dual<N>* xs, Dual<N>* us) {
  for (int i = 0; i < QM_NX; ++i) xs[i] = (sp.xs[i] >= 0) ? Dual<N>::seed(x[i], sp.xs[i]) : Dual<N>(x[i]);
  for (int i = 0; i < QM_NU; ++i) us[i] = (sp.us[i] >= 0) ? Dual<N>::seed(u[i], sp.us[i]) : Dual<N>(u[i]);
}
// a3: dynamics linearisation by AD (what CppAD does in the reference)
inline void flowMapLinear(const Model& M, const Vec& x, const Vec& u, Vec& f, Mat& A, Mat& B) {
  f.assign(QM_NX, 0.0); A = Mat(QM_NX, QM_NX); B = Mat(QM_NX, QM_NU);
  if (qm_ad_full_seeding) {
    D60 xs[QM_NX], us[QM_NU], dx[QM_NX]; seedXU(x, u, xs, us);
    flowMap<D60>(M, xs, us, dx);
    for (int i = 0; i < QM_NX; ++i) { f[i] = dx[i].v; for (int j = 0; j < 30; ++j) { A(i, j) = dx[i].d[j]; B(i, j) = dx[i].d[30 + j]; } }
    return;
  }
  typedef Dual<33> D; static const SeedPlan sp = seedPlanFlow();
  D xs[QM_NX], us[QM_NU], dx[QM_NX]; seedPlanned<33>(sp, x, u, xs, us);
  flowMap<D>(M, xs, us, dx);
  for (int i = 0; i < QM_NX; ++i) {
    f[i] = dx[i].v;
    for (int j = 0; j < 30; ++j) { if (sp.xs[j] >= 0) A(i, j) = dx[i].d[sp.xs[j]]; if (sp.us[j] >= 0) B(i, j) = dx[i].d[sp.us[j]]; }
  }
  for (int j = 0; j < QM_NJ; ++j) B(12 + j, 12 + j) = 1.0;      // xdot_{12+j} = u_{12+j}
}
// feet positions and LOCAL_WORLD_ALIGNED velocities, all four from ONE kinematics pass (same arithmetic per foot as footPosVel)
template <class T> inline void feetPosVel(const Model& M, const T* x, const T* u, V3<T>* pos, V3<T>* vel) {
  const T* q = x + 6;
  Kin<T> k; forwardKinematics(M, q, k);
  Srbd<T> c; srbd(M, q, c);
  T v[QM_NQ];
  baseVelocity(M, c, x, v);
  for (int j = 0; j < QM_NJ; ++j) v[6 + j] = u[12 + j];
  for (int i = 0; i < 4; ++i) { V3<T> ang; frameVelocity(M, k, q, v, i, vel[i], ang); pos[i] = k.fp[i]; }
}

// a8: equality constraints at (t,x,u): rows ordered per foot LF,RF,LH,RH as added in
// QMInterface.cpp:116-131: [zeroForce(swing,3) | zeroVelocity(stance,3) | normalVelocity(swing,1)]
inline void equalityConstraints(const Problem& P, double t, const Vec& x, const Vec& u, bool linear, Vec& e, Mat& C, Mat& D) {
  const Model& M = *P.M; bool fl[4]; modeToFlags(P.ms.modeAt(t), fl);
  int nc = 0; for (int i = 0; i < 4; ++i) nc += fl[i] ? 3 : 4;
  e.assign(nc, 0.0); if (linear) { C = Mat(nc, QM_NX); D = Mat(nc, QM_NU); }
  const double gain = M.st[ST_POS_ERR_GAIN];
  // derivative of foot quantity `a` with respect to x_j / u_j under the active seeding
  typedef Dual<36> D36; static const SeedPlan sp = seedPlanFeet();
  V3<D60> pdF[4], vdF[4]; V3<D36> pdC[4], vdC[4]; V3<double> pv[4], vv[4];
  if (linear && qm_ad_full_seeding) { D60 xs[QM_NX], us[QM_NU]; seedXU(x, u, xs, us); feetPosVel<D60>(M, xs, us, pdF, vdF); }
  else if (linear) { D36 xs[QM_NX], us[QM_NU]; seedPlanned<36>(sp, x, u, xs, us); feetPosVel<D36>(M, xs, us, pdC, vdC); }
  else feetPosVel<double>(M, x.data(), u.data(), pv, vv);
  const bool full = qm_ad_full_seeding;
  auto dX = [&](const D60& aF, const D36& aC, int j) { return full ? aF.d[j] : (sp.xs[j] >= 0 ? aC.d[sp.xs[j]] : 0.0); };
  auto dU = [&](const D60& aF, const D36& aC, int j) { return full ? aF.d[30 + j] : (sp.us[j] >= 0 ? aC.d[sp.us[j]] : 0.0); };
  int r = 0;
  for (int i = 0; i < 4; ++i) {
    if (!fl[i]) {   // ZeroForceConstraint [upstream]
      for (int k = 0; k < 3; ++k) { e[r + k] = u[3 * i + k]; if (linear) D(r + k, 3 * i + k) = 1.0; }
      r += 3;
    }
    V3<double> p, v;
    if (linear) { for (int k = 0; k < 3; ++k) { p[k] = full ? pdF[i][k].v : pdC[i][k].v; v[k] = full ? vdF[i][k].v : vdC[i][k].v; } }
    else { p = pv[i]; v = vv[i]; }
    const V3<D60>& pd = pdF[i]; const V3<D60>& vd = vdF[i]; const V3<D36>& pc = pdC[i]; const V3<D36>& vc = vdC[i];
    if (fl[i]) {    // zero velocity: Av = I, b = 0, Ax = diag(0,0,gain) if gain != 0 (QMInterface.cpp:324-339)
      for (int k = 0; k < 3; ++k) {
        e[r + k] = v[k] + ((k == 2 && gain != 0.0) ? gain * p[2] : 0.0);
        if (linear) for (int j = 0; j < 30; ++j) {
          C(r + k, j) = dX(vd[k], vc[k], j) + ((k == 2 && gain != 0.0) ? gain * dX(pd[2], pc[2], j) : 0.0);
          D(r + k, j) = dU(vd[k], vc[k], j) + ((k == 2 && gain != 0.0) ? gain * dU(pd[2], pc[2], j) : 0.0);
        }
      }
      r += 3;
    } else {        // normal velocity: Av = [0 0 1], b = −zvel_ref(t) (− gain z_ref), Ax = [0 0 gain] (QMPreComputation.cpp:56-65)
      double b = -P.swing.zVel(i, t); if (gain != 0.0) b -= gain * P.swing.zPos(i, t);
      e[r] = b + v[2] + (gain != 0.0 ? gain * p[2] : 0.0);
      if (linear) for (int j = 0; j < 30; ++j) {
        C(r, j) = dX(vd[2], vc[2], j) + (gain != 0.0 ? gain * dX(pd[2], pc[2], j) : 0.0);
        D(r, j) = dU(vd[2], vc[2], j) + (gain != 0.0 ? gain * dU(pd[2], pc[2], j) : 0.0);
      }
      r += 1;
    }
  }
}

// a2 + a6 + a7 + a5: intermediate cost L(t,x,u) and its quadratic model (not yet scaled by dt)
struct CostQuad { double f; Vec q, r; Mat Q, R, P; };   // P = d²L/du dx (nu x nx)
inline void eeSoftCost(const Problem& P, double t, const Vec& x, double muPos, double muOri, bool quad, double& f, Vec* gx, Mat* Hxx) {
  const Model& M = *P.M; double pref[3], qref[4]; P.target.eePose(t, pref, qref);
  double mu[6] = {muPos, muPos, muPos, muOri, muOri, muOri};
  if (!quad) { double g[6]; eePoseError<double>(M, x.data(), pref, qref, g); for (int i = 0; i < 6; ++i) f += 0.5 * mu[i] * g[i] * g[i]; return; }
  if (qm_ad_full_seeding) {
    Dual<30> xs[QM_NX], g[6]; for (int i = 0; i < QM_NX; ++i) xs[i] = Dual<30>::seed(x[i], i);
    eePoseError<Dual<30>>(M, xs, pref, qref, g);
    for (int i = 0; i < 6; ++i) {
      f += 0.5 * mu[i] * g[i].v * g[i].v;
      for (int a = 0; a < 30; ++a) { (*gx)[a] += mu[i] * g[i].v * g[i].d[a]; for (int b = 0; b < 30; ++b) (*Hxx)(a, b) += mu[i] * g[i].d[a] * g[i].d[b]; }
    }
    return;
  }
  // the end-effector pose depends on the base pose (x[6..11]) and the arm joints (x[24..29]) only: 12 slots, the other rows / columns of the Gauss-Newton term are zero
  Dual<12> xs[QM_NX], g[6]; int var[12];
  for (int i = 0; i < QM_NX; ++i) xs[i] = Dual<12>(x[i]);
  for (int k = 0; k < 6; ++k) { var[k] = 6 + k; var[6 + k] = 24 + k; }
  for (int k = 0; k < 12; ++k) xs[var[k]] = Dual<12>::seed(x[var[k]], k);
  eePoseError<Dual<12>>(M, xs, pref, qref, g);
  for (int i = 0; i < 6; ++i) {
    f += 0.5 * mu[i] * g[i].v * g[i].v;
    for (int a = 0; a < 12; ++a) { (*gx)[var[a]] += mu[i] * g[i].v * g[i].d[a]; for (int b = 0; b < 12; ++b) (*Hxx)(var[a], var[b]) += mu[i] * g[i].d[a] * g[i].d[b]; }
  }
}
// softIneq = false: without the relaxed-barrier terms a6 / a7 (the hard-inequality interior-point solver carries the arm boxes and the friction cones as constraints, ipm.h)
inline void intermediateCost(const Problem& P, double t, const Vec& x, const Vec& u, bool quad, CostQuad& c, bool softIneq = true) {
  const Model& M = *P.M; const double* st = M.st;
  bool fl[4]; modeToFlags(P.ms.modeAt(t), fl);
  c.f = 0.0;
  if (quad) { c.q.assign(QM_NX, 0.0); c.r.assign(QM_NU, 0.0); c.Q = Mat(QM_NX, QM_NX); c.R = Mat(QM_NU, QM_NU); c.P = Mat(QM_NU, QM_NX); }
  // a2: tracking cost (LeggedRobotQuadraticTrackingCost.h:34-40)
  Vec xr = P.target.desiredState(t); Vec un = weightCompensatingInput(M, fl);
  Vec dx(QM_NX), du(QM_NU); for (int i = 0; i < QM_NX; ++i) dx[i] = x[i] - xr[i]; for (int i = 0; i < QM_NU; ++i) du[i] = u[i] - un[i];
  for (int i = 0; i < QM_NX; ++i) { c.f += 0.5 * st[ST_Q + i] * dx[i] * dx[i]; if (quad) { c.q[i] += st[ST_Q + i] * dx[i]; c.Q(i, i) += st[ST_Q + i]; } }
  for (int i = 0; i < QM_NU; ++i) { double s = 0; for (int j = 0; j < QM_NU; ++j) s += st[ST_R + 30 * i + j] * du[j]; c.f += 0.5 * du[i] * s; if (quad) { c.r[i] += s; for (int j = 0; j < QM_NU; ++j) c.R(i, j) += st[ST_R + 30 * i + j]; } }
  // a6: arm joint position / velocity soft box (QMInterface.cpp:177-259), offset term affects only the value
  if (softIneq) {
    Barrier bp{st[ST_JPOS_MU], st[ST_JPOS_DELTA]}, bv{st[ST_JVEL_MU], st[ST_JVEL_DELTA]};
    for (int i = 0; i < 6; ++i) {
      const double lo = M.mb[MB_QLO + 12 + i], hi = M.mb[MB_QHI + 12 + i], z = x[24 + i];
      c.f += bp.value(z - lo) + bp.value(hi - z) - (bp.value(0.0 - lo) + bp.value(hi - 0.0));
      if (quad) { c.q[24 + i] += bp.d1(z - lo) - bp.d1(hi - z); c.Q(24 + i, 24 + i) += bp.d2(z - lo) + bp.d2(hi - z); }
      const double vlo = st[ST_JVEL_LO + i], vhi = st[ST_JVEL_HI + i], w = u[24 + i];
      c.f += bv.value(w - vlo) + bv.value(vhi - w) - (bv.value(0.0 - vlo) + bv.value(vhi - 0.0));
      if (quad) { c.r[24 + i] += bv.d1(w - vlo) - bv.d1(vhi - w); c.R(24 + i, 24 + i) += bv.d2(w - vlo) + bv.d2(vhi - w); }
    }
  }
  // a7: friction cone soft constraint per stance foot (QMInterface.cpp:344-358 -> [upstream FrictionConeConstraint])
  if (softIneq) {
    Barrier bf{st[ST_FRIC_MU], st[ST_FRIC_DELTA]};
    const double muf = st[ST_FRIC_COEF], reg = st[ST_FRIC_REG], shift = st[ST_FRIC_SHIFT];
    for (int i = 0; i < 4; ++i) if (fl[i]) {
      const double Fx = u[3 * i], Fy = u[3 * i + 1], Fz = u[3 * i + 2];
      const double T2 = Fx * Fx + Fy * Fy + reg, Tn = std::sqrt(T2), T3 = Tn * Tn * Tn;
      const double h = muf * Fz - Tn;
      c.f += bf.value(h);
      if (quad) {
        const double dh[3] = {-Fx / Tn, -Fy / Tn, muf};
        double ddh[3][3] = {{-(Fy * Fy + reg) / T3, Fx * Fy / T3, 0.0}, {Fx * Fy / T3, -(Fx * Fx + reg) / T3, 0.0}, {0.0, 0.0, 0.0}};
        const double p1 = bf.d1(h), p2 = bf.d2(h);
        for (int a = 0; a < 3; ++a) { c.r[3 * i + a] += p1 * dh[a]; for (int b = 0; b < 3; ++b) c.R(3 * i + a, 3 * i + b) += p2 * dh[a] * dh[b] + p1 * ddh[a][b]; }
        // hessianDiagonalShift: d²h/du² and d²h/dx² get −shift on their whole diagonals
        for (int a = 0; a < QM_NU; ++a) c.R(a, a) += p1 * (-shift);
        for (int a = 0; a < QM_NX; ++a) c.Q(a, a) += p1 * (-shift);
      }
    }
  }
  // a5: EE pose soft constraint (QMInterface.cpp:103, 147-172)
  eeSoftCost(P, t, x, st[ST_MU_EE_POS], st[ST_MU_EE_ORI], quad, c.f, quad ? &c.q : nullptr, quad ? &c.Q : nullptr);
}
inline void terminalCost(const Problem& P, double t, const Vec& x, bool quad, CostQuad& c) {
  c.f = 0.0; if (quad) { c.q.assign(QM_NX, 0.0); c.Q = Mat(QM_NX, QM_NX); }
  eeSoftCost(P, t, x, P.M->st[ST_MU_EEF_POS], P.M->st[ST_MU_EEF_ORI], quad, c.f, quad ? &c.q : nullptr, quad ? &c.Q : nullptr);
}
