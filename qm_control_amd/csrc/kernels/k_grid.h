// k_grid.h — K0: per-instance time grid, contact-mode lookup, swing-height references, reference
// interpolation and cold-start trajectories.  One thread per instance (integer / scalar work; B threads).
//
// Restates [upstream] timeDiscretizationWithEvents, ModeSchedule::modeAtTime and
// SwingTrajectoryPlanner (B.2, B.3; config qm_controllers/config/task.info:23-30), TargetTrajectories
// interpolation incl. EndEffectorConstraint::interpolateEndEffectorPose
// (qm_interface/src/constraint/EndEffectorConstraint.cpp:82-113) and QMInitializer::compute
// (qm_interface/src/initialization/QMInitializer.cpp:33-41).  The integer outputs (node event tags,
// modes) must be bit-exact with the reference semantics; event times are consumed as given.
#pragma once
#include "qm_dev_common.h"

#define QM_WEAK_EPS 1e-6                       /* numeric_traits::weakEpsilon<double>  */
#define QM_LIMIT_EPS 2.220446049250313e-16     /* numeric_traits::limitEpsilon<double> */

struct QmGridArgs {
  const double* mb; const double* st;
  int B, nmax, nref, nev;
  const double* t0;       // [B]
  const double* x0;       // [B][30]
  const double* ref_t;    // [B][nref]
  const double* ref_x;    // [B][nref][37]
  const double* ev;       // [B][nev]      event times of the mode schedule
  const int* modes;       // [B][nev+1]
  double horizon;
  int* n_nodes;           // [B]
  double* node_t;         // [nmax][B]
  double* node_ts;        // [nmax][B]
  double* node_dt;        // [nmax][B]
  int* node_ev;           // [nmax][B]
  int* node_mode;         // [nmax][B]
  double* zvel; double* zpos;   // [nmax][B][4]
  double* xref;           // [nmax][B][30]
  double* eeref;          // [nmax][B][7]
  int* ncap_dev; volatile int* host_ncap;   // optional: {max n_nodes, tickets, all-stance flag} on the device; host-visible word the last block publishes the batch's largest node count in
                                            // (the per-node launches that follow cover only that many nodes per instance: empty workgroups are not free)
  double* x; double* u;   // [nmax][B][30] initial guess (cold start, or warm start from the previous primal solution)
  // warm start ([upstream ocs2_sqp multiple_shooting::initializeStateInputTrajectories]): previous grid + primal solution; warm == 0 -> cold
  int warm; const int* prev_n; const double* prev_t; const int* prev_ev; const double* prev_xs; const double* prev_us;
  int* status;            // [B] 0 ok, -1 too many nodes / bad step or horizon, -2 swing phase not enclosed by stance, -3..-5 the gait front-end's sticky status
  const int* front_status; // [B] or null: status of the device-resident GaitSchedule of this batch (k_front.h) — a failed schedule update must not be reported as 0
};

__device__ __forceinline__ int grid_find_index(const double* ev, int nev, double t) {   // std::lower_bound
  int k = 0; while (k < nev && ev[k] < t) ++k; return k;
}
__device__ __forceinline__ void cubic_eval(double t0, double p0, double v0, double t1, double p1, double v1, double t, double* pos, double* vel) {
  const double dt = t1 - t0, dp = p1 - p0, dv = v1 - v0;
  const double c0 = p0, c1 = v0 * dt, c2 = -(3.0 * v0 + dv) * dt + 3.0 * dp, c3 = (2.0 * v0 + dv) * dt - 2.0 * dp;
  const double s = (t - t0) / dt;
  *pos = c3 * s * s * s + c2 * s * s + c1 * s + c0;
  *vel = (3.0 * c3 * s * s + 2.0 * c2 * s + c1) / dt;
}
// [upstream LinearInterpolation::timeSegment]
__device__ __forceinline__ void grid_time_segment(const double* ta, int n, double t, int* index, double* alpha) {
  if (n <= 1) { *index = 0; *alpha = 1.0; return; }
  const int part = grid_find_index(ta, n, t);
  const int interval = (part == 0 && t == ta[0]) ? 0 : part - 1;
  const int last = n - 1;
  if (interval >= 0) {
    if (interval < last) {
      const double len = ta[interval + 1] - ta[interval], till = ta[interval + 1] - t;
      *index = interval;
      *alpha = (len > 2.0 * QM_WEAK_EPS) ? till / len : ((till > 0.5 * len) ? 1.0 : 0.0);
    } else { *index = (last - 1 > 0) ? last - 1 : 0; *alpha = 0.0; }
  } else { *index = 0; *alpha = 1.0; }
}

__global__ void qm_grid_kernel(QmGridArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  int n = 0, has18 = 0;
  if (b < a.B) {
  const double* ev = a.ev + (size_t)b * a.nev; const int* modes = a.modes + (size_t)b * (a.nev + 1);
  const double t0 = a.t0[b], tf = t0 + a.horizon, dt = qm_ms_param(a.st, ST_SQP_DT);
  const double dtMin = a.st[ST_GRID_DT_MIN];  // [upstream] dt_min (default 10 limitEpsilon): a closer node overwrites its predecessor; QM_GRID_DT_MIN_ROBUST is the opt-in for fixed-rate loops
  int status = a.front_status ? a.front_status[b] : 0;
  // ---- time discretisation with events ----
  a.node_t[0 * a.B + b] = t0; a.node_ev[0 * a.B + b] = QM_EV_NONE; n = 1;
  int k = grid_find_index(ev, a.nev, t0);
  double nt = t0; double backT = t0;
  // a step that is not a positive finite number (or a NaN time) would never reach tf: one-node grid, status -1 (the solve kernels skip n < 2)
  const bool sane = dt > 0.0 && dt < 1.0e300 && t0 == t0 && tf > t0 && tf < 1.0e300;
  if (!sane) status = (status == 0) ? -1 : status;
  while (sane && backT < tf) {
    nt = nt + dt; int nevt = QM_EV_NONE; bool post = false;
    if (k < a.nev && nt >= ev[k]) { nt = ev[k]; nevt = QM_EV_PRE; post = true; ++k; }
    if (nt >= tf) { nt = tf; nevt = QM_EV_NONE; post = false; }
    if (nt > backT + dtMin) { if (n >= a.nmax) { status = (status == 0) ? -1 : status; break; } a.node_t[n * a.B + b] = nt; a.node_ev[n * a.B + b] = nevt; ++n; }
    else { a.node_t[(n - 1) * a.B + b] = nt; a.node_ev[(n - 1) * a.B + b] = nevt; }
    backT = nt;
    if (post) { if (n >= a.nmax) { status = (status == 0) ? -1 : status; break; } a.node_t[n * a.B + b] = nt; a.node_ev[n * a.B + b] = QM_EV_POST; ++n; }
  }
  a.n_nodes[b] = n;
  a.status[b] = status;
  // does the horizon hold a phase with three or four feet on the ground (17 / 18 reduced inputs: K1b's second instance, k_lq.h)?  Conservative: every phase between t0 and tf
  { const int k0 = grid_find_index(ev, a.nev, t0), k1e = grid_find_index(ev, a.nev, tf + QM_WEAK_EPS), k1 = (k1e < a.nev) ? k1e : a.nev;      // (a PostEvent node looks its mode up at t + weakEpsilon: an event within weakEpsilon behind tf still counts)
    for (int q = k0; q <= k1; ++q) { const int mq = modes[q]; if ((int)mode_flag(mq, 0) + (int)mode_flag(mq, 1) + (int)mode_flag(mq, 2) + (int)mode_flag(mq, 3) >= 3) has18 = 1; } }      // m = 14 + (feet on the ground) > 16
  }
  if (a.ncap_dev) {     // largest node count of the batch -> host (64-thread blocks = one wavefront each; the block that arrives last publishes and re-arms the counters)
    double m = (double)n, h = (double)has18;
    for (int off = 32; off > 0; off >>= 1) { m = fmax(m, __shfl_xor(m, off, 64)); h = fmax(h, __shfl_xor(h, off, 64)); }
    if ((threadIdx.x & 63) == 0) {
      atomicMax(a.ncap_dev, (int)m); if (h > 0.0) atomicOr(a.ncap_dev + 2, 1); __threadfence();
      // published word: largest node count | (some horizon holds an all-stance phase) << 16
      if (atomicAdd(a.ncap_dev + 1, 1) == (int)gridDim.x - 1) { const int f18 = atomicOr(a.ncap_dev + 2, 0); a.host_ncap[0] = atomicMax(a.ncap_dev, 0) | (f18 << 16); a.ncap_dev[0] = 0; a.ncap_dev[1] = 0; a.ncap_dev[2] = 0; __threadfence_system(); }
    }
  }
}

// [upstream PrimalSolution interpolation] segment of time t on a stored node grid of instance b: PreEvent nodes nudged down, PostEvent nodes
// nudged up by limitEpsilon (toInterpolationTime), then LinearInterpolation::timeSegment without materialising the array
__device__ __forceinline__ void grid_policy_segment(const double* node_t, const int* node_ev, int n, int B, int b, double t, int* index, double* alpha) {
  auto tt = [&](int i) { const int e = node_ev[i * B + b]; return node_t[i * B + b] + (e == QM_EV_POST ? QM_LIMIT_EPS : (e == QM_EV_PRE ? -QM_LIMIT_EPS : 0.0)); };
  int part = 0; while (part < n && tt(part) < t) ++part;
  const int interval = (part == 0 && t == tt(0)) ? 0 : part - 1; const int last = n - 1;
  if (n <= 1) { *index = 0; *alpha = 1.0; }
  else if (interval >= 0) {
    if (interval < last) { const double len = tt(interval + 1) - tt(interval), till = tt(interval + 1) - t; *index = interval; *alpha = (len > 2.0 * QM_WEAK_EPS) ? till / len : ((till > 0.5 * len) ? 1.0 : 0.0); }
    else { *index = (last - 1 > 0) ? last - 1 : 0; *alpha = 0.0; }
  } else { *index = 0; *alpha = 1.0; }
}

// K0b: per (node, instance): interval start/duration, mode, swing-z references, reference interpolation, cold start
// 72 registers: the WBC of the previous step runs beside this kernel on its own stream with 440 of a SIMD's 512 registers allocated (437 rounded up to the
// allocation granule of 8) — a wave of this kernel must fit into what is left or the whole launch waits for WBC wavefronts to retire (tests/test_kernel_budgets.py)
__global__ void QM_MAX_VGPRS(72) qm_grid_nodes_kernel(QmGridArgs a) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = g / a.B, b = g - i * a.B;
  if (i >= a.nmax) return;
  const int n = a.n_nodes[b];
  if (i >= n) return;
  const double* ev = a.ev + (size_t)b * a.nev; const int* modes = a.modes + (size_t)b * (a.nev + 1);
  int status = 0;
  const double mass = a.mb[MB_ROBOTMASS];
  const double liftV = a.st[ST_LIFTOFF_VEL], touchV = a.st[ST_TOUCHDOWN_VEL], swingH = a.st[ST_SWING_HEIGHT], tScale = a.st[ST_SWING_TIME_SCALE];
  const double* rt = a.ref_t + (size_t)b * a.nref; const double* rx = a.ref_x + (size_t)b * a.nref * QM_NREF;
  {
    const int nb = i * a.B + b;
    const double t = a.node_t[nb]; const int e = a.node_ev[nb];
    const double ts = (e == QM_EV_POST) ? t + QM_WEAK_EPS : t;
    double d = 0.0;
    if (i < n - 1 && e != QM_EV_PRE) { const double tn = a.node_t[(i + 1) * a.B + b]; const int en = a.node_ev[(i + 1) * a.B + b]; d = ((en == QM_EV_PRE) ? tn - QM_WEAK_EPS : tn) - ts; }
    a.node_ts[nb] = ts; a.node_dt[nb] = d;
    const int p = grid_find_index(ev, a.nev, ts); const int mode = modes[p];
    a.node_mode[nb] = mode;
    // swing z references per contact
    for (int c = 0; c < 4; ++c) {
      double zp = 0.0, zv = 0.0;
      if (!mode_flag(mode, c)) {
        int s = -1; for (int ip = p - 1; ip >= 0; --ip) if (mode_flag(modes[ip], c)) { s = ip; break; }
        int f = -1; for (int ip = p + 1; ip <= a.nev; ++ip) if (mode_flag(modes[ip], c)) { f = ip - 1; break; }
        if (s < 0 || f < 0) { status = (status == 0) ? -2 : status; }
        else {
          const double tsw = ev[s], tew = ev[f]; const double sc = fmin(1.0, (tew - tsw) / tScale); const double mid = (tsw + tew) / 2.0;
          if (ts < mid) cubic_eval(tsw, 0.0, sc * liftV, mid, sc * swingH, 0.0, ts, &zp, &zv);
          else cubic_eval(mid, sc * swingH, 0.0, tew, 0.0, sc * touchV, ts, &zp, &zv);
        }
      }
      a.zvel[nb * 4 + c] = zv; a.zpos[nb * 4 + c] = zp;
    }
    // reference state / EE pose at ts
    {
      int idx; double al; grid_time_segment(rt, a.nref, ts, &idx, &al);
      if (a.nref == 1) { for (int q = 0; q < 30; ++q) a.xref[nb * 30 + q] = rx[q]; for (int q = 0; q < 7; ++q) a.eeref[nb * 7 + q] = rx[30 + q]; }
      else {
        const double* l = rx + idx * QM_NREF; const double* r = rx + (idx + 1) * QM_NREF;
        for (int q = 0; q < 30; ++q) a.xref[nb * 30 + q] = al * l[q] + (1.0 - al) * r[q];
        for (int q = 0; q < 3; ++q) a.eeref[nb * 7 + q] = al * l[30 + q] + (1.0 - al) * r[30 + q];
        // Eigen slerp(t = 1 - alpha)
        const double* ql = l + 33; const double* qr = r + 33; const double tt = 1.0 - al; const double one = 1.0 - QM_LIMIT_EPS;
        const double dd = ql[0] * qr[0] + ql[1] * qr[1] + ql[2] * qr[2] + ql[3] * qr[3]; const double ad = fabs(dd); double s0, s1;
        if (ad >= one) { s0 = 1.0 - tt; s1 = tt; } else { const double th = acos(ad), sth = sin(th); s0 = sin((1.0 - tt) * th) / sth; s1 = sin(tt * th) / sth; }
        if (dd < 0.0) s1 = -s1;
        for (int q = 0; q < 4; ++q) a.eeref[nb * 7 + 3 + q] = s0 * ql[q] + s1 * qr[q];
      }
    }
    // initial guess.  Cold start: x_i = x0, u_i = weight compensating input of the node's mode (QMInitializer.cpp:33-41).
    // Warm start: x_0 = x0; interval j takes u_j = u_prev(start of j) and x_{j+1} = x_prev(end of j) while the previous solution covers it,
    // the initializer (u = weight compensation, x_{j+1} = x_j) beyond; PreEvent nodes carry no input and copy their state forward.
    int nst = 0; for (int c = 0; c < 4; ++c) nst += mode_flag(mode, c);
    const int np = (a.warm && a.prev_n) ? a.prev_n[b] : 0;
    const double tend = (np >= 2) ? a.prev_t[(np - 1) * a.B + b] : 0.0;
    auto istart = [&](int j) { const double tj = a.node_t[j * a.B + b]; return (a.node_ev[j * a.B + b] == QM_EV_POST) ? tj + QM_WEAK_EPS : tj; };
    auto iend = [&](int j) { const double tj = a.node_t[j * a.B + b]; return (a.node_ev[j * a.B + b] == QM_EV_PRE) ? tj - QM_WEAK_EPS : tj; };
    auto covered = [&](int j) { return np >= 2 && j < n - 1 && a.node_ev[j * a.B + b] != QM_EV_PRE && !(istart(j) > tend || iend(j + 1) > tend); };
    int j = i; while (j > 0 && !covered(j - 1)) --j;                 // x_i = x_j: the closest earlier node whose state the previous solution supplies
    if (j == 0) { for (int q = 0; q < 30; ++q) a.x[nb * 30 + q] = a.x0[(size_t)b * 30 + q]; }
    else {
      int idx; double al; grid_policy_segment(a.prev_t, a.prev_ev, np, a.B, b, iend(j), &idx, &al);
      const double* x0p = a.prev_xs + (size_t)(idx * a.B + b) * 30; const double* x1p = a.prev_xs + (size_t)((idx + 1) * a.B + b) * 30;
      for (int q = 0; q < 30; ++q) a.x[nb * 30 + q] = al * x0p[q] + (1.0 - al) * x1p[q];
    }
    if (covered(i)) {
      int idx; double al; grid_policy_segment(a.prev_t, a.prev_ev, np, a.B, b, ts, &idx, &al);
      const double* u0p = a.prev_us + (size_t)(idx * a.B + b) * 30; const double* u1p = a.prev_us + (size_t)((idx + 1) * a.B + b) * 30;
      for (int q = 0; q < 30; ++q) a.u[nb * 30 + q] = al * u0p[q] + (1.0 - al) * u1p[q];
    } else {
      for (int q = 0; q < 30; ++q) a.u[nb * 30 + q] = 0.0;
      if (e != QM_EV_PRE && nst > 0) for (int c = 0; c < 4; ++c) if (mode_flag(mode, c)) a.u[nb * 30 + 3 * c + 2] = mass * 9.81 / nst;
    }
  }
  if (status != 0 && a.status[b] == 0) a.status[b] = status;       // (K0 wrote the instance's status before this launch; racing writers of the same instance all carry -2)
}

// keeps the grid of the solve that produced the current primal solution (xs, us) before K0 overwrites it: the warm start of the
// next solve interpolates on it
struct QmSaveGridArgs { int B, nmax; const int* n_nodes; const double* node_t; const int* node_ev; int* prev_n; double* prev_t; int* prev_ev; };
__global__ void qm_save_grid_kernel(QmSaveGridArgs a) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = g / a.B, b = g - i * a.B;
  if (i >= a.nmax) return;
  if (i == 0) a.prev_n[b] = a.n_nodes[b];
  if (i < a.n_nodes[b]) { a.prev_t[g] = a.node_t[g]; a.prev_ev[g] = a.node_ev[g]; }
}
// closed-loop advance (perfect-tracking plant): t0 += dt and x0 <- the policy state at the new t0
struct QmAdvanceArgs { int B, nmax; const int* n_nodes; const double* node_t; const int* node_ev; const double* xs; double dt; double* t0; double* x0; };
__global__ void qm_advance_kernel(QmAdvanceArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.B) return;
  const double t = a.t0[b] + a.dt;
  int idx; double al; grid_policy_segment(a.node_t, a.node_ev, a.n_nodes[b], a.B, b, t, &idx, &al);
  const int i1 = (a.n_nodes[b] > 1) ? idx + 1 : idx;
  for (int q = 0; q < 30; ++q) a.x0[(size_t)b * 30 + q] = al * a.xs[(size_t)(idx * a.B + b) * 30 + q] + (1.0 - al) * a.xs[(size_t)(i1 * a.B + b) * 30 + q];
  a.t0[b] = t;
}
