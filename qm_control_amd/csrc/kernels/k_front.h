// k_front.h — reference / gait front-end, batched and device resident.
//
// What the reference does on the host, once per command, for ONE robot
//   * qm_controllers/src/QmTargetTrajectoriesPublisher_node.cpp:25-208  cmd_vel / EE cmd_vel / EE goal pose -> 2-knot TargetTrajectories
//   * qm_controllers/src/GaitJoyPublisher.cpp:35-60 + gait.info:1-255     a gait name -> ModeSequenceTemplate message
//   * [upstream ocs2_legged_robot] GaitReceiver::preSolverRun -> GaitSchedule::insertModeSequenceTemplate, and
//     SwitchedModelReferenceManager::modifyReferences -> GaitSchedule::getModeSchedule(t − T, t + 2T) before every MPC call
//     (constructed at qm_interface/src/QMInterface.cpp:455-480 with phaseTransitionStanceTime, task.info:11)
// is done here for B instances at once, one thread per instance, on state that never leaves HBM: the mode schedule of every
// instance lives in [slot][B] arrays (instances are the fast index: a wave's accesses are contiguous), the target knots and the
// exported schedule go straight into the buffers K0 (k_grid.h) reads.  This is integer / byte work: event times are produced by the
// same f64 additions in the same order as the reference's std::vector code, so schedules are bit-exact.
#pragma once
#include "qm_dev_kin.h"

#define QM_GAIT_MAX_PHASES 16          /* longest template of gait.info: lindyhop, 12 phases */
#define QM_MODE_STANCE 15

// ---- template table (all gaits of gait.info), read-only ----
struct QmGaitTable {
  int n_gaits;
  const int* n_phases;          // [n_gaits]
  const double* times;          // [n_gaits][QM_GAIT_MAX_PHASES + 1] switchingTimes
  const int* modes;             // [n_gaits][QM_GAIT_MAX_PHASES]     modeSequence
};

// ---- per-instance GaitSchedule state: ModeSchedule{eventTimes[n], modeSequence[n + 1]} + current template ----
struct QmGaitState {
  int B, cap;                   // cap: event slots per instance
  int* n;                       // [B]
  double* ev;                   // [cap][B]
  int* mode;                    // [cap + 1][B]
  int* tpl;                     // [B] index into the table
  int* status;                  // [B] 0 ok, -3 schedule capacity exceeded, -4 tiling start not after the last event (upstream throws), -5 empty schedule.
                                //     STICKY: the first failure of an instance stays until qmhip_gait_reset; K0 copies it into the solver status of every
                                //     later MPC call (a failed instance keeps solving on its last good schedule, but never reports status 0 again)
};
__device__ __forceinline__ void gait_fail(const QmGaitState& s, int b, int st) { if (s.status[b] == 0) s.status[b] = st; }

// GaitSchedule::tileModeSequenceTemplate [upstream]: push startTime, then template phases until the last event >= finalTime, then STANCE.
// On entry the instance holds n events and n + 1 modes (slots 0..n); on success s.n[b] is the new event count.
__device__ __forceinline__ int gait_tile(const QmGaitTable& T, const QmGaitState& s, int b, int n, double startTime, double finalTime) {
  const int g = s.tpl[b]; const int np = T.n_phases[g];
  const double* tt = T.times + (size_t)g * (QM_GAIT_MAX_PHASES + 1); const int* tm = T.modes + (size_t)g * QM_GAIT_MAX_PHASES;
  // on every failure the instance is left with a CONSISTENT schedule (n events, modes 0..n): what was tiled so far, closed by STANCE
  if (np == 0) { s.n[b] = n; return 0; }                                // "the last subsystem continues for ever": nothing appended
  if (n > 0 && startTime <= s.ev[(size_t)(n - 1) * s.B + b]) { s.n[b] = n; return -4; }
  if (n >= s.cap) { s.n[b] = n; return -3; }
  s.ev[(size_t)n * s.B + b] = startTime; ++n;                              // events n, modes n
  double last = startTime;
  while (last < finalTime) {
    for (int i = 0; i < np; ++i) {
      if (n >= s.cap) { s.mode[(size_t)n * s.B + b] = QM_MODE_STANCE; s.n[b] = n; return -3; }
      s.mode[(size_t)n * s.B + b] = tm[i];
      last = last + (tt[i + 1] - tt[i]);                                   // eventTimes.back() + deltaTime: the reference's addition order
      s.ev[(size_t)n * s.B + b] = last; ++n;
    }
  }
  s.mode[(size_t)n * s.B + b] = QM_MODE_STANCE;                            // default final phase
  s.n[b] = n;
  return 0;
}
// std::lower_bound on the instance's event times
__device__ __forceinline__ int gait_lower_bound(const QmGaitState& s, int b, int n, double t) {
  int lo = 0, hi = n;
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (s.ev[(size_t)mid * s.B + b] < t) lo = mid + 1; else hi = mid; }
  return lo;
}

// reset every instance to the initial mode schedule (reference.info:28-39) and the default template (reference.info:41-52)
struct QmGaitResetArgs { QmGaitState s; int n0; const double* ev0; const int* mode0; int tpl0; };
__global__ void qm_gait_reset_kernel(QmGaitResetArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.s.B) return;
  for (int k = 0; k < a.n0; ++k) a.s.ev[(size_t)k * a.s.B + b] = a.ev0[k];
  for (int k = 0; k <= a.n0; ++k) a.s.mode[(size_t)k * a.s.B + b] = a.mode0[k];
  a.s.n[b] = a.n0; a.s.tpl[b] = a.tpl0; a.s.status[b] = 0;
}

// GaitSchedule::insertModeSequenceTemplate(template, startTime, finalTime) [upstream] for the instances that request one (req_tpl >= 0)
struct QmGaitInsertArgs { QmGaitTable T; QmGaitState s; const int* req_tpl; const double* start; const double* final_t; double phase_transition_stance_time; };
__global__ void qm_gait_insert_kernel(QmGaitInsertArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.s.B) return;
  const int g = a.req_tpl[b];
  if (g < 0 || g >= a.T.n_gaits) return;
  const QmGaitState& s = a.s;
  s.tpl[b] = g;
  int n = s.n[b]; const double startTime = a.start[b];
  const int index = gait_lower_bound(s, b, n, startTime);
  if (index < n) n = index;                                               // erase events [index, end) and modes [index + 1, end)
  double pts = a.phase_transition_stance_time;
  if (s.mode[(size_t)n * s.B + b] == QM_MODE_STANCE) pts = 0.0;           // modeSequence.back() (never empty: n + 1 >= 1 modes)
  if (pts > 0.0) {                                                        // intermediate stance phase
    if (n >= s.cap) { s.n[b] = n; gait_fail(s, b, -3); return; }
    s.ev[(size_t)n * s.B + b] = startTime; ++n; s.mode[(size_t)n * s.B + b] = QM_MODE_STANCE;
  }
  const int st = gait_tile(a.T, s, b, n, startTime + pts, a.final_t[b]);
  if (st != 0) gait_fail(s, b, st);
}

// GaitSchedule::getModeSchedule(lowerBoundTime, upperBoundTime) [upstream] with the bounds SwitchedModelReferenceManager::modifyReferences
// asks for, [t0 − T, t0 + 2T] (T = finalTime − initTime), followed by the export of the schedule into the solver's buffers
// ev[B][nev], modes[B][nev + 1] (unused slots: far-future events, STANCE — the layout scenarios/_pad_schedules uses).
// A failure (status != 0, sticky) leaves the solver's buffers on the last good schedule; K0 reports it (QmGridArgs::front_status).
struct QmGaitScheduleArgs { QmGaitTable T; QmGaitState s; const double* t0; double horizon; int nev; double* ev_out; int* modes_out; };
__global__ void qm_gait_schedule_kernel(QmGaitScheduleArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.s.B) return;
  const QmGaitState& s = a.s; const size_t B = (size_t)s.B;
  // MPC_BASE::run: finalTime = initTime + horizon; modifyReferences: timeHorizon = finalTime − initTime (as rounded), bounds initTime − T, finalTime + T
  const double initTime = a.t0[b], finalTime = initTime + a.horizon, th = finalTime - initTime, lower = initTime - th, upper = finalTime + th;
  int n = s.n[b];
  const int index = gait_lower_bound(s, b, n, lower);
  if (index > 0) {
    const int sh = index - 1;                                               // erase [begin, begin + index − 1): keep the event before `lower`
    if (sh > 0) {
      for (int k = 0; k + sh < n; ++k) s.ev[(size_t)k * B + b] = s.ev[(size_t)(k + sh) * B + b];
      for (int k = 0; k + sh < n + 1; ++k) s.mode[(size_t)k * B + b] = s.mode[(size_t)(k + sh) * B + b];
      n -= sh;
    }
    s.mode[b] = QM_MODE_STANCE;                                             // modeSequence.front(): the default initial phase
  }
  int st = 0;
  if (n == 0) st = -5;                                                      // upstream would erase end() − 1 of an empty vector
  else {
    const double tilingStart = s.ev[(size_t)(n - 1) * B + b];               // eventTimes.back()
    n -= 1;                                                                 // drop the trailing default STANCE phase (one event, one mode)
    st = gait_tile(a.T, s, b, n, tilingStart, upper);
  }
  if (st != 0) { gait_fail(s, b, st); return; }
  if (s.status[b] != 0) return;                                             // failed earlier: the export stays on the last good schedule
  n = s.n[b];
  double* eo = a.ev_out + (size_t)b * a.nev; int* mo = a.modes_out + (size_t)b * (a.nev + 1);
  if (n > a.nev) { gait_fail(s, b, -3); return; }
  for (int k = 0; k < n; ++k) eo[k] = s.ev[(size_t)k * B + b];
  for (int k = 0; k <= n; ++k) mo[k] = s.mode[(size_t)k * B + b];
  const double lastev = (n > 0) ? eo[n - 1] : upper;
  for (int k = n; k < a.nev; ++k) { eo[k] = lastev + 1.0e3 * (double)(k - n + 1); mo[k + 1] = QM_MODE_STANCE; }
}

// ---- commands -> TargetTrajectories (2 knots of [0_6, base pose(6), defaultJointState(18), EE pose xyz + quat xyzw(7)]) ----
#define QM_CMD_NONE   0   /* keep the current target */
#define QM_CMD_VEL    1   /* cmd[0..3] = vx, vy, vz (base frame), yaw rate      cmdVelToTargetTrajectories,     _node.cpp:71-116 */
#define QM_CMD_EE_VEL 2   /* cmd[0..2] = EE linear velocity (tool frame)        EeCmdVelToTargetTrajectories,   _node.cpp:121-165 */
#define QM_CMD_EE_GOAL 3  /* cmd[0..6] = EE goal position, quaternion xyzw      EEgoalPoseToTargetTrajectories, _node.cpp:172-208 */
struct QmTargetArgs {
  const double* mb;
  int B, nref;
  const int* kind;              // [B]
  const double* cmd;            // [B][7]
  const double* t0;             // [B]    observation.time
  const double* x0;             // [B][30] observation.state
  const double* ee_state;       // [B][7] or null: forward kinematics of x0
  int ee_through_float;         // 1: EE state rounded to float as the qm_msgs::ee_state message does (QMController.cpp:246-256)
  double time_to_target;        // mpc.timeHorizon (task.info:140)
  double disp_velocity, rot_velocity, com_height;   // reference.info:1-4
  double* last_ee;              // [B][7] lastEeTarget_ (QmTargetTrajectoriesPublisher.h:52-54,  .cpp:108)
  double* ref_t;                // [B][nref]
  double* ref_x;                // [B][nref][37]
};
__device__ __forceinline__ void quat_to_R(const double* q /*xyzw*/, double* R) {   // Eigen::Quaternion::toRotationMatrix
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z, twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1.0 - (txx + tyy);
}
__global__ void __launch_bounds__(64) qm_target_kernel(QmTargetArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.B) return;
  const int kind = a.kind[b];
  if (kind == QM_CMD_NONE) return;
  const double* mb = a.mb; const double* x = a.x0 + (size_t)b * 30; const double* cmd = a.cmd + (size_t)b * 7; double* last = a.last_ee + (size_t)b * 7;
  const double t0 = a.t0[b], T = a.time_to_target;
  double ee[7];
  if (a.ee_state) { for (int q = 0; q < 7; ++q) ee[q] = a.ee_state[(size_t)b * 7 + q]; }
  else { double Kb[KW_LEG], A[KW_SIZE - KW_ARM]; kin_base(mb, x, Kb); kin_arm_block(mb, x, Kb, A); for (int q = 0; q < 3; ++q) ee[q] = A[36 + q]; mat_to_quat(A + 39, ee + 3); }
  if (a.ee_through_float) for (int q = 0; q < 7; ++q) ee[q] = (double)(float)ee[q];
  double baseCur[6]; for (int q = 0; q < 6; ++q) baseCur[q] = x[6 + q];
  double baseTarget[6], eeTarget[7], eeFirst[7], reach = t0 + T, vlin[3] = {0.0, 0.0, 0.0};
  if (kind == QM_CMD_VEL) {
    double R[9]; rot_zyx(baseCur[3], baseCur[4], baseCur[5], R); m3_mulv(R, cmd, vlin);            // world-frame velocity
    baseTarget[0] = baseCur[0] + vlin[0] * T; baseTarget[1] = baseCur[1] + vlin[1] * T; baseTarget[2] = a.com_height;
    baseTarget[3] = baseCur[3] + cmd[3] * T; baseTarget[4] = 0.0; baseTarget[5] = 0.0;
    const double d0 = last[0] - ee[0], d1 = last[1] - ee[1], d2 = last[2] - ee[2];
    if (sqrt(d0 * d0 + d1 * d1 + d2 * d2) > 0.1) { last[0] = ee[0]; last[1] = ee[1]; last[2] = ee[2]; }
    for (int q = 0; q < 7; ++q) { eeTarget[q] = last[q]; eeFirst[q] = last[q]; }                   // eeStateLast.state = EeTargetPose
  } else if (kind == QM_CMD_EE_VEL) {
    const double qinit[4] = {0.5, -0.5, 0.5, -0.5};                                                // Quaterniond(w −0.5, 0.5, −0.5, 0.5) as xyzw
    double Rq[9], Ri[9], Rt[9], M[9]; quat_to_R(ee + 3, Rq); quat_to_R(qinit, Ri);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Rt[3 * i + j] = Ri[3 * j + i];
    m3_mul(Rq, Rt, M); m3_mulv(M, cmd, vlin);
    for (int q = 0; q < 7; ++q) { eeFirst[q] = ee[q]; eeTarget[q] = last[q]; }
    eeTarget[0] = ee[0] + vlin[0] * T; eeTarget[1] = ee[1] + vlin[1] * T;                          // z and orientation stay at the last target
    for (int q = 0; q < 6; ++q) baseTarget[q] = baseCur[q];
    baseTarget[0] = eeTarget[0] - 0.52; baseTarget[1] = eeTarget[1] - 0.09; baseTarget[2] = a.com_height; baseTarget[4] = 0.0; baseTarget[5] = 0.0;
    vlin[0] = vlin[1] = vlin[2] = 0.0;                                                             // only cmd_vel writes the momentum reference
  } else {
    for (int q = 0; q < 7; ++q) { eeFirst[q] = ee[q]; eeTarget[q] = cmd[q]; }
    for (int q = 0; q < 6; ++q) baseTarget[q] = baseCur[q];
    baseTarget[0] = cmd[0] - 0.52; baseTarget[1] = cmd[1] - 0.09; baseTarget[2] = a.com_height; baseTarget[4] = 0.0; baseTarget[5] = 0.0;
    // estimateTimeToTarget of [position error, quaternionDistance(q_current, q_target)]
    const double dp[3] = {cmd[0] - ee[0], cmd[1] - ee[1], cmd[2] - ee[2]};
    const double* qc = ee + 3; const double* qt = cmd + 3; double cx[3]; v3_cross(qc, qt, cx);
    const double dr[3] = {qc[3] * qt[0] - qt[3] * qc[0] + cx[0], qc[3] * qt[1] - qt[3] * qc[1] + cx[1], qc[3] * qt[2] - qt[3] * qc[2] + cx[2]};
    const double tdis = sqrt(dp[0] * dp[0] + dp[1] * dp[1] + dp[2] * dp[2]) / a.disp_velocity, trot = sqrt(dr[0] * dr[0] + dr[1] * dr[1] + dr[2] * dr[2]) / a.rot_velocity;
    reach = t0 + (trot > tdis ? trot : tdis);
    for (int q = 0; q < 7; ++q) last[q] = cmd[q];                                                  // processFeedback: lastEeTarget_ << position, orientation
  }
  // targetPoseToTargetTrajectories (_node.cpp:44-68)
  baseCur[2] = a.com_height; baseCur[4] = 0.0; baseCur[5] = 0.0;
  double* rt = a.ref_t + (size_t)b * a.nref; double* rx = a.ref_x + (size_t)b * a.nref * QM_NREF;
  rt[0] = t0; rt[1] = reach;
  for (int k = 0; k < 2; ++k) {
    double* r = rx + (size_t)k * QM_NREF;
    for (int q = 0; q < 3; ++q) { r[q] = vlin[q]; r[3 + q] = 0.0; }
    for (int q = 0; q < 6; ++q) r[6 + q] = k ? baseTarget[q] : baseCur[q];
    for (int j = 0; j < QM_NJ; ++j) r[12 + j] = mb[MB_QNOM + j];
    for (int q = 0; q < 7; ++q) r[30 + q] = k ? eeTarget[q] : eeFirst[q];
  }
  for (int k = 2; k < a.nref; ++k) {                                                               // spare knots of the buffer: hold the target
    rt[k] = reach + 1.0e3 * (double)(k - 1);
    for (int q = 0; q < QM_NREF; ++q) rx[(size_t)k * QM_NREF + q] = rx[QM_NREF + q];
  }
}
