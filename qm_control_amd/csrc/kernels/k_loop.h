// k_loop.h — glue kernels of the device-resident control loop around the plant (SURVEY.md §8(f) rank 1 + 3), one thread per instance:
//   qm_observe_kernel  QMController::updateStateEstimation (qm_controllers/src/QMController.cpp:202-244) with the "ground truth" estimator: the plant's rbd
//                      state IS the measured state; currentObservation_.state = computeCentroidalStateFromRbdModel(rbd) [upstream ocs2_centroidal_model
//                      CentroidalModelRbdConversions, SRBD branch: normalized momentum = A_b(q) v_base / m], currentObservation_.time = plant time
//   qm_command_kernel  QMController::updateControlLaw (QMController.cpp:177-190): legs setCommand(posDes, velDes, 0, 3, tau) once time > 10, arm
//                      setCommand(posDes, 0, arm_kp_wbc, arm_kd_wbc, tau); posDes / velDes = joint part of the evaluated policy (QMController.cpp:156-157);
//                      or QMMpcController::updateControlLaw (QMController.cpp:431-445), see below
#pragma once
#include "qm_dev_kin.h"
#include "k_sim.h"

struct QmObserveArgs { const double* mb; int B; const double* rbd; const double* time; double* x0; double* t0; };
__global__ void qm_observe_kernel(QmObserveArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.B) return;
  const double* r = a.rbd + (size_t)b * QM_NRBD; double* x = a.x0 + (size_t)b * 30;
  double R[9], T[9], Rt[9], Iw[9]; rot_zyx(r[0], r[1], r[2], R);
  m3_mul(R, a.mb + MB_INOM, T); for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Rt[3 * i + j] = R[3 * j + i]; m3_mul(T, Rt, Iw);
  const double w[3] = {r[24], r[25], r[26]}; double rw[3], c[3], Lw[3];
  m3_mulv(R, a.mb + MB_RNOM, rw); v3_cross(rw, w, c); m3_mulv(Iw, w, Lw);
  const double im = 1.0 / a.mb[MB_ROBOTMASS];
  for (int i = 0; i < 3; ++i) { x[i] = r[27 + i] + c[i]; x[3 + i] = Lw[i] * im; x[6 + i] = r[3 + i]; x[9 + i] = r[i]; }
  for (int j = 0; j < QM_NJ; ++j) x[12 + j] = r[6 + j];
  a.t0[b] = a.time[b];
}

// controller 0: QMController::updateControlLaw (QMController.cpp:177-190).
// controller 1: QMMpcController::updateControlLaw (QMController.cpp:431-445) — the variant for the real arm: legs through the hybrid joint handles on EVERY tick
//   (no time > 10 gate), the arm as POSITION commands to its own joint position controllers, published at arm_control_loop_hz = 100:
//   when time − last_time_ > 1/100:  q_cmd_j = currentObservation_.state(24 + j) + velDes(12 + j) * 1.0 / 100.0,  last_time_ = time.
//   In the plant the position controller is the hybrid joint law with (posDes = the held q_cmd_j, velDes = 0, kp = arm_kp, kd = arm_kd, ff = 0): the WBC's arm
//   torques are NOT applied.  arm_hold / arm_last are per-instance, per-joint state (reset: hold the current arm pose, last_time_ = time, QMController.cpp:127).
struct QmCommandArgs { int B; const double* x_des; const double* u_des; const double* wbc_out; const double* time; double arm_kp, arm_kd; double* cmd;
                       int controller; const double* rbd; double* arm_hold; double* arm_last; };
__global__ void qm_command_kernel(QmCommandArgs a) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = g / QM_NJ, j = g - b * QM_NJ;
  if (b >= a.B) return;
  double* c = a.cmd + (size_t)b * (QM_SIM_CMD - 1);
  const double pos = a.x_des[(size_t)b * 30 + 12 + j], vel = a.u_des[(size_t)b * 30 + 12 + j], tau = a.wbc_out[(size_t)b * QM_NWBC_OUT + 36 + j];
  if (a.controller == 1) {
    if (j < 12) { c[j] = pos; c[18 + j] = vel; c[36 + j] = 0.0; c[54 + j] = 3.0; c[72 + j] = tau; }
    else {
      const int k = (int)((size_t)b * 6 + (j - 12)); const double t = a.time[b];
      double hold = a.arm_hold[k];
      if (t - a.arm_last[k] > 1.0 / 100.0) { hold = a.rbd[(size_t)b * QM_NRBD + 6 + j] + vel * 1.0 / 100.0; a.arm_hold[k] = hold; a.arm_last[k] = t; }
      c[j] = hold; c[18 + j] = 0.0; c[36 + j] = a.arm_kp; c[54 + j] = a.arm_kd; c[72 + j] = 0.0;
    }
    return;
  }
  if (j < 12) { if (a.time[b] > 10.0) { c[j] = pos; c[18 + j] = vel; c[36 + j] = 0.0; c[54 + j] = 3.0; c[72 + j] = tau; } }
  else { c[j] = pos; c[18 + j] = 0.0; c[36 + j] = a.arm_kp; c[54 + j] = a.arm_kd; c[72 + j] = tau; }
}
// reset of the QMMpcController arm state: hold the current arm pose, last_time_ = the plant time
struct QmArmResetArgs { int B; const double* q; const double* time; double* arm_hold; double* arm_last; };
__global__ void qm_arm_reset_kernel(QmArmResetArgs a) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= a.B * 6) return;
  const int b = g / 6, j = g - 6 * b;
  a.arm_hold[g] = a.q[(size_t)b * 24 + 18 + j]; a.arm_last[g] = a.time[b];
}
