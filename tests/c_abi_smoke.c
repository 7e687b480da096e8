/* tests/c_abi_smoke.c — a plain-C client of libqmhip.so (compiled with gcc by tests/test_gpu_files.py, no Python / C++ in the loop):
 *   qmhip_create(urdf, task.info, reference.info)  ->  qmhip_mpc_step  ->  qmhip_policy_eval  ->  qmhip_wbc_step
 * on configuration C1 of BASELINE.md (stance gait, horizon N = 20, nominal initial state, t0 = 0).  What a cgo / JNI / ctypes binding or the
 * C++ adaptors under adaptors/ do, reduced to the calls.  Prints the results as "key: v0 v1 ..." lines; exit code 0 on success.
 * usage: c_abi_smoke robot.urdf task.info reference.info */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "qmhip.h"

static void print_vec(const char* key, const double* v, int n) { printf("%s:", key); for (int i = 0; i < n; ++i) printf(" %.17g", v[i]); printf("\n"); }

int main(int argc, char** argv) {
  if (argc != 4) { fprintf(stderr, "usage: %s robot.urdf task.info reference.info\n", argv[0]); return 2; }
  enum { MAXN = 32, NREF = 2, NEV = 1 };
  qmhip_ctx* ctx = NULL;
  int rc = qmhip_create(argv[1], argv[2], argv[3], 0, 1, MAXN, NREF, NEV, &ctx);
  if (rc != QMHIP_OK) { fprintf(stderr, "qmhip_create failed (%d): %s\n", rc, qmhip_last_error(NULL)); return 1; }
  static double mb[MB_SIZE], st[ST_SIZE];
  qmhip_export_blobs(ctx, mb, st);
  const double dt = st[ST_SQP_DT], horizon = 20 * dt, t0 = 0.0;
  /* initial state = task.info's initialState; target: hold the nominal pose (2 knots: QmTargetTrajectoriesPublisher_node.cpp:44-68) */
  double x0[QM_NX]; memcpy(x0, st + ST_XINIT, sizeof(x0));
  double ref_t[NREF] = {t0, t0 + horizon}, ref_x[NREF][QM_NREF];
  const double ee[7] = {0.52, 0.09, 0.38 + 0.4, 0.5, -0.5, 0.5, -0.5};        /* QMController.cpp:107-108 */
  for (int k = 0; k < NREF; ++k) {
    memset(ref_x[k], 0, sizeof(ref_x[k]));
    for (int i = 0; i < 6; ++i) ref_x[k][6 + i] = x0[6 + i];
    if (k == 0) { ref_x[k][8] = 0.4; ref_x[k][10] = 0.0; ref_x[k][11] = 0.0; }  /* current pose at comHeight, level; the goal knot is the nominal pose itself (scenarios.make_config("C1")) */
    for (int j = 0; j < QM_NJ; ++j) ref_x[k][12 + j] = mb[MB_QNOM + j];
    memcpy(ref_x[k] + 30, ee, sizeof(ee));
  }
  /* stance throughout: one event long before the horizon, modes {STANCE, STANCE} (reference.info:28-39) */
  double ev[NEV] = {t0 - 2.0 * horizon - 1.0}; int32_t modes[NEV + 1] = {QM_MODE_STANCE, QM_MODE_STANCE};
  int32_t num_nodes = 0, status = -99, out_event[MAXN], out_mode[MAXN];
  static double out_t[MAXN], out_x[MAXN][QM_NX], out_u[MAXN][QM_NU], perf[10];
  rc = qmhip_mpc_step(ctx, 1, &t0, x0, NREF, ref_t, &ref_x[0][0], NEV, ev, modes, horizon, &num_nodes, out_t, out_event, out_mode, &out_x[0][0], &out_u[0][0], perf, &status);
  if (rc != QMHIP_OK) { fprintf(stderr, "qmhip_mpc_step failed (%d): %s\n", rc, qmhip_last_error(ctx)); return 1; }
  double x_des[QM_NX], u_des[QM_NU]; int32_t mode = -1;
  rc = qmhip_policy_eval(ctx, 1, &t0, x_des, u_des, &mode);
  if (rc != QMHIP_OK) { fprintf(stderr, "qmhip_policy_eval failed (%d): %s\n", rc, qmhip_last_error(ctx)); return 1; }
  /* measured rbd state (StateEstimateBase.cpp:41-103 layout) from x0: zyx, position, joints; zero velocities.  The EE pose slots [48, 55) are
   * not read by WbcBase::update (it runs its own kinematics on q). */
  double rbd[QM_NRBD]; memset(rbd, 0, sizeof(rbd));
  for (int i = 0; i < 3; ++i) { rbd[i] = x0[9 + i]; rbd[3 + i] = x0[6 + i]; }
  for (int j = 0; j < QM_NJ; ++j) rbd[6 + j] = x0[12 + j];
  double wbc_out[QM_NWBC_OUT], time = 20.0; int32_t qps[3] = {-9, -9, -9};
  qmhip_wbc_reset(ctx);
  rc = qmhip_wbc_step(ctx, 1, x_des, u_des, rbd, &mode, 0.002, &time, 0, wbc_out, qps);
  if (rc != QMHIP_OK) { fprintf(stderr, "qmhip_wbc_step failed (%d): %s\n", rc, qmhip_last_error(ctx)); return 1; }
  printf("status: %d\nnum_nodes: %d\nmode: %d\nqp_status: %d %d %d\n", status, num_nodes, mode, qps[0], qps[1], qps[2]);
  print_vec("x_des", x_des, QM_NX); print_vec("u_des", u_des, QM_NU); print_vec("wbc_out", wbc_out, QM_NWBC_OUT); print_vec("perf", perf, 10);
  double fz = 0.0; for (int c = 0; c < 4; ++c) fz += wbc_out[24 + 3 * c + 2];
  printf("sum_fz_over_weight: %.12f\n", fz / (mb[MB_ROBOTMASS] * 9.81));
  qmhip_destroy(ctx);
  return (status == 0 && qps[0] == 0 && qps[1] == 0 && qps[2] == 0 && isfinite(fz)) ? 0 : 3;
}
