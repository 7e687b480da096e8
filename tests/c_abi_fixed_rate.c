/* tests/c_abi_fixed_rate.c — a fixed-rate MPC loop through the C ABI from plain C (compiled by tests/test_gpu_threads.py): what mpcThread_ does for minutes on end
 * (qm_controllers/src/QMController.cpp:315-330: advanceMpc on the current observation, `mpc.mpcDesiredFrequency 100`, task.info:146) — observation, the reference manager's
 * sliding mode-schedule window (getModeSchedule(t − T, t + 2T), trot of gait.info:30-43), warm-started SQP iteration, primal solution to the host — for `seconds` of
 * CONTROLLER time, back to back (no sleeping: 60 s are 6000 solves).  The observation follows the plan (state of the last policy at the new time).
 * Three observation-time rasters, one after the other:
 *   raster 0: t0 = 0.1013 + 0.01 k                 — the review's raster (10 ms, offset 1.3 ms): never near an event
 *   raster 1: t0 = 0.005 k'                         — shares a raster with the gait events (0.35 n) and the grid step (0.015): grid nodes land ON events up to rounding,
 *                                                     i.e. every few solves a node falls a few ulp BEFORE an event — inside (event − weakEpsilon, event)
 *   raster 2: as raster 0, but once per gait event the observation is delayed (by less than one period) so that grid node 3 lands 5e-7 s before the event
 * A solve FAILS when a call returns an error or the instance's status is negative; status > 0 (QM_MPC_WARN_PIVOT: the negative-duration stage in front of an event was
 * solved with zeroed pivots) is counted as a warning.  Exit code 0 iff no solve failed and raster 2 did meet the degenerate window (one warning per gait event).
 * usage: c_abi_fixed_rate robot.urdf task.info reference.info [controller seconds per raster, default 60] */
#define _POSIX_C_SOURCE 200809L
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "qmhip.h"

enum { MAXN = 160, NREF = 2, NEV = 24 };
static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

/* getModeSchedule(t − T, t + 2T) of a trot that started at time 0 after an initial stance: events e_n = n * 0.35 accumulated the way GaitSchedule tiles them
 * (repeated addition), modes alternate LF_RH / RF_LH; unused slots are far-future events in STANCE (the layout K0 expects) */
static void schedule_window(double t, double T, double* ev, int32_t* modes) {
  double e = 0.0; int n = 0;
  while (e + 0.35 <= t - T) { e += 0.35; ++n; }                 /* e = last event at or before t − T (event 0 at time 0) */
  int cnt = 0;
  modes[0] = QM_MODE_STANCE;                                     /* [upstream GaitSchedule::getModeSchedule] forces the first mode of the window to STANCE: every swing is enclosed */
  for (; cnt < NEV - 2 && e < t + 2.0 * T + 0.35; ++cnt, ++n) { ev[cnt] = e; modes[cnt + 1] = (n & 1) ? QM_MODE_RF_LH : QM_MODE_LF_RH; e += 0.35; }
  ev[cnt] = e; modes[cnt + 1] = QM_MODE_STANCE; ++cnt;          /* the trailing stance phase every swing must be enclosed by */
  for (; cnt < NEV; ++cnt) { ev[cnt] = ev[cnt - 1] + 1.0e3; modes[cnt + 1] = QM_MODE_STANCE; }
}

int main(int argc, char** argv) {
  if (argc < 4) { fprintf(stderr, "usage: %s robot.urdf task.info reference.info [controller seconds]\n", argv[0]); return 2; }
  const double seconds = argc > 4 ? atof(argv[4]) : 60.0; const int n_solves = (int)(seconds * 100.0);
  qmhip_ctx* ctx = NULL;
  if (qmhip_create(argv[1], argv[2], argv[3], 0, 1, MAXN, NREF, NEV, &ctx) != QMHIP_OK) { fprintf(stderr, "qmhip_create: %s\n", qmhip_last_error(NULL)); return 1; }
  static double mb[MB_SIZE], st[ST_SIZE]; qmhip_export_blobs(ctx, mb, st);
  if (st[ST_GRID_DT_MIN] != QM_GRID_DT_MIN_UPSTREAM || st[ST_RICCATI_STRICT] != 0.0) { fprintf(stderr, "the shipped defaults changed\n"); return 1; }
  const double dt = st[ST_SQP_DT], horizon = 100 * dt;
  static double out_t[MAXN], out_x[MAXN][QM_NX], out_u[MAXN][QM_NU], perf[10]; static int32_t out_event[MAXN], out_mode[MAXN];
  int fail = 0;
  for (int raster = 0; raster < 3; ++raster) {
    double x0[QM_NX]; memcpy(x0, st + ST_XINIT, sizeof(x0));
    double ref_x[NREF][QM_NREF]; const double ee[7] = {0.52, 0.09, 0.38 + 0.4, 0.5, -0.5, 0.5, -0.5};
    for (int k = 0; k < NREF; ++k) {
      memset(ref_x[k], 0, sizeof(ref_x[k]));
      for (int i = 0; i < 6; ++i) ref_x[k][6 + i] = x0[6 + i];
      ref_x[k][8] = 0.4; ref_x[k][10] = ref_x[k][11] = 0.0; ref_x[k][6] += 0.3;      /* base target 0.3 m ahead of the initial pose, held */
      for (int q = 0; q < QM_NJ; ++q) ref_x[k][12 + q] = mb[MB_QNOM + q];
      memcpy(ref_x[k] + 30, ee, sizeof(ee));
    }
    int failed = 0, warned = 0, errors = 0, max_nodes = 0; double ms_sum = 0, ms_max = 0, worst_gap = 1.0; const double wall0 = now_s();
    for (int k = 0; k < n_solves; ++k) {
      double t0 = (raster == 1) ? 0.105 + 0.005 * k : 0.1013 + 0.01 * k;
      double ev[NEV]; int32_t modes[NEV + 1];
      if (raster == 2) {                          /* once per gait event: the observation is delayed (by less than one period) so that grid node 3 lands 5e-7 s before the event */
        double e = 0.0; while (e - 5e-7 - 3.0 * dt < t0) e += 0.35;
        if (e - 5e-7 - 3.0 * dt < t0 + 0.01) t0 = e - 5e-7 - 3.0 * dt;
      }
      schedule_window(t0, horizon, ev, modes);
      double ref_t[NREF] = {t0, t0 + horizon};
      const double a = now_s(); int32_t nn = 0, status = -99; int rc;
      if (k == 0) {
        rc = qmhip_mpc_upload(ctx, 1, &t0, x0, NREF, ref_t, &ref_x[0][0], NEV, ev, modes);
        if (rc == QMHIP_OK) rc = qmhip_mpc_solve_resident(ctx, 1, horizon);
      } else {
        int32_t md; double xd[QM_NX], ud[QM_NU];
        rc = qmhip_policy_eval(ctx, 1, &t0, xd, ud, &md);                                   /* the plant tracks the plan: observation = planned state at the new time */
        if (rc == QMHIP_OK) { memcpy(x0, xd, sizeof(x0)); rc = qmhip_mpc_update_references(ctx, 1, NREF, ref_t, &ref_x[0][0], NEV, ev, modes); }
        if (rc == QMHIP_OK) rc = qmhip_mpc_set_initial(ctx, 1, &t0, x0);
        if (rc == QMHIP_OK) rc = qmhip_mpc_solve_resident_warm(ctx, 1, horizon);
      }
      if (rc == QMHIP_OK) rc = qmhip_mpc_download(ctx, 1, &nn, out_t, out_event, out_mode, &out_x[0][0], &out_u[0][0], perf, &status);
      const double ms = 1e3 * (now_s() - a); ms_sum += ms; if (ms > ms_max) ms_max = ms;
      if (rc != QMHIP_OK) { if (!errors) fprintf(stderr, "raster %d solve %d: call failed (%d): %s\n", raster, k, rc, qmhip_last_error(ctx)); ++errors; continue; }
      if (status < 0) { if (!failed) fprintf(stderr, "raster %d solve %d (t0 = %.9f): status %d\n", raster, k, t0, status); ++failed; }
      else if (status > 0) ++warned;
      if (nn > max_nodes) max_nodes = nn;
      for (int i = 0; i + 1 < nn; ++i) if (out_event[i + 1] == QM_EV_PRE && out_event[i] != QM_EV_POST) { const double g = out_t[i + 1] - out_t[i]; if (g < worst_gap) worst_gap = g; }
      for (int i = 0; i < nn; ++i) for (int q = 0; q < QM_NX; ++q) if (!isfinite(out_x[i][q]) || !isfinite(out_u[i][q])) { if (!failed) fprintf(stderr, "raster %d solve %d: non-finite solution\n", raster, k); ++failed; i = nn; break; }
    }
    printf("raster_%d: solves %d failed %d warnings %d call_errors %d smallest_gap_before_an_event %.3e max_nodes %d solve_ms_mean %.4f solve_ms_max %.4f controller_seconds %.1f wall_seconds %.1f base_x_travel %.4f\n",
           raster, n_solves, failed, warned, errors, worst_gap, max_nodes, ms_sum / n_solves, ms_max, seconds, now_s() - wall0, x0[6] - st[ST_XINIT + 6]);
    if (failed || errors || (raster == 2 && warned < (int)(seconds / 0.35) - 2)) fail = 1;
  }
  qmhip_destroy(ctx);
  printf("result: %s\n", fail ? "FAIL" : "ok");
  return fail ? 3 : 0;
}
