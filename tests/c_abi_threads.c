/* tests/c_abi_threads.c — the reference's two-thread layout driven through the C ABI from plain C + pthreads (compiled by tests/test_gpu_threads.py):
 *   thread A = mpcThread_   (qm_controllers/src/QMController.cpp:315-333): observation -> warm MPC solve -> primal solution download, 100 Hz;
 *   thread B = ros_control  (QMController.cpp:128-147):                    WbcBase::update = qmhip_wbc_step, 500 Hz (controller period 0.002 s).
 * Single robot (B = 1), trot, horizon N = 100.  The tick sequence is first run ALONE (reference outputs + undisturbed latency), then twice
 * beside the MPC thread: (1) the control thread on its own context (qmhip_create_wbc_context — the layout adaptors/QmhipController.h installs), (2) both threads
 * on ONE context (the entry points serialise on the context).  Every WBC output must equal the single-threaded one bit for bit; the tick latency is printed.
 * usage: c_abi_threads robot.urdf task.info reference.info [seconds per phase, default 2.5] */
#define _POSIX_C_SOURCE 200809L
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "qmhip.h"

enum { MAXN = 160, NREF = 2, NEV = 24, NIN = 64 };
static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
static void sleep_until(double t) { struct timespec ts; ts.tv_sec = (time_t)t; ts.tv_nsec = (long)((t - (double)ts.tv_sec) * 1e9); clock_nanosleep(CLOCK_MONOTONIC, TIMER_ABSTIME, &ts, NULL); }

static double in_x[NIN][QM_NX], in_u[NIN][QM_NU], in_rbd[NIN][QM_NRBD]; static int32_t in_mode[NIN];
static double x0[QM_NX], horizon, t_first;
static volatile int stop_mpc = 0;

typedef struct { qmhip_ctx* ctx; int n_ticks; double* out; int32_t* qps; double* lat; double lat_max, lat_sum; int err, late, max_at; } tick_job;
typedef struct { qmhip_ctx* ctx; int solves, bad_status, err; double ms_sum, ms_max; } mpc_job;

/* the control thread: n_ticks WBC updates on a 2 ms raster */
static void* tick_thread(void* p) {
  tick_job* j = (tick_job*)p; j->lat_max = j->lat_sum = 0; j->err = j->late = 0;
  const double t_begin = now_s() + 0.01;
  for (int i = 0; i < j->n_ticks; ++i) {
    const double due = t_begin + 0.002 * i; if (now_s() > due + 0.002) j->late++; sleep_until(due);
    const int k = i % NIN; const double time = 20.0 + 0.002 * i, a = now_s();
    const int rc = qmhip_wbc_step(j->ctx, 1, in_x[k], in_u[k], in_rbd[k], &in_mode[k], 0.002, &time, 0, j->out + (size_t)i * QM_NWBC_OUT, j->qps + 3 * (size_t)i);
    const double ms = 1e3 * (now_s() - a);
    if (rc != QMHIP_OK) { if (!j->err) fprintf(stderr, "qmhip_wbc_step failed (%d): %s\n", rc, qmhip_last_error(j->ctx)); j->err++; }
    j->lat[i] = ms; j->lat_sum += ms; if (ms > j->lat_max) { j->lat_max = ms; j->max_at = i; }
  }
  return NULL;
}
/* the MPC thread: observation, warm-started SQP iteration, primal solution to the host — what QmhipSolver::runImpl does per MPC_BASE::run — at 100 Hz */
static void* mpc_thread(void* p) {
  mpc_job* j = (mpc_job*)p; j->solves = j->bad_status = j->err = 0; j->ms_sum = j->ms_max = 0;
  static double out_t[MAXN], out_x[MAXN][QM_NX], out_u[MAXN][QM_NU], perf[10]; static int32_t out_event[MAXN], out_mode[MAXN];
  const double t_begin = now_s();
  for (int k = 0; !stop_mpc; ++k) {
    sleep_until(t_begin + 0.01 * k);
    const double t0 = t_first + 0.0013 + 0.01 * k, a = now_s(); int32_t nn = 0, status = -99;      /* never exactly on a gait event; stays inside the uploaded schedule */
    int rc = qmhip_mpc_set_initial(j->ctx, 1, &t0, x0);
    if (rc == QMHIP_OK) rc = k ? qmhip_mpc_solve_resident_warm(j->ctx, 1, horizon) : qmhip_mpc_solve_resident(j->ctx, 1, horizon);      /* each phase starts cold */
    if (rc == QMHIP_OK) rc = qmhip_mpc_download(j->ctx, 1, &nn, out_t, out_event, out_mode, &out_x[0][0], &out_u[0][0], perf, &status);
    const double ms = 1e3 * (now_s() - a);
    if (rc != QMHIP_OK) { if (!j->err) fprintf(stderr, "MPC thread: call failed (%d): %s\n", rc, qmhip_last_error(j->ctx)); j->err++; }
    else if (status != 0) j->bad_status++;
    j->solves++; j->ms_sum += ms; if (ms > j->ms_max) j->ms_max = ms;
  }
  return NULL;
}

static int cmp_double(const void* a, const void* b) { const double x = *(const double*)a, y = *(const double*)b; return (x > y) - (x < y); }

static int run_phase(const char* name, qmhip_ctx* mpc_ctx, qmhip_ctx* tick_ctx, int n_ticks, const double* ref_out, double* out, int32_t* qps) {
  pthread_t ta, tb; tick_job tj; mpc_job mj; memset(&tj, 0, sizeof(tj)); memset(&mj, 0, sizeof(mj));
  tj.ctx = tick_ctx; tj.n_ticks = n_ticks; tj.out = out; tj.qps = qps; mj.ctx = mpc_ctx; tj.lat = (double*)calloc((size_t)n_ticks, sizeof(double));
  if (qmhip_wbc_reset(tick_ctx) != QMHIP_OK) { fprintf(stderr, "qmhip_wbc_reset: %s\n", qmhip_last_error(tick_ctx)); return 1; }
  stop_mpc = 0;
  if (mpc_ctx) pthread_create(&ta, NULL, mpc_thread, &mj);
  pthread_create(&tb, NULL, tick_thread, &tj);
  pthread_join(tb, NULL); stop_mpc = 1; if (mpc_ctx) pthread_join(ta, NULL);
  int mismatches = 0, bad_qp = 0;
  if (ref_out) for (int i = 0; i < n_ticks; ++i) if (memcmp(ref_out + (size_t)i * QM_NWBC_OUT, out + (size_t)i * QM_NWBC_OUT, QM_NWBC_OUT * sizeof(double))) mismatches++;
  for (int i = 0; i < 3 * n_ticks; ++i) if (qps[i] != 0) bad_qp++;
  int over_half = 0, over_one = 0; for (int i = 0; i < n_ticks; ++i) { over_half += tj.lat[i] > 0.5; over_one += tj.lat[i] > 1.0; }
  qsort(tj.lat, (size_t)n_ticks, sizeof(double), cmp_double); const double p99 = tj.lat[(int)(0.99 * (n_ticks - 1))], p999 = tj.lat[(int)(0.999 * (n_ticks - 1))]; free(tj.lat);
  printf("%s: ticks %d mismatches %d bad_qp %d late_ticks %d wbc_ms_mean %.4f wbc_ms_p99 %.4f wbc_ms_p999 %.4f wbc_ms_max %.4f max_at_tick %d ticks_over_0.5ms %d ticks_over_1ms %d tick_errors %d mpc_solves %d mpc_ms_mean %.4f mpc_ms_max %.4f mpc_bad_status %d mpc_errors %d\n",
         name, n_ticks, mismatches, bad_qp, tj.late, tj.lat_sum / n_ticks, p99, p999, tj.lat_max, tj.max_at, over_half, over_one, tj.err, mj.solves, mj.solves ? mj.ms_sum / mj.solves : 0.0, mj.ms_max, mj.bad_status, mj.err);
  return (mismatches || tj.err || mj.err || mj.bad_status || (mpc_ctx && mj.solves < 10)) ? 1 : 0;
}

int main(int argc, char** argv) {
  if (argc < 4) { fprintf(stderr, "usage: %s robot.urdf task.info reference.info [seconds]\n", argv[0]); return 2; }
  const double seconds = argc > 4 ? atof(argv[4]) : 2.5; const int n_ticks = (int)(seconds * 500.0);
  qmhip_ctx* ctx = NULL; qmhip_ctx* wctx = NULL;
  if (qmhip_create(argv[1], argv[2], argv[3], 0, 1, MAXN, NREF, NEV, &ctx) != QMHIP_OK) { fprintf(stderr, "qmhip_create: %s\n", qmhip_last_error(NULL)); return 1; }
  if (qmhip_create_wbc_context(ctx, 1, &wctx) != QMHIP_OK) { fprintf(stderr, "qmhip_create_wbc_context: %s\n", qmhip_last_error(NULL)); return 1; }
  static double mb[MB_SIZE], st[ST_SIZE]; qmhip_export_blobs(ctx, mb, st);
  /* C2 of BASELINE.md: trot (gait.info: 0.35 s phases LF_RH / RF_LH), N = 100, nominal state, t0 = 0.1; target 0.3 m ahead */
  horizon = 100 * st[ST_SQP_DT]; t_first = 0.1; memcpy(x0, st + ST_XINIT, sizeof(x0));
  double ev[NEV]; int32_t modes[NEV + 1]; ev[0] = 0.0; modes[0] = QM_MODE_STANCE;
  for (int k = 1; k < NEV; ++k) { ev[k] = ev[k - 1] + 0.35; modes[k] = (k & 1) ? QM_MODE_LF_RH : QM_MODE_RF_LH; }
  modes[NEV] = QM_MODE_STANCE;       /* last event 8.05 s: beyond the last observation time + 2 horizons */
  double ref_t[NREF] = {t_first, t_first + horizon}, ref_x[NREF][QM_NREF]; const double ee[7] = {0.52, 0.09, 0.38 + 0.4, 0.5, -0.5, 0.5, -0.5};
  for (int k = 0; k < NREF; ++k) {
    memset(ref_x[k], 0, sizeof(ref_x[k]));
    for (int i = 0; i < 6; ++i) ref_x[k][6 + i] = x0[6 + i];
    ref_x[k][8] = 0.4; ref_x[k][10] = ref_x[k][11] = 0.0;
    if (k == 1) ref_x[k][6] += 0.3;
    for (int q = 0; q < QM_NJ; ++q) ref_x[k][12 + q] = mb[MB_QNOM + q];
    memcpy(ref_x[k] + 30, ee, sizeof(ee));
  }
  if (qmhip_mpc_upload(ctx, 1, &t_first, x0, NREF, ref_t, &ref_x[0][0], NEV, ev, modes) != QMHIP_OK || qmhip_mpc_solve_resident(ctx, 1, horizon) != QMHIP_OK) {
    fprintf(stderr, "cold solve: %s\n", qmhip_last_error(ctx)); return 1; }
  /* NIN distinct tick inputs: the policy along the first 0.13 s of the plan (crosses the 0.35 s gait event? no: 0.1 .. 0.226; contact mode LF_RH), measured state =
   * the planned pose with a deterministic perturbation and non-zero velocities, so that the QP's active sets differ from tick to tick */
  for (int k = 0; k < NIN; ++k) {
    const double t = t_first + 0.002 * k;
    if (qmhip_policy_eval(ctx, 1, &t, in_x[k], in_u[k], &in_mode[k]) != QMHIP_OK) { fprintf(stderr, "policy: %s\n", qmhip_last_error(ctx)); return 1; }
    memset(in_rbd[k], 0, sizeof(in_rbd[k]));
    for (int i = 0; i < 3; ++i) { in_rbd[k][i] = in_x[k][9 + i] + 0.01 * sin(1.3 * k + i); in_rbd[k][3 + i] = in_x[k][6 + i] + 0.004 * cos(0.7 * k + 2 * i); }
    for (int q = 0; q < QM_NJ; ++q) { in_rbd[k][6 + q] = in_x[k][12 + q] + 0.02 * sin(0.9 * k + 0.37 * q); in_rbd[k][QM_NQ + 6 + q] = 0.3 * cos(1.1 * k + 0.53 * q); }
    for (int i = 0; i < 6; ++i) in_rbd[k][QM_NQ + i] = 0.05 * sin(0.8 * k + i);
  }
  double* ref_out = (double*)malloc((size_t)n_ticks * QM_NWBC_OUT * sizeof(double)); double* out = (double*)malloc((size_t)n_ticks * QM_NWBC_OUT * sizeof(double));
  int32_t* qps = (int32_t*)malloc((size_t)n_ticks * 3 * sizeof(int32_t));
  int fail = 0;
  fail |= run_phase("alone_wbc_context", NULL, wctx, n_ticks, NULL, ref_out, qps);             /* reference outputs, undisturbed latency */
  fail |= run_phase("threads_two_contexts", ctx, wctx, n_ticks, ref_out, out, qps);          /* the installed layout */
  fail |= run_phase("alone_shared_context", NULL, ctx, n_ticks / 2, ref_out, out, qps);      /* same tick sequence on the MPC context (bit-equal across contexts) */
  fail |= run_phase("threads_one_context", ctx, ctx, n_ticks / 2, ref_out, out, qps);        /* serialised on the context's lock */
  double cs = 0; for (size_t i = 0; i < (size_t)n_ticks * QM_NWBC_OUT; ++i) cs += ref_out[i];
  printf("checksum: %.17g\nfirst_tick_tau:", cs); for (int q = 0; q < QM_NJ; ++q) printf(" %.17g", ref_out[36 + q]); printf("\n");
  qmhip_destroy(wctx); qmhip_destroy(ctx); free(ref_out); free(out); free(qps);
  printf("result: %s\n", fail ? "FAIL" : "ok");
  return fail ? 3 : 0;
}
