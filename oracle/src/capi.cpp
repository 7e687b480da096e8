// oracle/src/capi.cpp — TEST INFRASTRUCTURE (CPU oracle): C entry points for ctypes (tests/, bench.py
// cpu_baseline leg, __graft_entry__.smoke()).  Nothing under qm_control_amd/ may link or load this.
#include "sqp.h"
#include "ilqr.h"
#include "ipm.h"
#include "wbc.h"
#include "sim.h"
#include <cstdio>
#include <thread>
#include <atomic>

struct Oracle {
  Model M; Problem P; SqpResult R; WbcState W; WbcDebug dbg; SimState sim; SimParams simp;
  Oracle() { P.M = &M; }
};

extern "C" {

void* qmo_create(const double* mb, const double* st) {
  Oracle* o = new Oracle();
  std::memcpy(o->M.mb, mb, sizeof(double) * MB_SIZE); std::memcpy(o->M.st, st, sizeof(double) * ST_SIZE);
  return o;
}
void qmo_destroy(void* h) { delete (Oracle*)h; }
void qmo_set_setting(void* h, int idx, double v) { ((Oracle*)h)->M.st[idx] = v; }
// worker threads over shooting nodes of the calling thread's later mpc_step calls (sqp.nThreads, task.info:77)
void qmo_set_threads(int n) { n = n < 1 ? 1 : n; if (n != oracleThreads()) { oracleThreads() = n; nodePool().resize(n - 1); } }

void qmo_flow_map(void* h, const double* x, const double* u, double* f, double* A, double* B) {
  Oracle* o = (Oracle*)h; Vec xv(x, x + QM_NX), uv(u, u + QM_NU), fv; Mat Am, Bm;
  if (A || B) { flowMapLinear(o->M, xv, uv, fv, Am, Bm); if (A) std::memcpy(A, Am.a.data(), 900 * 8); if (B) std::memcpy(B, Bm.a.data(), 900 * 8); }
  else flowMapValue(o->M, xv, uv, fv);
  std::memcpy(f, fv.data(), QM_NX * 8);
}
void qmo_foot_pos_vel(void* h, const double* x, const double* u, int i, double* pos, double* vel) {
  V3<double> p, v; footPosVel<double>(((Oracle*)h)->M, x, u, i, p, v); for (int k = 0; k < 3; ++k) { pos[k] = p[k]; vel[k] = v[k]; }
}
void qmo_ee_pose_error(void* h, const double* x, const double* pref, const double* qref, double* g) { eePoseError<double>(((Oracle*)h)->M, x, pref, qref, g); }
void qmo_frame_pose(void* h, const double* q, int f, double* pos, double* Rm) {
  Kin<double> k; forwardKinematics(((Oracle*)h)->M, q, k); for (int i = 0; i < 3; ++i) pos[i] = k.fp[f][i]; for (int i = 0; i < 9; ++i) Rm[i] = k.fR[f].m[i];
}
void qmo_mat_to_quat(const double* Rm, double* q) { M3<double> R; for (int i = 0; i < 9; ++i) R.m[i] = Rm[i]; matToQuat<double>(R, q); }

double qmo_get_setting(void* h, int idx) { return (idx < 0 || idx >= ST_SIZE) ? 0.0 : ((Oracle*)h)->M.st[idx]; }
int qmo_time_grid(double t0, double tf, double dt, int nev, const double* ev, int maxn, double* out_t, int* out_ev, double dt_min) {
  Vec e(ev, ev + nev); auto g = timeDiscretizationWithEvents(t0, tf, dt, e, dt_min);
  if ((int)g.size() > maxn) return -(int)g.size();
  for (size_t i = 0; i < g.size(); ++i) { out_t[i] = g[i].t; out_ev[i] = g[i].ev; }
  return (int)g.size();
}
int qmo_set_schedule(void* h, int nev, const double* ev, const int* modes) {
  Oracle* o = (Oracle*)h; o->P.ms.ev.assign(ev, ev + nev); o->P.ms.modes.assign(modes, modes + nev + 1);
  o->P.swing.update(o->M, o->P.ms); return o->P.swing.status;
}
void qmo_set_target(void* h, int K, const double* t, const double* x37) {
  Oracle* o = (Oracle*)h; o->P.target.t.assign(t, t + K); o->P.target.x.clear();
  for (int k = 0; k < K; ++k) o->P.target.x.emplace_back(x37 + QM_NREF * k, x37 + QM_NREF * (k + 1));
}
double qmo_swing_zvel(void* h, int leg, double t) { return ((Oracle*)h)->P.swing.zVel(leg, t); }
int qmo_mode_at(void* h, double t) { return ((Oracle*)h)->P.ms.modeAt(t); }
void qmo_desired_state(void* h, double t, double* x37, double* eepos, double* eequat) {
  Oracle* o = (Oracle*)h; Vec r = o->P.target.desiredState(t); std::memcpy(x37, r.data(), QM_NREF * 8); o->P.target.eePose(t, eepos, eequat);
}

// perf[10] = baseline {merit,cost,dyn,eq}, after {merit,cost,dyn,eq}, alpha, armijo
int qmo_mpc_step(void* h, double t0, double tf, const double* x0, int maxn, int* n_nodes, double* node_t, int* node_ev, int* node_mode,
                 double* x_out, double* u_out, double* perf) {
  Oracle* o = (Oracle*)h; Vec x0v(x0, x0 + QM_NX);
  o->R = SqpResult(); sqpIteration(o->P, t0, tf, x0v, nullptr, nullptr, o->R);
  if (o->R.status != 0) return o->R.status;
  const int n = (int)o->R.grid.size(); if (n > maxn) return -100;
  *n_nodes = n;
  for (int i = 0; i < n; ++i) {
    node_t[i] = o->R.grid[i].t; node_ev[i] = o->R.grid[i].ev; node_mode[i] = o->R.mode[i];
    std::memcpy(x_out + QM_NX * i, o->R.x[i].data(), QM_NX * 8); std::memcpy(u_out + QM_NU * i, o->R.u[i].data(), QM_NU * 8);
  }
  const Performance& b = o->R.baseline; const Performance& a = o->R.after;
  double p[10] = {b.merit, b.cost, b.dynSSE, b.eqSSE, a.merit, a.cost, a.dynSSE, a.eqSSE, o->R.alpha, o->R.armijo};
  std::memcpy(perf, p, sizeof(p));
  return 0;
}
static void copyPad(const Mat& A, double* out, int R, int C) { std::fill(out, out + R * C, 0.0); for (int i = 0; i < A.r; ++i) for (int j = 0; j < A.c; ++j) out[i * C + j] = A(i, j); }
static void copyPadV(const Vec& a, double* out, int n) { std::fill(out, out + n, 0.0); for (size_t i = 0; i < a.size(); ++i) out[i] = a[i]; }
// unprojected LQ data of node i: A,B(900) b(30) Q,R,P(900) q,r(30) c, dt | nc, event | C,D (16x30 padded) e(16)
int qmo_get_node_lq(void* h, int i, double* A, double* B, double* b, double* Q, double* Rr, double* P, double* q, double* r, double* scal /*c,dt,nc,event*/, double* C, double* D, double* e) {
  Oracle* o = (Oracle*)h; if (i < 0 || i >= (int)o->R.lq.size()) return -1; const NodeLQ& n = o->R.lq[i];
  scal[0] = n.c; scal[1] = n.dt; scal[2] = n.nc; scal[3] = n.event;
  copyPad(n.A, A, 30, 30); copyPadV(n.b, b, 30); copyPad(n.Q, Q, 30, 30); copyPadV(n.q, q, 30);
  if (n.event) { std::fill(B, B + 900, 0.0); std::fill(Rr, Rr + 900, 0.0); std::fill(P, P + 900, 0.0); std::fill(r, r + 30, 0.0); std::fill(C, C + 480, 0.0); std::fill(D, D + 480, 0.0); std::fill(e, e + 16, 0.0); return 0; }
  copyPad(n.B, B, 30, 30); copyPad(n.R, Rr, 30, 30); copyPad(n.P, P, 30, 30); copyPadV(n.r, r, 30);
  copyPad(n.C, C, 16, 30); copyPad(n.D, D, 16, 30); copyPadV(n.e, e, 16);
  return 0;
}
// projected data: Px(30x30) Pu(30x30 padded cols) Pe(30) Ap Bp(30x30 padded cols) bp Qp Rp(30x30 padded) Pp(30x30 padded rows m) qp rp(30 padded) scal{cp, m} K(30x30 padded rows m) kff(30)
int qmo_get_node_proj(void* h, int i, double* Px, double* Pu, double* Pe, double* Ap, double* Bp, double* bp, double* Qp, double* Rp, double* Pp, double* qp, double* rp, double* scal, double* K, double* kff) {
  Oracle* o = (Oracle*)h; if (i < 0 || i >= (int)o->R.lq.size()) return -1; const NodeLQ& n = o->R.lq[i];
  scal[0] = n.cp; scal[1] = n.m;
  copyPad(n.Px, Px, 30, 30); copyPad(n.Pu, Pu, 30, 30); copyPadV(n.Pe, Pe, 30); copyPad(n.Ap, Ap, 30, 30); copyPad(n.Bp, Bp, 30, 30); copyPadV(n.bp, bp, 30);
  copyPad(n.Qp, Qp, 30, 30); copyPad(n.Rp, Rp, 30, 30); copyPad(n.Pp, Pp, 30, 30); copyPadV(n.qp, qp, 30); copyPadV(n.rp, rp, 30); copyPad(n.K, K, 30, 30); copyPadV(n.kff, kff, 30);
  return 0;
}
int qmo_get_terminal(void* h, double* Q, double* q, double* c) { Oracle* o = (Oracle*)h; copyPad(o->R.terminal.Qp, Q, 30, 30); copyPadV(o->R.terminal.qp, q, 30); *c = o->R.terminal.cp; return 0; }
int qmo_get_step(void* h, double* dx, double* du) {
  Oracle* o = (Oracle*)h; const int n = (int)o->R.grid.size();
  for (int i = 0; i < n; ++i) std::memcpy(dx + QM_NX * i, o->R.dx[i].data(), QM_NX * 8);
  for (int i = 0; i < n - 1; ++i) std::memcpy(du + QM_NU * i, o->R.du[i].data(), QM_NU * 8);
  return n;
}
int qmo_ls_trials(void* h) { return ((Oracle*)h)->R.lsTrials; }
// diagnostics (tools/warm_ls_histogram.py): rows of 5 doubles, baseline first, then one row per trial (sqp.h: SqpResult::lsTrace); returns the number of rows copied
int qmo_ls_trace(void* h, double* out, int max_rows) { const std::vector<double>& t = ((Oracle*)h)->R.lsTrace; const int n = (int)(t.size() / 5), k = n < max_rows ? n : max_rows; for (int i = 0; i < 5 * k; ++i) out[i] = t[i]; return k; }
// tests only: 1 = every Jacobian from the full 60-slot forward mode of rounds 1-4 (the seeded evaluation must reproduce it entry by entry), 0 = seeded (default)
void qmo_set_full_seeding(int on) { qm_ad_full_seeding = on != 0; }
int qmo_last_warn(void* h) { return ((Oracle*)h)->R.warn; }      // warning bits of the last (valid) solve: QM_MPC_WARN_PIVOT
void qmo_phase_ms(void* h, double* ms3) { for (int i = 0; i < 3; ++i) ms3[i] = ((Oracle*)h)->R.phaseMs[i]; }
// one more SQP iteration on the iterate the last call left (sqp.sqpIteration > 1, [upstream SqpSolver::runImpl loop]); same outputs as qmo_mpc_step
int qmo_mpc_iterate(void* h, double t0, double tf, const double* x0, int maxn, int* n_nodes, double* node_t, int* node_ev, int* node_mode, double* xs, double* us, double* perf) {
  Oracle* o = (Oracle*)h; Vec x0v(x0, x0 + QM_NX);
  if (o->R.grid.size() < 2) return -3;
  std::vector<Vec> xi = o->R.x, ui(o->R.u.begin(), o->R.u.end() - 1);
  for (size_t i = 0; i < ui.size(); ++i) if (o->R.grid[i].ev == QM_EV_PRE) ui[i] = Vec(QM_NU, 0.0);
  o->R = SqpResult();
  try { sqpIteration(o->P, t0, tf, x0v, &xi, &ui, o->R); } catch (const std::exception&) { return -2; }
  const SqpResult& R = o->R; const int n = (int)R.grid.size(); if (n > maxn) return -1;
  *n_nodes = n;
  for (int i = 0; i < n; ++i) { node_t[i] = R.grid[i].t; node_ev[i] = R.grid[i].ev; node_mode[i] = R.mode[i]; std::memcpy(xs + QM_NX * i, R.x[i].data(), QM_NX * 8); std::memcpy(us + QM_NU * i, R.u[i].data(), QM_NU * 8); }
  const Performance* pf[2] = {&R.baseline, &R.after};
  for (int k = 0; k < 2; ++k) { perf[4 * k] = pf[k]->merit; perf[4 * k + 1] = pf[k]->cost; perf[4 * k + 2] = pf[k]->dynSSE; perf[4 * k + 3] = pf[k]->eqSSE; }
  perf[8] = R.alpha; perf[9] = R.armijo;
  return 0;
}
// warm-started iteration: the previous solution of this oracle is the initial guess (cold start if there is none)
int qmo_mpc_step_warm(void* h, double t0, double tf, const double* x0, int maxn, int* n_nodes, double* node_t, int* node_ev, int* node_mode, double* xs, double* us, double* perf) {
  Oracle* o = (Oracle*)h; Vec x0v(x0, x0 + QM_NX);
  SqpResult prev = o->R; o->R = SqpResult();
  try { sqpIteration(o->P, t0, tf, x0v, nullptr, nullptr, o->R, &prev); } catch (const std::exception&) { return -2; }
  if (o->R.status != 0) return o->R.status;      // e.g. -4: Riccati recursion not positive definite (a failed solve has no trajectories to copy)
  const SqpResult& R = o->R; const int n = (int)R.grid.size(); if (n > maxn) return -1;
  *n_nodes = n;
  for (int i = 0; i < n; ++i) { node_t[i] = R.grid[i].t; node_ev[i] = R.grid[i].ev; node_mode[i] = R.mode[i]; std::memcpy(xs + QM_NX * i, R.x[i].data(), QM_NX * 8); std::memcpy(us + QM_NU * i, R.u[i].data(), QM_NU * 8); }
  const Performance* pf[2] = {&R.baseline, &R.after};
  for (int k = 0; k < 2; ++k) { perf[4 * k] = pf[k]->merit; perf[4 * k + 1] = pf[k]->cost; perf[4 * k + 2] = pf[k]->dynSSE; perf[4 * k + 3] = pf[k]->eqSSE; }
  perf[8] = R.alpha; perf[9] = R.armijo;
  return 0;
}
// one discrete iLQR iteration (oracle/src/ilqr.h); warm != 0: inputs from this oracle's previous solution; same outputs as qmo_mpc_step
int qmo_ilqr_step(void* h, int warm, double t0, double tf, const double* x0, int maxn, int* n_nodes, double* node_t, int* node_ev, int* node_mode, double* xs, double* us, double* perf) {
  Oracle* o = (Oracle*)h; Vec x0v(x0, x0 + QM_NX);
  SqpResult prev = o->R; o->R = SqpResult();
  try { ilqrIteration(o->P, t0, tf, x0v, o->R, warm ? &prev : nullptr); } catch (const std::exception&) { return -2; }
  const SqpResult& R = o->R; if (R.status != 0) return R.status; const int n = (int)R.grid.size(); if (n > maxn) return -1;
  *n_nodes = n;
  for (int i = 0; i < n; ++i) { node_t[i] = R.grid[i].t; node_ev[i] = R.grid[i].ev; node_mode[i] = R.mode[i]; std::memcpy(xs + QM_NX * i, R.x[i].data(), QM_NX * 8); std::memcpy(us + QM_NU * i, R.u[i].data(), QM_NU * 8); }
  const Performance* pf[2] = {&R.baseline, &R.after};
  for (int k = 0; k < 2; ++k) { perf[4 * k] = pf[k]->merit; perf[4 * k + 1] = pf[k]->cost; perf[4 * k + 2] = pf[k]->dynSSE; perf[4 * k + 3] = pf[k]->eqSSE; }
  perf[8] = R.alpha; perf[9] = R.armijo;
  return 0;
}
// hard-inequality interior-point iteration (ipm.h); mode 0: cold, 1: warm from the previous solution, 2: one more iteration on the iterate of the last call (slack / dual / barrier kept)
static IpmDebug g_ipm_dbg;
int qmo_ipm_step(void* h, int mode, double t0, double tf, const double* x0, int maxn, int* n_nodes, double* node_t, int* node_ev, int* node_mode, double* xs, double* us, double* perf) {
  Oracle* o = (Oracle*)h; Vec x0v(x0, x0 + QM_NX);
  try {
    if (mode == 2) {
      if (o->R.grid.size() < 2) return -3;
      std::vector<Vec> xi = o->R.x, ui(o->R.u.begin(), o->R.u.end() - 1);
      for (size_t i = 0; i < ui.size(); ++i) if (o->R.grid[i].ev == QM_EV_PRE) ui[i] = Vec(QM_NU, 0.0);
      ipmIteration(o->P, t0, tf, x0v, &xi, &ui, o->R, nullptr, &g_ipm_dbg);
    } else { SqpResult prev = o->R; o->R = SqpResult(); ipmIteration(o->P, t0, tf, x0v, nullptr, nullptr, o->R, mode == 1 ? &prev : nullptr, &g_ipm_dbg); }
  } catch (const std::exception&) { return -2; }
  const SqpResult& R = o->R; if (R.status != 0) return R.status; const int n = (int)R.grid.size(); if (n > maxn) return -1;
  *n_nodes = n;
  for (int i = 0; i < n; ++i) { node_t[i] = R.grid[i].t; node_ev[i] = R.grid[i].ev; node_mode[i] = R.mode[i]; std::memcpy(xs + QM_NX * i, R.x[i].data(), QM_NX * 8); std::memcpy(us + QM_NU * i, R.u[i].data(), QM_NU * 8); }
  const Performance* pf[2] = {&R.baseline, &R.after};
  for (int k = 0; k < 2; ++k) { perf[4 * k] = pf[k]->merit; perf[4 * k + 1] = pf[k]->cost; perf[4 * k + 2] = pf[k]->dynSSE; perf[4 * k + 3] = pf[k]->eqSSE; }
  perf[8] = R.alpha; perf[9] = R.armijo;
  return 0;
}
// info[5] = barrier parameter after the iteration, primal / dual step limits, dual step taken, barrier parameter the iteration ran on (slot 4 filled by the caller's bookkeeping: see pyoracle)
void qmo_ipm_info(void* h, double* info) { const SqpResult& R = ((Oracle*)h)->R; info[0] = R.barrier; info[1] = R.alphaPrimalMax; info[2] = R.alphaDualMax; info[3] = R.alphaDual; }
// node i of the last interior-point iteration: slack / dual AFTER the step, their directions, the linearised rows (value, Hx, Hu, active) the step was computed on, dx / du,
// and the UNCONDENSED cost blocks (the dense KKT check of tests/test_ipm.py assembles the Newton system from these)
int qmo_ipm_node(void* h, int i, double* slack, double* dual, double* dslack, double* ddual, double* hv, double* Hx, double* Hu, int* on, double* dx, double* du, double* Q, double* Rm, double* q, double* r) {
  const SqpResult& R = ((Oracle*)h)->R; if (i < 0 || i >= (int)R.slack.size() || i >= (int)g_ipm_dbg.g.size()) return -1;
  const IneqLin& g = g_ipm_dbg.g[i]; const NodeLQ& n = g_ipm_dbg.lqUncondensed[i];
  for (int k = 0; k < QM_NH; ++k) { slack[k] = R.slack[i][k]; dual[k] = R.dual[i][k]; dslack[k] = R.dslack[i][k]; ddual[k] = R.ddual[i][k]; hv[k] = g.h[k]; on[k] = g.on[k] ? 1 : 0; }
  if (g.Hx.a.size()) { std::memcpy(Hx, g.Hx.a.data(), QM_NH * QM_NX * 8); std::memcpy(Hu, g.Hu.a.data(), QM_NH * QM_NU * 8); } else { std::memset(Hx, 0, QM_NH * QM_NX * 8); std::memset(Hu, 0, QM_NH * QM_NU * 8); }
  std::memcpy(dx, R.dx[i].data(), QM_NX * 8); std::memcpy(du, R.du[i].data(), QM_NU * 8);
  if (n.Q.a.size()) { std::memcpy(Q, n.Q.a.data(), 900 * 8); std::memcpy(Rm, n.R.a.data(), 900 * 8); std::memcpy(q, n.q.data(), 240); std::memcpy(r, n.r.data(), 240); }
  return 0;
}
// terminal node of the last iteration: Q_N, q_N (final end-effector soft constraint)
void qmo_terminal_lq(void* h, double* Q, double* q) { const SqpResult& R = ((Oracle*)h)->R; std::memcpy(Q, R.terminal.Qp.a.data(), 900 * 8); std::memcpy(q, R.terminal.qp.data(), 240); }
void qmo_eval_policy(void* h, double t, double* x, double* u, int* mode) {
  Oracle* o = (Oracle*)h; Vec xv, uv; int m; evaluatePolicy(o->R, o->P.ms, t, xv, uv, m);
  std::memcpy(x, xv.data(), QM_NX * 8); std::memcpy(u, uv.data(), QM_NU * 8); *mode = m;
}

void qmo_wbc_iters(void* h, int* it3) { for (int i = 0; i < 3; ++i) it3[i] = ((Oracle*)h)->dbg.iters[i]; }
void qmo_wbc_reset(void* h) { ((Oracle*)h)->W = WbcState(); }
void qmo_wbc_set_input_last(void* h, const double* u) { ((Oracle*)h)->W.inputLast.assign(u, u + QM_NU); }
// dbg (may be null): qMeas(24) vMeas(24) qDes(24) vDes(24) baseAcc(6) nle(24) x0(36) x1(36) x2(36) M(576) J(288) dJ(288)
int qmo_wbc(void* h, const double* xdes, const double* udes, const double* rbd, int mode, double period, double time, int mpc_variant, double* out54, int* status3, double* dbg) {
  Oracle* o = (Oracle*)h; Vec xd(xdes, xdes + QM_NX), ud(udes, udes + QM_NU), rb(rbd, rbd + QM_NRBD);
  Vec out = wbcUpdate(o->M, o->W, xd, ud, rb, mode, period, time, mpc_variant != 0, &o->dbg);
  std::memcpy(out54, out.data(), QM_NWBC_OUT * 8);
  if (status3) for (int i = 0; i < 3; ++i) status3[i] = o->dbg.status[i];
  if (dbg) {
    double* p = dbg; const WbcDebug& d = o->dbg;
    auto put = [&](const Vec& v) { std::memcpy(p, v.data(), v.size() * 8); p += v.size(); };
    put(d.qMeas); put(d.vMeas); put(d.qDes); put(d.vDes); put(d.baseAcc); put(d.nle); put(d.x0); put(d.x1); put(d.x2); put(d.Mq.a); put(d.J.a); put(d.dJ.a);
  }
  return 0;
}

// the three priority levels of the last qmo_wbc call as (A, b, D, f) — what WbcBase hands to HoQp (HierarchicalWbc.cpp:18-44); dims = {rows A, rows D};
// A / D row-major with 36 columns, caller buffers sized for 64 / 128 rows
int qmo_wbc_task(void* h, int level, int* dims, double* A, double* b, double* D, double* f) {
  Oracle* o = (Oracle*)h; if (level < 0 || level > 2) return -1; const Task& t = o->dbg.task[level];
  if (t.A.r > 64 || t.D.r > 128) return -2;
  dims[0] = t.A.r; dims[1] = t.D.r;
  for (int i = 0; i < t.A.r; ++i) { for (int j = 0; j < QM_NWBC; ++j) A[i * QM_NWBC + j] = t.A(i, j); b[i] = t.b[i]; }
  for (int i = 0; i < t.D.r; ++i) { for (int j = 0; j < QM_NWBC; ++j) D[i * QM_NWBC + j] = t.D(i, j); f[i] = t.f[i]; }
  return 0;
}

// ---- batched-plant restatement (oracle/src/sim.h) ----
void qmo_sim_params(void* h, const double* p) { SimParams& q = ((Oracle*)h)->simp; q.k_n = p[0]; q.d_n = p[1]; q.mu = p[2]; q.v_eps = p[3]; q.foot_radius = p[4]; q.delay = p[5]; q.saturate = p[6] != 0.0; }
void qmo_sim_reset(void* h, const double* q, const double* v, double time) { Oracle* o = (Oracle*)h; o->sim = SimState(); for (int i = 0; i < QM_NQ; ++i) { o->sim.q[i] = q[i]; o->sim.v[i] = v[i]; } o->sim.time = time; }
void qmo_sim_command(void* h, const double* pos, const double* vel, const double* kp, const double* kd, const double* ff) {
  SimCmd& c = ((Oracle*)h)->sim.held; for (int j = 0; j < QM_NJ; ++j) { c.pos[j] = pos[j]; c.vel[j] = vel[j]; c.kp[j] = kp[j]; c.kd[j] = kd[j]; c.ff[j] = ff[j]; }
}
int qmo_sim_step(void* h, double period, int nsub, double* q, double* v, double* time, double* force, int* contact) {
  Oracle* o = (Oracle*)h; simStep(o->M, o->simp, o->sim, period, nsub);
  for (int i = 0; i < QM_NQ; ++i) { q[i] = o->sim.q[i]; v[i] = o->sim.v[i]; } *time = o->sim.time; for (int i = 0; i < 12; ++i) force[i] = o->sim.force[i]; for (int i = 0; i < 4; ++i) contact[i] = o->sim.contact[i];
  return o->sim.status;
}
// qm::HoQp on an ARBITRARY cascade (HoQp.h:17-36: HoQp(task, higherProblem) level by level, getSolutions() of the last one): n decision variables, per level
// ma[k] equality rows (A row-major [ma][n], b) and md[k] inequality rows (D, f), concatenated over the levels from the highest priority down.  status[k] per level.
int qmo_hoqp(int n_levels, int n, const int* ma, const int* md, const double* A, const double* b, const double* D, const double* f, double* x_out, int* status, int* iters) {
  std::vector<HoLevel> lv; lv.reserve(n_levels);
  size_t oa = 0, ob = 0, od = 0, of = 0;
  for (int k = 0; k < n_levels; ++k) {
    Task t; t.A = Mat(ma[k], n); t.b.assign(b + ob, b + ob + ma[k]); t.D = Mat(md[k], n); t.f.assign(f + of, f + of + md[k]);
    for (int i = 0; i < ma[k]; ++i) for (int j = 0; j < n; ++j) t.A(i, j) = A[oa + (size_t)i * n + j];
    for (int i = 0; i < md[k]; ++i) for (int j = 0; j < n; ++j) t.D(i, j) = D[od + (size_t)i * n + j];
    oa += (size_t)ma[k] * n; ob += ma[k]; od += (size_t)md[k] * n; of += md[k];
    lv.push_back(solveHoLevel(t, k ? &lv[k - 1] : nullptr, n));
    if (status) status[k] = lv[k].status;
    if (iters) iters[k] = lv[k].iters;
  }
  for (int j = 0; j < n; ++j) x_out[j] = lv.back().x[j];
  return lv.back().status;
}

// rbd state (55) from generalized coordinates: zero velocities, EE pose by FK (StateEstimateBase.cpp:41-103 layout)
void qmo_rbd_from_q(void* h, const double* q, const double* v /*24 pinocchio, may be null*/, double* rbd) {
  Oracle* o = (Oracle*)h; std::fill(rbd, rbd + QM_NRBD, 0.0);
  for (int i = 0; i < 3; ++i) { rbd[i] = q[3 + i]; rbd[3 + i] = q[i]; }
  for (int j = 0; j < QM_NJ; ++j) rbd[6 + j] = q[6 + j];
  if (v) { M3<double> E = eulerZyxE(q[3], q[4]); V3<double> w = E * v3<double>(v[3], v[4], v[5]); for (int i = 0; i < 3; ++i) { rbd[24 + i] = w[i]; rbd[27 + i] = v[i]; } for (int j = 0; j < QM_NJ; ++j) rbd[30 + j] = v[6 + j]; }
  Kin<double> k; forwardKinematics(o->M, q, k); for (int i = 0; i < 3; ++i) rbd[48 + i] = k.fp[4][i];
  double qq[4]; matToQuat<double>(k.fR[4], qq); for (int i = 0; i < 4; ++i) rbd[51 + i] = qq[i];
}

// ---- batch driver for the cpu_baseline leg of bench.py: nthreads over instances; each instance = MPC step + policy eval at t0 + WBC ----
// inst arrays are per instance; schedule & target given per instance with fixed strides.
int qmo_batch_step(const double* mb, const double* st, int B, int nthreads, const double* t0, double horizon, const double* x0 /*[B][30]*/,
                   int K, const double* ref_t /*[B][K]*/, const double* ref_x /*[B][K][37]*/, int nev, const double* ev /*[B][nev]*/, const int* modes /*[B][nev+1]*/,
                   double period, double time, double* x_first /*[B][30]*/, double* u_first /*[B][30]*/, double* wbc_out /*[B][54]*/,
                   int maxn, int* n_nodes /*[B]*/, double* node_t /*[B][maxn]*/, int* node_ev, int* node_mode, double* xs /*[B][maxn][30]*/, double* us /*[B][maxn][30]*/) {
  // the trailing arrays (any may be null, maxn 0) receive each instance's whole primal solution: the full-size parity test compares trajectories, not only the policy at t0
  std::atomic<int> next(0), bad(0);
  auto work = [&]() {
    Oracle o; std::memcpy(o.M.mb, mb, sizeof(double) * MB_SIZE); std::memcpy(o.M.st, st, sizeof(double) * ST_SIZE);
    for (int b = next++; b < B; b = next++) {
      o.P.ms.ev.assign(ev + (size_t)b * nev, ev + (size_t)(b + 1) * nev); o.P.ms.modes.assign(modes + (size_t)b * (nev + 1), modes + (size_t)(b + 1) * (nev + 1));
      o.P.swing.update(o.M, o.P.ms);
      o.P.target.t.assign(ref_t + (size_t)b * K, ref_t + (size_t)(b + 1) * K); o.P.target.x.clear();
      for (int k = 0; k < K; ++k) o.P.target.x.emplace_back(ref_x + ((size_t)b * K + k) * QM_NREF, ref_x + ((size_t)b * K + k + 1) * QM_NREF);
      Vec x0v(x0 + (size_t)b * QM_NX, x0 + (size_t)(b + 1) * QM_NX);
      o.R = SqpResult(); sqpIteration(o.P, t0[b], t0[b] + horizon, x0v, nullptr, nullptr, o.R);
      if (o.R.status != 0) { ++bad; continue; }
      { const int n = (int)o.R.grid.size(); if (n_nodes) n_nodes[b] = n;
        for (int i = 0; i < n && i < maxn; ++i) {
          const size_t k = (size_t)b * maxn + i;
          if (node_t) node_t[k] = o.R.grid[i].t;
          if (node_ev) node_ev[k] = o.R.grid[i].ev;
          if (node_mode) node_mode[k] = o.R.mode[i];
          if (xs) std::memcpy(xs + k * QM_NX, o.R.x[i].data(), QM_NX * 8);
          if (us) std::memcpy(us + k * QM_NU, o.R.u[i].data(), QM_NU * 8);
        } }
      Vec xd, ud; int mode; evaluatePolicy(o.R, o.P.ms, t0[b], xd, ud, mode);
      double rbd[QM_NRBD]; qmo_rbd_from_q(&o, x0v.data() + 6, nullptr, rbd);
      o.W = WbcState(); Vec rb(rbd, rbd + QM_NRBD);
      Vec out = wbcUpdate(o.M, o.W, xd, ud, rb, mode, period, time, false, nullptr);
      if (x_first) std::memcpy(x_first + (size_t)b * QM_NX, xd.data(), QM_NX * 8);
      if (u_first) std::memcpy(u_first + (size_t)b * QM_NU, ud.data(), QM_NU * 8);
      if (wbc_out) std::memcpy(wbc_out + (size_t)b * QM_NWBC_OUT, out.data(), QM_NWBC_OUT * 8);
    }
  };
  std::vector<std::thread> th; for (int i = 0; i < nthreads; ++i) th.emplace_back(work); for (auto& t : th) t.join();
  return bad.load();
}

}  // extern "C"
