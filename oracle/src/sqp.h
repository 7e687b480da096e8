// oracle/src/sqp.h — TEST INFRASTRUCTURE (CPU oracle). One multiple-shooting SQP iteration as the
// reference runs it through ocs2::SqpMpc (qm_controllers/src/QMController.cpp:287-288; settings
// qm_controllers/config/task.info:75-92).  Restates [upstream ocs2_sqp SqpSolver::runImpl] per
// SURVEY.md §8 a11 / Appendix B.1, B.6.  PARITY UNPINNED.
#pragma once
#include "ocp.h"
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <functional>
#include <thread>
#include <pthread.h>
#include <sched.h>

// worker threads over shooting nodes for the LQ approximation and the line-search performance evaluation: what `sqp.nThreads 3`
// (qm_controllers/config/task.info:77) does in [upstream ocs2_sqp SqpSolver] (the Riccati / HPIPM solve stays serial there too).
// Results do not depend on the thread count: every node writes its own slot, sums are taken afterwards in node order.
// A persistent pool (the way OCS2's ThreadPool lives for the solver's lifetime): on this class of host a thread that is created per call and
// lives ~100 ms is often never migrated off its parent's core.
struct NodePool {
  std::vector<std::thread> th; std::mutex m; std::condition_variable cvWork, cvDone;
  const std::function<void(int)>* fn = nullptr; int n = 0; std::atomic<int> next{0}; int gen = 0, busy = 0; bool stop = false;
  void resize(int workers) {
    { std::unique_lock<std::mutex> l(m); stop = true; ++gen; } cvWork.notify_all(); for (auto& t : th) t.join(); th.clear(); stop = false;
    // each worker is pinned to its own allowed CPU (not the caller's): this host's scheduler otherwise leaves freshly woken workers stacked on the waker's core
    cpu_set_t allowed; CPU_ZERO(&allowed); std::vector<int> cpus;
    if (sched_getaffinity(0, sizeof(allowed), &allowed) == 0) { const int self = sched_getcpu(); for (int c = 0; c < CPU_SETSIZE; ++c) if (CPU_ISSET(c, &allowed) && c != self) cpus.push_back(c); }
    const int g0 = gen;                                   // (a worker that reads `gen` itself could start after the first run() bumped it and sleep through that job)
    for (int i = 0; i < workers; ++i) th.emplace_back([this, g0] { int seen = g0; for (;;) { std::unique_lock<std::mutex> l(m); cvWork.wait(l, [&] { return gen != seen; }); seen = gen; if (stop) return;
                                                                   l.unlock(); for (int k = next++; k < n; k = next++) (*fn)(k); l.lock(); if (--busy == 0) cvDone.notify_one(); } });
    if (!cpus.empty()) for (int i = 0; i < workers; ++i) { cpu_set_t one; CPU_ZERO(&one); CPU_SET(cpus[i % cpus.size()], &one); pthread_setaffinity_np(th[i].native_handle(), sizeof(one), &one); }
  }
  void run(int count, const std::function<void(int)>& f) {
    { std::unique_lock<std::mutex> l(m); fn = &f; n = count; next = 0; busy = (int)th.size(); ++gen; } cvWork.notify_all();
    for (int k = next++; k < count; k = next++) f(k);
    std::unique_lock<std::mutex> l(m); cvDone.wait(l, [&] { return busy == 0; });
  }
  ~NodePool() { resize(0); }
};
inline NodePool& nodePool() { static NodePool p; return p; }
inline int& oracleThreads() { static int n = 1; return n; }          // process-wide (one solver at a time uses the pool; the batch driver runs with 1)
inline void parallelFor(int n, const std::function<void(int)>& fn) {
  if (oracleThreads() <= 1 || n <= 1) { for (int i = 0; i < n; ++i) fn(i); return; }
  nodePool().run(n, fn);
}

static const double kWeakEps = 1e-6;                       // numeric_traits::weakEpsilon<double>()
static const double kLimitEps = 2.220446049250313e-16;     // numeric_traits::limitEpsilon<double>()

// parameter set of the multiple-shooting solver selected by ST_SOLVER: 2 = the `ipm` block (task.info:94-125) — with no hard inequality constraints in this OCP
// (friction cones / joint limits are soft costs, QMInterface.cpp:79-142) an interior-point iteration has no slack / dual variables and is the SQP step on these parameters
enum class MsParam { Dt, Iterations, DeltaTol, GMax, GMin };
inline double msParam(const double* st, MsParam p) {
  const bool ipm = st[ST_SOLVER] >= 2.0;      // 2: the SQP step on the `ipm` block's parameters; 3: the hard-inequality interior-point method (ipm.h)
  switch (p) {
    case MsParam::Dt: return ipm ? st[ST_IPM_DT] : st[ST_SQP_DT];
    case MsParam::Iterations: return ipm ? st[ST_IPM_ITER] : st[ST_SQP_ITER];
    case MsParam::DeltaTol: return ipm ? st[ST_IPM_DELTA_TOL] : st[ST_DELTA_TOL];
    case MsParam::GMax: return ipm ? st[ST_IPM_G_MAX] : st[ST_G_MAX];
    default: return ipm ? st[ST_IPM_G_MIN] : st[ST_G_MIN];
  }
}
struct Node { double t; int ev; };
inline double intervalStart(const Node& n) { return n.ev == QM_EV_POST ? n.t + kWeakEps : n.t; }
inline double intervalEnd(const Node& n) { return n.ev == QM_EV_PRE ? n.t - kWeakEps : n.t; }

// [upstream timeDiscretizationWithEvents]
// dtMin: [upstream]'s `dt_min` argument, default 10 * limitEpsilon (settings slot ST_GRID_DT_MIN).  With that default a grid node that falls within weakEpsilon BEFORE an
// event opens an interval whose adapted duration (intervalEnd − intervalStart, ∓ weakEpsilon at events) is NEGATIVE — any observation time in a 1 µs window per event and grid
// phase, about one MPC call in 3700 at 100 Hz.  That stage's cost blocks (× duration) are negative definite; riccatiSolve below survives it the way [upstream, recalled]
// HPIPM / BLASFEO does (zeroed pivots) and flags SqpResult::warn.  QM_GRID_DT_MIN_ROBUST (10 weakEpsilon: the node merges into the event node) is the opt-in variant.
inline std::vector<Node> timeDiscretizationWithEvents(double t0, double tf, double dt, const Vec& ev, double dtMin = 10.0 * kLimitEps) {
  std::vector<Node> g; g.push_back({t0, QM_EV_NONE});
  int k = findIndexInTimeArray(ev, t0);
  Node next = g.back();
  while (g.back().t < tf) {
    next.t = next.t + dt; next.ev = QM_EV_NONE; bool post = false;
    if (k < (int)ev.size() && next.t >= ev[k]) { next.t = ev[k]; next.ev = QM_EV_PRE; post = true; ++k; }
    if (next.t >= tf) { next.t = tf; next.ev = QM_EV_NONE; post = false; }
    if (next.t > g.back().t + dtMin) g.push_back(next); else g.back() = next;
    if (post) g.push_back({next.t, QM_EV_POST});
  }
  return g;
}

struct NodeLQ {
  // unprojected (K1 output)
  Mat A, B; Vec b;                 // dx+ = A dx + B du + b
  double c; Vec q, r; Mat Q, R, P; // cost model (already × dt)
  Mat C, D; Vec e; int nc = 0;     // equality rows
  double dt = 0; int event = 0;    // event==1: zero-duration PreEvent->PostEvent node (identity jump)
  // projection (K2 output): du = Pe + Px dx + Pu ut
  Mat Px, Pu; Vec Pe; int m = 0;
  Mat Ap, Bp; Vec bp; double cp; Vec qp, rp; Mat Qp, Rp, Pp;
  // Riccati (K3)
  Mat K; Vec kff;
};
struct Performance { double merit = 0, cost = 0, dynSSE = 0, eqSSE = 0; };
struct SqpResult {
  std::vector<Node> grid; std::vector<int> mode; std::vector<Vec> x, u;   // u has grid.size() entries (primal solution)
  std::vector<Vec> dx, du; std::vector<NodeLQ> lq; NodeLQ terminal;
  Performance baseline, after; double alpha = 0; int lsTrials = 0; double armijo = 0; int status = 0;
  int warn = 0;                    // warning bits of a VALID solution: QM_MPC_WARN_PIVOT = some stage's Huu had non-positive pivots, zeroed (riccatiSolve)
  // hard-inequality interior-point solver (ipm.h): slack / dual of every node's QM_NH inequality rows, their Newton directions, the barrier parameter and the step limits of the last iteration
  std::vector<Vec> slack, dual, dslack, ddual; double barrier = 0.0, alphaPrimalMax = 1.0, alphaDualMax = 1.0, alphaDual = 0.0;
  std::vector<double> lsTrace;     // diagnostics: per line-search trial {alpha, merit, theta, filter branch (0: theta > gMax, 1: Armijo, 2: cost-or-constraint decrease), accepted}; [0..4] of the baseline: {0, merit, theta0, armijo, -1}
  double phaseMs[3] = {0, 0, 0};   // wall time of the last iteration: LQ approximation + projection, Riccati solve, line search (the timers ocs2's benchmark prints)
};

// RK2 (Heun) flow value: x + dt/2 (k1 + k2)
inline Vec rk2Step(const Model& M, const Vec& x, const Vec& u, double dt) {
  Vec k1, k2; flowMapValue(M, x, u, k1);
  Vec x2(QM_NX); for (int i = 0; i < QM_NX; ++i) x2[i] = x[i] + dt * k1[i];
  flowMapValue(M, x2, u, k2);
  Vec r(QM_NX); for (int i = 0; i < QM_NX; ++i) r[i] = x[i] + 0.5 * dt * k1[i] + 0.5 * dt * k2[i];
  return r;
}

// K1: setupIntermediateNode
inline void setupIntermediateNode(const Problem& P, double t, double dt, const Vec& x, const Vec& xn, const Vec& u, NodeLQ& n, bool softIneq = true) {
  const Model& M = *P.M;
  Vec f1, f2; Mat A1, B1, A2, B2;
  flowMapLinear(M, x, u, f1, A1, B1);
  Vec x2(QM_NX); for (int i = 0; i < QM_NX; ++i) x2[i] = x[i] + dt * f1[i];
  flowMapLinear(M, x2, u, f2, A2, B2);
  Mat A2A1 = matmul(A2, A1), A2B1 = matmul(A2, B1);
  n.A = Mat(QM_NX, QM_NX); n.B = Mat(QM_NX, QM_NU); n.b.assign(QM_NX, 0.0);
  for (int i = 0; i < QM_NX; ++i) {
    for (int j = 0; j < QM_NX; ++j) n.A(i, j) = 0.5 * dt * A1(i, j) + 0.5 * dt * (A2(i, j) + dt * A2A1(i, j)) + (i == j ? 1.0 : 0.0);
    for (int j = 0; j < QM_NU; ++j) n.B(i, j) = 0.5 * dt * B1(i, j) + 0.5 * dt * (B2(i, j) + dt * A2B1(i, j));
    n.b[i] = x[i] + 0.5 * dt * f1[i] + 0.5 * dt * f2[i] - xn[i];
  }
  CostQuad c; intermediateCost(P, t, x, u, true, c, softIneq);
  n.c = c.f * dt; n.q = vscaled(c.q, dt); n.r = vscaled(c.r, dt); n.Q = scaled(c.Q, dt); n.R = scaled(c.R, dt); n.P = scaled(c.P, dt);
  equalityConstraints(P, t, x, u, true, n.e, n.C, n.D); n.nc = (int)n.e.size();
  n.dt = dt; n.event = 0;
}

// K2: projectTranscription with the QR null-space projection
inline void projectNode(NodeLQ& n) {
  const int nc = n.nc, nu = QM_NU; n.m = nu - nc;
  Mat Qf, Rf; householderQR(transpose(n.D), Qf, Rf);   // Dᵀ = Q1 R
  // D† = Q1 R⁻ᵀ ; Pu = Q2
  Mat Q1(nu, nc), Q2(nu, n.m);
  for (int i = 0; i < nu; ++i) { for (int j = 0; j < nc; ++j) Q1(i, j) = Qf(i, j); for (int j = 0; j < n.m; ++j) Q2(i, j) = Qf(i, nc + j); }
  // solve Rᵀ Y = [C e]  (Rᵀ lower triangular) -> Px = −Q1 Y_C, Pe = −Q1 y_e
  Mat Y(nc, QM_NX); Vec ye(nc);
  for (int col = 0; col <= QM_NX; ++col) {
    for (int i = 0; i < nc; ++i) {
      double s = (col < QM_NX) ? n.C(i, col) : n.e[i];
      for (int k = 0; k < i; ++k) s -= Rf(k, i) * ((col < QM_NX) ? Y(k, col) : ye[k]);
      s /= Rf(i, i);
      if (col < QM_NX) Y(i, col) = s; else ye[i] = s;
    }
  }
  n.Pu = Q2; n.Px = scaled(matmul(Q1, Y), -1.0); n.Pe = vscaled(matvec(Q1, ye), -1.0);
  // dynamics
  n.bp = vadd(n.b, matvec(n.B, n.Pe)); n.Ap = add(n.A, matmul(n.B, n.Px)); n.Bp = matmul(n.B, n.Pu);
  // cost (changeOfInputVariables)
  Vec RPe = matvec(n.R, n.Pe);
  n.cp = n.c + vdot(n.r, n.Pe) + 0.5 * vdot(n.Pe, RPe);
  Vec rr = vadd(n.r, RPe);
  n.qp = vadd(vadd(n.q, matvecT(n.Px, rr)), matvecT(n.P, n.Pe));
  n.rp = matvecT(n.Pu, rr);
  Mat PRPx = add(n.P, matmul(n.R, n.Px));          // P + R Px   (nu x nx)
  Mat PxtP = matmulTN(n.Px, n.P);
  n.Qp = add(add(n.Q, PxtP), add(transpose(PxtP), matmulTN(n.Px, matmul(n.R, n.Px))));
  n.Pp = matmulTN(n.Pu, PRPx);
  n.Rp = matmulTN(n.Pu, matmul(n.R, n.Pu));
}

// performance of a trajectory (computePerformance)
inline Performance computePerformance(const Problem& P, const std::vector<Node>& g, const Vec& x0, const std::vector<Vec>& x, const std::vector<Vec>& u) {
  const Model& M = *P.M; const int N = (int)g.size() - 1; Performance p;
  std::vector<double> dyn(N, 0.0), cost(N, 0.0), eq(N, 0.0);
  parallelFor(N, [&](int i) {
    if (g[i].ev == QM_EV_PRE) {
      double s = 0; for (int k = 0; k < QM_NX; ++k) { const double d = x[i][k] - x[i + 1][k]; s += d * d; }
      dyn[i] = s;
    } else {
      const double ti = intervalStart(g[i]); const double dt = intervalEnd(g[i + 1]) - ti;
      Vec xe = rk2Step(M, x[i], u[i], dt);
      double s = 0; for (int k = 0; k < QM_NX; ++k) { const double d = xe[k] - x[i + 1][k]; s += d * d; }
      dyn[i] = dt * s;
      CostQuad c; intermediateCost(P, ti, x[i], u[i], false, c); cost[i] = c.f * dt;
      Vec e; Mat C, D; equalityConstraints(P, ti, x[i], u[i], false, e, C, D);
      double se = 0; for (double v : e) se += v * v; eq[i] = dt * se;
    }
  });
  for (int i = 0; i < N; ++i) { p.dynSSE += dyn[i]; if (g[i].ev != QM_EV_PRE) { p.cost += cost[i]; p.eqSSE += eq[i]; } }
  { CostQuad c; terminalCost(P, intervalStart(g[N]), x[N], false, c); p.cost += c.f; }
  { double s = 0; for (int k = 0; k < QM_NX; ++k) { const double d = x0[k] - x[0][k]; s += d * d; } p.dynSSE += s; }
  p.merit = p.cost;
  return p;
}

inline double trajectoryNorm(const std::vector<Vec>& v) { double s = 0; for (auto& a : v) for (double z : a) s += z * z; return std::sqrt(s); }

// QP sub-problem of the projected LQ model: Riccati backward sweep (feedback gains K, feed-forward kff per node) and the LINEAR forward rollout
// (dx, du, Armijo descent metric).  Shared by the SQP iteration and the discrete iLQR iteration.  
// A stage whose Huu is NOT positive definite (the negative-duration interval in front of a gait event, see timeDiscretizationWithEvents) does not abort the solve:
// [upstream, recalled — HPIPM's d_ocp_qp_fact_solve_kkt_unconstr factorises with BLASFEO dpotrf kernels, which store a zero diagonal entry and a zero reciprocal for a pivot
// that is not positive instead of failing] the pivot's column of L is zero, so that reduced input gets K_j = 0, kff_j = 0 (no update on this stage) and drops out of the
// Schur complement S' = Q + AᵀSA + Huxᵀ K; the other inputs are solved as if it were not there.  The pivot ORDER is the order of the reduced inputs ũ, i.e. of this restatement's
// null-space basis (OCS2's differs), so WHICH directions are dropped is not pinned to upstream — on the stage this is there for (duration ≈ −5e-7 s, Huu ≈ duration · R)
// every pivot is negative and every order drops them all.  strict (settings slot ST_RICCATI_STRICT): report the hard failure of rounds 1-3 instead (status -2 here, -4 on the device).
inline bool riccatiSolve(SqpResult& R, const Vec& x0, const std::vector<Vec>& x, bool strict = false) {
  const int N = (int)R.grid.size() - 1;
  // ---- QP solve: Riccati ----
  Mat S = R.terminal.Qp; Vec s = R.terminal.qp;
  for (int k = N - 1; k >= 0; --k) {
    NodeLQ& n = R.lq[k];
    Vec Sb = matvec(S, n.bp); Vec spSb = vadd(s, Sb);
    Mat SA = matmul(S, n.Ap);
    if (n.event) { S = matmulTN(n.Ap, SA); s = matvecT(n.Ap, spSb); }
    else {
      Mat BtS = matmulTN(n.Bp, S);
      Mat Huu = add(n.Rp, matmul(BtS, n.Bp));
      Mat Hux = add(n.Pp, matmul(BtS, n.Ap));
      Vec hu = vadd(n.rp, matvecT(n.Bp, spSb));
      for (int i = 0; i < Huu.r; ++i) for (int j = i + 1; j < Huu.c; ++j) { const double a = 0.5 * (Huu(i, j) + Huu(j, i)); Huu(i, j) = Huu(j, i) = a; }
      // benign only on a stage of NON-POSITIVE duration with a finite Huu; a non-positive pivot on a stage of positive duration (Huu genuinely indefinite) or an entry that is
      // not a number (a NaN in the observation) is the hard failure [upstream: SqpSolver throws on HPIPM's NaN status]
      bool finite = true; for (int i = 0; i < Huu.r; ++i) for (int j = 0; j < Huu.c; ++j) if (!std::isfinite(Huu(i, j))) finite = false;
      Mat L; if (choleskyZeroPivots(Huu, L) > 0 || !finite) { if (strict || !finite || n.dt > 0.0) { R.status = -2; return false; } R.warn |= QM_MPC_WARN_PIVOT; }
      n.K = scaled(cholSolve(L, Hux), -1.0); n.kff = vscaled(cholSolve(L, hu), -1.0);
      Mat Snew = add(add(n.Qp, matmulTN(n.Ap, SA)), matmulTN(Hux, n.K));
      for (int i = 0; i < QM_NX; ++i) for (int j = i + 1; j < QM_NX; ++j) { const double a = 0.5 * (Snew(i, j) + Snew(j, i)); Snew(i, j) = Snew(j, i) = a; }
      s = vadd(vadd(n.qp, matvecT(n.Ap, spSb)), matvecT(Hux, n.kff));
      S = Snew;
    }
  }
  R.dx.assign(N + 1, Vec(QM_NX, 0.0)); R.du.assign(N, Vec(QM_NU, 0.0));
  for (int k = 0; k < QM_NX; ++k) R.dx[0][k] = x0[k] - x[0][k];
  double armijo = 0;
  for (int k = 0; k < N; ++k) {
    NodeLQ& n = R.lq[k];
    if (n.event) { R.dx[k + 1] = vadd(matvec(n.Ap, R.dx[k]), n.bp); armijo += vdot(n.qp, R.dx[k]); continue; }
    Vec ut = vadd(matvec(n.K, R.dx[k]), n.kff);
    R.dx[k + 1] = vadd(vadd(matvec(n.Ap, R.dx[k]), matvec(n.Bp, ut)), n.bp);
    armijo += vdot(n.qp, R.dx[k]) + vdot(n.rp, ut);
    R.du[k] = vadd(vadd(n.Pe, matvec(n.Px, R.dx[k])), matvec(n.Pu, ut));   // remapProjectedInput
  }
  armijo += vdot(R.terminal.qp, R.dx[N]);
  R.armijo = armijo;
  if (!std::isfinite(armijo) || !std::isfinite(trajectoryNorm(R.dx)) || !std::isfinite(trajectoryNorm(R.du))) { R.status = -2; return false; }      // a step that is not finite is a failed solve (the device: -4)
  return true;
}

struct SqpResult;
inline void evaluatePolicy(const SqpResult& R, const ModeSchedule& ms, double t, Vec& x, Vec& u, int& mode);

// initializeStateInputTrajectories: cold start QMInitializer::compute (QMInitializer.cpp:33-41), warm start from `prev` (see sqpIteration); shared with ipm.h
inline void initialGuess(const Problem& P, const SqpResult& R, const Vec& x0, const SqpResult* prev, std::vector<Vec>& x, std::vector<Vec>& u) {
  const Model& M = *P.M; const int N = (int)R.grid.size() - 1;

    const bool warm = prev && prev->grid.size() >= 2;
    const double tend = warm ? prev->grid.back().t : 0.0;
    x[0] = x0;
    for (int i = 0; i < N; ++i) {
      if (R.grid[i].ev == QM_EV_PRE) { u[i] = Vec(QM_NU, 0.0); x[i + 1] = x[i]; continue; }
      const double time = intervalStart(R.grid[i]), nextTime = intervalEnd(R.grid[i + 1]);
      if (warm && !(time > tend || nextTime > tend)) {
        Vec xa, ua, xb, ub; int md; evaluatePolicy(*prev, P.ms, time, xa, ua, md); evaluatePolicy(*prev, P.ms, nextTime, xb, ub, md);
        u[i] = ua; x[i + 1] = xb;
      } else { bool fl[4]; modeToFlags(P.ms.modeAt(time), fl); u[i] = weightCompensatingInput(M, fl); x[i + 1] = x[i]; }
    }
}


// one SQP iteration; initial guess: xInit/uInit if given, else warm start from `prev` (a previous primal solution, may be null /
// empty -> cold start).  Warm start restates [upstream ocs2_sqp multiple_shooting::initializeStateInputTrajectories]: x_0 = x0; interval i
// takes u_i = u_prev(intervalStart(i)) and x_{i+1} = x_prev(intervalEnd(i+1)) while the previous solution covers both times, the
// initializer a9 (weight-compensating input, x_{i+1} = x_i) beyond it; PreEvent nodes have no input and copy their state forward.
inline void sqpIteration(const Problem& P, double t0, double tf, const Vec& x0, const std::vector<Vec>* xInit, const std::vector<Vec>* uInit, SqpResult& R, const SqpResult* prev = nullptr);
inline void sqpIteration(const Problem& P, double t0, double tf, const Vec& x0, const std::vector<Vec>* xInit, const std::vector<Vec>* uInit, SqpResult& R, const SqpResult* prev) {
  const Model& M = *P.M; const double* st = M.st;
  R.grid = timeDiscretizationWithEvents(t0, tf, msParam(st, MsParam::Dt), P.ms.ev, st[ST_GRID_DT_MIN]);
  const int N = (int)R.grid.size() - 1;
  R.mode.resize(N + 1); for (int i = 0; i <= N; ++i) R.mode[i] = P.ms.modeAt(intervalStart(R.grid[i]));
  // initializeStateInputTrajectories, cold start: QMInitializer::compute (QMInitializer.cpp:33-41)
  std::vector<Vec> x(N + 1), u(N);
  if (xInit) { x = *xInit; u = *uInit; }
  else initialGuess(P, R, x0, prev, x, u);
  // ---- setupQuadraticSubproblem ----
  const auto tq0 = std::chrono::steady_clock::now();
  R.lq.assign(N, NodeLQ()); Performance base;
  parallelFor(N, [&](int i) {
    NodeLQ& n = R.lq[i];
    if (R.grid[i].ev == QM_EV_PRE) {   // setupEventNode: identity jump map, no pre-jump cost/constraints (QMInterface.cpp:79-142)
      n.event = 1; n.m = 0; n.nc = 0; n.dt = 0;
      n.Ap = Mat::identity(QM_NX); n.A = n.Ap; n.bp.assign(QM_NX, 0.0);
      for (int k = 0; k < QM_NX; ++k) n.bp[k] = x[i][k] - x[i + 1][k];
      n.b = n.bp; n.Qp = Mat(QM_NX, QM_NX); n.Q = n.Qp; n.qp.assign(QM_NX, 0.0); n.q = n.qp; n.cp = n.c = 0;
    } else {
      const double ti = intervalStart(R.grid[i]); const double dt = intervalEnd(R.grid[i + 1]) - ti;
      setupIntermediateNode(P, ti, dt, x[i], x[i + 1], u[i], n);
      projectNode(n);
    }
  });
  for (int i = 0; i < N; ++i) {          // baseline performance, summed in node order (independent of the thread count)
    const NodeLQ& n = R.lq[i];
    if (n.event) { double s = 0; for (double v : n.bp) s += v * v; base.dynSSE += s; }
    else {
      double s = 0; for (double v : n.b) s += v * v; base.dynSSE += n.dt * s;
      base.cost += n.c;
      double se = 0; for (double v : n.e) se += v * v; base.eqSSE += n.dt * se;
    }
  }
  { CostQuad c; terminalCost(P, intervalStart(R.grid[N]), x[N], true, c); R.terminal.Qp = c.Q; R.terminal.qp = c.q; R.terminal.cp = c.f; base.cost += c.f; }
  { double s = 0; for (int k = 0; k < QM_NX; ++k) { const double d = x0[k] - x[0][k]; s += d * d; } base.dynSSE += s; }
  base.merit = base.cost; R.baseline = base;
  const auto tq1 = std::chrono::steady_clock::now();
  if (!riccatiSolve(R, x0, x, st[ST_RICCATI_STRICT] != 0.0)) return;
  const double armijo = R.armijo;
  const auto tq2 = std::chrono::steady_clock::now();
  // ---- takeStep: filter line-search ----
  const double gMax = msParam(st, MsParam::GMax), gMin = msParam(st, MsParam::GMin), gammaC = 1e-6, armijoFactor = 1e-4, alphaDecay = 0.5, alphaMin = 1e-4;
  const double theta0 = std::sqrt(base.dynSSE + base.eqSSE);
  const double duNorm = trajectoryNorm(R.du), dxNorm = trajectoryNorm(R.dx);
  double alpha = 1.0; bool accepted = false; std::vector<Vec> xn(N + 1), un(N); Performance pn; R.lsTrials = 0;
  R.lsTrace.assign({0.0, base.merit, theta0, armijo, -1.0});
  do {
    for (int i = 0; i <= N; ++i) { xn[i] = x[i]; for (int k = 0; k < QM_NX; ++k) xn[i][k] += alpha * R.dx[i][k]; }
    for (int i = 0; i < N; ++i) { un[i] = u[i]; if (R.grid[i].ev != QM_EV_PRE) for (int k = 0; k < QM_NU; ++k) un[i][k] += alpha * R.du[i][k]; }
    pn = computePerformance(P, R.grid, x0, xn, un); ++R.lsTrials;
    const double theta = std::sqrt(pn.dynSSE + pn.eqSSE);
    if (theta > gMax) accepted = theta < (1.0 - gammaC) * theta0;
    else if (theta < gMin && theta0 < gMin && alpha * armijo < 0.0) accepted = pn.merit < base.merit + armijoFactor * alpha * armijo;
    else accepted = pn.merit < (base.merit - gammaC * theta0) || theta < (1.0 - gammaC) * theta0;
    { const double br = (theta > gMax) ? 0.0 : ((theta < gMin && theta0 < gMin && alpha * armijo < 0.0) ? 1.0 : 2.0); const double row[5] = {alpha, pn.merit, theta, br, accepted ? 1.0 : 0.0}; R.lsTrace.insert(R.lsTrace.end(), row, row + 5); }
    if (accepted) break;
    alpha *= alphaDecay;
    if (alpha * duNorm < msParam(st, MsParam::DeltaTol) && alpha * dxNorm < msParam(st, MsParam::DeltaTol)) break;
  } while (alpha >= alphaMin);
  if (accepted) { x = xn; u = un; R.alpha = alpha; R.after = pn; } else { R.alpha = 0.0; R.after = base; }
  // ---- toPrimalSolution: u at PreEvent nodes copied from the previous node, last u repeated
  R.x = x; R.u.assign(N + 1, Vec(QM_NU, 0.0));
  for (int i = 0; i < N; ++i) { if (R.grid[i].ev == QM_EV_PRE && i > 0) R.u[i] = R.u[i - 1]; else R.u[i] = u[i]; }
  R.u[N] = R.u[N - 1];
  R.status = 0;
  const auto tq3 = std::chrono::steady_clock::now();
  auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  R.phaseMs[0] = ms(tq0, tq1); R.phaseMs[1] = ms(tq1, tq2); R.phaseMs[2] = ms(tq2, tq3);
}

// a12: MPC_MRT_Interface::evaluatePolicy [upstream]: linear interpolation of the primal solution
inline void evaluatePolicy(const SqpResult& R, const ModeSchedule& ms, double t, Vec& x, Vec& u, int& mode) {
  const int n = (int)R.grid.size(); Vec ta(n);
  for (int i = 0; i < n; ++i) ta[i] = R.grid[i].t + (R.grid[i].ev == QM_EV_POST ? kLimitEps : (R.grid[i].ev == QM_EV_PRE ? -kLimitEps : 0.0));
  int i; double a; timeSegment(t, ta, i, a);
  x.assign(QM_NX, 0.0); u.assign(QM_NU, 0.0);
  for (int k = 0; k < QM_NX; ++k) x[k] = a * R.x[i][k] + (1.0 - a) * R.x[i + 1][k];
  for (int k = 0; k < QM_NU; ++k) u[k] = a * R.u[i][k] + (1.0 - a) * R.u[i + 1][k];
  mode = ms.modeAt(t);
}
