// oracle/src/ipm.h — TEST INFRASTRUCTURE (CPU oracle).  One iteration of a primal-dual INTERIOR-POINT multiple-shooting solver with HARD inequality constraints
// ("IPM for hard friction cones", settings block `ipm`, qm_controllers/config/task.info:94-125, loaded at qm_interface/src/QMInterface.cpp:72).
// The reference registers friction cones and arm joint limits as SOFT costs only (QMInterface.cpp:116-131, 177-259) and instantiates no IpmMpc, so this solver solves a
// DIFFERENT problem than the controller's — the same OCP with
//     h(x, u) >= 0 :  arm joint position boxes (12 rows), arm joint velocity boxes (12 rows), the friction cone of every stance foot (<= 4 rows)      (QM_NH = 28, qmhip_layout.h)
// as constraints and WITHOUT their relaxed-barrier cost terms — and there is no reference behaviour to match: it restates the structure of [upstream ocs2_ipm IpmSolver +
// IpmHelpers, recalled] — PARITY UNPINNED by reference data — and is pinned instead by a dense primal-dual KKT solve of the whole-horizon Newton system (tests/test_ipm.py).
//
//   0. grid, modes, initial (x, u) as the SQP (sqp.h initialGuess); first iteration of a solve: slack s = (1 + marginRate) max(h(x, u), initialSlackLowerBound),
//      dual lam = (1 + marginRate) max(mu / s, initialDualLowerBound), mu = initialBarrierParameter          [ipm::initializeSlackVariable / initializeDualVariable]
//   1. per node: LQ model of dynamics, cost (tracking + end-effector term), equality rows as the SQP; linearised inequality rows h, Hx, Hu;
//      CONDENSING [ipm::condenseIneqConstraints]:  W = diag(lam / s),  c = (lam ∘ h − mu) / s − lam
//          Q += Hxᵀ W Hx,  P += Huᵀ W Hx,  R += Huᵀ W Hu,  q += Hxᵀ c,  r += Huᵀ c          (constraint rows are NOT scaled by dt; the cost blocks they are added to are)
//      then the SQP's equality-constraint projection (projectNode) and Riccati solve (riccatiSolve) -> dx, du
//   2. directions  ds = h + Hx dx + Hu du − s,   dlam = −(lam ∘ ds + (s ∘ lam − mu)) / s          [ipm::retrieveSlackDirection / retrieveDualDirection]
//      step limits (ONE pair per solve, over all nodes)  alphaP = fractionToBoundary(s, ds),  alphaD = fractionToBoundary(lam, dlam),
//          fractionToBoundary(v, dv) = min(1, 1 / max_i(−dv_i / (margin v_i)))  (1 when that maximum is not positive)
//   3. filter line search as the SQP's, from alpha = alphaP, on  merit = cost − mu Σ ln s,  theta² = dynamics SSE + dt (|e|² + |h − s|²)   per node          [ipm::computePerformanceIndex]
//      accepted:  x += alpha dx, u += alpha du, s += alpha ds,  lam += alphaDual dlam with alphaDual = usePrimalStepSizeForDual ? min(alpha, alphaD) : alphaD
//   4. barrier update [IpmSolver::updateBarrierParameter]: when |merit_before − merit_after| < barrierReductionCostTol and theta_after < barrierReductionConstraintTol:
//          mu <- max(targetBarrierParameter, min(barrierLinearDecreaseFactor mu, mu ^ barrierSuperlinearDecreasePower))
// Slack, dual and mu persist across the iterations of ONE solve; every solve (cold or warm MPC call) re-initialises them at its initial iterate, as the device does (qm_pipeline.h: ipm_fresh).
#pragma once
#include "sqp.h"

struct IneqLin { Vec h; Mat Hx, Hu; bool on[QM_NH]; };
// rows in the order of qmhip_layout.h (QM_NH); an inactive row (cone of a swing foot) has on = false, h = 1, zero Jacobian rows
inline void inequalityConstraints(const Problem& P, double t, const Vec& x, const Vec& u, bool linear, IneqLin& g) {
  const Model& M = *P.M; const double* st = M.st; bool fl[4]; modeToFlags(P.ms.modeAt(t), fl);
  g.h.assign(QM_NH, 1.0); if (linear) { g.Hx = Mat(QM_NH, QM_NX); g.Hu = Mat(QM_NH, QM_NU); }
  for (int k = 0; k < 6; ++k) {
    const double lo = M.mb[MB_QLO + 12 + k], hi = M.mb[MB_QHI + 12 + k], vlo = st[ST_JVEL_LO + k], vhi = st[ST_JVEL_HI + k];
    g.h[2 * k] = x[24 + k] - lo; g.h[2 * k + 1] = hi - x[24 + k]; g.h[12 + 2 * k] = u[24 + k] - vlo; g.h[13 + 2 * k] = vhi - u[24 + k];
    for (int r = 0; r < 2; ++r) { g.on[2 * k + r] = true; g.on[12 + 2 * k + r] = true; }
    if (linear) { g.Hx(2 * k, 24 + k) = 1.0; g.Hx(2 * k + 1, 24 + k) = -1.0; g.Hu(12 + 2 * k, 24 + k) = 1.0; g.Hu(13 + 2 * k, 24 + k) = -1.0; }
  }
  const double muf = st[ST_FRIC_COEF], reg = st[ST_FRIC_REG];
  for (int c = 0; c < 4; ++c) {
    g.on[24 + c] = fl[c]; if (!fl[c]) continue;
    const double Fx = u[3 * c], Fy = u[3 * c + 1], Fz = u[3 * c + 2], Tn = std::sqrt(Fx * Fx + Fy * Fy + reg);
    g.h[24 + c] = muf * Fz - Tn;                                   // [upstream FrictionConeConstraint::coneConstraint], gripper force 0
    if (linear) { g.Hu(24 + c, 3 * c) = -Fx / Tn; g.Hu(24 + c, 3 * c + 1) = -Fy / Tn; g.Hu(24 + c, 3 * c + 2) = muf; }
  }
}

inline double fractionToBoundary(const std::vector<Vec>& v, const std::vector<Vec>& dv, const std::vector<IneqLin>& g, double margin) {
  double worst = 0.0;
  for (size_t i = 0; i < v.size(); ++i) for (int r = 0; r < (int)v[i].size(); ++r) if (g[i].on[r]) worst = std::max(worst, -dv[i][r] / (margin * v[i][r]));
  return worst > 0.0 ? std::min(1.0, 1.0 / worst) : 1.0;
}

// performance of a trajectory with its slacks (ipm::computePerformanceIndex [upstream, recalled]): cost without the soft inequality terms − mu Σ ln s; theta² gains dt |h − s|²
inline Performance computePerformanceIpm(const Problem& P, const std::vector<Node>& g, const Vec& x0, const std::vector<Vec>& x, const std::vector<Vec>& u, const std::vector<Vec>& s, double mu) {
  const Model& M = *P.M; const int N = (int)g.size() - 1; Performance p;
  std::vector<double> dyn(N, 0.0), cost(N, 0.0), eq(N, 0.0);
  parallelFor(N, [&](int i) {
    if (g[i].ev == QM_EV_PRE) { double d2 = 0; for (int k = 0; k < QM_NX; ++k) { const double d = x[i][k] - x[i + 1][k]; d2 += d * d; } dyn[i] = d2; return; }
    const double ti = intervalStart(g[i]); const double dt = intervalEnd(g[i + 1]) - ti;
    Vec xe = rk2Step(M, x[i], u[i], dt);
    double d2 = 0; for (int k = 0; k < QM_NX; ++k) { const double d = xe[k] - x[i + 1][k]; d2 += d * d; }
    dyn[i] = dt * d2;
    CostQuad c; intermediateCost(P, ti, x[i], u[i], false, c, false); double cv = c.f * dt;
    Vec e; Mat C, D; equalityConstraints(P, ti, x[i], u[i], false, e, C, D);
    double se = 0; for (double v : e) se += v * v;
    IneqLin q; inequalityConstraints(P, ti, x[i], u[i], false, q);
    for (int r = 0; r < QM_NH; ++r) if (q.on[r]) { cv -= mu * std::log(s[i][r]); const double d = q.h[r] - s[i][r]; se += d * d; }
    cost[i] = cv; eq[i] = dt * se;
  });
  for (int i = 0; i < N; ++i) { p.dynSSE += dyn[i]; if (g[i].ev != QM_EV_PRE) { p.cost += cost[i]; p.eqSSE += eq[i]; } }
  { CostQuad c; terminalCost(P, intervalStart(g[N]), x[N], false, c); p.cost += c.f; }
  { double d2 = 0; for (int k = 0; k < QM_NX; ++k) { const double d = x0[k] - x[0][k]; d2 += d * d; } p.dynSSE += d2; }
  p.merit = p.cost;
  return p;
}

// per-node linearised inequality rows of the last iteration (tests: the dense KKT check needs them next to R.lq's uncondensed blocks)
struct IpmDebug { std::vector<IneqLin> g; std::vector<NodeLQ> lqUncondensed; };

// one interior-point iteration.  xInit / uInit: continue on the given iterate with R's slack / dual / barrier (ipm.ipmIteration > 1); else initial guess from `prev` (warm) or cold
inline void ipmIteration(const Problem& P, double t0, double tf, const Vec& x0, const std::vector<Vec>* xInit, const std::vector<Vec>* uInit, SqpResult& R, const SqpResult* prev = nullptr, IpmDebug* dbg = nullptr) {
  const Model& M = *P.M; const double* st = M.st;
  R.grid = timeDiscretizationWithEvents(t0, tf, st[ST_IPM_DT], P.ms.ev, st[ST_GRID_DT_MIN]);
  const int N = (int)R.grid.size() - 1;
  R.mode.resize(N + 1); for (int i = 0; i <= N; ++i) R.mode[i] = P.ms.modeAt(intervalStart(R.grid[i]));
  std::vector<Vec> x(N + 1), u(N);
  const bool cont = xInit != nullptr && (int)R.slack.size() == N && R.barrier > 0.0;
  if (xInit) { x = *xInit; u = *uInit; } else initialGuess(P, R, x0, prev, x, u);
  auto tOf = [&](int i) { return intervalStart(R.grid[i]); };
  auto dtOf = [&](int i) { return intervalEnd(R.grid[i + 1]) - intervalStart(R.grid[i]); };
  // ---- 0. slack / dual ----
  if (!cont) {
    R.barrier = st[ST_IPM_MU]; R.slack.assign(N, Vec(QM_NH, 1.0)); R.dual.assign(N, Vec(QM_NH, 0.0));
    for (int i = 0; i < N; ++i) {
      if (R.grid[i].ev == QM_EV_PRE) continue;
      IneqLin q; inequalityConstraints(P, tOf(i), x[i], u[i], false, q);
      for (int r = 0; r < QM_NH; ++r) if (q.on[r]) {
        R.slack[i][r] = (1.0 + st[ST_IPM_SLACK_MARGIN]) * std::max(q.h[r], st[ST_IPM_SLACK_LB]);
        R.dual[i][r] = (1.0 + st[ST_IPM_DUAL_MARGIN]) * std::max(R.barrier / R.slack[i][r], st[ST_IPM_DUAL_LB]);
      }
    }
  }
  const double mu = R.barrier;
  // ---- 1. LQ model, condensing, projection ----
  const auto tq0 = std::chrono::steady_clock::now();
  R.lq.assign(N, NodeLQ()); std::vector<IneqLin> G(N); Performance base; std::vector<double> nodeBarrier(N, 0.0), nodeIneq(N, 0.0);
  if (dbg) dbg->lqUncondensed.assign(N, NodeLQ());
  parallelFor(N, [&](int i) {
    NodeLQ& n = R.lq[i];
    if (R.grid[i].ev == QM_EV_PRE) {
      n.event = 1; n.m = 0; n.nc = 0; n.dt = 0; n.Ap = Mat::identity(QM_NX); n.A = n.Ap; n.bp.assign(QM_NX, 0.0);
      for (int k = 0; k < QM_NX; ++k) n.bp[k] = x[i][k] - x[i + 1][k];
      n.b = n.bp; n.Qp = Mat(QM_NX, QM_NX); n.Q = n.Qp; n.qp.assign(QM_NX, 0.0); n.q = n.qp; n.cp = n.c = 0;
      G[i].h.assign(QM_NH, 1.0); for (int r = 0; r < QM_NH; ++r) G[i].on[r] = false;
      return;
    }
    setupIntermediateNode(P, tOf(i), dtOf(i), x[i], x[i + 1], u[i], n, false);
    IneqLin& q = G[i]; inequalityConstraints(P, tOf(i), x[i], u[i], true, q);
    if (dbg) dbg->lqUncondensed[i] = n;
    const Vec& s = R.slack[i]; const Vec& lam = R.dual[i];
    for (int r = 0; r < QM_NH; ++r) if (q.on[r]) {
      nodeBarrier[i] -= mu * std::log(s[r]); const double d = q.h[r] - s[r]; nodeIneq[i] += d * d;
      const double w = lam[r] / s[r], c = (lam[r] * q.h[r] - mu) / s[r] - lam[r];
      for (int a = 0; a < QM_NX; ++a) { const double ha = q.Hx(r, a); if (ha == 0.0) continue; n.q[a] += ha * c; for (int b = 0; b < QM_NX; ++b) n.Q(a, b) += ha * w * q.Hx(r, b); }
      for (int a = 0; a < QM_NU; ++a) { const double ha = q.Hu(r, a); if (ha == 0.0) continue; n.r[a] += ha * c; for (int b = 0; b < QM_NU; ++b) n.R(a, b) += ha * w * q.Hu(r, b); for (int b = 0; b < QM_NX; ++b) n.P(a, b) += ha * w * q.Hx(r, b); }
    }
    projectNode(n);
  });
  for (int i = 0; i < N; ++i) {
    const NodeLQ& n = R.lq[i];
    if (n.event) { double d2 = 0; for (double v : n.bp) d2 += v * v; base.dynSSE += d2; continue; }
    double d2 = 0; for (double v : n.b) d2 += v * v; base.dynSSE += n.dt * d2;
    base.cost += n.c + nodeBarrier[i];
    double se = 0; for (double v : n.e) se += v * v; base.eqSSE += n.dt * (se + nodeIneq[i]);
  }
  { CostQuad c; terminalCost(P, intervalStart(R.grid[N]), x[N], true, c); R.terminal.Qp = c.Q; R.terminal.qp = c.q; R.terminal.cp = c.f; base.cost += c.f; }
  { double d2 = 0; for (int k = 0; k < QM_NX; ++k) { const double d = x0[k] - x[0][k]; d2 += d * d; } base.dynSSE += d2; }
  base.merit = base.cost; R.baseline = base;
  if (dbg) dbg->g = G;
  const auto tq1 = std::chrono::steady_clock::now();
  if (!riccatiSolve(R, x0, x, st[ST_RICCATI_STRICT] != 0.0)) return;
  const double armijo = R.armijo;
  const auto tq2 = std::chrono::steady_clock::now();
  // ---- 2. slack / dual directions, fraction to the boundary ----
  R.dslack.assign(N, Vec(QM_NH, 0.0)); R.ddual.assign(N, Vec(QM_NH, 0.0));
  for (int i = 0; i < N; ++i) {
    if (R.grid[i].ev == QM_EV_PRE) continue;
    const IneqLin& q = G[i];
    for (int r = 0; r < QM_NH; ++r) if (q.on[r]) {
      double ds = q.h[r] - R.slack[i][r];
      for (int a = 0; a < QM_NX; ++a) ds += q.Hx(r, a) * R.dx[i][a];
      for (int a = 0; a < QM_NU; ++a) ds += q.Hu(r, a) * R.du[i][a];
      R.dslack[i][r] = ds;
      R.ddual[i][r] = -(R.dual[i][r] * ds + (R.slack[i][r] * R.dual[i][r] - mu)) / R.slack[i][r];
    }
  }
  R.alphaPrimalMax = fractionToBoundary(R.slack, R.dslack, G, st[ST_IPM_FTB_MARGIN]);
  R.alphaDualMax = fractionToBoundary(R.dual, R.ddual, G, st[ST_IPM_FTB_MARGIN]);
  // ---- 3. filter line search from the primal step limit ----
  const double gMax = st[ST_IPM_G_MAX], gMin = st[ST_IPM_G_MIN], gammaC = 1e-6, armijoFactor = 1e-4, alphaDecay = 0.5, alphaMin = 1e-4;
  const double theta0 = std::sqrt(base.dynSSE + base.eqSSE);
  const double duNorm = trajectoryNorm(R.du), dxNorm = trajectoryNorm(R.dx);
  double alpha = R.alphaPrimalMax; bool accepted = false; std::vector<Vec> xn(N + 1), un(N), sn(N); Performance pn; R.lsTrials = 0;
  do {
    for (int i = 0; i <= N; ++i) { xn[i] = x[i]; for (int k = 0; k < QM_NX; ++k) xn[i][k] += alpha * R.dx[i][k]; }
    for (int i = 0; i < N; ++i) { un[i] = u[i]; sn[i] = R.slack[i]; if (R.grid[i].ev != QM_EV_PRE) { for (int k = 0; k < QM_NU; ++k) un[i][k] += alpha * R.du[i][k]; for (int r = 0; r < QM_NH; ++r) sn[i][r] += alpha * R.dslack[i][r]; } }
    pn = computePerformanceIpm(P, R.grid, x0, xn, un, sn, mu); ++R.lsTrials;
    const double theta = std::sqrt(pn.dynSSE + pn.eqSSE);
    if (theta > gMax) accepted = theta < (1.0 - gammaC) * theta0;
    else if (theta < gMin && theta0 < gMin && alpha * armijo < 0.0) accepted = pn.merit < base.merit + armijoFactor * alpha * armijo;
    else accepted = pn.merit < (base.merit - gammaC * theta0) || theta < (1.0 - gammaC) * theta0;
    if (accepted) break;
    alpha *= alphaDecay;
    if (alpha * duNorm < st[ST_IPM_DELTA_TOL] && alpha * dxNorm < st[ST_IPM_DELTA_TOL]) break;
  } while (alpha >= alphaMin);
  R.alphaDual = 0.0;
  if (accepted) {
    x = xn; u = un; R.slack = sn; R.alpha = alpha; R.after = pn;
    R.alphaDual = (st[ST_IPM_PRIMAL_FOR_DUAL] != 0.0) ? std::min(alpha, R.alphaDualMax) : R.alphaDualMax;
    for (int i = 0; i < N; ++i) for (int r = 0; r < QM_NH; ++r) R.dual[i][r] += R.alphaDual * R.ddual[i][r];
  } else { R.alpha = 0.0; R.after = base; }
  // ---- 4. barrier parameter ----
  { const double thetaAfter = std::sqrt(R.after.dynSSE + R.after.eqSSE);
    if (std::fabs(base.merit - R.after.merit) < st[ST_IPM_RED_COST_TOL] && thetaAfter < st[ST_IPM_RED_CON_TOL])
      R.barrier = std::max(st[ST_IPM_MU_TARGET], std::min(st[ST_IPM_MU_LINEAR] * mu, std::pow(mu, st[ST_IPM_MU_POWER]))); }
  // ---- primal solution as the SQP's ----
  R.x = x; R.u.assign(N + 1, Vec(QM_NU, 0.0));
  for (int i = 0; i < N; ++i) { if (R.grid[i].ev == QM_EV_PRE && i > 0) R.u[i] = R.u[i - 1]; else R.u[i] = u[i]; }
  R.u[N] = R.u[N - 1];
  R.status = 0;
  const auto tq3 = std::chrono::steady_clock::now();
  auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  R.phaseMs[0] = ms(tq0, tq1); R.phaseMs[1] = ms(tq1, tq2); R.phaseMs[2] = ms(tq2, tq3);
}
