// oracle/src/ilqr.h — TEST INFRASTRUCTURE (CPU oracle).  One iteration of a DISCRETE iLQR on the same transcription as the SQP solver:
// the alternative solver the reference loads settings for (`ddp { algorithm … }`, qm_controllers/config/task.info:33-71, read at
// qm_interface/src/QMInterface.cpp:70) but never instantiates (QMController::setupMpc installs SqpMpc, QMController.cpp:287-288).  There is therefore no
// reference behaviour to match: this restates the structure of [upstream ocs2_ddp] ILQR + LineSearchStrategy as recalled — PARITY UNPINNED —
// and exists as the checker of the product's iLQR path (same algorithm, independent code).
//
//   1. grid, modes, initial INPUTS exactly as the SQP iteration (initializer, or the previous solution's inputs when warm);
//      the nominal STATES come from a forward rollout x_{i+1} = RK2(x_i, u_i) (single shooting: no defects; event nodes are identity jumps)
//   2. LQ approximation + equality-constraint projection of every node (the SQP's setupIntermediateNode / projectNode: one shared transcription)
//   3. Riccati backward sweep -> feedback gains K_i, feed-forward kff_i of the projected input (riccatiSolve of sqp.h)
//   4. line search over the step length a = maxStepLength * 0.5^k >= minStepLength (ddp.lineSearch, task.info:63-69) on NONLINEAR rollouts with feedback:
//        x~_0 = x0,   ũ_i = K_i (x~_i − x_i) + a kff_i,   u~_i = u_i + a Pe_i + Px_i (x~_i − x_i) + Pu_i ũ_i,   x~_{i+1} = RK2(x~_i, u~_i)
//      merit = cost + rho * sqrt(equality-constraint SSE), rho = ddp.constraintPenaltyInitialValue (task.info:56); accepted when
//        merit(a) < merit(0) + 1e-4 * a * (expected decrease of the LQ model);   no step length accepted: the nominal rollout is kept (alpha = 0)
//   5. primal solution as the SQP's (inputs of PreEvent nodes copied from the previous node, last input repeated)
#pragma once
#include "sqp.h"

inline double ilqrMerit(const Problem& P, const Performance& p) { return p.cost + P.M->st[ST_DDP_PENALTY] * std::sqrt(p.eqSSE); }

inline void ilqrIteration(const Problem& P, double t0, double tf, const Vec& x0, SqpResult& R, const SqpResult* prev = nullptr) {
  const Model& M = *P.M; const double* st = M.st;
  R.grid = timeDiscretizationWithEvents(t0, tf, st[ST_SQP_DT], P.ms.ev, st[ST_GRID_DT_MIN]);
  const int N = (int)R.grid.size() - 1;
  R.mode.resize(N + 1); for (int i = 0; i <= N; ++i) R.mode[i] = P.ms.modeAt(intervalStart(R.grid[i]));
  // ---- 1. inputs as the SQP's initial guess, states by rollout ----
  std::vector<Vec> x(N + 1), u(N);
  const bool warm = prev && prev->grid.size() >= 2;
  const double tend = warm ? prev->grid.back().t : 0.0;
  auto dtOf = [&](int i) { return intervalEnd(R.grid[i + 1]) - intervalStart(R.grid[i]); };
  x[0] = x0;
  for (int i = 0; i < N; ++i) {
    if (R.grid[i].ev == QM_EV_PRE) { u[i] = Vec(QM_NU, 0.0); x[i + 1] = x[i]; continue; }
    const double time = intervalStart(R.grid[i]), nextTime = intervalEnd(R.grid[i + 1]);
    if (warm && !(time > tend || nextTime > tend)) { Vec xa, ua; int md; evaluatePolicy(*prev, P.ms, time, xa, ua, md); u[i] = ua; }
    else { bool fl[4]; modeToFlags(P.ms.modeAt(time), fl); u[i] = weightCompensatingInput(M, fl); }
    x[i + 1] = rk2Step(M, x[i], u[i], dtOf(i));
  }
  // ---- 2. LQ model + projection ----
  R.lq.assign(N, NodeLQ());
  parallelFor(N, [&](int i) {
    NodeLQ& n = R.lq[i];
    if (R.grid[i].ev == QM_EV_PRE) {
      n.event = 1; n.m = 0; n.nc = 0; n.dt = 0; n.Ap = Mat::identity(QM_NX); n.A = n.Ap; n.bp.assign(QM_NX, 0.0);
      for (int k = 0; k < QM_NX; ++k) n.bp[k] = x[i][k] - x[i + 1][k];
      n.b = n.bp; n.Qp = Mat(QM_NX, QM_NX); n.Q = n.Qp; n.qp.assign(QM_NX, 0.0); n.q = n.qp; n.cp = n.c = 0;
    } else {
      setupIntermediateNode(P, intervalStart(R.grid[i]), dtOf(i), x[i], x[i + 1], u[i], n);
      projectNode(n);
    }
  });
  { CostQuad c; terminalCost(P, intervalStart(R.grid[N]), x[N], true, c); R.terminal.Qp = c.Q; R.terminal.qp = c.q; R.terminal.cp = c.f; }
  const Performance base = computePerformance(P, R.grid, x0, x, u);
  R.baseline = base; R.baseline.merit = ilqrMerit(P, base);
  // ---- 3. Riccati ----
  if (!riccatiSolve(R, x0, x, st[ST_RICCATI_STRICT] != 0.0)) return;
  // ---- 4. line search on nonlinear rollouts with feedback ----
  const double armijoCoefficient = 1e-4, contraction = 0.5;
  std::vector<Vec> xt(N + 1), ut(N); Performance pt; bool accepted = false; double alpha = st[ST_DDP_MAX_STEP]; R.lsTrials = 0;
  while (alpha >= st[ST_DDP_MIN_STEP]) {
    xt[0] = x0;
    for (int i = 0; i < N; ++i) {
      const NodeLQ& n = R.lq[i];
      if (n.event) { ut[i] = Vec(QM_NU, 0.0); xt[i + 1] = xt[i]; continue; }
      Vec dxi = vsub(xt[i], x[i]);
      Vec uproj = vadd(matvec(n.K, dxi), vscaled(n.kff, alpha));
      Vec du = vadd(vadd(vscaled(n.Pe, alpha), matvec(n.Px, dxi)), matvec(n.Pu, uproj));
      ut[i] = vadd(u[i], du);
      xt[i + 1] = rk2Step(M, xt[i], ut[i], dtOf(i));
    }
    pt = computePerformance(P, R.grid, x0, xt, ut); ++R.lsTrials;
    if (ilqrMerit(P, pt) < R.baseline.merit + armijoCoefficient * alpha * R.armijo) { accepted = true; break; }
    alpha *= contraction;
  }
  if (accepted) { x = xt; u = ut; R.alpha = alpha; R.after = pt; R.after.merit = ilqrMerit(P, pt); } else { R.alpha = 0.0; R.after = R.baseline; }
  // ---- 5. primal solution ----
  R.x = x; R.u.assign(N + 1, Vec(QM_NU, 0.0));
  for (int i = 0; i < N; ++i) { if (R.grid[i].ev == QM_EV_PRE && i > 0) R.u[i] = R.u[i - 1]; else R.u[i] = u[i]; }
  R.u[N] = R.u[N - 1];
  R.status = 0;
}
