// adaptors/compile_check.cpp — translation unit for the syntax / type check of the adaptors against adaptors/stubs (tests/test_adaptors.py)
#define QMHIP_ADAPTOR_STUBS 1
#include "QmhipController.h"
int qmhip_adaptors_compile_check() {
  qm::QmhipSolver::Sizes s{160, 2, 64};
  ocs2::PrimalSolution p; ocs2::ModeSchedule ms;
  double t[2] = {0, 1}, x[60] = {0}, u[60] = {0}; int32_t ev[2] = {0, 0};
  qm::toPrimalSolution(2, t, ev, x, u, ms, p);
  dynamic_reconfigure::Config cfg; cfg.doubles.push_back({"kp_swing", 350.0});
  int (qm::QmhipWbc::*apply)(const dynamic_reconfigure::Config&) = &qm::QmhipWbc::applyReconfigure; (void)apply;
  return (int)p.timeTrajectory_.size() + s.maxNodes + (int)cfg.doubles.size();
}
