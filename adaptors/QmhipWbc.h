// adaptors/QmhipWbc.h — qm::WbcBase-shaped whole-body controller backed by libqmhip.
//
// Seam: qm_wbc/include/qm_wbc/WbcBase.h:28-34 — the constructor the controller calls in setupWbc (QMController.cpp:272-276, 410-414), `update()` called on
// every control tick (QMController.cpp:145-147) and `loadTasksSetting()`.  `update` returns [v̇(24); F(12); τ(18)]; the controller takes tail(18).
// The joint-acceleration state inputLast_ (WbcBase.cpp:212-213) lives in the device context.  `variant` selects the hierarchy:
//   0 = HierarchicalWbc (HierarchicalWbc.cpp:18-44, incl. the time < 10 arm-joint branch), 1 = HierarchicalMpcWbc (HierarchicalMpcWbc.cpp:18-34).
#pragma once
#include <stdexcept>
#include <string>

#include "qmhip.h"
#ifdef QMHIP_ADAPTOR_STUBS
#include "stubs/reference_stubs.h"
#else
#include <qm_wbc/WbcBase.h>
#endif

namespace qm {

class QmhipWbc : public WbcBase {
 public:
  QmhipWbc(const ocs2::PinocchioInterface& pinocchioInterface, ocs2::CentroidalModelInfo info, const ocs2::PinocchioEndEffectorKinematics& eeKinematics,
           const ocs2::PinocchioEndEffectorKinematics& armEeKinematics, ros::NodeHandle& controllerNh, qmhip_ctx* ctx, int variant)
      : WbcBase(pinocchioInterface, std::move(info), eeKinematics, armEeKinematics, controllerNh), ctx_(ctx), variant_(variant) {
    if (!ctx_) throw std::invalid_argument("[QmhipWbc] null device context");
    if (qmhip_wbc_reset(ctx_) != QMHIP_OK) throw std::runtime_error(std::string("[QmhipWbc] qmhip_wbc_reset: ") + qmhip_last_error(ctx_));
  }

  ocs2::vector_t update(const ocs2::vector_t& stateDesired, const ocs2::vector_t& inputDesired, const ocs2::vector_t& rbdStateMeasured, size_t mode,
                        ocs2::scalar_t period, ocs2::scalar_t time) override {
    if (stateDesired.size() != QM_NX || inputDesired.size() != QM_NU || rbdStateMeasured.size() != QM_NRBD)
      throw std::runtime_error("[QmhipWbc] update(): expected 30 / 30 / 55 entries (StateEstimateBase.cpp:41-103 layout for the measured state)");
    const int32_t m = (int32_t)mode; const double t = time;
    ocs2::vector_t out(QM_NWBC_OUT);
    const int rc = qmhip_wbc_step(ctx_, 1, stateDesired.data(), inputDesired.data(), rbdStateMeasured.data(), &m, period, &t, variant_, out.data(), qpStatus_);
    if (rc != QMHIP_OK) throw std::runtime_error(std::string("[QmhipWbc] qmhip_wbc_step: ") + qmhip_last_error(ctx_));
    // the reference ignores qpOASES' return value (HoQp.cpp:135-150); the statuses are kept for whoever wants to look: 0 ok, 1 iteration limit
    return out;
  }

  // task.info's torqueLimitsTask / frictionConeTask blocks (WbcBase.cpp:565-595) were read by qmhip_create from the same file; nothing to load here.
  void loadTasksSetting(const std::string& /*taskFile*/, bool /*verbose*/) override {}

  // dynamic_reconfigure (WbcBase.cpp:69-116 — the base class keeps its own server; a node that wants the gains on the device forwards them here)
  void setGain(int settingsIndex /* ST_KP_SWING ... ST_KD_EE_ANG, include/qmhip_layout.h */, double value) {
    if (qmhip_set_setting(ctx_, settingsIndex, value) != QMHIP_OK) throw std::runtime_error(std::string("[QmhipWbc] qmhip_set_setting: ") + qmhip_last_error(ctx_));
  }
  const int32_t* lastQpStatus() const { return qpStatus_; }

 private:
  qmhip_ctx* ctx_; int variant_; int32_t qpStatus_[3] = {0, 0, 0};
};

}  // namespace qm
