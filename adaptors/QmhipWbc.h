// adaptors/QmhipWbc.h — qm::WbcBase-shaped whole-body controller backed by libqmhip.
//
// Seam: qm_wbc/include/qm_wbc/WbcBase.h:28-34 — the constructor the controller calls in setupWbc (QMController.cpp:272-276, 410-414), `update()` called on
// every control tick (QMController.cpp:145-147) and `loadTasksSetting()`.  `update` returns [v̇(24); F(12); τ(18)]; the controller takes tail(18).
// The joint-acceleration state inputLast_ (WbcBase.cpp:212-213) lives in the device context.  `variant` selects the hierarchy:
//   0 = HierarchicalWbc (HierarchicalWbc.cpp:18-44, incl. the time < 10 arm-joint branch), 1 = HierarchicalMpcWbc (HierarchicalMpcWbc.cpp:18-34).
//
// dynamic_reconfigure.  The reference is tuned through rqt_reconfigure on `<controller>/wbc`: WbcBase's constructor starts a
// dynamic_reconfigure::Server<qm_wbc::WbcWeightConfig> there (WbcBase.cpp:60-65) whose callback writes the 32 gains into WbcBase's members (WbcBase.cpp:69-116).
// Server, callback and members are PRIVATE (WbcBase.h:60-62), so a subclass can neither replace the callback nor read the gains — but every
// dynamic_reconfigure server also publishes each accepted configuration on the latched topic `<ns>/parameter_updates` (dynamic_reconfigure/Config: name / value
// pairs) [upstream dynamic_reconfigure::Server::updateConfigInternal].  QmhipWbc subscribes to `<controller>/wbc/parameter_updates` and forwards every gain the
// reference's callback reads to the device (qmhip_wbc_gain_index -> qmhip_set_setting on the WBC context; names it does not read — d_ee_*, da_ee_* — are skipped, as
// in the reference).  The topic is latched: the configuration the server applied in setCallback (the cfg defaults or the parameter server's values) arrives right
// after the subscription, so the device starts from the same gains as WbcBase's members; rqt_reconfigure keeps working unchanged.  The subscriber callback runs on
// the node's spinner thread; qmhip_set_setting serialises with qmhip_wbc_step on the context's lock (include/qmhip.h "Threads").
#pragma once
#include <atomic>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>

#include "qmhip.h"
#ifdef QMHIP_ADAPTOR_STUBS
#include "stubs/reference_stubs.h"
#else
#include <dynamic_reconfigure/Config.h>
#include <qm_wbc/WbcBase.h>
#include <ros/callback_queue.h>
#include <ros/ros.h>
#endif

namespace qm {

class QmhipWbc : public WbcBase {
 public:
  QmhipWbc(const ocs2::PinocchioInterface& pinocchioInterface, ocs2::CentroidalModelInfo info, const ocs2::PinocchioEndEffectorKinematics& eeKinematics,
           const ocs2::PinocchioEndEffectorKinematics& armEeKinematics, ros::NodeHandle& controllerNh, qmhip_ctx* ctx, int variant)
      : WbcBase(pinocchioInterface, std::move(info), eeKinematics, armEeKinematics, controllerNh), ctx_(ctx), variant_(variant) {
    if (!ctx_) throw std::invalid_argument("[QmhipWbc] null device context");
    if (qmhip_wbc_reset(ctx_) != QMHIP_OK) throw std::runtime_error(std::string("[QmhipWbc] qmhip_wbc_reset: ") + qmhip_last_error(ctx_));
    // the base class has just started its server on <controller>/wbc (WbcBase.cpp:60-65): follow its (latched) update topic
    // on a PRIVATE callback queue: this constructor runs inside the controller_manager's load-controller service callback, where ros::spinOnce() would run the global
    // queue's other callbacks re-entrantly, and a controller handle with a custom queue would never deliver to the global one at all
    gainNh_ = ros::NodeHandle(controllerNh); gainNh_.setCallbackQueue(&gainQueue_);
    gainSub_ = gainNh_.subscribe<dynamic_reconfigure::Config>("wbc/parameter_updates", 4, [this](const dynamic_reconfigure::Config::ConstPtr& msg) { applyReconfigure(*msg); });
    // Until that latched message is delivered the device runs on the gains of the settings blob (the cfg defaults) while WbcBase's members may already hold the parameter
    // server's overrides (applied in setCallback): wait — bounded in WALL time: under use_sim_time with Gazebo started paused (empty_world*.launch: paused = true) ROS time
    // stands still and ros::Duration::sleep() would not return — for the first configuration, so that the first control ticks use the server's gains.
    const ros::WallTime deadline = ros::WallTime::now() + ros::WallDuration(1.0);
    while (reconfigureCount_.load() == 0 && ros::ok() && ros::WallTime::now() < deadline) gainQueue_.callAvailable(ros::WallDuration(0.005));
    gainSpinner_.reset(new ros::AsyncSpinner(1, &gainQueue_)); gainSpinner_->start();      // later reconfigure messages: one thread on the private queue (applyReconfigure is thread-safe)
  }
  ~QmhipWbc() { if (gainSpinner_) gainSpinner_->stop(); }      // (WbcBase declares no virtual destructor, WbcBase.h:23-34: the controller holds the object through a shared_ptr made from the concrete type)

  // one configuration of the reference's server -> device gains; returns how many gains were written.  Never throws (it runs in a subscriber callback):
  // a refused value is counted in gainErrors() and the text kept in lastGainError()
  int applyReconfigure(const dynamic_reconfigure::Config& config) {
    int n = 0;
    for (const auto& d : config.doubles) {
      const int idx = qmhip_wbc_gain_index(d.name.c_str());
      if (idx < 0) continue;                                           // d_ee_x ... da_ee_x: not read by WbcBase::dynamicCallback either
      if (qmhip_set_setting(ctx_, idx, d.value) == QMHIP_OK) ++n; else { const std::string why = qmhip_last_error(ctx_); std::lock_guard<std::mutex> lk(errMutex_); lastGainError_ = why; ++gainErrors_; }
    }
    ++reconfigureCount_;
    return n;
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

  // a single gain by settings index (tests, nodes without the reconfigure server)
  void setGain(int settingsIndex /* ST_KP_SWING ... ST_KD_EE_ANG, include/qmhip_layout.h */, double value) {
    if (qmhip_set_setting(ctx_, settingsIndex, value) != QMHIP_OK) throw std::runtime_error(std::string("[QmhipWbc] qmhip_set_setting: ") + qmhip_last_error(ctx_));
  }
  const int32_t* lastQpStatus() const { return qpStatus_; }
  // (written by the subscriber callback on the node's spinner thread, read from any thread: atomics, and the text by value under a mutex)
  int reconfigureCount() const { return reconfigureCount_.load(); }   // configurations received from <controller>/wbc/parameter_updates
  int gainErrors() const { return gainErrors_.load(); }
  std::string lastGainError() const { std::lock_guard<std::mutex> lk(errMutex_); return lastGainError_; }

 private:
  qmhip_ctx* ctx_; int variant_; int32_t qpStatus_[3] = {0, 0, 0};
  ros::CallbackQueue gainQueue_; ros::NodeHandle gainNh_; ros::Subscriber gainSub_; std::unique_ptr<ros::AsyncSpinner> gainSpinner_;      // (the queue outlives everything that refers to it)
  std::atomic<int> reconfigureCount_{0}, gainErrors_{0}; mutable std::mutex errMutex_; std::string lastGainError_;
};

}  // namespace qm
