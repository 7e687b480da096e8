// adaptors/QmhipInterface.h — qm::QMInterface-shaped owner of the libqmhip context.
//
// Seam: qm_interface/include/qm_interface/QMInterface.h:31-54 (constructor from the three files, setupOptimalControlProblem, the getters the controller
// uses).  The adaptor DERIVES from the reference class, so every getter qm_controllers calls (getPinocchioInterface, getCentroidalModelInfo,
// getSwitchedModelReferenceManagerPtr, getInitializer, getRollout, mpcSettings, sqpSettings, ...) keeps working unchanged, and additionally owns the
// device context built from the SAME three files by libqmhip's own ingestion (qmhip_create, include/qmhip.h).  What no longer runs on the host is the
// CppAD model generation inside setupOptimalControlProblem (QMInterface.cpp:93-131): the HIP kernels carry hand-derived derivatives, so the adaptor's
// setupOptimalControlProblem only builds the reference manager / initializer parts the controller still needs (see below).
//
// Compiles inside the reference's catkin workspace (needs the OCS2 / Pinocchio headers qm_interface already depends on) and links libqmhip.so.
// In this repository it is syntax-checked against adaptors/stubs (tests/test_adaptors.py).
#pragma once
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "qmhip.h"
#ifdef QMHIP_ADAPTOR_STUBS
#include "stubs/reference_stubs.h"
#else
#include <qm_interface/QMInterface.h>
#endif

namespace qm {

class QmhipInterface : public QMInterface {
 public:
  struct DeviceOptions {
    int device = 0;          // HIP device ordinal
    int maxBatch = 1;        // the ros_control plugin drives one robot; batch studies pass more
    int maxNodes = 160;      // timeHorizon / sqp.dt + 2 nodes per gait event inside the horizon (K0 reports status -1 when it does not fit)
    int maxRefKnots = 2;     // targetPoseToTargetTrajectories publishes 2 knots (QmTargetTrajectoriesPublisher_node.cpp:44-68)
    int maxEvents = 64;      // events of getModeSchedule(t − T, t + 2T) of the busiest gait
    bool robustTimeGrid = false;   // false: [upstream] timeDiscretizationWithEvents' dt_min (node schedules bit-equal to OCS2's; a node within 1e-6 s in front of a gait event is kept and
                                   // its negative-duration stage solved with zeroed pivots, status warning QM_MPC_WARN_PIVOT).  true: ST_GRID_DT_MIN = QM_GRID_DT_MIN_ROBUST — such a node
                                   // is merged into the event node (no warning, one node less than upstream on those calls)
  };

  QmhipInterface(const std::string& taskFile, const std::string& urdfFile, const std::string& referenceFile, DeviceOptions opt)
      : QMInterface(taskFile, urdfFile, referenceFile), opt_(opt) {
    // the reference constructor has already thrown std::invalid_argument for missing files (QMInterface.cpp:45,53,61); qmhip_create repeats the checks
    qmhip_ctx* raw = nullptr;
    const int rc = qmhip_create(urdfFile.c_str(), taskFile.c_str(), referenceFile.c_str(), opt.device, opt.maxBatch, opt.maxNodes, opt.maxRefKnots, opt.maxEvents, &raw);
    if (rc == QMHIP_ERR_FILE) throw std::invalid_argument(qmhip_last_error(nullptr));
    if (rc != QMHIP_OK) throw std::runtime_error(std::string("[QmhipInterface] qmhip_create failed: ") + qmhip_last_error(nullptr));
    ctx_.reset(raw);
    if (opt.robustTimeGrid && qmhip_set_setting(raw, ST_GRID_DT_MIN, QM_GRID_DT_MIN_ROBUST) != QMHIP_OK) throw std::runtime_error(std::string("[QmhipInterface] qmhip_set_setting: ") + qmhip_last_error(raw));
    // the control thread's own context (include/qmhip.h "Threads"): WbcBase::update runs on the ros_control thread while mpcThread_ is inside MPC_BASE::run
    // (QMController.cpp:128-147 beside :315-333)
    qmhip_ctx* rawWbc = nullptr;
    if (qmhip_create_wbc_context(raw, opt.maxBatch, &rawWbc) != QMHIP_OK) throw std::runtime_error(std::string("[QmhipInterface] qmhip_create_wbc_context failed: ") + qmhip_last_error(nullptr));
    wbcCtx_.reset(rawWbc);
    modelBlob_.resize(MB_SIZE); settingsBlob_.resize(ST_SIZE);
    qmhip_export_blobs(ctx_.get(), modelBlob_.data(), settingsBlob_.data());
  }
  QmhipInterface(const std::string& taskFile, const std::string& urdfFile, const std::string& referenceFile)
      : QmhipInterface(taskFile, urdfFile, referenceFile, DeviceOptions()) {}
  ~QmhipInterface() override = default;

  // Same call the controller makes (QMController.cpp:336-340).  The base implementation builds the Pinocchio interface, the centroidal model info, the
  // reference manager (gait schedule + swing planner), the initializer AND the CppAD-generated OCP terms.  Everything but the last is still consumed by
  // qm_controllers (state estimate, visualizer, WbcBase constructor arguments, GaitReceiver), so the base is called as is; a maintainer who wants to skip
  // the code generation overrides the protected setupModel / term factories instead.  Afterwards the two ingestions are cross-checked.
  void setupOptimalControlProblem(const std::string& taskFile, const std::string& urdfFile, const std::string& referenceFile, bool verbose) override {
    QMInterface::setupOptimalControlProblem(taskFile, urdfFile, referenceFile, verbose);
    const auto& info = getCentroidalModelInfo();
    if (std::abs(info.robotMass - modelBlob_[MB_ROBOTMASS]) > 1e-9 * info.robotMass)
      throw std::runtime_error("[QmhipInterface] robot mass of the Pinocchio model and of the device model differ");
    const auto& x0 = getInitialState();
    for (int i = 0; i < QM_NX; ++i)
      if (std::abs(x0(i) - settingsBlob_[ST_XINIT + i]) > 1e-12) throw std::runtime_error("[QmhipInterface] initialState of task.info parsed differently by the two ingestions");
  }

  qmhip_ctx* hipContext() const { return ctx_.get(); }        // MPC side: used by mpcThread_ (and by the thread that calls MPC_BASE::reset)
  qmhip_ctx* wbcContext() const { return wbcCtx_.get(); }     // control-tick side: used by the ros_control thread only
  const DeviceOptions& deviceOptions() const { return opt_; }
  const std::vector<double>& modelBlob() const { return modelBlob_; }        // include/qmhip_layout.h: MB_*
  const std::vector<double>& settingsBlob() const { return settingsBlob_; }  // include/qmhip_layout.h: ST_*

 private:
  struct CtxDeleter { void operator()(qmhip_ctx* c) const { qmhip_destroy(c); } };
  DeviceOptions opt_;
  std::unique_ptr<qmhip_ctx, CtxDeleter> ctx_, wbcCtx_;
  std::vector<double> modelBlob_, settingsBlob_;
};

}  // namespace qm
