// adaptors/QmhipController.h — how the three adaptors are installed: QMController's setup functions are protected virtuals
// (qm_controllers/include/qm_controllers/QMController.h:50-58), so a derived plugin swaps the implementation and nothing else in qm_controllers changes.
// Register it next to the existing ones:  PLUGINLIB_EXPORT_CLASS(qm::QmhipController, controller_interface::ControllerBase)   (QMController.cpp:447-448)
#pragma once
#include <memory>
#include <string>

#include "QmhipInterface.h"
#include "QmhipMpc.h"
#include "QmhipWbc.h"
#ifndef QMHIP_ADAPTOR_STUBS
#include <ocs2_legged_robot_ros/gait/GaitReceiver.h>
#include <ocs2_ros_interfaces/synchronized_module/RosReferenceManager.h>
#include <qm_controllers/QMController.h>
#endif

namespace qm {

class QmhipController : public QMController {
 protected:
  // QMController.cpp:336-340
  void setupInterface(const std::string& taskFile, const std::string& urdfFile, const std::string& referenceFile, bool verbose) override {
    auto itf = std::make_shared<QmhipInterface>(taskFile, urdfFile, referenceFile);
    itf->setupOptimalControlProblem(taskFile, urdfFile, referenceFile, verbose);
    hip_ = itf; qmInterface_ = itf;
  }
  // QMController.cpp:286-306, with the device MPC in the MPC_BASE slot; the gait receiver and the ROS reference manager are attached exactly as before
  void setupMpc(ros::NodeHandle& controllerNh) override {
    const auto& o = hip_->deviceOptions();
    mpc_ = std::make_shared<QmhipMpc>(qmInterface_->mpcSettings(), hip_->hipContext(), QmhipSolver::Sizes{o.maxNodes, o.maxRefKnots, o.maxEvents}, &qmInterface_->getOptimalControlProblem());
    rbdConversions_ = std::make_shared<ocs2::CentroidalModelRbdConversions>(qmInterface_->getPinocchioInterface(), qmInterface_->getCentroidalModelInfo());
    const std::string robotName = "qm", gaitName = "legged_robot";
    ros::NodeHandle nh;
    auto gaitReceiverPtr = std::make_shared<ocs2::legged_robot::GaitReceiver>(nh, qmInterface_->getSwitchedModelReferenceManagerPtr()->getGaitSchedule(), gaitName);
    auto rosReferenceManagerPtr = std::make_shared<ocs2::RosReferenceManager>(robotName, qmInterface_->getReferenceManagerPtr());
    rosReferenceManagerPtr->subscribe(nh);
    mpc_->getSolverPtr()->addSynchronizedModule(gaitReceiverPtr);
    mpc_->getSolverPtr()->setReferenceManager(rosReferenceManagerPtr);
    observationPublisher_ = nh.advertise<ocs2_msgs::mpc_observation>(robotName + "_mpc_observation", 1);
    eeStatePublisher_ = nh.advertise<qm_msgs::ee_state>(robotName + "_mpc_observation_ee_state", 1);
    (void)controllerNh;
  }
  // QMController.cpp:272-276 (variant 0; a QMMpcController-style plugin passes 1, QMController.cpp:410-414)
  void setupWbc(ros::NodeHandle& controllerNh, const std::string& taskFile) override {
    wbc_ = std::make_shared<QmhipWbc>(qmInterface_->getPinocchioInterface(), qmInterface_->getCentroidalModelInfo(), *eeKinematicsPtr_, *armEeKinematicsPtr_, controllerNh,
                                      hip_->wbcContext(), /*variant*/ 0);      // its own context: a control tick never queues behind the MPC solve in flight
    wbc_->loadTasksSetting(taskFile, true);
  }

 private:
  std::shared_ptr<QmhipInterface> hip_;
};

}  // namespace qm
