// adaptors/stubs/reference_stubs.h — COMPILE-CHECK STAND-INS, not OCS2 / ROS / qm_control code.
// The adaptors need the reference's headers (OCS2, Pinocchio, ROS, qm_*), none of which exist in this repository's build container.  So that the adaptor
// sources are at least parsed and type-checked here (tests/test_adaptors.py: g++ -fsyntax-only -DQMHIP_ADAPTOR_STUBS), this header declares the handful
// of names they touch with the member signatures recalled from the upstream headers named in each adaptor ([upstream], unpinned in the reference:
// README.md:35).  Bodies are empty; nothing here is linked into any product or test binary.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <memory>
#include <string>
#include <utility>
#include <vector>

namespace ros {
struct Publisher {};
struct Subscriber {};
// roscpp: wall-clock time (independent of /clock), a private callback queue and a spinner thread on it
struct WallDuration { explicit WallDuration(double s = 0) : s_(s) {} bool sleep() const { return true; } double s_; };
struct WallTime { static WallTime now() { return WallTime(); } WallTime operator+(const WallDuration& d) const { WallTime t; t.s_ = s_ + d.s_; return t; } bool operator<(const WallTime& o) const { return s_ < o.s_; } double s_ = 0; };
struct CallbackQueue { void callAvailable(WallDuration) {} };
struct AsyncSpinner { AsyncSpinner(unsigned, CallbackQueue*) {} void start() {} void stop() {} };
inline void init(int&, char**, const std::string&) {}
struct NodeHandle {
  NodeHandle() = default;
  explicit NodeHandle(const std::string&) {}
  NodeHandle(const NodeHandle&, const std::string&) {}
  void setCallbackQueue(CallbackQueue*) {}
  template <class M> Publisher advertise(const std::string&, int) { return Publisher(); }
  // roscpp: subscribe<M>(topic, queue_size, boost::function<void(const boost::shared_ptr<M const>&)>)
  template <class M, class F> Subscriber subscribe(const std::string&, int, F callback) { if (false) callback(typename M::ConstPtr()); return Subscriber(); }      // type-checks the callback against M::ConstPtr
};
inline bool ok() { return false; }                 // roscpp: ros::ok(), ros::spinOnce(), ros::Duration(s).sleep()
inline void spinOnce() {}
struct Duration { explicit Duration(double) {} bool sleep() const { return true; } };
}  // namespace ros
namespace dynamic_reconfigure {                 // dynamic_reconfigure/Config.msg: bools / ints / strs / doubles / groups; the adaptor reads `doubles`
struct DoubleParameter { std::string name; double value = 0; };
struct Config { typedef std::shared_ptr<const Config> ConstPtr; std::vector<DoubleParameter> doubles; };
}  // namespace dynamic_reconfigure
namespace ocs2_msgs { struct mpc_observation {}; }
namespace qm_msgs { struct ee_state {}; }

namespace ocs2 {
using scalar_t = double;
struct vector_t {                               // Eigen::VectorXd stand-in
  std::vector<double> v;
  vector_t() = default;
  explicit vector_t(long n) : v((size_t)n, 0.0) {}
  long size() const { return (long)v.size(); }
  double* data() { return v.data(); }
  const double* data() const { return v.data(); }
  double& operator()(long i) { return v[(size_t)i]; }
  double operator()(long i) const { return v[(size_t)i]; }
};
using scalar_array_t = std::vector<scalar_t>;
using vector_array_t = std::vector<vector_t>;
using size_array_t = std::vector<size_t>;
struct ModeSchedule { scalar_array_t eventTimes; size_array_t modeSequence; size_t modeAtTime(scalar_t) const { return modeSequence.empty() ? 0 : modeSequence.front(); } };
struct TargetTrajectories {
  TargetTrajectories() = default;
  TargetTrajectories(scalar_array_t t, vector_array_t x, vector_array_t u) : timeTrajectory(std::move(t)), stateTrajectory(std::move(x)), inputTrajectory(std::move(u)) {}
  scalar_array_t timeTrajectory; vector_array_t stateTrajectory, inputTrajectory;
};
struct ControllerBase { virtual ~ControllerBase() = default; virtual ControllerBase* clone() const = 0; virtual vector_t computeInput(scalar_t t, const vector_t& x) = 0; };
namespace LinearInterpolation { inline vector_t interpolate(scalar_t, const scalar_array_t&, const vector_array_t& v) { return v.empty() ? vector_t() : v.front(); } }
struct FeedforwardController : ControllerBase {
  FeedforwardController(scalar_array_t t, vector_array_t u) : t_(std::move(t)), u_(std::move(u)) {}
  FeedforwardController* clone() const override { return new FeedforwardController(*this); }
  vector_t computeInput(scalar_t, const vector_t&) override { return u_.empty() ? vector_t() : u_.front(); }
  scalar_array_t t_; vector_array_t u_;
};
struct PrimalSolution {
  PrimalSolution() = default;
  PrimalSolution(const PrimalSolution& o) { *this = o; }
  PrimalSolution& operator=(const PrimalSolution& o) {
    timeTrajectory_ = o.timeTrajectory_; stateTrajectory_ = o.stateTrajectory_; inputTrajectory_ = o.inputTrajectory_; postEventIndices_ = o.postEventIndices_; modeSchedule_ = o.modeSchedule_;
    controllerPtr_.reset(o.controllerPtr_ ? o.controllerPtr_->clone() : nullptr); return *this;
  }
  void clear() { timeTrajectory_.clear(); stateTrajectory_.clear(); inputTrajectory_.clear(); postEventIndices_.clear(); controllerPtr_.reset(); }
  scalar_array_t timeTrajectory_; vector_array_t stateTrajectory_, inputTrajectory_; size_array_t postEventIndices_; ModeSchedule modeSchedule_; std::unique_ptr<ControllerBase> controllerPtr_;
};
struct PerformanceIndex { scalar_t merit = 0, cost = 0, dualFeasibilitiesSSE = 0, dynamicsViolationSSE = 0, equalityConstraintsSSE = 0, equalityLagrangian = 0, inequalityLagrangian = 0; };
struct OptimalControlProblem {};
struct ScalarFunctionQuadraticApproximation {};
struct ReferenceManagerInterface {
  virtual ~ReferenceManagerInterface() = default;
  virtual const ModeSchedule& getModeSchedule() const = 0;
  virtual const TargetTrajectories& getTargetTrajectories() const = 0;
  virtual void setTargetTrajectories(TargetTrajectories) {}
};
struct SolverSynchronizedModule { virtual ~SolverSynchronizedModule() = default; };
class SolverBase {                              // ocs2_oc/oc_solver/SolverBase.h
 public:
  virtual ~SolverBase() = default;
  virtual void reset() = 0;
  void run(scalar_t initTime, const vector_t& initState, scalar_t finalTime) { runImpl(initTime, initState, finalTime); }     // upstream: pre/postSolverRun around runImpl
  void setReferenceManager(std::shared_ptr<ReferenceManagerInterface> p) { ref_ = std::move(p); }
  ReferenceManagerInterface& getReferenceManager() { return *ref_; }
  const ReferenceManagerInterface& getReferenceManager() const { return *ref_; }
  void addSynchronizedModule(std::shared_ptr<SolverSynchronizedModule> m) { modules_.push_back(std::move(m)); }
  virtual scalar_t getFinalTime() const = 0;
  PrimalSolution primalSolution(scalar_t finalTime) const { PrimalSolution p; getPrimalSolution(finalTime, &p); return p; }
  virtual void getPrimalSolution(scalar_t finalTime, PrimalSolution* primalSolutionPtr) const = 0;
  virtual size_t getNumIterations() const = 0;
  virtual const OptimalControlProblem& getOptimalControlProblem() const = 0;
  virtual const PerformanceIndex& getPerformanceIndeces() const = 0;
  virtual const std::vector<PerformanceIndex>& getIterationsLog() const = 0;
  virtual ScalarFunctionQuadraticApproximation getValueFunction(scalar_t time, const vector_t& state) const = 0;
  virtual ScalarFunctionQuadraticApproximation getHamiltonian(scalar_t time, const vector_t& state, const vector_t& input) = 0;
  virtual vector_t getStateInputEqualityConstraintLagrangian(scalar_t time, const vector_t& state) const = 0;
 private:
  virtual void runImpl(scalar_t initTime, const vector_t& initState, scalar_t finalTime) = 0;
  virtual void runImpl(scalar_t initTime, const vector_t& initState, scalar_t finalTime, const ControllerBase* externalControllerPtr) = 0;
  std::shared_ptr<ReferenceManagerInterface> ref_; std::vector<std::shared_ptr<SolverSynchronizedModule>> modules_;
};
namespace mpc { struct Settings { bool coldStart_ = false; scalar_t timeHorizon_ = 1.0; scalar_t mpcDesiredFrequency_ = 100, mrtDesiredFrequency_ = 500; }; }
class MPC_BASE {                                // ocs2_mpc/MPC_BASE.h
 public:
  explicit MPC_BASE(mpc::Settings s) : settings_(std::move(s)) {}
  virtual ~MPC_BASE() = default;
  virtual void reset() {}
  virtual bool run(scalar_t currentTime, const vector_t& currentState) { calculateController(currentTime, currentState, currentTime + settings_.timeHorizon_); return true; }
  virtual SolverBase* getSolverPtr() = 0;
  virtual const SolverBase* getSolverPtr() const = 0;
  const mpc::Settings& settings() const { return settings_; }
 protected:
  virtual void calculateController(scalar_t initTime, const vector_t& initState, scalar_t finalTime) = 0;
 private:
  mpc::Settings settings_;
};
struct PinocchioInterface {};
struct CentroidalModelInfo { scalar_t robotMass = 0; };
struct CentroidalModelPinocchioMapping { explicit CentroidalModelPinocchioMapping(const CentroidalModelInfo&) {} };
struct PinocchioEndEffectorKinematics {
  PinocchioEndEffectorKinematics() = default;
  PinocchioEndEffectorKinematics(const PinocchioInterface&, const CentroidalModelPinocchioMapping&, std::vector<std::string>) {}
};
struct Initializer {};
namespace sqp { struct Settings { int threadPriority = 0; }; }
class SqpSolverStub final : public SolverBase {     // stands for ocs2::SqpSolver behind SqpMpc::getSolverPtr()
 public:
  void reset() override {}
  scalar_t getFinalTime() const override { return 0; }
  void getPrimalSolution(scalar_t, PrimalSolution*) const override {}
  size_t getNumIterations() const override { return 0; }
  const OptimalControlProblem& getOptimalControlProblem() const override { return problem_; }
  const PerformanceIndex& getPerformanceIndeces() const override { return perf_; }
  const std::vector<PerformanceIndex>& getIterationsLog() const override { return log_; }
  ScalarFunctionQuadraticApproximation getValueFunction(scalar_t, const vector_t&) const override { return {}; }
  ScalarFunctionQuadraticApproximation getHamiltonian(scalar_t, const vector_t&, const vector_t&) override { return {}; }
  vector_t getStateInputEqualityConstraintLagrangian(scalar_t, const vector_t&) const override { return {}; }
 private:
  void runImpl(scalar_t, const vector_t&, scalar_t) override {}
  void runImpl(scalar_t, const vector_t&, scalar_t, const ControllerBase*) override {}
  OptimalControlProblem problem_; PerformanceIndex perf_; std::vector<PerformanceIndex> log_;
};
class SqpMpc final : public MPC_BASE {              // ocs2_sqp/SqpMpc.h
 public:
  SqpMpc(mpc::Settings mpcSettings, sqp::Settings, const OptimalControlProblem&, const Initializer&) : MPC_BASE(std::move(mpcSettings)) {}
  SolverBase* getSolverPtr() override { return &solver_; }
  const SolverBase* getSolverPtr() const override { return &solver_; }
 protected:
  void calculateController(scalar_t, const vector_t&, scalar_t) override {}
 private:
  SqpSolverStub solver_;
};
struct CentroidalModelRbdConversions { CentroidalModelRbdConversions(const PinocchioInterface&, const CentroidalModelInfo&) {} };
struct RosReferenceManager : ReferenceManagerInterface {
  RosReferenceManager(std::string, std::shared_ptr<ReferenceManagerInterface> p) : p_(std::move(p)) {}
  void subscribe(ros::NodeHandle&) {}
  const ModeSchedule& getModeSchedule() const override { return p_->getModeSchedule(); }
  const TargetTrajectories& getTargetTrajectories() const override { return p_->getTargetTrajectories(); }
  std::shared_ptr<ReferenceManagerInterface> p_;
};
namespace legged_robot {
struct ModeSequenceTemplate { ModeSequenceTemplate(scalar_array_t t, size_array_t m) : switchingTimes(std::move(t)), modeSequence(std::move(m)) {} scalar_array_t switchingTimes; size_array_t modeSequence; };
struct GaitSchedule {                           // ocs2_legged_robot/gait/GaitSchedule.h
  GaitSchedule() : tmpl_({0.0, 1.0}, {size_t(15)}) {}
  GaitSchedule(ModeSchedule initModeSchedule, ModeSequenceTemplate initModeSequenceTemplate, scalar_t phaseTransitionStanceTime) : ms_(std::move(initModeSchedule)), tmpl_(std::move(initModeSequenceTemplate)), t_(phaseTransitionStanceTime) {}
  ModeSchedule ms_; ModeSequenceTemplate tmpl_; scalar_t t_ = 0;
};
struct ModelSettings { std::vector<std::string> contactNames3DoF; struct { std::string eeFrame; } info; };
struct SwitchedModelReferenceManager : ReferenceManagerInterface { std::shared_ptr<GaitSchedule> getGaitSchedule() { return nullptr; } };
struct GaitReceiver : SolverSynchronizedModule { GaitReceiver(ros::NodeHandle, std::shared_ptr<GaitSchedule>, const std::string&) {} };
}  // namespace legged_robot
}  // namespace ocs2

namespace qm {
class QMInterface {                             // qm_interface/include/qm_interface/QMInterface.h:29-54
 public:
  QMInterface(const std::string&, const std::string&, const std::string&) {}
  virtual ~QMInterface() = default;
  virtual void setupOptimalControlProblem(const std::string&, const std::string&, const std::string&, bool) {}
  const ocs2::OptimalControlProblem& getOptimalControlProblem() const { return problem_; }
  const ocs2::mpc::Settings& mpcSettings() const { return mpcSettings_; }
  const ocs2::sqp::Settings& sqpSettings() { return sqpSettings_; }
  const ocs2::legged_robot::ModelSettings& modelSettings() const { return modelSettings_; }
  const ocs2::Initializer& getInitializer() const { return initializer_; }
  const ocs2::vector_t& getInitialState() const { return initialState_; }
  ocs2::PinocchioInterface& getPinocchioInterface() { return pin_; }
  const ocs2::CentroidalModelInfo& getCentroidalModelInfo() const { return info_; }
  std::shared_ptr<ocs2::legged_robot::SwitchedModelReferenceManager> getSwitchedModelReferenceManagerPtr() const { return nullptr; }
  std::shared_ptr<ocs2::ReferenceManagerInterface> getReferenceManagerPtr() const { return nullptr; }
 private:
  ocs2::OptimalControlProblem problem_; ocs2::mpc::Settings mpcSettings_; ocs2::vector_t initialState_{30}; ocs2::PinocchioInterface pin_; ocs2::CentroidalModelInfo info_;
  ocs2::sqp::Settings sqpSettings_; ocs2::legged_robot::ModelSettings modelSettings_; ocs2::Initializer initializer_;
};
class WbcBase {                                 // qm_wbc/include/qm_wbc/WbcBase.h:26-34
 public:
  WbcBase(const ocs2::PinocchioInterface&, ocs2::CentroidalModelInfo, const ocs2::PinocchioEndEffectorKinematics&, const ocs2::PinocchioEndEffectorKinematics&, ros::NodeHandle&) {}
  virtual ocs2::vector_t update(const ocs2::vector_t&, const ocs2::vector_t&, const ocs2::vector_t&, size_t, ocs2::scalar_t, ocs2::scalar_t) { return ocs2::vector_t(); }
  virtual void loadTasksSetting(const std::string&, bool) {}
};
class HierarchicalWbc : public WbcBase { public: using WbcBase::WbcBase; };        // qm_wbc/include/qm_wbc/HierarchicalWbc.h
class HierarchicalMpcWbc : public WbcBase { public: using WbcBase::WbcBase; };     // qm_wbc/include/qm_wbc/HierarchicalMpcWbc.h
class QMController {                            // qm_controllers/include/qm_controllers/QMController.h:37-95 (the members the adaptor touches)
 public:
  virtual ~QMController() = default;
 protected:
  virtual void setupInterface(const std::string&, const std::string&, const std::string&, bool) {}
  virtual void setupMpc(ros::NodeHandle&) {}
  virtual void setupWbc(ros::NodeHandle&, const std::string&) {}
  std::shared_ptr<QMInterface> qmInterface_;
  std::shared_ptr<ocs2::PinocchioEndEffectorKinematics> eeKinematicsPtr_, armEeKinematicsPtr_;
  std::shared_ptr<ocs2::CentroidalModelRbdConversions> rbdConversions_;
  std::shared_ptr<ocs2::MPC_BASE> mpc_;
  std::shared_ptr<WbcBase> wbc_;
  ros::Publisher observationPublisher_, eeStatePublisher_;
};
}  // namespace qm
