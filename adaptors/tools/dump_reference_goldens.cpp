// adaptors/tools/dump_reference_goldens.cpp — REFERENCE-SIDE golden generator (the only route by which this repository's parity can be pinned to the
// reference itself).  A maintainer compiles it INSIDE the reference's catkin workspace, next to qm_controllers (it needs OCS2, Pinocchio, ROS, qm_interface, qm_wbc — none of
// which exist in this repository's build container, where the file is only parsed and type-checked against adaptors/stubs: tests/test_adaptors.py):
//
//   add_executable(dump_reference_goldens /path/to/repo/adaptors/tools/dump_reference_goldens.cpp)
//   target_link_libraries(dump_reference_goldens ${catkin_LIBRARIES})            # in qm_controllers/CMakeLists.txt
//   rosrun qm_controllers dump_reference_goldens <task.info> <robot.urdf> <reference.info> /path/to/repo/adaptors/tools/reference_cases.txt reference_goldens.txt
//   python /path/to/repo/tools/import_reference_goldens.py reference_goldens.txt          # -> tests/golden_ref/*.npz;  then: pytest tests/test_reference_goldens.py
//
// What it runs is the reference's own objects, constructed the way the controller constructs them:
//   qm::QMInterface + setupOptimalControlProblem           (QMController::setupInterface, qm_controllers/src/QMController.cpp:336-340)
//   ocs2::SqpMpc(mpcSettings, sqpSettings, problem, initializer)   (QMController::setupMpc, QMController.cpp:286-288) — one per case because the case sets the horizon
//   the reference manager of the interface                  (QMController.cpp:296-303, without the ROS subscribers: targets and gait come from the case file)
//   qm::HierarchicalWbc / qm::HierarchicalMpcWbc + loadTasksSetting   (QMController::setupWbc, QMController.cpp:272-276, 410-414)
// and, per MPC case, the sequence of QMController::starting / update (QMController.cpp:98-175): target trajectories -> MPC_BASE::run(t0, x0) (the first call:
// cold start, `sqp.sqpIteration 1`) -> primal solution; policy at t0; WbcBase::update on the measured state built from x0 (zero velocities — the benchmark's step,
// SURVEY.md §8(d)).  The mode schedule of a case is installed by overwriting the interface's GaitSchedule with (case schedule, STANCE template): getModeSchedule then
// returns the case's events inside [t0 − T, tf + T] unchanged ([upstream ocs2_legged_robot GaitSchedule::getModeSchedule]).
// Everything marked [upstream] is an OCS2 API recalled from the headers named beside it; the reference pins no OCS2 revision (README.md:35).
#ifdef QMHIP_ADAPTOR_STUBS
#include "../stubs/reference_stubs.h"
#else
#include <pinocchio/fwd.hpp>  // forward declarations must be included first.

#include <ocs2_centroidal_model/CentroidalModelPinocchioMapping.h>
#include <ocs2_centroidal_model/CentroidalModelRbdConversions.h>
#include <ocs2_legged_robot/gait/GaitSchedule.h>
#include <ocs2_legged_robot/gait/ModeSequenceTemplate.h>
#include <ocs2_pinocchio_interface/PinocchioEndEffectorKinematics.h>
#include <ocs2_sqp/SqpMpc.h>
#include <qm_interface/QMInterface.h>
#include <qm_wbc/HierarchicalMpcWbc.h>
#include <qm_wbc/HierarchicalWbc.h>
#include <ros/ros.h>
#endif

#include <cstdio>
#include <fstream>
#include <iostream>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace {

using ocs2::scalar_t;
using ocs2::vector_t;

struct MpcCase {
  std::string name; int instance = 0, intervals = 0; scalar_t horizon = 0, t0 = 0, period = 0, time = 0;
  vector_t x0; ocs2::scalar_array_t refT; ocs2::vector_array_t refX; ocs2::scalar_array_t events; ocs2::size_array_t modes;
};
struct WbcCase { std::string name; int variant = 0; size_t mode = 0; scalar_t period = 0, time = 0; vector_t xDes, uDes, rbd, inputLast; };

vector_t readVector(std::istream& in, long n) { vector_t v(n); for (long i = 0; i < n; ++i) { scalar_t s; if (!(in >> s)) throw std::runtime_error("case file: number expected"); v(i) = s; } return v; }
void expect(std::istream& in, const char* word) { std::string w; if (!(in >> w) || w != word) throw std::runtime_error(std::string("case file: expected '") + word + "', got '" + w + "'"); }

// adaptors/tools/reference_cases.txt (tools/export_reference_cases.py)
void readCases(const std::string& file, std::vector<MpcCase>& mpc, std::vector<WbcCase>& wbc) {
  std::ifstream in(file); if (!in) throw std::runtime_error("cannot open " + file);
  expect(in, "QM_REFERENCE_CASES"); int version; in >> version; if (version != 1) throw std::runtime_error("case file: unknown version");
  std::string w;
  while (in >> w) {
    if (w == "END") return;
    if (w == "MPC") {
      MpcCase c; in >> c.name >> c.instance; expect(in, "intervals"); in >> c.intervals; expect(in, "horizon"); in >> c.horizon; expect(in, "t0"); in >> c.t0;
      expect(in, "period"); in >> c.period; expect(in, "time"); in >> c.time;
      expect(in, "X0"); c.x0 = readVector(in, 30);
      expect(in, "TARGET"); int n; in >> n;
      for (int k = 0; k < n; ++k) { scalar_t t; in >> t; c.refT.push_back(t); c.refX.push_back(readVector(in, 37)); }
      expect(in, "SCHEDULE"); int m; in >> m;
      for (int k = 0; k < m; ++k) { scalar_t t; in >> t; c.events.push_back(t); }
      for (int k = 0; k <= m; ++k) { size_t md; in >> md; c.modes.push_back(md); }
      mpc.push_back(c);
    } else if (w == "WBC") {
      WbcCase c; in >> c.name; expect(in, "variant"); in >> c.variant; expect(in, "mode"); in >> c.mode; expect(in, "period"); in >> c.period; expect(in, "time"); in >> c.time;
      expect(in, "XDES"); c.xDes = readVector(in, 30); expect(in, "UDES"); c.uDes = readVector(in, 30); expect(in, "RBD"); c.rbd = readVector(in, 55); expect(in, "INPUTLAST"); c.inputLast = readVector(in, 30);
      wbc.push_back(c);
    } else throw std::runtime_error("case file: unknown record '" + w + "'");
  }
  throw std::runtime_error("case file: END missing");
}

void put(std::FILE* f, const vector_t& v) { for (long i = 0; i < v.size(); ++i) std::fprintf(f, " %.17g", v(i)); }

// the benchmark's synthetic measured state: generalized coordinates of x0, zero velocities, in the estimator's layout
// (qm_estimation/src/StateEstimateBase.cpp:41-103: [zyx(3) pos(3) joints(18) | angular vel(3) linear vel(3) joint vel(18) | EE pos(3) EE quat xyzw(4)]).
// The EE entries are not read by WbcBase::update (WbcBase.cpp:150-191 uses head(48)); they are left zero.
vector_t measuredStateFrom(const vector_t& x0) {
  vector_t rbd(55);
  for (int i = 0; i < 3; ++i) { rbd(i) = x0(9 + i); rbd(3 + i) = x0(6 + i); }
  for (int j = 0; j < 18; ++j) rbd(6 + j) = x0(12 + j);
  return rbd;
}

}  // namespace

int main(int argc, char** argv) {
  if (argc < 6) { std::cerr << "usage: dump_reference_goldens task.info robot.urdf reference.info reference_cases.txt out.txt\n"; return 2; }
  ros::init(argc, argv, "dump_reference_goldens");      // WbcBase's constructor starts a dynamic_reconfigure server (WbcBase.cpp:60-65): a node handle is needed
  ros::NodeHandle nh("~");
  const std::string taskFile = argv[1], urdfFile = argv[2], referenceFile = argv[3];
  std::vector<MpcCase> mpcCases; std::vector<WbcCase> wbcCases;
  readCases(argv[4], mpcCases, wbcCases);
  std::FILE* out = std::fopen(argv[5], "w"); if (!out) { std::cerr << "cannot write " << argv[5] << "\n"; return 1; }
  std::fprintf(out, "QM_REFERENCE_GOLDENS 1 source qm_control\n");

  // QMController::setupInterface (QMController.cpp:336-340)
  qm::QMInterface interface(taskFile, urdfFile, referenceFile);
  interface.setupOptimalControlProblem(taskFile, urdfFile, referenceFile, false);
  const auto& info = interface.getCentroidalModelInfo();
  // QMController::init (QMController.cpp:50-57): end-effector kinematics of the feet and of the arm
  ocs2::CentroidalModelPinocchioMapping pinocchioMapping(info);
  ocs2::PinocchioEndEffectorKinematics eeKinematics(interface.getPinocchioInterface(), pinocchioMapping, interface.modelSettings().contactNames3DoF);
  const std::vector<std::string> eeName{interface.modelSettings().info.eeFrame};
  ocs2::PinocchioEndEffectorKinematics armEeKinematics(interface.getPinocchioInterface(), pinocchioMapping, eeName);
  // QMController::setupWbc (QMController.cpp:272-276) and QMMpcController::setupWbc (QMController.cpp:410-414)
  qm::HierarchicalWbc wbc(interface.getPinocchioInterface(), info, eeKinematics, armEeKinematics, nh);
  wbc.loadTasksSetting(taskFile, false);
  ros::NodeHandle nhMpcWbc(nh, "mpc_variant");          // a second server namespace: two WbcBase objects in one process
  qm::HierarchicalMpcWbc mpcWbc(interface.getPinocchioInterface(), info, eeKinematics, armEeKinematics, nhMpcWbc);
  mpcWbc.loadTasksSetting(taskFile, false);

  for (const MpcCase& c : mpcCases) {
    // the case's horizon: MPC_BASE::run solves [t, t + mpc.timeHorizon] ([upstream ocs2_mpc MPC_BASE::run]); C1 / C2 / C5 use 0.3 / 1.5 / 2.25 s, task.info:140 ships 1.0
    ocs2::mpc::Settings mpcSettings = interface.mpcSettings();
    mpcSettings.timeHorizon_ = c.horizon;
    mpcSettings.coldStart_ = false;                       // task.info:142; irrelevant for the first call
    // QMController::setupMpc (QMController.cpp:286-288)
    ocs2::SqpMpc mpc(mpcSettings, interface.sqpSettings(), interface.getOptimalControlProblem(), interface.getInitializer());
    mpc.getSolverPtr()->setReferenceManager(interface.getReferenceManagerPtr());          // QMController.cpp:303 without the ROS wrapper: targets / gait come from the case
    // mode schedule of the case -> the interface's GaitSchedule (what GaitReceiver / the initial schedule of reference.info:28-52 would have produced).
    // [upstream] GaitSchedule(ModeSchedule initModeSchedule, ModeSequenceTemplate initModeSequenceTemplate, scalar_t phaseTransitionStanceTime)
    {
      ocs2::ModeSchedule schedule; schedule.eventTimes = c.events; schedule.modeSequence = c.modes;
      ocs2::legged_robot::ModeSequenceTemplate stanceTemplate({0.0, 0.5}, {size_t(15)});      // STANCE, like reference.info:41-52
      *interface.getSwitchedModelReferenceManagerPtr()->getGaitSchedule() = ocs2::legged_robot::GaitSchedule(schedule, stanceTemplate, 0.1 /* task.info:11 */);
    }
    // QMController::starting (QMController.cpp:105-116): target trajectories of 37-dim states (30 + EE position + EE quaternion xyzw), zero inputs
    ocs2::vector_array_t inputs(c.refT.size(), vector_t(30)); for (auto& u : inputs) for (int i = 0; i < 30; ++i) u(i) = 0.0;
    interface.getReferenceManagerPtr()->setTargetTrajectories(ocs2::TargetTrajectories(c.refT, c.refX, inputs));
    // mpcMrtInterface_->advanceMpc() -> MPC_BASE::run(currentObservation.time, currentObservation.state) (QMController.cpp:117-121, 315-323): ONE call = one SQP iteration
    mpc.reset();
    if (!mpc.run(c.t0, c.x0)) throw std::runtime_error("MPC_BASE::run returned false for case " + c.name);
    const scalar_t finalTime = c.t0 + c.horizon;
    const ocs2::PrimalSolution primal = mpc.getSolverPtr()->primalSolution(finalTime);      // [upstream SolverBase::primalSolution]
    const ocs2::PerformanceIndex& perf = mpc.getSolverPtr()->getPerformanceIndeces();
    const size_t n = primal.timeTrajectory_.size();
    std::fprintf(out, "MPC %s %d nodes %zu\n", c.name.c_str(), c.instance, n);
    std::fprintf(out, "POSTEVENT %zu", primal.postEventIndices_.size()); for (size_t i : primal.postEventIndices_) std::fprintf(out, " %zu", i); std::fprintf(out, "\n");
    for (size_t i = 0; i < n; ++i) { std::fprintf(out, "%.17g", primal.timeTrajectory_[i]); put(out, primal.stateTrajectory_[i]); put(out, primal.inputTrajectory_[i]); std::fprintf(out, "\n"); }
    std::fprintf(out, "PERF %.17g %.17g %.17g %.17g\n", perf.merit, perf.cost, perf.dynamicsViolationSSE, perf.equalityConstraintsSSE);
    // MPC_MRT_Interface::evaluatePolicy(t0, ...) (QMController.cpp:139-142): the feed-forward controller + linear interpolation of the state trajectory
    // [upstream LinearInterpolation::interpolate on PrimalSolution, FeedforwardController::computeInput]; mode = modeSchedule.modeAtTime(t0)
    const vector_t xDes = ocs2::LinearInterpolation::interpolate(c.t0, primal.timeTrajectory_, primal.stateTrajectory_);
    const vector_t uDes = primal.controllerPtr_->computeInput(c.t0, xDes);
    const size_t plannedMode = primal.modeSchedule_.modeAtTime(c.t0);
    std::fprintf(out, "POLICY %zu", plannedMode); put(out, xDes); put(out, uDes); std::fprintf(out, "\n");
    // WbcBase::update on the synthetic measured state (QMController.cpp:145-147); a fresh object per case so that inputLast_ starts at zero (WbcBase.cpp:41)
    {
      ros::NodeHandle nhCase(nh, "case_" + c.name + "_" + std::to_string(c.instance));
      qm::HierarchicalWbc caseWbc(interface.getPinocchioInterface(), info, eeKinematics, armEeKinematics, nhCase);
      caseWbc.loadTasksSetting(taskFile, false);
      const vector_t x = caseWbc.update(xDes, uDes, measuredStateFrom(c.x0), plannedMode, c.period, c.time);
      std::fprintf(out, "STEPWBC"); put(out, x); std::fprintf(out, "\n");
    }
  }
  for (const WbcCase& c : wbcCases) {
    qm::WbcBase& w = c.variant ? static_cast<qm::WbcBase&>(mpcWbc) : static_cast<qm::WbcBase&>(wbc);
    w.update(c.xDes, c.inputLast, c.rbd, c.mode, c.period, c.time);                       // primes inputLast_ (WbcBase.cpp:212-213) exactly as the tests of this repository do
    const vector_t x = w.update(c.xDes, c.uDes, c.rbd, c.mode, c.period, c.time);
    std::fprintf(out, "WBC %s variant %d\nOUT", c.name.c_str(), c.variant); put(out, x); std::fprintf(out, "\n");
  }
  std::fprintf(out, "END\n");
  std::fclose(out);
  std::cout << "wrote " << argv[5] << ": " << mpcCases.size() << " MPC cases, " << wbcCases.size() << " WBC cases\n";
  return 0;
}
