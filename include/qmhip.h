/*
 * qmhip.h — C ABI of libqmhip.so: the MI355X-native MPC + whole-body-control step of qm_control.
 *
 * The reference has no FFI; its seams are three C++ interfaces.  Each entry point below
 * names the reference interface it stands behind; INTEGRATION.md shows the thin C++ adaptors
 * (qm::QMInterface-, ocs2::MPC_BASE-, qm::WbcBase-shaped) a maintainer adds on the reference side.
 *
 * Conventions: plain pointers + sizes, no C++/torch types; f64 everywhere (ocs2::scalar_t); arrays are
 * instance-major and caller owned; pointers are HOST memory unless the name says `_dev`.
 * Return value 0 = ok, negative = error (qmhip_last_error gives the text).
 *
 * Threads.  Every entry point that takes a context locks it: calls on ONE context from several threads are safe and run one after another (a blocked
 * caller waits for the other call's host work, e.g. the launches of an MPC solve — since round 6 the SQP solve only enqueues, its line search no longer waits for the device —).  The reference runs the MPC in `mpcThread_` (advanceMpc,
 * qm_controllers/src/QMController.cpp:315-333) beside the ros_control thread's WbcBase::update (QMController.cpp:128-147); for that layout the control thread
 * gets its OWN context from qmhip_create_wbc_context: own streams, own device copies of the model, own error string — a control tick then never waits for,
 * and is never reordered by, an MPC solve in flight (tests/c_abi_threads.c).  Different contexts share nothing mutable; qmhip_last_error(NULL) is per thread.
 */
#ifndef QMHIP_H
#define QMHIP_H
#include <stdint.h>
#include "qmhip_layout.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct qmhip_ctx qmhip_ctx;

/* ---- construction: replaces qm::QMInterface(taskFile, urdfFile, referenceFile) + setupOptimalControlProblem
 *      (qm_interface/include/qm_interface/QMInterface.h:31-35, qm_interface/src/QMInterface.cpp:37-142) and the
 *      WBC constructor / loadTasksSetting (qm_wbc/include/qm_wbc/WbcBase.h:28-34).
 *      Missing files -> QMHIP_ERR_FILE, like the reference's std::invalid_argument (QMInterface.cpp:45,53,61).
 *      3 <= max_nodes <= 512 (horizon nodes incl. event-split nodes; the per-instance node list lives in LDS), else QMHIP_ERR_ARG.
 *      Device memory: ~ 35 KB per (instance, node) — the stage record (30 KB; rounds 1-4: 58.9), the kinematics record (3 KB), iterate / step / reference arrays — measured
 *      (free device memory around the call, profiles/r05_context_footprint.json): 35.8 KB per (instance, node), 4.7 GB for max_batch 1024 x max_nodes 128, 34 GB for BASELINE config 4 on
 *      one device at 8192 x 116 (its instances use <= 110 nodes) and 37.5 GB at 8192 x 128; an allocation that does not fit returns QMHIP_ERR_HIP with the runtime's message. */
int qmhip_create(const char* urdf_file, const char* task_file, const char* reference_file,
                 int device, int max_batch, int max_nodes, int max_ref_knots, int max_events, qmhip_ctx** out);
/* same, from the flat MODEL / SETTINGS blobs of qmhip_layout.h (no file I/O).  The blobs carry no size stamp: model_blob must hold MB_SIZE and settings_blob ST_SIZE doubles of THIS
 *      header's layout (a shorter array of an older layout is read past its end); the slots whose garbage would steer a kernel's control flow are range-checked (QMHIP_ERR_MODEL) */
int qmhip_create_from_blobs(const double* model_blob, const double* settings_blob,
                            int device, int max_batch, int max_nodes, int max_ref_knots, int max_events, qmhip_ctx** out);
/* WBC-only context for the control thread (see "Threads" above): the parent's model and settings VALUES (copied; later qmhip_set_setting calls go to whichever
 * context they name — WBC gains to this one), the WBC buffers for max_batch instances, the WBC stream.  It serves qmhip_wbc_step / qmhip_wbc_reset / qmhip_wbc_download /
 * qmhip_set_setting / the instrumentation calls; every MPC, front-end and plant entry point returns QMHIP_ERR_STATE on it.  Replaces nothing new in the reference:
 * it is the `wbc_` object of QMController (QMController.h:80, constructed in setupWbc, QMController.cpp:272-276) as opposed to its `mpc_`. */
int qmhip_create_wbc_context(const qmhip_ctx* parent, int max_batch, qmhip_ctx** out);
void qmhip_destroy(qmhip_ctx* ctx);
const char* qmhip_last_error(const qmhip_ctx* ctx);            /* ctx may be NULL: error of the last failed create */
/* parse only (host): what getPinocchioInterface()/getCentroidalModelInfo()/settings getters expose
 * (QMInterface.h:37-54), as blobs */
int qmhip_parse_model(const char* urdf_file, const char* task_file, const char* reference_file,
                      double* model_blob /*[MB_SIZE]*/, double* settings_blob /*[ST_SIZE]*/);
int qmhip_export_blobs(const qmhip_ctx* ctx, double* model_blob, double* settings_blob);
int qmhip_set_setting(qmhip_ctx* ctx, int settings_index, double value);   /* e.g. WBC gains: dynamic_reconfigure callback, WbcBase.cpp:69-116 */
/* settings index of a WBC gain by the name it carries in the reference's dynamic_reconfigure config (qm_wbc/cfg/wbcWigeht.cfg:7-47, assigned in
 * WbcBase::dynamicCallback, qm_wbc/src/WbcBase.cpp:69-116): "kp_swing" -> ST_KP_SWING, "baseHeightKp" -> ST_KP_BASE_H, "kp_arm_joint_3" -> ST_KP_ARM_J + 2,
 * "kd_ee_angular_y" -> ST_KD_EE_ANG + 1, ...; -1 for a name the callback does not read (d_ee_x ... da_ee_x) or an unknown one.  Host-only, no context:
 * adaptors/QmhipWbc.h feeds every entry of the server's `parameter_updates` message through it into qmhip_set_setting. */
int qmhip_wbc_gain_index(const char* reconfigure_name);

/* ---- MPC: replaces ocs2::MPC_BASE::run(t, x) on the SqpMpc the reference installs
 *      (qm_controllers/src/QMController.cpp:287-288,315-323) for B independent instances: one multiple-shooting
 *      SQP iteration (task.info:75-92; `sqp.sqpIteration` of them, shipped 1 — settable through qmhip_set_setting(ST_SQP_ITER)) from a cold start.
 *      Inputs per instance: initial time/state, target trajectory knots (37-dim, QmTargetTrajectoriesPublisher_node.cpp:44-68),
 *      contact-mode schedule (event times + mode ids, modes has n_events+1 entries).
 *      Outputs (any may be NULL): node count, node times / event tags / modes (int, bit-exact), optimal state and
 *      input trajectories (primal solution: input of event nodes copied from the previous node, last input repeated),
 *      perf[10] = baseline{merit,cost,dynSSE,eqSSE}, after-step{...}, step size alpha, armijo metric.
 *      status[b]: 0 ok; < 0 failure: -1 node buffer too small, -2 swing phase not enclosed by stance in the schedule, -3 device gait schedule over capacity,
 *      -4 Riccati failed (indefinite stage of positive duration / NaN; with ST_RICCATI_STRICT also the degenerate stage); > 0 WARNING bits on a valid solution: QM_MPC_WARN_PIVOT (1) = a stage's Huu had non-positive pivots, which were
 *      zeroed ([upstream, recalled] BLASFEO / HPIPM behaviour; qmhip_layout.h) — the stage in front of a gait event when a shooting node falls within weakEpsilon
 *      before it, i.e. about one call in 3700 for an MPC thread on continuous time (QMController.cpp:315-330); on a fixed-rate clock that shares a raster with the gait
 *      events 3 % ... 12 % of the calls (profiles/r04_fixed_rate_report.txt).  Such a solve lies within 5e-6 (per block) of the solve on the robust grid everywhere but the
 *      degenerate interval's own input, its policy at t0 within 5e-6 (measured 2.8e-6; tests/test_gpu_mpc.py); rastered controllers should set ST_GRID_DT_MIN = QM_GRID_DT_MIN_ROBUST
 *      (INTEGRATION.md section 3).  Callers treat status >= 0 as success.  -4 also reports: a non-positive pivot on a stage of POSITIVE duration, a pivot or a step that is
 *      not a finite number (e.g. a NaN in x0) — never a warning.
 *      Solver slot (qmhip_set_setting(ST_SOLVER, .)): 0 the SQP above; 1 a discrete iLQR on the `ddp` block (task.info:33-71); 2 the SAME multiple-shooting step as
 *      slot 0, run with the `ipm` block's parameters (task.info:94-125: ipm.dt / ipmIteration / deltaTol / g_max / g_min, ST_IPM_*) — kept for compatibility, NOT an
 *      interior-point method; 3 (round 5) a primal-dual INTERIOR-POINT method with HARD inequality constraints: the friction cone of every stance foot and the arm joint
 *      position / velocity boxes are constraints h(x, u) >= 0 (QM_NH = 28 rows per node, qmhip_layout.h) instead of relaxed-barrier costs — slack and dual per row,
 *      condensing into the stage cost, fraction-to-the-boundary step limits, the filter line search on the barrier merit, barrier-parameter update, all on the rest of the
 *      `ipm` block (ST_IPM_MU ... ST_IPM_DUAL_MARGIN); csrc/kernels/k_ipm.h, oracle/src/ipm.h.  The reference registers cones and limits as soft costs only
 *      (QMInterface.cpp:79-142) and instantiates no IpmMpc (it loads both blocks, QMInterface.cpp:70-72): slot 3 solves a DIFFERENT problem than the controller's, has no
 *      reference instance to match, and is checked against the oracle, which is pinned by a dense solve of the horizon's primal-dual Newton system (tests/test_ipm.py).
 *      Slack / dual / per-instance {barrier, alphaP, alphaD, dual step} can be read through qmhip_debug_read("ipm_s" | "ipm_l" | "ipm_info", ...). */
int qmhip_mpc_step(qmhip_ctx* ctx, int B, const double* t0, const double* x0 /*[B][30]*/,
                   int n_ref, const double* ref_t /*[B][n_ref]*/, const double* ref_x /*[B][n_ref][37]*/,
                   int n_events, const double* event_times /*[B][n_events]*/, const int32_t* modes /*[B][n_events+1]*/,
                   double horizon,
                   int32_t* out_num_nodes /*[B]*/, double* out_t /*[B][max_nodes]*/, int32_t* out_event /*[B][max_nodes]*/,
                   int32_t* out_mode /*[B][max_nodes]*/, double* out_x /*[B][max_nodes][30]*/, double* out_u /*[B][max_nodes][30]*/,
                   double* out_perf /*[B][10]*/, int32_t* status /*[B]*/);

/* split form used by the benchmark (inputs resident in HBM before the timed region):
 * upload -> solve (device only, asynchronous on the context stream) -> download */
int qmhip_mpc_upload(qmhip_ctx* ctx, int B, const double* t0, const double* x0, int n_ref, const double* ref_t, const double* ref_x,
                     int n_events, const double* event_times, const int32_t* modes);
int qmhip_mpc_solve_resident(qmhip_ctx* ctx, int B, double horizon);
int qmhip_mpc_download(qmhip_ctx* ctx, int B, int32_t* out_num_nodes, double* out_t, int32_t* out_event, int32_t* out_mode,
                       double* out_x, double* out_u, double* out_perf, int32_t* status);

/* ---- receding horizon: what ocs2::MPC_BASE::run does on every call after the first one
 *      (`mpc.coldStart false`, task.info:142): the SqpSolver keeps its primal solution and the next iteration starts from its
 *      interpolation ([upstream ocs2_sqp multiple_shooting::initializeStateInputTrajectories]; the a9 initializer only beyond it).
 *      set_initial: new observation (t, x) per instance — MPC_MRT_Interface::setCurrentObservation (QMController.cpp:133-137); references,
 *      schedule and the previous solution stay resident.  solve_resident_warm falls back to the cold start if there is no previous solve.
 *      advance_resident: perfect-tracking plant for back-to-back steps without host round trips: t0 += dt, x0 <- policy state at the new t0.
 *      closed_loop_resident: n_steps x [advance (not on the first step), warm solve, policy at t0, WBC on the state built from x0]; when the gait
 *      front-end below has been reset for this batch, every step first refreshes the mode schedule (qmhip_gait_update_resident). */
int qmhip_mpc_set_initial(qmhip_ctx* ctx, int B, const double* t0, const double* x0 /*[B][30]*/);
/*      update_references: new target trajectories and / or mode schedule for the NEXT call, the previous primal solution kept for its warm start — what
 *      ReferenceManager::preSolverRun leaves behind before every MPC_BASE::run (RosReferenceManager's target subscriber, GaitReceiver's template insertion,
 *      QMController.cpp:296-303).  Either pair may be NULL (left as is).  B must be the batch of the last upload. */
int qmhip_mpc_update_references(qmhip_ctx* ctx, int B, int n_ref, const double* ref_t /*[B][n_ref]*/, const double* ref_x /*[B][n_ref][37]*/,
                                int n_events, const double* event_times /*[B][n_events]*/, const int32_t* modes /*[B][n_events+1]*/);
int qmhip_mpc_solve_resident_warm(qmhip_ctx* ctx, int B, double horizon);
int qmhip_mpc_advance_resident(qmhip_ctx* ctx, int B, double dt);
int qmhip_closed_loop_resident(qmhip_ctx* ctx, int B, int n_steps, double mpc_dt, double horizon, double period, double time0);

/* ---- reference / gait front-end, batched and device resident.
 *      Gait side — replaces, per instance, the reference's GaitSchedule object (constructed at qm_interface/src/QMInterface.cpp:455-480 from
 *      reference.info:28-52 with phaseTransitionStanceTime, task.info:11) and the two calls made on it around every MPC iteration
 *      ([upstream ocs2_legged_robot] GaitReceiver::preSolverRun -> GaitSchedule::insertModeSequenceTemplate(template, finalTime, timeHorizon);
 *      SwitchedModelReferenceManager::modifyReferences -> GaitSchedule::getModeSchedule(initTime - T, finalTime + T)), fed by the templates
 *      GaitJoyPublisher loads from gait.info (qm_controllers/src/GaitJoyPublisher.cpp:17-33).
 *      set_templates: the table of mode-sequence templates, n_phases[g] <= QMHIP_GAIT_MAX_PHASES, switching_times[g][QMHIP_GAIT_MAX_PHASES + 1],
 *                     mode_sequence[g][QMHIP_GAIT_MAX_PHASES] (mode id = 8 LF + 4 RF + 2 LH + RH).
 *      reset:         every instance starts from the initial mode schedule (event_times[n_events], mode_sequence[n_events + 1]) and template
 *                     `default_template`; B is the batch size of all later gait calls.
 *      insert_template: insertModeSequenceTemplate for the instances with template_id[b] >= 0 (host arrays [B]).
 *      update_resident: getModeSchedule(t0 - T, t0 + 2T) on every instance (t0 = the resident observation time, T = horizon); the result
 *                     becomes the solver's mode schedule (same buffers qmhip_mpc_upload fills; needs n_events <= max_events, else status -3).
 *                     Event times are produced by the reference's additions in the reference's order: bit-exact.
 *      download / schedule_download: test access to the per-instance GaitSchedule state ([B][QMHIP_GAIT_EVENT_SLOTS], [B][.. + 1]) and to
 *                     the solver's schedule buffers ([B][max_events], [B][max_events + 1]). */
#define QMHIP_GAIT_MAX_PHASES 16
#define QMHIP_GAIT_EVENT_SLOTS 256
int qmhip_gait_set_templates(qmhip_ctx* ctx, int n_gaits, const int32_t* n_phases, const double* switching_times, const int32_t* mode_sequence);
int qmhip_gait_reset(qmhip_ctx* ctx, int B, int n_events, const double* event_times, const int32_t* mode_sequence, int default_template);
int qmhip_gait_insert_template(qmhip_ctx* ctx, int B, const int32_t* template_id, const double* start_time, const double* final_time);
int qmhip_gait_update_resident(qmhip_ctx* ctx, int B, double horizon);
int qmhip_gait_download(qmhip_ctx* ctx, int B, int32_t* n_events, double* event_times, int32_t* mode_sequence, int32_t* template_id, int32_t* status);
int qmhip_schedule_download(qmhip_ctx* ctx, int B, double* event_times /*[B][max_events]*/, int32_t* modes /*[B][max_events + 1]*/);

/*      Target side — replaces the three command callbacks of QmTargetTrajectoriesInteractiveMarker
 *      (qm_controllers/include/qm_controllers/QmTargetTrajectoriesPublisher.h:75-112, QmTargetTrajectoriesPublisher.cpp:94-109) and the
 *      conversion functions of qm_controllers/src/QmTargetTrajectoriesPublisher_node.cpp: kind 1 cmdVelToTargetTrajectories (:71-116),
 *      2 EeCmdVelToTargetTrajectories (:121-165), 3 EEgoalPoseToTargetTrajectories (:172-208), all through targetPoseToTargetTrajectories
 *      (:44-68); kind 0 leaves the instance's target untouched.  The observation is the resident (t0, x0); ee_state[B][7] (position, quaternion
 *      xyzw) may be null: forward kinematics of x0.  ee_through_float != 0 rounds the EE state to float like the qm_msgs::ee_state message does
 *      (QMController.cpp:246-256).  The 2 knots are written into the solver's target buffers (spare knots of max_ref_knots hold the last one);
 *      lastEeTarget_ (QmTargetTrajectoriesPublisher.h:52-54) is per-instance device state, set by target_reset. */
typedef struct qmhip_target_params {
  double time_to_target;                 /* mpc.timeHorizon, task.info:140 (TIME_TO_TARGET, _node.cpp:226) */
  double target_displacement_velocity;   /* reference.info:1 */
  double target_rotation_velocity;       /* reference.info:2 */
  double com_height;                     /* reference.info:4 */
} qmhip_target_params;
int qmhip_target_reset(qmhip_ctx* ctx, int B, const double* last_ee_target /*[7]*/);
int qmhip_target_from_command(qmhip_ctx* ctx, int B, const int32_t* kind /*[B]*/, const double* cmd /*[B][7]*/, const double* ee_state /*[B][7] or NULL*/,
                              int ee_through_float, const qmhip_target_params* params);
int qmhip_target_download(qmhip_ctx* ctx, int B, double* ref_t /*[B][max_ref_knots]*/, double* ref_x /*[B][max_ref_knots][37]*/, double* last_ee_target /*[B][7]*/);

/* ---- policy evaluation: replaces MPC_MRT_Interface::evaluatePolicy (call site QMController.cpp:139-142):
 *      linear interpolation of the last primal solution at time t[b] */
int qmhip_policy_eval(qmhip_ctx* ctx, int B, const double* t, double* x_des /*[B][30]*/, double* u_des /*[B][30]*/, int32_t* mode /*[B]*/);

/* ---- WBC: replaces qm::WbcBase::update / HierarchicalWbc::update (qm_wbc/include/qm_wbc/WbcBase.h:31-32,
 *      qm_wbc/src/HierarchicalWbc.cpp:18-44; variant 1 = HierarchicalMpcWbc.cpp:18-34).
 *      out[b] = [vdot(24), F(12), tau(18)]; qp_status[b][3] per priority level: 0 ok, 1 iteration limit (qpOASES' nWSR = 100, HoQp.cpp:141), 2 working set larger
 *      than the level's null space (a degenerate vertex).  `variant` selects one of the two hierarchies the reference ships; both put inequality rows into their first
 *      level only, and the cascade kernel is specialised to that shape (slack eliminated analytically at level 0, hard rows below).  The GENERAL stacking of
 *      HoQp.cpp:92-124 — own inequality rows at a lower level, with the reference's current-first / previous-first pairing of stacked rows and slack solutions — is not
 *      reachable through this entry point; it is what qmhip_hoqp_solve below offers on explicit task matrices (a generic kernel); restated in the oracle (oracle/src/wbc.h: solveHoLevel) and pinned against
 *      the literal cascade (tests/test_hoqp_literal.py).
 *      The joint-acceleration state `inputLast_` (WbcBase.cpp:212-213) lives in the context per instance;
 *      qmhip_wbc_reset zeroes it.  The call enqueues on the context's WBC stream only and waits for that stream only (pinned staging, asynchronous copies). */
int qmhip_wbc_step(qmhip_ctx* ctx, int B, const double* x_des, const double* u_des, const double* rbd_meas /*[B][55]*/,
                   const int32_t* mode, double period, const double* time /*[B]*/, int variant,
                   double* out /*[B][54]*/, int32_t* qp_status /*[B][3]*/);
int qmhip_wbc_reset(qmhip_ctx* ctx);

/* ---- qm::HoQp on ARBITRARY task hierarchies: replaces the cascade HoQp(task_k, HoQp(task_k-1, ...)) + getSolutions() of qm_wbc/include/qm_wbc/HoQp.h:17-36 /
 *      qm_wbc/src/HoQp.cpp:12-158 for a WbcBase subclass that stacks its tasks differently from the two shipped hierarchies — the general stacking, own inequality
 *      rows below the first level included, with the reference's pairing of stacked rows (current level first, HoQp.cpp:46) and stacked slack solutions (previous
 *      levels first, HoQp.cpp:152-158).  B independent cascades of ONE shape: n decision variables (<= 36), n_levels (<= 8) tasks from the highest priority down with
 *      ma[k] equality rows A x = b (<= 36) and md[k] inequality rows D x <= f (<= 64; <= 128 over all levels); arrays instance-major, the levels concatenated:
 *      A [B][sum ma][n], b [B][sum ma], D [B][sum md][n], f [B][sum md].  x [B][n] = getSolutions() of the last level; status [B][n_levels]: 0 ok, 1 iteration limit
 *      (qpOASES' nWSR = 100), 2 working set larger than the problem (degenerate vertex), 3 the higher levels' rows do not hold at the previous solution (the
 *      row / slack pairing quirk with inequality rows on two higher levels and a non-zero slack: the reference would hand qpOASES that problem and ignore the return
 *      code).  A generic kernel (dense factorisations on a per-problem workspace, one wavefront per problem): not on the benchmark's path. */
int qmhip_hoqp_solve(qmhip_ctx* ctx, int B, int n_levels, int n, const int32_t* ma, const int32_t* md,
                     const double* A, const double* b, const double* D, const double* f, double* x /*[B][n]*/, int32_t* status /*[B][n_levels]*/);

/* ---- whole control step on resident data (benchmark "step"): SQP iteration + policy evaluation at t0 + WBC with
 *      the measured state built from x0 (zero velocities, EE pose by FK).  Everything is enqueued, nothing waited for: the kernels that decide
 *      the line search's step length also write the policy at t0 (what MPC_MRT_Interface::evaluatePolicy reads from the primal solution at its first node), the WBC starts
 *      behind them on its own stream, and the primal solution on all nodes (what qmhip_mpc_download / qmhip_policy_eval / the next warm start read) is written beside it.
 *      Same results as qmhip_mpc_solve_resident + qmhip_policy_eval(t0) + qmhip_wbc_step, bit for bit (tests/test_gpu_mpc.py) */
int qmhip_control_step_resident(qmhip_ctx* ctx, int B, double horizon, double period, double time);
int qmhip_wbc_download(qmhip_ctx* ctx, int B, double* out /*[B][54]*/, int32_t* qp_status /*[B][3]*/);

/* ---- batched rigid-body plant: stands where Gazebo + qm_gazebo::QMHWSim stand in the reference.
 *      sim_set_command = HybridJointHandle::setCommand as QMController::updateControlLaw issues it (qm_controllers/src/QMController.cpp:177-190):
 *        per joint posDes, velDes, kp, kd, ff in the reference's joint order (LF, LH, RF, RH, arm).
 *      sim_step = one simulation step: QMHWSim::writeSim (qm_gazebo/src/QMHWSim.cpp:98-116: the held command enters the delay buffer, commands older than
 *        `delay` — qm_gazebo/config/default.yaml:2 — are dropped, the oldest survivor is applied: tau = kp (posDes − q) + kd (velDes − qd) + ff, saturated at
 *        the URDF effort limit), then n_substeps semi-implicit Euler steps of the floating-base forward dynamics with penalty ground contact of the four
 *        feet, then QMHWSim::readSim (QMHWSim.cpp:60-75): rbd[b] is the state in the estimator's layout (qm_estimation/src/StateEstimateBase.cpp:41-103),
 *        contact[b] the four contact flags (LF RF LH RH).  rbd / contact may be null (results stay resident).
 *      sim_reset: generalized coordinates q = [pos(3), zyx(3), joints(18)], v = [world linear velocity, zyx rates, joint rates], time per instance;
 *        clears the delay buffer ("Simulation reset", QMHWSim.cpp:101-103).  A new episode starts cold: the MPC's previous primal solution is dropped, so until the next
 *        solve qmhip_policy_eval / qmhip_mpc_download / qmhip_mpc_advance_resident return QMHIP_ERR_STATE and the next warm solve is a cold one.
 *      sim_set_params: {contact stiffness [N/m], contact damping [N s/m], friction coefficient, friction regularisation speed [m/s], foot radius [m],
 *        command delay [s], saturate efforts (0/1)}; Gazebo's ODE contact solver is not part of the reference's sources, the contact model is this library's own.
 *      sim_get_state: q, v, time, contact forces [B][12] (world frame) of the last sub-step, status [B] (0 ok, 1 mass matrix not positive definite); any may be null. */
int qmhip_sim_set_params(qmhip_ctx* ctx, const double* params, int n);
/*      sim_set_controller: which controller plugin the closed loops below run — 0 qm::QMController (HierarchicalWbc; updateControlLaw QMController.cpp:177-190),
 *        1 qm::QMMpcController (HierarchicalMpcWbc, QMController.cpp:410-414; updateControlLaw QMController.cpp:431-445: legs commanded on every tick, the arm as
 *        position commands q_meas + velDes / 100 re-published when more than 1/100 s have passed — arm_kp / arm_kd of the loop calls are then the gains of the arm's
 *        position controllers and no arm torque is fed forward).  The control law switches with the next tick; the arm's held position command / publication
 *        times (QMController::starting, QMController.cpp:127) are only re-initialised by qmhip_sim_reset, so switch before a reset. */
int qmhip_sim_set_controller(qmhip_ctx* ctx, int controller);
int qmhip_sim_reset(qmhip_ctx* ctx, int B, const double* q /*[B][24]*/, const double* v /*[B][24]*/, const double* time /*[B]*/);
int qmhip_sim_set_command(qmhip_ctx* ctx, int B, const double* pos_des /*[B][18]*/, const double* vel_des, const double* kp, const double* kd, const double* ff);
int qmhip_sim_step(qmhip_ctx* ctx, int B, double period, int n_substeps, double* rbd /*[B][55]*/, int32_t* contact /*[B][4]*/);
int qmhip_sim_get_state(qmhip_ctx* ctx, int B, double* q, double* v, double* time, double* force /*[B][12]*/, int32_t* status /*[B]*/);
/*      closed_loop_sim: n_ticks of the whole controller around the plant, device resident — QMController::update + mpcThread_
 *        (qm_controllers/src/QMController.cpp:128-175, 202-244, 315-332): state estimate = the plant's state ("ground truth" estimator,
 *        currentObservation_.state = computeCentroidalStateFromRbdModel), an MPC call on that observation every mpc_every ticks (warm-started SQP; the
 *        gait front-end refreshes the mode schedule first when it is active), policy evaluation at the plant time, WBC on the measured state,
 *        updateControlLaw (QMController.cpp:177-190: legs kp 0 / kd 3 once time > 10, arm arm_kp / arm_kd, WBC torque as feed-forward), one
 *        simulation step.  Needs qmhip_mpc_upload (reference, schedule) and qmhip_sim_reset before; the tick counter restarts at sim_reset.
 *        The reference runs the MPC in its own thread; here it is synchronous with the tick that triggers it.  On the first tick after a reset the
 *        WBC's joint-acceleration state inputLast_ (WbcBase.cpp:212-213) is primed with the planned input (zero joint acceleration). */
int qmhip_closed_loop_sim(qmhip_ctx* ctx, int B, int n_ticks, double period, int n_substeps, int mpc_every, double horizon, double arm_kp, double arm_kd);
/*      closed_loop_sim_pipelined: the same loop with the MPC BESIDE the control ticks, like mpcThread_ beside QMController::update (QMController.cpp:315-332): the
 *        MPC call triggered at a tick observes the plant at that tick and computes on its own stream while the next mpc_every ticks run on the policy published
 *        before; its solution is published (MPC_MRT_Interface's policy buffer) when those ticks are done — a latency of one MPC period, deterministic instead of
 *        thread-timing dependent.  The first call after a reset is synchronous.  n_ticks and the tick counter must be multiples of mpc_every. */
int qmhip_closed_loop_sim_pipelined(qmhip_ctx* ctx, int B, int n_ticks, double period, int n_substeps, int mpc_every, double horizon, double arm_kp, double arm_kd);

/* ---- instrumentation (ocs2 benchmark::RepeatedTimer analogue, QMController.cpp:145-147,321-323) ----
 * per-kernel HIP-event timing on the stream each kernel runs on; names: "grid","lq_kin","lq","riccati","ls_eval","ls_misc","policy","wbc","sim".
 * enable: 0 off, 1 a span around every launch, 2 only around the three modelled kernels "lq","riccati","wbc", 3 only around "lq" — the dominant kernel, all the
 * bench's timed region carries (two event records cost about one launch) */
int qmhip_set_profiling(qmhip_ctx* ctx, int enable);
int qmhip_get_kernel_ms(qmhip_ctx* ctx, const char* name, double* total_ms, int* launches);
int qmhip_reset_kernel_ms(qmhip_ctx* ctx);
int qmhip_synchronize(qmhip_ctx* ctx);
/* line-search trials of the last SQP iteration (the longest search of the batch).  Since round 6 the trials after the first run on the device without the host
 * (qm_ls_tail_kernel): the count is read back from the device's trial counters on demand — this call synchronises the context's streams */
int qmhip_last_ls_trials(const qmhip_ctx* ctx);
/* debug/parity access to a device buffer by name (see QmMpcBuffers); copies `bytes` to host */
int qmhip_debug_read(qmhip_ctx* ctx, const char* buffer, void* dst, size_t bytes);
/* profiling / parity switches: "riccati_skip" bit mask of kernel phases to skip (results are then meaningless; 20 = no backward stage and no rollout, i.e. the stage
 * records stay as the LQ kernel wrote them), "wbc_stop", "lq_prof", "lq_debug" (1: the LQ kernel also writes the unprojected LQ model of every interval into the
 * buffer "lqdbg" and the null-space basis into the stage record — tests/test_gpu_lq_records.py); launch-order switches for same-box A/Bs and the bit-identity tests
 * (results do not depend on them): "ls_device_tail" (1: the line search's later trials in one launch on the device; 0: one host round trip per trial, rounds 1-5),
 * "fused_policy" (1: the policy at t0 from the deciding kernels, apply beside the WBC; 0: apply -> policy kernel -> WBC), "r_dense" (1: the dense instances of the trial
 * evaluation and of the LQ kernel's input-weight product although the table's R is block diagonal — same bits; readable: "r_blocks" = the structured instances run),
 * "wbc_defer", "filler_*" (scheduling experiments) */
int qmhip_debug_set(qmhip_ctx* ctx, const char* key, int value);
/* read such a switch back (bench.py asserts they are all 0 before it times anything) */
int qmhip_debug_get(const qmhip_ctx* ctx, const char* key, int* value);
/* profiling only: a latency-bound filler kernel on the second stream (co-residency experiments, tools/coresidency_probe.py) */
int qmhip_debug_filler(qmhip_ctx* ctx, int waves, int iters, int wait, double* ms);
int qmhip_debug_lq_with_filler(qmhip_ctx* ctx, int B, double horizon, int waves, int iters, double* ms /*[3]: LQ kernel, filler, Riccati kernel (ms)*/);
/* micro-benchmarks used to anchor the FP64 roofline: returns achieved TFLOP/s */
int qmhip_microbench_fp64(qmhip_ctx* ctx, int use_mfma, double* tflops);

#define QMHIP_OK 0
#define QMHIP_ERR_ARG  -1
#define QMHIP_ERR_FILE -2
#define QMHIP_ERR_MODEL -3
#define QMHIP_ERR_HIP  -4
#define QMHIP_ERR_STATE -5

#ifdef __cplusplus
}
#endif
#endif
