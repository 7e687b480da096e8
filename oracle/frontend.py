"""oracle/frontend.py — TEST INFRASTRUCTURE (CPU oracle of the reference / gait front-end).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this; the product never does.

Restates, for ONE robot at a time with plain Python lists / numpy (the reference's own data structures are std::vector):

* GaitSchedule  [upstream ocs2_legged_robot/gait/GaitSchedule.cpp, recalled — not vendored in /root/reference; SURVEY.md B.2]
    constructed by qm_interface/src/QMInterface.cpp:455-480 from reference.info:28-52 (initialModeSchedule, defaultModeSequenceTemplate)
    and task.info:11 (phaseTransitionStanceTime); templates published by qm_controllers/src/GaitJoyPublisher.cpp:17-33 from gait.info.
    insertModeSequenceTemplate / getModeSchedule / tileModeSequenceTemplate follow the upstream statements one by one; event times are
    Python floats (IEEE double) added in the same order, so they are bit-identical to what the C++ produces.
* the call pattern around one MPC iteration [upstream]: GaitReceiver::preSolverRun(initTime, finalTime) inserts a received template with
    (startTime = finalTime, finalTime = finalTime − initTime)  — the upstream call passes the horizon LENGTH as the tiling bound —
    and SwitchedModelReferenceManager::modifyReferences asks getModeSchedule(initTime − T, finalTime + T), T = finalTime − initTime.
* the command -> TargetTrajectories functions of qm_controllers/src/QmTargetTrajectoriesPublisher_node.cpp:25-208 and the lastEeTarget_
    bookkeeping of QmTargetTrajectoriesPublisher.h:52-54 / QmTargetTrajectoriesPublisher.cpp:94-109.

Parity unpinned (as for the rest of the oracle): the reference ships no tests or golden vectors for these functions; tests/test_frontend.py
pins this file with hand-computed known answers (trot tiling, stance insertion, erase rules).
"""
import bisect
import math
import numpy as np

STANCE = 15
MODE_NAMES = {"FLY": 0, "RH": 1, "LH": 2, "LH_RH": 3, "RF": 4, "RF_RH": 5, "RF_LH": 6, "RF_LH_RH": 7, "LF": 8, "LF_RH": 9, "LF_LH": 10,
              "LF_LH_RH": 11, "LF_RF": 12, "LF_RF_RH": 13, "LF_RF_LH": 14, "STANCE": 15}


class GaitSchedule:
    """ModeSchedule {eventTimes[K], modeSequence[K + 1]} + the current ModeSequenceTemplate {switchingTimes[P + 1], modeSequence[P]}."""

    def __init__(self, event_times, mode_sequence, template_times, template_modes, phase_transition_stance_time):
        self.event_times = [float(t) for t in event_times]
        self.mode_sequence = [int(m) for m in mode_sequence]
        assert len(self.mode_sequence) == len(self.event_times) + 1
        self.template_times = [float(t) for t in template_times]
        self.template_modes = [int(m) for m in template_modes]
        self.pts = float(phase_transition_stance_time)

    # GaitSchedule::tileModeSequenceTemplate(startTime, finalTime)
    def _tile(self, start_time, final_time):
        n = len(self.template_modes)
        if n == 0:
            return
        if self.event_times and start_time <= self.event_times[-1]:
            raise RuntimeError("The initial time for template-tiling is not greater than the last event time.")
        self.event_times.append(start_time)
        while self.event_times[-1] < final_time:
            for i in range(n):
                self.mode_sequence.append(self.template_modes[i])
                delta = self.template_times[i + 1] - self.template_times[i]
                self.event_times.append(self.event_times[-1] + delta)
        self.mode_sequence.append(STANCE)

    # GaitSchedule::insertModeSequenceTemplate(modeSequenceTemplate, startTime, finalTime)
    def insert_mode_sequence_template(self, template_times, template_modes, start_time, final_time):
        self.template_times = [float(t) for t in template_times]
        self.template_modes = [int(m) for m in template_modes]
        index = bisect.bisect_left(self.event_times, start_time)            # std::lower_bound
        if index < len(self.event_times):
            del self.event_times[index:]
            del self.mode_sequence[index + 1:]
        pts = self.pts
        if self.mode_sequence and self.mode_sequence[-1] == STANCE:
            pts = 0.0
        if pts > 0.0:
            self.event_times.append(start_time)
            self.mode_sequence.append(STANCE)
        self._tile(start_time + pts, final_time)

    # GaitSchedule::getModeSchedule(lowerBoundTime, upperBoundTime)
    def get_mode_schedule(self, lower, upper):
        index = bisect.bisect_left(self.event_times, lower)
        if index > 0:
            del self.event_times[:index - 1]
            del self.mode_sequence[:index - 1]
            self.mode_sequence[0] = STANCE
        tiling_start = upper if not self.event_times else self.event_times[-1]
        del self.event_times[-1:]
        del self.mode_sequence[-1:]
        self._tile(tiling_start, upper)
        return list(self.event_times), list(self.mode_sequence)

    # --- the reference's call pattern around one MPC iteration ---
    def pre_solver_run_insert(self, template_times, template_modes, init_time, final_time):
        """GaitReceiver::preSolverRun with a freshly received template."""
        self.insert_mode_sequence_template(template_times, template_modes, final_time, final_time - init_time)

    def modify_references(self, init_time, horizon):
        """MPC_BASE::run: finalTime = initTime + horizon; SwitchedModelReferenceManager::modifyReferences."""
        final_time = init_time + horizon
        th = final_time - init_time
        return self.get_mode_schedule(init_time - th, final_time + th)


# ---- command -> TargetTrajectories ----
def _zyx_to_R(z, y, x):
    cz, sz, cy, sy, cx, sx = math.cos(z), math.sin(z), math.cos(y), math.sin(y), math.cos(x), math.sin(x)
    return np.array([[cz * cy, cz * sy * sx - sz * cx, cz * sy * cx + sz * sx],
                     [sz * cy, sz * sy * sx + cz * cx, sz * sy * cx - cz * sx],
                     [-sy, cy * sx, cy * cx]])


def _quat_to_R(q):  # xyzw
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _quaternion_distance(q, qref):  # [upstream ocs2_robotic_tools quaternionDistance]; xyzw
    return q[3] * qref[:3] - qref[3] * q[:3] + np.cross(q[:3], qref[:3])


class TargetPublisher:
    """QmTargetTrajectoriesInteractiveMarker + the conversion functions of QmTargetTrajectoriesPublisher_node.cpp for one robot."""

    def __init__(self, default_joint_state, com_height, disp_velocity, rot_velocity, time_to_target):
        self.qnom = np.asarray(default_joint_state, dtype=float)
        self.com_height, self.vd, self.vr, self.T = float(com_height), float(disp_velocity), float(rot_velocity), float(time_to_target)
        # QmTargetTrajectoriesPublisher.h:52-54
        self.last_ee = np.array([0.52, 0.09, 0.44, 0.5, -0.5, 0.5, -0.5])

    def _target_pose(self, ee_target, base_target, t0, x, ee_first, reach):     # targetPoseToTargetTrajectories, _node.cpp:44-68
        base_cur = np.array(x[6:12], dtype=float)
        base_cur[2] = self.com_height; base_cur[4] = 0.0; base_cur[5] = 0.0
        xs = np.zeros((2, 37))
        xs[0, 6:12] = base_cur; xs[1, 6:12] = base_target
        xs[:, 12:30] = self.qnom
        xs[0, 30:37] = ee_first; xs[1, 30:37] = ee_target
        return np.array([t0, reach]), xs

    def cmd_vel(self, cmd, t0, x, ee):                                           # _node.cpp:71-116
        base = np.array(x[6:12], dtype=float)
        v = _zyx_to_R(base[3], base[4], base[5]) @ np.asarray(cmd[:3], dtype=float)
        T = self.T
        base_target = np.array([base[0] + v[0] * T, base[1] + v[1] * T, self.com_height, base[3] + cmd[3] * T, 0.0, 0.0])
        if np.linalg.norm(self.last_ee[:3] - np.asarray(ee[:3])) > 0.1:
            self.last_ee[:3] = ee[:3]
        ee_target = self.last_ee.copy()
        rt, xs = self._target_pose(ee_target, base_target, t0, x, ee_target, t0 + T)
        xs[0, :3] = v; xs[1, :3] = v
        return rt, xs

    def ee_cmd_vel(self, cmd, t0, x, ee):                                        # _node.cpp:121-165
        ee = np.asarray(ee, dtype=float)
        base = np.array(x[6:12], dtype=float)
        qinit = np.array([0.5, -0.5, 0.5, -0.5])                                 # Quaterniond(w −0.5, 0.5, −0.5, 0.5)
        v = _quat_to_R(ee[3:7]) @ _quat_to_R(qinit).T @ np.asarray(cmd[:3], dtype=float)
        T = self.T
        ee_target = ee.copy()
        ee_target[0] = ee[0] + v[0] * T; ee_target[1] = ee[1] + v[1] * T
        ee_target[2:7] = self.last_ee[2:7]
        base_target = base.copy()
        base_target[0] = ee_target[0] - 0.52; base_target[1] = ee_target[1] - 0.09; base_target[2] = self.com_height; base_target[4] = 0.0; base_target[5] = 0.0
        return self._target_pose(ee_target, base_target, t0, x, ee, t0 + T)

    def ee_goal(self, goal, t0, x, ee):                                          # _node.cpp:172-208 + processFeedback (.cpp:94-109)
        ee = np.asarray(ee, dtype=float); goal = np.asarray(goal, dtype=float)
        base = np.array(x[6:12], dtype=float)
        base_target = base.copy()
        base_target[0] = goal[0] - 0.52; base_target[1] = goal[1] - 0.09; base_target[2] = self.com_height; base_target[4] = 0.0; base_target[5] = 0.0
        dp = goal[:3] - ee[:3]
        dr = _quaternion_distance(ee[3:7], goal[3:7])
        reach = t0 + max(np.linalg.norm(dr) / self.vr, np.linalg.norm(dp) / self.vd)   # estimateTimeToTarget, _node.cpp:25-41
        out = self._target_pose(goal.copy(), base_target, t0, x, ee, reach)
        self.last_ee = goal.copy()
        return out


def ee_state_through_float(ee):
    """qm_msgs::ee_state carries float32 (QMController.cpp:246-256)."""
    return np.asarray(ee, dtype=np.float32).astype(np.float64)
