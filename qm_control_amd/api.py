"""Host-side mirror of the reference's interfaces for the hot path, over the C ABI of libqmhip.so.

    QMInterface   <- qm::QMInterface            (qm_interface/include/qm_interface/QMInterface.h:31-54)
    SqpMpc        <- ocs2::MPC_BASE / SqpMpc    (qm_controllers/src/QMController.cpp:287-288; run(), policy)
    HierarchicalWbc <- qm::HierarchicalWbc      (qm_wbc/src/HierarchicalWbc.cpp:18-44; update(), loadTasksSetting)

There is NO CPU fallback: if libqmhip.so is missing, or no HIP device is present, construction raises.
Arrays are numpy f64, instance-major (batch first).  torch is not needed by this module.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libqmhip.so")
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
from . import layout as L
MB_SIZE, ST_SIZE = L.MB_SIZE, L.ST_SIZE

EXPORTS = ["qmhip_create", "qmhip_create_from_blobs", "qmhip_create_wbc_context", "qmhip_destroy", "qmhip_last_error", "qmhip_parse_model", "qmhip_export_blobs",
           "qmhip_set_setting", "qmhip_wbc_gain_index", "qmhip_mpc_step", "qmhip_mpc_upload", "qmhip_mpc_solve_resident", "qmhip_mpc_set_initial", "qmhip_mpc_update_references", "qmhip_mpc_solve_resident_warm",
           "qmhip_mpc_advance_resident", "qmhip_closed_loop_resident", "qmhip_mpc_download", "qmhip_policy_eval",
           "qmhip_wbc_step", "qmhip_wbc_reset", "qmhip_hoqp_solve", "qmhip_control_step_resident", "qmhip_wbc_download", "qmhip_set_profiling",
           "qmhip_get_kernel_ms", "qmhip_reset_kernel_ms", "qmhip_synchronize", "qmhip_last_ls_trials", "qmhip_debug_read", "qmhip_debug_set", "qmhip_debug_get", "qmhip_debug_filler", "qmhip_debug_lq_with_filler", "qmhip_microbench_fp64",
           "qmhip_gait_set_templates", "qmhip_gait_reset", "qmhip_gait_insert_template", "qmhip_gait_update_resident", "qmhip_gait_download", "qmhip_schedule_download",
           "qmhip_target_reset", "qmhip_target_from_command", "qmhip_target_download",
           "qmhip_sim_set_params", "qmhip_sim_set_controller", "qmhip_sim_reset", "qmhip_sim_set_command", "qmhip_sim_step", "qmhip_sim_get_state", "qmhip_closed_loop_sim", "qmhip_closed_loop_sim_pipelined"]


class QmhipError(RuntimeError):
    pass


_lib = None


def load_library():
    """dlopen libqmhip.so (built by __graft_entry__.build()); raises if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise QmhipError("libqmhip.so not found at %s — build it with `python -c 'import __graft_entry__ as g; g.build()'`; there is no CPU fallback" % LIB_PATH)
        _lib = C.CDLL(LIB_PATH)
        _lib.qmhip_last_error.restype = C.c_char_p
        _lib.qmhip_last_error.argtypes = [C.c_void_p]
        _lib.qmhip_destroy.argtypes = [C.c_void_p]
        _lib.qmhip_destroy.restype = None
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _pi(a):
    return None if a is None else a.ctypes.data_as(_ip)


def _f(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        assert a.shape == tuple(shape), (a.shape, shape)
    return a


def parse_model(urdf_file, task_file, reference_file):
    """host-only parse (no GPU needed): returns (model_blob, settings_blob); raises ValueError for missing files
    like the reference constructor's std::invalid_argument (QMInterface.cpp:45,53,61)."""
    lib = load_library()
    mb = np.zeros(MB_SIZE); st = np.zeros(ST_SIZE)
    rc = lib.qmhip_parse_model(urdf_file.encode(), task_file.encode(), reference_file.encode(), _p(mb), _p(st))
    if rc != 0:
        msg = lib.qmhip_last_error(None).decode()
        raise (ValueError if rc == -2 else QmhipError)(msg)
    return mb, st


class QMInterface:
    """Owns the device context (model + settings + all HBM buffers) — qm::QMInterface's role."""

    def __init__(self, task_file=None, urdf_file=None, reference_file=None, *, blobs=None, device=0, max_batch=1, max_nodes=128, max_ref_knots=2, max_events=8):
        self.lib = load_library()
        h = C.c_void_p()
        if blobs is not None:
            mb, st = _f(blobs[0], (MB_SIZE,)), _f(blobs[1], (ST_SIZE,))
            rc = self.lib.qmhip_create_from_blobs(_p(mb), _p(st), device, max_batch, max_nodes, max_ref_knots, max_events, C.byref(h))
        else:
            rc = self.lib.qmhip_create(urdf_file.encode(), task_file.encode(), reference_file.encode(), device, max_batch, max_nodes, max_ref_knots, max_events, C.byref(h))
        if rc != 0:
            msg = self.lib.qmhip_last_error(None).decode()
            raise (ValueError if rc == -2 else QmhipError)("qmhip_create failed (%d): %s" % (rc, msg))
        self.h = h
        self.max_batch, self.max_nodes, self.max_ref_knots, self.max_events = max_batch, max_nodes, max_ref_knots, max_events
        self.model_blob = np.zeros(MB_SIZE); self.settings_blob = np.zeros(ST_SIZE)
        self.lib.qmhip_export_blobs(self.h, _p(self.model_blob), _p(self.settings_blob))

    def wbc_context(self, max_batch=None):
        """WBC-only context for the control thread (include/qmhip.h "Threads"): an interface object whose handle serves HierarchicalWbc only"""
        h = C.c_void_p()
        mbatch = self.max_batch if max_batch is None else int(max_batch)
        rc = self.lib.qmhip_create_wbc_context(self.h, mbatch, C.byref(h))
        if rc != 0:
            raise QmhipError("qmhip_create_wbc_context failed (%d): %s" % (rc, self.lib.qmhip_last_error(None).decode()))
        o = object.__new__(QMInterface)
        o.lib = self.lib; o.h = h; o.max_batch = mbatch; o.max_nodes, o.max_ref_knots, o.max_events = 3, 1, 1
        o.model_blob = self.model_blob.copy(); o.settings_blob = self.settings_blob.copy()
        return o

    def close(self):
        if getattr(self, "h", None):
            self.lib.qmhip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise QmhipError("%s failed (%d): %s" % (what, rc, self.lib.qmhip_last_error(self.h).decode()))

    # getters named after the reference's
    def getInitialState(self):
        return self.settings_blob[L.ST_XINIT:L.ST_XINIT + 30].copy()

    def getCentroidalModelInfo(self):
        mb = self.model_blob
        return dict(robotMass=mb[L.MB_ROBOTMASS], centroidalInertiaNominal=mb[L.MB_INOM:L.MB_INOM + 9].reshape(3, 3).copy(), comToBasePositionNominal=mb[L.MB_RNOM:L.MB_RNOM + 3].copy(),
                    qPinocchioNominal=np.concatenate([np.zeros(6), mb[L.MB_QNOM:L.MB_QNOM + 18]]), stateDim=30, inputDim=30, generalizedCoordinatesNum=24, actuatedDofNum=18, numThreeDofContacts=4)

    def set_gain(self, name, value):
        """one field of the reference's dynamic_reconfigure config qm_wbc::WbcWeightConfig by NAME (WbcBase::dynamicCallback, WbcBase.cpp:69-116); False for a field
        the callback does not read (d_ee_x ...)"""
        idx = self.lib.qmhip_wbc_gain_index(name.encode())
        if idx < 0:
            return False
        self.set_setting(idx, value)
        return True

    def set_setting(self, index, value):
        self._check(self.lib.qmhip_set_setting(self.h, C.c_int(index), C.c_double(value)), "qmhip_set_setting")
        self.settings_blob[index] = value

    # instrumentation
    def set_profiling(self, on):
        """False / True: spans around no / every launch; 2: only around the modelled kernels (lq, riccati, wbc)"""
        self.lib.qmhip_set_profiling(self.h, int(on))

    def kernel_ms(self, name):
        ms = C.c_double(0); n = C.c_int(0)
        self.lib.qmhip_get_kernel_ms(self.h, name.encode(), C.byref(ms), C.byref(n))
        return ms.value, n.value

    def reset_kernel_ms(self):
        self.lib.qmhip_reset_kernel_ms(self.h)

    def synchronize(self):
        self._check(self.lib.qmhip_synchronize(self.h), "qmhip_synchronize")

    def microbench_fp64(self, use_mfma):
        v = C.c_double(0)
        self._check(self.lib.qmhip_microbench_fp64(self.h, int(use_mfma), C.byref(v)), "qmhip_microbench_fp64")
        return v.value

    def debug_set(self, key, value):
        self._check(self.lib.qmhip_debug_set(self.h, key.encode(), C.c_int(value)), "qmhip_debug_set")

    def debug_filler(self, waves, iters, wait=True):
        ms = C.c_double(0)
        self._check(self.lib.qmhip_debug_filler(self.h, int(waves), int(iters), int(bool(wait)), C.byref(ms)), "qmhip_debug_filler")
        return ms.value

    def debug_lq_with_filler(self, B, horizon, waves, iters):
        ms = (C.c_double * 3)()
        self._check(self.lib.qmhip_debug_lq_with_filler(self.h, int(B), C.c_double(horizon), int(waves), int(iters), ms), "qmhip_debug_lq_with_filler")
        return list(ms)

    def debug_get(self, key):
        v = C.c_int(0)
        self._check(self.lib.qmhip_debug_get(self.h, key.encode(), C.byref(v)), "qmhip_debug_get")
        return v.value

    def debug_read(self, name, shape, dtype=np.float64):
        out = np.zeros(shape, dtype=dtype)
        self._check(self.lib.qmhip_debug_read(self.h, name.encode(), out.ctypes.data_as(C.c_void_p), C.c_size_t(out.nbytes)), "qmhip_debug_read")
        return out


class SqpMpc:
    """ocs2::MPC_BASE-shaped front: run() = one multiple-shooting SQP iteration per instance (cold start)."""

    def __init__(self, interface):
        self.itf = interface
        self.lib = interface.lib
        self.B = 0

    def set_problem(self, t0, x0, ref_t, ref_x, event_times, modes):
        B = len(t0)
        t0 = _f(t0, (B,)); x0 = _f(x0, (B, 30)); ref_t = _f(ref_t); ref_x = _f(ref_x); ev = _f(event_times)
        modes = np.ascontiguousarray(modes, dtype=np.int32)
        assert ref_x.shape == (B, ref_t.shape[1], 37) and modes.shape == (B, ev.shape[1] + 1)
        self.itf._check(self.lib.qmhip_mpc_upload(self.itf.h, B, _p(t0), _p(x0), ref_t.shape[1], _p(ref_t), _p(ref_x), ev.shape[1], _p(ev), _pi(modes)), "qmhip_mpc_upload")
        self.B = B

    def solve_resident(self, horizon, warm=False):
        """one SQP iteration; warm=True starts from the previous primal solution (what MPC_BASE::run does on every call after the first)"""
        if warm:
            self.itf._check(self.lib.qmhip_mpc_solve_resident_warm(self.itf.h, self.B, C.c_double(horizon)), "qmhip_mpc_solve_resident_warm")
        else:
            self.itf._check(self.lib.qmhip_mpc_solve_resident(self.itf.h, self.B, C.c_double(horizon)), "qmhip_mpc_solve_resident")

    def set_initial(self, t0, x0):
        """new observation (MPC_MRT_Interface::setCurrentObservation); references, schedule and the previous solution stay resident"""
        t0 = _f(t0, (self.B,)); x0 = _f(x0, (self.B, 30))
        self.itf._check(self.lib.qmhip_mpc_set_initial(self.itf.h, self.B, _p(t0), _p(x0)), "qmhip_mpc_set_initial")

    def update_references(self, ref_t=None, ref_x=None, event_times=None, modes=None):
        """new targets and / or mode schedule for the next call; the previous primal solution stays (warm start) — ReferenceManager::preSolverRun"""
        B = self.B
        rt = rx = ev = mo = None; n_ref = self.itf.max_ref_knots; n_ev = self.itf.max_events
        if ref_t is not None:
            rt = _f(ref_t, (B, n_ref)); rx = _f(ref_x, (B, n_ref, 37))
        if event_times is not None:
            ev = _f(event_times, (B, n_ev)); mo = np.ascontiguousarray(modes, dtype=np.int32); assert mo.shape == (B, n_ev + 1)
        self.itf._check(self.lib.qmhip_mpc_update_references(self.itf.h, B, n_ref, _p(rt), _p(rx), n_ev, _p(ev), _pi(mo)), "qmhip_mpc_update_references")

    def advance(self, dt):
        """perfect-tracking plant on the device: t0 += dt, x0 <- policy state at the new t0"""
        self.itf._check(self.lib.qmhip_mpc_advance_resident(self.itf.h, self.B, C.c_double(dt)), "qmhip_mpc_advance_resident")

    def closed_loop_resident(self, n_steps, mpc_dt, horizon, period, time0):
        self.itf._check(self.lib.qmhip_closed_loop_resident(self.itf.h, self.B, int(n_steps), C.c_double(mpc_dt), C.c_double(horizon), C.c_double(period), C.c_double(time0)), "qmhip_closed_loop_resident")

    def control_step_resident(self, horizon, period, time):
        self.itf._check(self.lib.qmhip_control_step_resident(self.itf.h, self.B, C.c_double(horizon), C.c_double(period), C.c_double(time)), "qmhip_control_step_resident")

    def download(self):
        B, nm = self.B, self.itf.max_nodes
        nn = np.zeros(B, np.int32); t = np.zeros((B, nm)); ev = np.zeros((B, nm), np.int32); mode = np.zeros((B, nm), np.int32)
        x = np.zeros((B, nm, 30)); u = np.zeros((B, nm, 30)); perf = np.zeros((B, 10)); status = np.zeros(B, np.int32)
        self.itf._check(self.lib.qmhip_mpc_download(self.itf.h, B, _pi(nn), _p(t), _pi(ev), _pi(mode), _p(x), _p(u), _p(perf), _pi(status)), "qmhip_mpc_download")
        return dict(num_nodes=nn, t=t, event=ev, mode=mode, x=x, u=u, perf=perf, status=status, ls_trials=self.lib.qmhip_last_ls_trials(self.itf.h))

    def run(self, t0, x0, ref_t, ref_x, event_times, modes, horizon):
        self.set_problem(t0, x0, ref_t, ref_x, event_times, modes)
        self.solve_resident(horizon)
        return self.download()

    def evaluatePolicy(self, t):
        t = _f(t, (self.B,))
        x = np.zeros((self.B, 30)); u = np.zeros((self.B, 30)); mode = np.zeros(self.B, np.int32)
        self.itf._check(self.lib.qmhip_policy_eval(self.itf.h, self.B, _p(t), _p(x), _p(u), _pi(mode)), "qmhip_policy_eval")
        return x, u, mode


class HierarchicalWbc:
    """qm::HierarchicalWbc-shaped front: update(stateDesired, inputDesired, rbdStateMeasured, mode, period, time) -> [x(36); tau(18)]."""

    def __init__(self, interface, mpc_variant=False):
        self.itf = interface
        self.lib = interface.lib
        self.variant = 1 if mpc_variant else 0

    def reset(self):
        self.itf._check(self.lib.qmhip_wbc_reset(self.itf.h), "qmhip_wbc_reset")

    def update(self, stateDesired, inputDesired, rbdStateMeasured, mode, period, time):
        xd = _f(stateDesired); B = xd.shape[0]
        ud = _f(inputDesired, (B, 30)); rbd = _f(rbdStateMeasured, (B, 55)); mode = np.ascontiguousarray(mode, np.int32); time = _f(np.broadcast_to(time, (B,)))
        out = np.zeros((B, 54)); st = np.zeros((B, 3), np.int32)
        self.itf._check(self.lib.qmhip_wbc_step(self.itf.h, B, _p(xd), _p(ud), _p(rbd), _pi(mode), C.c_double(period), _p(time), self.variant, _p(out), _pi(st)), "qmhip_wbc_step")
        return out, st

    def download(self, B):
        out = np.zeros((B, 54)); st = np.zeros((B, 3), np.int32)
        self.itf._check(self.lib.qmhip_wbc_download(self.itf.h, B, _p(out), _pi(st)), "qmhip_wbc_download")
        return out, st


class HoQp:
    """qm::HoQp-shaped front of the general cascade (qm_wbc/include/qm_wbc/HoQp.h:17-36): tasks from the highest priority down, every task dict(A, b, D, f) with
    arrays [B][rows][n] / [B][rows] (or without the batch axis for one problem); getSolutions() of the last level and the per-level status"""

    def __init__(self, interface):
        self.itf = interface; self.lib = interface.lib

    def solve(self, tasks):
        first = np.asarray(tasks[0]["A"] if np.asarray(tasks[0]["A"]).size else tasks[0]["D"], float)
        single = first.ndim == 2
        n = first.shape[-1]
        arr = lambda t, k, w: np.asarray(t[k], float).reshape((1, -1, n) if w else (1, -1)) if single else np.asarray(t[k], float).reshape((len(np.asarray(t[k])), -1, n) if w else (len(np.asarray(t[k])), -1))
        As = [arr(t, "A", True) for t in tasks]; bs = [arr(t, "b", False) for t in tasks]; Ds = [arr(t, "D", True) for t in tasks]; fs = [arr(t, "f", False) for t in tasks]
        B = max(a.shape[0] for a in As + Ds)
        ma = np.array([a.shape[1] for a in As], np.int32); md = np.array([d.shape[1] for d in Ds], np.int32)
        A = _f(np.concatenate(As, axis=1)) if ma.sum() else np.zeros(1); b = _f(np.concatenate(bs, axis=1)) if ma.sum() else np.zeros(1)
        D = _f(np.concatenate(Ds, axis=1)) if md.sum() else np.zeros(1); f = _f(np.concatenate(fs, axis=1)) if md.sum() else np.zeros(1)
        x = np.zeros((B, n)); st = np.zeros((B, len(tasks)), np.int32)
        self.itf._check(self.lib.qmhip_hoqp_solve(self.itf.h, B, len(tasks), n, _pi(ma), _pi(md), _p(A), _p(b), _p(D), _p(f), _p(x), _pi(st)), "qmhip_hoqp_solve")
        return (x[0], st[0]) if single else (x, st)


class QMHWSim:
    """qm_gazebo::QMHWSim-shaped front of the batched rigid-body plant (qm_gazebo/src/QMHWSim.cpp:60-116): `setCommand` is what
    QMController::updateControlLaw issues through the hybrid joint handles, `step(period)` = writeSim (delay buffer + joint PD law) + the physics step +
    readSim (rbd state in the estimator's layout, contact flags)."""
    PARAMS = ("contact_stiffness", "contact_damping", "friction", "friction_speed_eps", "foot_radius", "delay", "saturate_effort")

    def __init__(self, interface, robust_grid=False, **params):
        """robust_grid: opt into the SQP time grid's robust minimum step (ST_GRID_DT_MIN = QM_GRID_DT_MIN_ROBUST, include/qmhip_layout.h) — for long fixed-rate loops whose
        1 ms observation raster can land within weakEpsilon of a gait event; the default keeps [upstream]'s 10 * limitEpsilon"""
        self.itf = interface
        self.lib = interface.lib
        self.B = 0
        if robust_grid:
            interface.set_setting(L.ST_GRID_DT_MIN, L.QM_GRID_DT_MIN_ROBUST)
        if params:
            self.set_params(**params)

    def set_params(self, **params):
        cur = dict(contact_stiffness=4.0e4, contact_damping=200.0, friction=0.8, friction_speed_eps=1.0e-2, foot_radius=0.02, delay=0.009, saturate_effort=1.0)
        cur.update(getattr(self, "params", {}))
        for k in params:
            if k not in cur:
                raise KeyError(k)
        cur.update(params); self.params = cur
        v = _f([float(cur[k]) for k in self.PARAMS])
        self.itf._check(self.lib.qmhip_sim_set_params(self.itf.h, _p(v), 7), "qmhip_sim_set_params")

    def reset(self, q, v, time=0.0):
        q = _f(q); B = q.shape[0]; v = _f(v, (B, 24)); t = _f(np.broadcast_to(time, (B,)))
        assert q.shape == (B, 24)
        self.itf._check(self.lib.qmhip_sim_reset(self.itf.h, B, _p(q), _p(v), _p(t)), "qmhip_sim_reset")
        self.B = B

    def setCommand(self, posDes, velDes, kp, kd, ff):
        B = self.B; a = [_f(np.broadcast_to(x, (B, 18))) for x in (posDes, velDes, kp, kd, ff)]
        self.itf._check(self.lib.qmhip_sim_set_command(self.itf.h, B, *[_p(x) for x in a]), "qmhip_sim_set_command")

    def step(self, period, n_substeps=2, download=True):
        B = self.B; rbd = np.zeros((B, 55)) if download else None; contact = np.zeros((B, 4), np.int32) if download else None
        self.itf._check(self.lib.qmhip_sim_step(self.itf.h, B, C.c_double(period), int(n_substeps), _p(rbd), _pi(contact)), "qmhip_sim_step")
        return rbd, contact

    def set_controller(self, kind):
        """0: qm::QMController, 1: qm::QMMpcController (HierarchicalMpcWbc + arm position commands at 100 Hz); call before reset()"""
        self.itf._check(self.lib.qmhip_sim_set_controller(self.itf.h, int(kind)), "qmhip_sim_set_controller")

    def closed_loop(self, n_ticks, period, horizon, n_substeps=2, mpc_every=10, arm_kp=0.0, arm_kd=0.5, pipelined=False):
        """n_ticks of [state estimate -> MPC every mpc_every ticks -> policy -> WBC -> updateControlLaw -> simulation step] on the device (QMController::update);
        arm gains default to the reference's dynamic-reconfigure defaults (qm_controllers/cfg/weight.cfg:7-8)"""
        fn = self.lib.qmhip_closed_loop_sim_pipelined if pipelined else self.lib.qmhip_closed_loop_sim     # pipelined: the MPC beside the ticks, one period of latency
        self.itf._check(fn(self.itf.h, self.B, int(n_ticks), C.c_double(period), int(n_substeps), int(mpc_every), C.c_double(horizon), C.c_double(arm_kp), C.c_double(arm_kd)), "qmhip_closed_loop_sim")

    def state(self):
        B = self.B; q = np.zeros((B, 24)); v = np.zeros((B, 24)); t = np.zeros(B); f = np.zeros((B, 12)); st = np.zeros(B, np.int32)
        self.itf._check(self.lib.qmhip_sim_get_state(self.itf.h, B, _p(q), _p(v), _p(t), _p(f), _pi(st)), "qmhip_sim_get_state")
        return dict(q=q, v=v, time=t, force=f, status=st)


GAIT_MAX_PHASES, GAIT_EVENT_SLOTS = 16, 256
CMD_NONE, CMD_VEL, CMD_EE_VEL, CMD_EE_GOAL = 0, 1, 2, 3


class _TargetParams(C.Structure):
    _fields_ = [("time_to_target", C.c_double), ("target_displacement_velocity", C.c_double), ("target_rotation_velocity", C.c_double), ("com_height", C.c_double)]


class GaitSchedule:
    """B device-resident copies of the reference's GaitSchedule ([upstream ocs2_legged_robot]; built at QMInterface.cpp:455-480) plus the
    template table GaitJoyPublisher loads from gait.info (GaitJoyPublisher.cpp:17-33).  Method names follow the upstream class."""

    def __init__(self, interface, gaits, batch, initial_event_times=(0.5,), initial_mode_sequence=(15, 15), default_gait="stance"):
        """gaits: {name: {"modeSequence": [...], "switchingTimes": [...]}} (scenarios.load_gaits()); initial schedule = reference.info:28-39."""
        self.itf = interface; self.lib = interface.lib; self.B = batch
        self.names = list(gaits.keys())
        G = len(self.names)
        n_ph = np.zeros(G, dtype=np.int32); times = np.zeros((G, GAIT_MAX_PHASES + 1)); modes = np.zeros((G, GAIT_MAX_PHASES), dtype=np.int32)
        for g, name in enumerate(self.names):
            seq, sw = gaits[name]["modeSequence"], gaits[name]["switchingTimes"]
            if len(seq) > GAIT_MAX_PHASES or len(sw) != len(seq) + 1:
                raise ValueError("gait template %r: at most %d phases, switchingTimes one longer than modeSequence" % (name, GAIT_MAX_PHASES))
            n_ph[g] = len(seq); times[g, :len(sw)] = sw; modes[g, :len(seq)] = seq
        self.itf._check(self.lib.qmhip_gait_set_templates(self.itf.h, G, _pi(n_ph), _p(times), _pi(modes)), "qmhip_gait_set_templates")
        ev = _f(initial_event_times); mo = np.ascontiguousarray(initial_mode_sequence, dtype=np.int32)
        self.itf._check(self.lib.qmhip_gait_reset(self.itf.h, batch, len(ev), _p(ev), _pi(mo), self.names.index(default_gait)), "qmhip_gait_reset")

    def insertModeSequenceTemplate(self, gait, start_time, final_time):
        """gait: one name for every instance, or a list of names / None per instance."""
        B = self.B
        req = [gait] * B if (gait is None or isinstance(gait, str)) else list(gait)
        ids = np.array([-1 if g is None else self.names.index(g) for g in req], dtype=np.int32)
        st = _f(np.broadcast_to(np.asarray(start_time, dtype=float), (B,)).copy()); fi = _f(np.broadcast_to(np.asarray(final_time, dtype=float), (B,)).copy())
        self.itf._check(self.lib.qmhip_gait_insert_template(self.itf.h, B, _pi(ids), _p(st), _p(fi)), "qmhip_gait_insert_template")

    def preSolverRun(self, gait, init_time, horizon):
        """GaitReceiver::preSolverRun [upstream]: a received template is inserted at the end of the current horizon; the upstream call passes the
        horizon LENGTH as the tiling bound (insertModeSequenceTemplate(template, finalTime, finalTime − initTime))."""
        init_time = np.broadcast_to(np.asarray(init_time, dtype=float), (self.B,))
        final = init_time + horizon
        self.insertModeSequenceTemplate(gait, final, final - init_time)

    def updateSolverSchedule(self, horizon):
        """SwitchedModelReferenceManager::modifyReferences [upstream]: getModeSchedule(t0 − T, t0 + 2T) for the resident observation times;
        the result becomes the mode schedule of the next MPC iteration."""
        self.itf._check(self.lib.qmhip_gait_update_resident(self.itf.h, self.B, C.c_double(horizon)), "qmhip_gait_update_resident")

    def download(self):
        B = self.B
        n = np.zeros(B, dtype=np.int32); ev = np.zeros((B, GAIT_EVENT_SLOTS)); mo = np.zeros((B, GAIT_EVENT_SLOTS + 1), dtype=np.int32); tp = np.zeros(B, dtype=np.int32); st = np.zeros(B, dtype=np.int32)
        self.itf._check(self.lib.qmhip_gait_download(self.itf.h, B, _pi(n), _p(ev), _pi(mo), _pi(tp), _pi(st)), "qmhip_gait_download")
        return dict(n=n, event_times=ev, mode_sequence=mo, template=tp, status=st)

    def solver_schedule(self):
        B, ne = self.B, self.itf.max_events
        ev = np.zeros((B, ne)); mo = np.zeros((B, ne + 1), dtype=np.int32)
        self.itf._check(self.lib.qmhip_schedule_download(self.itf.h, B, _p(ev), _pi(mo)), "qmhip_schedule_download")
        return ev, mo


class TargetTrajectoriesPublisher:
    """The command callbacks of QmTargetTrajectoriesInteractiveMarker for B robots (QmTargetTrajectoriesPublisher.h:75-112,
    QmTargetTrajectoriesPublisher_node.cpp:44-208): commands in, the solver's resident 2-knot TargetTrajectories out."""

    def __init__(self, interface, batch, time_to_target=None, target_displacement_velocity=0.3, target_rotation_velocity=0.1, com_height=0.4,
                 last_ee_target=(0.52, 0.09, 0.44, 0.5, -0.5, 0.5, -0.5)):
        self.itf = interface; self.lib = interface.lib; self.B = batch
        T = interface.settings_blob[996] if time_to_target is None else time_to_target       # mpc.timeHorizon
        self.params = _TargetParams(T, target_displacement_velocity, target_rotation_velocity, com_height)
        self.itf._check(self.lib.qmhip_target_reset(self.itf.h, batch, _p(_f(last_ee_target, (7,)))), "qmhip_target_reset")

    def publish(self, kind, cmd, ee_state=None, ee_through_float=False):
        """kind[B] in {CMD_NONE, CMD_VEL, CMD_EE_VEL, CMD_EE_GOAL}; cmd[B][7]; ee_state[B][7] (None: forward kinematics of the resident x0)."""
        B = self.B
        kd = np.ascontiguousarray(np.broadcast_to(np.asarray(kind, dtype=np.int32), (B,)), dtype=np.int32)
        cm = _f(cmd, (B, 7)); ee = None if ee_state is None else _f(ee_state, (B, 7))
        self.itf._check(self.lib.qmhip_target_from_command(self.itf.h, B, _pi(kd), _p(cm), _p(ee), int(bool(ee_through_float)), C.byref(self.params)), "qmhip_target_from_command")

    def download(self):
        B, nr = self.B, self.itf.max_ref_knots
        rt = np.zeros((B, nr)); rx = np.zeros((B, nr, 37)); le = np.zeros((B, 7))
        self.itf._check(self.lib.qmhip_target_download(self.itf.h, B, _p(rt), _p(rx), _p(le)), "qmhip_target_download")
        return rt, rx, le
