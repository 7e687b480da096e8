"""oracle/pyoracle.py — TEST INFRASTRUCTURE: ctypes binding of oracle/_build/libqm_oracle.so.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "libqm_oracle.so")
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def load_blobs():
    """the ORACLE's own model / settings blobs: written by the numpy front-end oracle/front.py (tools/gen_blobs.py), independent of the
    product's C++ ingestion, whose blobs (qm_control_amd/data) the product runs on"""
    return np.load(os.path.join(_HERE, "data", "model_blob.npy")), np.load(os.path.join(_HERE, "data", "settings_blob.npy"))


def build(force=False):
    if force or not os.path.exists(_LIB):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB


def _p(a):
    return a.ctypes.data_as(_dp)


def _pi(a):
    return a.ctypes.data_as(_ip)


class Oracle:
    MAXN = 512

    def __init__(self, model_blob, settings_blob):
        build()
        self.lib = C.CDLL(_LIB)
        L = self.lib
        L.qmo_create.restype = C.c_void_p
        L.qmo_create.argtypes = [_dp, _dp]
        L.qmo_swing_zvel.restype = C.c_double
        L.qmo_swing_zvel.argtypes = [C.c_void_p, C.c_int, C.c_double]
        L.qmo_mode_at.argtypes = [C.c_void_p, C.c_double]
        L.qmo_destroy.argtypes = [C.c_void_p]
        self.mb = np.ascontiguousarray(model_blob, dtype=np.float64)
        self.st = np.ascontiguousarray(settings_blob, dtype=np.float64)
        self.h = C.c_void_p(L.qmo_create(_p(self.mb), _p(self.st)))

    def __del__(self):
        try:
            self.lib.qmo_destroy(self.h)
        except Exception:
            pass

    def set_threads(self, n):
        """worker threads over shooting nodes (LQ approximation, line-search evaluation) — sqp.nThreads of task.info:77"""
        self.lib.qmo_set_threads(C.c_int(n))

    def set_full_seeding(self, on):
        """tests only (process-wide): True = every Jacobian from the full 60-slot forward mode of rounds 1-4, False = the seeded evaluation (default, oracle/src/ocp.h)"""
        self.lib.qmo_set_full_seeding(C.c_int(int(bool(on))))

    def phase_ms(self):
        """wall time of the last mpc_step: [LQ approximation + projection, Riccati solve, line search] in ms"""
        ms = np.zeros(3); self.lib.qmo_phase_ms(self.h, _p(ms)); return ms

    def set_setting(self, idx, v):
        """one settings slot of this oracle (the device side's qmhip_set_setting), e.g. ST_GRID_DT_MIN for the fixed-rate loops; returns the previous value"""
        self.lib.qmo_get_setting.restype = C.c_double
        old = self.lib.qmo_get_setting(self.h, C.c_int(idx))
        self.lib.qmo_set_setting(self.h, C.c_int(idx), C.c_double(v))
        return old

    # ---- probes ----
    def flow_map(self, x, u, jac=False):
        x = np.ascontiguousarray(x, float); u = np.ascontiguousarray(u, float)
        f = np.zeros(30)
        if jac:
            A = np.zeros((30, 30)); B = np.zeros((30, 30))
            self.lib.qmo_flow_map(self.h, _p(x), _p(u), _p(f), _p(A), _p(B))
            return f, A, B
        self.lib.qmo_flow_map(self.h, _p(x), _p(u), _p(f), None, None)
        return f

    def foot_pos_vel(self, x, u, i):
        x = np.ascontiguousarray(x, float); u = np.ascontiguousarray(u, float)
        p = np.zeros(3); v = np.zeros(3)
        self.lib.qmo_foot_pos_vel(self.h, _p(x), _p(u), C.c_int(i), _p(p), _p(v))
        return p, v

    def ee_pose_error(self, x, pref, qref):
        x = np.ascontiguousarray(x, float); pref = np.ascontiguousarray(pref, float); qref = np.ascontiguousarray(qref, float)
        g = np.zeros(6)
        self.lib.qmo_ee_pose_error(self.h, _p(x), _p(pref), _p(qref), _p(g))
        return g

    def frame_pose(self, q, f):
        q = np.ascontiguousarray(q, float); p = np.zeros(3); R = np.zeros((3, 3))
        self.lib.qmo_frame_pose(self.h, _p(q), C.c_int(f), _p(p), _p(R))
        return p, R

    def time_grid(self, t0, tf, dt, ev, dt_min=10.0 * 2.220446049250313e-16):
        """[upstream timeDiscretizationWithEvents]; dt_min defaults to upstream's 10 * limitEpsilon"""
        ev = np.ascontiguousarray(ev, float)
        t = np.zeros(self.MAXN); e = np.zeros(self.MAXN, np.int32)
        n = self.lib.qmo_time_grid(C.c_double(t0), C.c_double(tf), C.c_double(dt), C.c_int(len(ev)), _p(ev), C.c_int(self.MAXN), _p(t), _pi(e), C.c_double(dt_min))
        assert n > 0
        return t[:n].copy(), e[:n].copy()

    # ---- problem data ----
    def set_schedule(self, ev, modes):
        ev = np.ascontiguousarray(ev, float); modes = np.ascontiguousarray(modes, np.int32)
        assert len(modes) == len(ev) + 1
        return self.lib.qmo_set_schedule(self.h, C.c_int(len(ev)), _p(ev), _pi(modes))

    def set_target(self, t, x37):
        t = np.ascontiguousarray(t, float); x37 = np.ascontiguousarray(x37, float)
        self.lib.qmo_set_target(self.h, C.c_int(len(t)), _p(t), _p(x37))

    def swing_zvel(self, leg, t):
        return self.lib.qmo_swing_zvel(self.h, C.c_int(leg), C.c_double(t))

    def mode_at(self, t):
        return self.lib.qmo_mode_at(self.h, C.c_double(t))

    def desired_state(self, t):
        x = np.zeros(37); p = np.zeros(3); q = np.zeros(4)
        self.lib.qmo_desired_state(self.h, C.c_double(t), _p(x), _p(p), _p(q))
        return x, p, q

    # ---- MPC ----
    def mpc_step(self, t0, tf, x0, warm=False):
        """one SQP iteration; warm=True: initial guess from this oracle's previous solution (cold start if there is none);
        warm="iterate": one more iteration on the iterate of the last call (sqp.sqpIteration > 1)"""
        x0 = np.ascontiguousarray(x0, float)
        n = C.c_int(0)
        nt = np.zeros(self.MAXN); ne = np.zeros(self.MAXN, np.int32); nm = np.zeros(self.MAXN, np.int32)
        xo = np.zeros((self.MAXN, 30)); uo = np.zeros((self.MAXN, 30)); perf = np.zeros(10)
        fn = self.lib.qmo_mpc_iterate if warm == "iterate" else (self.lib.qmo_mpc_step_warm if warm else self.lib.qmo_mpc_step)
        rc = fn(self.h, C.c_double(t0), C.c_double(tf), _p(x0), C.c_int(self.MAXN), C.byref(n), _p(nt), _pi(ne), _pi(nm), _p(xo), _p(uo), _p(perf))
        if rc != 0:
            raise RuntimeError("oracle mpc_step failed rc=%d" % rc)
        k = n.value
        return dict(t=nt[:k].copy(), ev=ne[:k].copy(), mode=nm[:k].copy(), x=xo[:k].copy(), u=uo[:k].copy(), perf=perf,
                    alpha=perf[8], armijo=perf[9], ls_trials=self.lib.qmo_ls_trials(self.h), warn=self.lib.qmo_last_warn(self.h))

    def ls_trace(self):
        """diagnostics: rows {alpha, merit, theta, filter branch, accepted} of the last SQP iteration's line search; row 0 is the baseline {0, merit, theta0, armijo, -1}"""
        out = np.zeros((20, 5)); self.lib.qmo_ls_trace.restype = C.c_int
        k = self.lib.qmo_ls_trace(self.h, _p(out), C.c_int(20))
        return out[:k].copy()

    def ilqr_step(self, t0, tf, x0, warm=False):
        """one discrete iLQR iteration (oracle/src/ilqr.h): same result dict as mpc_step"""
        x0 = np.ascontiguousarray(x0, float); n = C.c_int(0)
        nt = np.zeros(self.MAXN); ne = np.zeros(self.MAXN, np.int32); nm = np.zeros(self.MAXN, np.int32)
        xo = np.zeros((self.MAXN, 30)); uo = np.zeros((self.MAXN, 30)); perf = np.zeros(10)
        rc = self.lib.qmo_ilqr_step(self.h, C.c_int(int(bool(warm))), C.c_double(t0), C.c_double(tf), _p(x0), C.c_int(self.MAXN), C.byref(n), _p(nt), _pi(ne), _pi(nm), _p(xo), _p(uo), _p(perf))
        if rc != 0:
            raise RuntimeError("oracle ilqr_step failed rc=%d" % rc)
        k = n.value
        return dict(t=nt[:k].copy(), ev=ne[:k].copy(), mode=nm[:k].copy(), x=xo[:k].copy(), u=uo[:k].copy(), perf=perf, alpha=perf[8], armijo=perf[9], ls_trials=self.lib.qmo_ls_trials(self.h), warn=self.lib.qmo_last_warn(self.h))

    def ipm_step(self, t0, tf, x0, mode="cold"):
        """one iteration of the hard-inequality interior-point solver (oracle/src/ipm.h); mode: "cold", "warm" (initial guess from the previous solution) or
        "iterate" (one more iteration on the last call's iterate, slack / dual / barrier kept).  Result dict as mpc_step + barrier, alpha_primal_max, alpha_dual_max, alpha_dual"""
        x0 = np.ascontiguousarray(x0, float); n = C.c_int(0)
        nt = np.zeros(self.MAXN); ne = np.zeros(self.MAXN, np.int32); nm = np.zeros(self.MAXN, np.int32)
        xo = np.zeros((self.MAXN, 30)); uo = np.zeros((self.MAXN, 30)); perf = np.zeros(10)
        rc = self.lib.qmo_ipm_step(self.h, C.c_int({"cold": 0, "warm": 1, "iterate": 2}[mode]), C.c_double(t0), C.c_double(tf), _p(x0), C.c_int(self.MAXN), C.byref(n), _p(nt), _pi(ne), _pi(nm), _p(xo), _p(uo), _p(perf))
        if rc != 0:
            raise RuntimeError("oracle ipm_step failed rc=%d" % rc)
        k = n.value; info = np.zeros(5); self.lib.qmo_ipm_info(self.h, _p(info))
        return dict(t=nt[:k].copy(), ev=ne[:k].copy(), mode=nm[:k].copy(), x=xo[:k].copy(), u=uo[:k].copy(), perf=perf, alpha=perf[8], armijo=perf[9], ls_trials=self.lib.qmo_ls_trials(self.h),
                    warn=self.lib.qmo_last_warn(self.h), barrier=info[0], alpha_primal_max=info[1], alpha_dual_max=info[2], alpha_dual=info[3])

    def ipm_node(self, i):
        """interval i of the last ipm_step: slack / dual after the step, directions, linearised inequality rows, dx / du, uncondensed cost blocks"""
        z = lambda *s: np.zeros(s)
        d = dict(slack=z(28), dual=z(28), dslack=z(28), ddual=z(28), h=z(28), Hx=z(28, 30), Hu=z(28, 30), on=np.zeros(28, np.int32), dx=z(30), du=z(30), Q=z(30, 30), R=z(30, 30), q=z(30), r=z(30))
        rc = self.lib.qmo_ipm_node(self.h, C.c_int(i), _p(d["slack"]), _p(d["dual"]), _p(d["dslack"]), _p(d["ddual"]), _p(d["h"]), _p(d["Hx"]), _p(d["Hu"]), _pi(d["on"]), _p(d["dx"]), _p(d["du"]),
                                   _p(d["Q"]), _p(d["R"]), _p(d["q"]), _p(d["r"]))
        if rc != 0:
            raise IndexError(i)
        return d

    def terminal_lq(self):
        Q = np.zeros((30, 30)); q = np.zeros(30); self.lib.qmo_terminal_lq(self.h, _p(Q), _p(q)); return dict(Q=Q, q=q)

    def node_lq(self, i):
        z = lambda *s: np.zeros(s)
        d = dict(A=z(30, 30), B=z(30, 30), b=z(30), Q=z(30, 30), R=z(30, 30), P=z(30, 30), q=z(30), r=z(30), scal=z(4), C=z(16, 30), D=z(16, 30), e=z(16))
        rc = self.lib.qmo_get_node_lq(self.h, C.c_int(i), *[_p(d[k]) for k in ("A", "B", "b", "Q", "R", "P", "q", "r", "scal", "C", "D", "e")])
        assert rc == 0
        d["c"], d["dt"], d["nc"], d["event"] = d["scal"][0], d["scal"][1], int(d["scal"][2]), int(d["scal"][3])
        return d

    def node_proj(self, i):
        z = lambda *s: np.zeros(s)
        keys = ("Px", "Pu", "Pe", "Ap", "Bp", "bp", "Qp", "Rp", "Pp", "qp", "rp", "scal", "K", "kff")
        d = dict(Px=z(30, 30), Pu=z(30, 30), Pe=z(30), Ap=z(30, 30), Bp=z(30, 30), bp=z(30), Qp=z(30, 30), Rp=z(30, 30), Pp=z(30, 30), qp=z(30), rp=z(30), scal=z(2), K=z(30, 30), kff=z(30))
        rc = self.lib.qmo_get_node_proj(self.h, C.c_int(i), *[_p(d[k]) for k in keys])
        assert rc == 0
        d["cp"], d["m"] = d["scal"][0], int(d["scal"][1])
        return d

    def terminal(self):
        Q = np.zeros((30, 30)); q = np.zeros(30); c = C.c_double(0)
        self.lib.qmo_get_terminal(self.h, _p(Q), _p(q), C.byref(c))
        return Q, q, c.value

    def step(self, n):
        dx = np.zeros((n, 30)); du = np.zeros((n, 30))
        self.lib.qmo_get_step(self.h, _p(dx), _p(du))
        return dx, du[: n - 1]

    def eval_policy(self, t):
        x = np.zeros(30); u = np.zeros(30); m = C.c_int(0)
        self.lib.qmo_eval_policy(self.h, C.c_double(t), _p(x), _p(u), C.byref(m))
        return x, u, m.value

    # ---- WBC ----
    def wbc_reset(self):
        self.lib.qmo_wbc_reset(self.h)

    def wbc_set_input_last(self, u):
        u = np.ascontiguousarray(u, float)
        self.lib.qmo_wbc_set_input_last(self.h, _p(u))

    def rbd_from_q(self, q, v=None):
        q = np.ascontiguousarray(q, float); rbd = np.zeros(55)
        vv = None if v is None else np.ascontiguousarray(v, float)
        self.lib.qmo_rbd_from_q(self.h, _p(q), None if vv is None else _p(vv), _p(rbd))
        return rbd

    # ---- batched-plant restatement (oracle/src/sim.h) ----
    def sim_params(self, **params):
        cur = dict(contact_stiffness=4.0e4, contact_damping=200.0, friction=0.8, friction_speed_eps=1.0e-2, foot_radius=0.02, delay=0.009, saturate_effort=1.0)
        cur.update(params)
        v = np.array([cur[k] for k in ("contact_stiffness", "contact_damping", "friction", "friction_speed_eps", "foot_radius", "delay", "saturate_effort")], float)
        self.lib.qmo_sim_params(self.h, _p(v))

    def sim_reset(self, q, v, time=0.0):
        q = np.ascontiguousarray(q, float); v = np.ascontiguousarray(v, float)
        self.lib.qmo_sim_reset(self.h, _p(q), _p(v), C.c_double(time))

    def sim_command(self, pos, vel, kp, kd, ff):
        a = [np.ascontiguousarray(np.broadcast_to(x, (18,)), float) for x in (pos, vel, kp, kd, ff)]
        self.lib.qmo_sim_command(self.h, *[_p(x) for x in a])

    def sim_step(self, period, nsub=2):
        q = np.zeros(24); v = np.zeros(24); t = C.c_double(0); f = np.zeros(12); c = np.zeros(4, np.int32)
        st = self.lib.qmo_sim_step(self.h, C.c_double(period), C.c_int(nsub), _p(q), _p(v), C.byref(t), _p(f), _pi(c))
        return dict(q=q, v=v, time=t.value, force=f, contact=c, status=st, rbd=self.rbd_from_q(q, v))

    def wbc(self, xdes, udes, rbd, mode, period, time, mpc_variant=False, debug=False):
        xdes = np.ascontiguousarray(xdes, float); udes = np.ascontiguousarray(udes, float); rbd = np.ascontiguousarray(rbd, float)
        out = np.zeros(54); st = np.zeros(3, np.int32); dbg = np.zeros(24 * 4 + 6 + 24 + 36 * 3 + 576 + 288 + 288)
        self.lib.qmo_wbc(self.h, _p(xdes), _p(udes), _p(rbd), C.c_int(mode), C.c_double(period), C.c_double(time), C.c_int(int(mpc_variant)), _p(out), _pi(st), _p(dbg))
        if not debug:
            return out, st
        o = 0
        d = {}
        for k, n in (("qMeas", 24), ("vMeas", 24), ("qDes", 24), ("vDes", 24), ("baseAcc", 6), ("nle", 24), ("x0", 36), ("x1", 36), ("x2", 36), ("M", 576), ("J", 288), ("dJ", 288)):
            d[k] = dbg[o:o + n].copy(); o += n
        d["M"] = d["M"].reshape(24, 24); d["J"] = d["J"].reshape(12, 24); d["dJ"] = d["dJ"].reshape(12, 24)
        return out, st, d


def _wbc_tasks(self):
    """the three priority levels of the last wbc() call as dicts A, b, D, f (what WbcBase hands to HoQp)"""
    out = []
    for level in range(3):
        dims = np.zeros(2, np.int32); A = np.zeros((64, 36)); b = np.zeros(64); D = np.zeros((128, 36)); f = np.zeros(128)
        rc = self.lib.qmo_wbc_task(self.h, C.c_int(level), _pi(dims), _p(A), _p(b), _p(D), _p(f))
        assert rc == 0
        out.append(dict(A=A[:dims[0]].copy(), b=b[:dims[0]].copy(), D=D[:dims[1]].copy(), f=f[:dims[1]].copy()))
    return out


Oracle.wbc_tasks = _wbc_tasks


def batch_step(model_blob, settings_blob, nthreads, t0, horizon, x0, ref_t, ref_x, ev, modes, period, time, traj_nodes=0):
    """cpu_baseline driver: B instances of (MPC step + policy eval at t0 + WBC) over nthreads.
    traj_nodes > 0: a fifth return value dict(num_nodes[B], t, event, mode [B][traj_nodes], x, u [B][traj_nodes][30]) with every instance's whole primal solution."""
    build()
    lib = C.CDLL(_LIB)
    mb = np.ascontiguousarray(model_blob, float); st = np.ascontiguousarray(settings_blob, float)
    t0 = np.ascontiguousarray(t0, float); x0 = np.ascontiguousarray(x0, float)
    ref_t = np.ascontiguousarray(ref_t, float); ref_x = np.ascontiguousarray(ref_x, float)
    ev = np.ascontiguousarray(ev, float); modes = np.ascontiguousarray(modes, np.int32)
    B = x0.shape[0]; K = ref_t.shape[1]; nev = ev.shape[1]
    xf = np.zeros((B, 30)); uf = np.zeros((B, 30)); w = np.zeros((B, 54))
    bad = lib.qmo_batch_step(_p(mb), _p(st), C.c_int(B), C.c_int(nthreads), _p(t0), C.c_double(horizon), _p(x0), C.c_int(K), _p(ref_t), _p(ref_x),
                             C.c_int(nev), _p(ev), _pi(modes), C.c_double(period), C.c_double(time), _p(xf), _p(uf), _p(w), *_traj_args(B, traj_nodes))
    if traj_nodes:
        return bad, xf, uf, w, _traj_last
    return bad, xf, uf, w


def hoqp(tasks):
    """qm::HoQp cascade on arbitrary tasks [dict(A, b, D, f), ...] from the highest priority down (oracle/src/wbc.h: solveHoLevel, general stacking included).
    Returns (x of the last level, status per level, active-set iterations per level)."""
    build()
    lib = C.CDLL(_LIB)
    n = int(np.asarray(tasks[0]["A"]).shape[1]) if np.asarray(tasks[0]["A"]).size else int(np.asarray(tasks[0]["D"]).shape[1])
    ma = np.array([np.asarray(t["A"]).reshape(-1, n).shape[0] for t in tasks], np.int32); md = np.array([np.asarray(t["D"]).reshape(-1, n).shape[0] for t in tasks], np.int32)
    cat = lambda k: np.ascontiguousarray(np.concatenate([np.asarray(t[k], float).ravel() for t in tasks] + [np.zeros(1)]))
    A, b, D, f = cat("A"), cat("b"), cat("D"), cat("f")
    x = np.zeros(n); st = np.zeros(len(tasks), np.int32); it = np.zeros(len(tasks), np.int32)
    lib.qmo_hoqp(C.c_int(len(tasks)), C.c_int(n), _pi(ma), _pi(md), _p(A), _p(b), _p(D), _p(f), _p(x), _pi(st), _pi(it))
    return x, st, it


_traj_last = None


def _traj_args(B, maxn):
    global _traj_last
    if not maxn:
        _traj_last = None
        return [C.c_int(0)] + [None] * 6
    tr = dict(num_nodes=np.zeros(B, np.int32), t=np.zeros((B, maxn)), event=np.zeros((B, maxn), np.int32), mode=np.zeros((B, maxn), np.int32), x=np.zeros((B, maxn, 30)), u=np.zeros((B, maxn, 30)))
    _traj_last = tr
    return [C.c_int(maxn), _pi(tr["num_nodes"]), _p(tr["t"]), _pi(tr["event"]), _pi(tr["mode"]), _p(tr["x"]), _p(tr["u"])]
