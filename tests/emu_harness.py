"""tests/emu_harness.py — TEST INFRASTRUCTURE: ctypes binding of the host-emulated kernels (tests/emu)."""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "emu", "_build", "libqm_emu.so")
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def build():
    subprocess.check_call(["make", "-C", os.path.join(_HERE, "emu"), "-s"])
    return _LIB


def _p(a):
    return a.ctypes.data_as(_dp)


def _pi(a):
    return a.ctypes.data_as(_ip)


class Emu:
    def __init__(self, mb, st, Bmax, nmax, nref, nev):
        build()
        self.lib = C.CDLL(_LIB)
        self.lib.emu_create.restype = C.c_void_p
        self.lib.emu_buffer.restype = C.c_void_p
        self.lib.emu_buffer.argtypes = [C.c_void_p, C.c_char_p]
        self.lib.emu_destroy.argtypes = [C.c_void_p]
        self.mb = np.ascontiguousarray(mb, float); self.st = np.ascontiguousarray(st, float)
        self.Bmax, self.nmax, self.nref, self.nev = Bmax, nmax, nref, nev
        self.h = C.c_void_p(self.lib.emu_create(_p(self.mb), _p(self.st), Bmax, nmax, nref, nev))
        self.SR, self.DBG, self.PF = (self.lib.emu_sizes(i) for i in range(3))
        self.B = 0

    def __del__(self):
        try:
            self.lib.emu_destroy(self.h)
        except Exception:
            pass

    def mpc_step(self, cfg, max_trials=14, batch=None):
        B = cfg["B"] if batch is None else batch
        self.B = B
        a = lambda k, t=float: np.ascontiguousarray(cfg[k][:B], t)
        t0, x0, rt, rx, ev, mo = a("t0"), a("x0"), a("ref_t"), a("ref_x"), a("ev"), a("modes", np.int32)
        assert rt.shape[1] == self.nref and ev.shape[1] == self.nev
        return self.lib.emu_mpc_step(self.h, C.c_int(B), _p(t0), _p(x0), _p(rt), _p(rx), _p(ev), _pi(mo), C.c_double(cfg["horizon"]), C.c_int(max_trials))

    def mpc_iterate(self, max_trials=14):
        """one more iteration on the committed iterate (multi-iteration solves)"""
        return self.lib.emu_mpc_iterate(self.h, C.c_int(self.B), C.c_int(max_trials))

    def set_solver(self, solver):
        self.lib.emu_set_solver(self.h, C.c_int(solver))

    def set_lq_debug(self, on):
        """False: the PRODUCT instance of the LQ kernel runs (qm_lq_kernel: no debug records); True (default): qm_lq_dbg_kernel"""
        self.lib.emu_set_lq_debug(self.h, C.c_int(int(bool(on))))

    def set_r_dense(self, on):
        """True: the dense instances of the trial-evaluation kernel and of K1b's R0 (u - u_nom) run although the input weight is block diagonal (qmhip_debug_set "r_dense")"""
        self.lib.emu_set_r_dense(self.h, C.c_int(int(bool(on))))

    def r_blocks(self):
        return bool(self.lib.emu_r_blocks(self.h))

    def set_riccati_skip(self, mask):
        """profiling / parity switch of the product (qmhip_debug_set "riccati_skip"): 16 | 4 = no backward stage, no rollout -> the stage records stay as K1b wrote them"""
        self.lib.emu_set_riccati_skip(self.h, C.c_int(mask))

    def grid_only(self, cfg, batch=None):
        """upload + K0 (time discretisation, modes, references, initial guess) without the SQP iteration"""
        B = cfg["B"] if batch is None else batch
        self.B = B
        a = lambda k, t=float: np.ascontiguousarray(cfg[k][:B], t)
        self.lib.emu_upload(self.h, C.c_int(B), _p(a("t0")), _p(a("x0")), _p(a("ref_t")), _p(a("ref_x")), _p(a("ev")), _pi(a("modes", np.int32)))
        self.lib.emu_grid(self.h, C.c_int(B), C.c_double(cfg["horizon"]))

    def mpc_step_warm(self, t0, x0, horizon, max_trials=14):
        """new observation + warm-started SQP iteration (inputs / schedule of the last mpc_step stay resident)"""
        t0 = np.ascontiguousarray(t0, float); x0 = np.ascontiguousarray(x0, float)
        return self.lib.emu_mpc_step_warm(self.h, C.c_int(self.B), _p(t0), _p(x0), C.c_double(horizon), C.c_int(max_trials))

    def advance(self, dt):
        self.lib.emu_advance(self.h, C.c_int(self.B), C.c_double(dt))

    # ---- reference / gait front-end ----
    def gait_setup(self, gaits, B, ev0=(0.5,), mode0=(15, 15), default="stance"):
        self.gait_names = list(gaits.keys()); self.B = B
        G = len(self.gait_names)
        n_ph = np.zeros(G, np.int32); times = np.zeros((G, 17)); modes = np.zeros((G, 16), np.int32)
        for g, name in enumerate(self.gait_names):
            seq, sw = gaits[name]["modeSequence"], gaits[name]["switchingTimes"]
            n_ph[g] = len(seq); times[g, :len(sw)] = sw; modes[g, :len(seq)] = seq
        self.lib.emu_gait_set_templates(self.h, G, _pi(n_ph), _p(times), _pi(modes))
        ev = np.ascontiguousarray(ev0, float); mo = np.ascontiguousarray(mode0, np.int32)
        self.lib.emu_gait_reset(self.h, B, len(ev), _p(ev), _pi(mo), self.gait_names.index(default))

    def gait_insert(self, names, start, final):
        B = self.B
        ids = np.array([-1 if g is None else self.gait_names.index(g) for g in names], np.int32)
        st = np.ascontiguousarray(np.broadcast_to(np.asarray(start, float), (B,))); fi = np.ascontiguousarray(np.broadcast_to(np.asarray(final, float), (B,)))
        self.lib.emu_gait_insert(self.h, B, _pi(ids), _p(st), _p(fi))

    def gait_update(self, t0, horizon):
        t0 = np.ascontiguousarray(t0, float)
        self.lib.emu_gait_update(self.h, self.B, _p(t0), C.c_double(horizon))

    def gait_download(self):
        B = self.B
        n = np.zeros(B, np.int32); ev = np.zeros((B, 256)); mo = np.zeros((B, 257), np.int32); tp = np.zeros(B, np.int32); st = np.zeros(B, np.int32)
        self.lib.emu_gait_download(self.h, B, _pi(n), _p(ev), _pi(mo), _pi(tp), _pi(st))
        return dict(n=n, event_times=ev, mode_sequence=mo, template=tp, status=st)

    def schedule_download(self):
        ev = np.zeros((self.B, self.nev)); mo = np.zeros((self.B, self.nev + 1), np.int32)
        self.lib.emu_schedule_download(self.h, self.B, _p(ev), _pi(mo))
        return ev, mo

    def target_reset(self, B, last7):
        self.B = B
        self.lib.emu_target_reset(self.h, B, _p(np.ascontiguousarray(last7, float)))

    def target_from_command(self, t0, x0, kind, cmd, ee, thru_float, T, vd, vr, ch):
        B = self.B
        a = lambda v, t=float: np.ascontiguousarray(v, t)
        self.lib.emu_target_from_command(self.h, B, _p(a(t0)), _p(a(x0)), _pi(a(kind, np.int32)), _p(a(cmd)), None if ee is None else _p(a(ee)), int(thru_float),
                                         C.c_double(T), C.c_double(vd), C.c_double(vr), C.c_double(ch))
        rt = np.zeros((B, self.nref)); rx = np.zeros((B, self.nref, 37)); last = np.zeros((B, 7))
        self.lib.emu_target_download(self.h, B, _p(rt), _p(rx), _p(last))
        return rt, rx, last

    def buf(self, name, shape, dtype=np.float64):
        ptr = self.lib.emu_buffer(self.h, name.encode())
        assert ptr, name
        n = int(np.prod(shape))
        ct = C.c_double if dtype == np.float64 else C.c_int
        arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), shape=(n,))
        return arr.reshape(shape).copy()

    # node-major [nmax][B][k] helpers (B = batch of the last call)
    def node_arr(self, name, k, dtype=np.float64):
        full = self.buf(name, (self.nmax * self.B * k,), dtype)
        return full.reshape(self.nmax, self.B, k) if k > 1 else full.reshape(self.nmax, self.B)

    def stage(self, b, i):
        full = self.buf("stage", (self.Bmax * self.nmax * self.SR,))
        return full.reshape(-1, self.SR)[b * self.nmax + i]

    def lqdbg(self, b, i):
        full = self.buf("lqdbg", (self.Bmax * self.nmax * self.DBG,))
        return full.reshape(-1, self.DBG)[b * self.nmax + i]

    # ---- policy / WBC ----
    def policy_eval(self, t):
        t = np.ascontiguousarray(t, float); B = len(t)
        xd = np.zeros((B, 30)); ud = np.zeros((B, 30)); mode = np.zeros(B, np.int32)
        self.lib.emu_policy_eval(self.h, C.c_int(B), _p(t), _p(xd), _p(ud), _pi(mode))
        return xd, ud, mode

    def wbc_reset(self):
        self.lib.emu_wbc_reset(self.h)

    def wbc_step(self, xd, ud, rbd, mode, period, time, variant=0):
        xd = np.ascontiguousarray(xd, float); B = xd.shape[0]
        ud = np.ascontiguousarray(ud, float); rbd = np.ascontiguousarray(rbd, float); mode = np.ascontiguousarray(mode, np.int32)
        time = np.ascontiguousarray(np.broadcast_to(time, (B,)), float)
        out = np.zeros((B, 54)); st = np.zeros((B, 3), np.int32); nd = self.lib.emu_sizes(5); dbg = np.zeros((B, nd))
        self.lib.emu_wbc_step(self.h, C.c_int(B), _p(xd), _p(ud), _p(rbd), _pi(mode), C.c_double(period), _p(time), C.c_int(variant), _p(out), _pi(st), _p(dbg))
        d = []
        for b in range(B):
            o = 0; e = {}
            for k, n in (("qMeas", 24), ("vMeas", 24), ("qDes", 24), ("vDes", 24), ("baseAcc", 6), ("nle", 24), ("x0", 36), ("x1", 36), ("x2", 36), ("M", 576), ("J", 288), ("dJv", 12)):
                e[k] = dbg[b, o:o + n].copy(); o += n
            e["M"] = e["M"].reshape(24, 24); e["J"] = e["J"].reshape(12, 24); d.append(e)
        return out, st, d

    # batched rigid-body plant
    def sim_params(self, **params):
        cur = dict(contact_stiffness=4.0e4, contact_damping=200.0, friction=0.8, friction_speed_eps=1.0e-2, foot_radius=0.02, delay=0.009, saturate_effort=1.0)
        cur.update(params)
        v = np.array([cur[k] for k in ("contact_stiffness", "contact_damping", "friction", "friction_speed_eps", "foot_radius", "delay", "saturate_effort")], float)
        self.lib.emu_sim_params(self.h, _p(v))

    def sim_reset(self, q, v, time=0.0):
        q = np.ascontiguousarray(q, float); B = q.shape[0]; v = np.ascontiguousarray(v, float); t = np.ascontiguousarray(np.broadcast_to(time, (B,)), float)
        self.lib.emu_sim_reset(self.h, C.c_int(B), _p(q), _p(v), _p(t)); self.simB = B

    def sim_command(self, pos, vel, kp, kd, ff):
        B = self.simB; cmd = np.concatenate([np.ascontiguousarray(np.broadcast_to(x, (B, 18)), float) for x in (pos, vel, kp, kd, ff)], axis=1)
        cmd = np.ascontiguousarray(cmd); self.lib.emu_sim_command(self.h, C.c_int(B), _p(cmd))

    def sim_set_controller(self, kind):
        self.lib.emu_sim_set_controller(self.h, C.c_int(kind))

    def closed_loop_sim(self, n_ticks, period, horizon, nsub=2, mpc_every=10, arm_kp=0.0, arm_kd=0.5, restart=False, pipelined=False):
        (self.lib.emu_closed_loop_sim_pipelined if pipelined else self.lib.emu_closed_loop_sim)(self.h, C.c_int(self.simB), C.c_int(n_ticks), C.c_double(period), C.c_int(nsub), C.c_int(mpc_every), C.c_double(horizon), C.c_double(arm_kp), C.c_double(arm_kd), C.c_int(int(restart)))

    def sim_state(self):
        B = self.simB; return dict(q=self.buf("sim_q", (B, 24)).copy(), v=self.buf("sim_v", (B, 24)).copy(), tau=self.buf("wbc_out", (B, 54)).copy()[:, 36:], wbc_status=self.buf("wbc_qp_status", (B, 3), np.int32).copy(), mpc_status=self.buf("status", (B,), np.int32).copy())

    def sim_step(self, period, nsub=2):
        B = self.simB; rbd = np.zeros((B, 55)); contact = np.zeros((B, 4), np.int32); q = np.zeros((B, 24)); v = np.zeros((B, 24)); f = np.zeros((B, 12)); st = np.zeros(B, np.int32)
        self.lib.emu_sim_step(self.h, C.c_int(B), C.c_double(period), C.c_int(nsub), _p(rbd), _pi(contact), _p(q), _p(v), _p(f), _pi(st))
        return dict(rbd=rbd, contact=contact, q=q, v=v, force=f, status=st)

    def control_step(self, cfg, batch=None):
        """upload + the benchmark's whole control step (MPC iteration, policy at t0, WBC on the synthetic measured state)"""
        B = cfg["B"] if batch is None else batch
        self.B = B
        a = lambda k, t=float: np.ascontiguousarray(cfg[k][:B], t)
        t0, x0, rt, rx, ev, mo = a("t0"), a("x0"), a("ref_t"), a("ref_x"), a("ev"), a("modes", np.int32)
        self.lib.emu_upload(self.h, C.c_int(B), _p(t0), _p(x0), _p(rt), _p(rx), _p(ev), _pi(mo))
        out = np.zeros((B, 54)); st = np.zeros((B, 3), np.int32); rbd = np.zeros((B, 55))
        self.lib.emu_control_step(self.h, C.c_int(B), C.c_double(cfg["horizon"]), C.c_double(cfg["period"]), C.c_double(cfg["time"]), _p(out), _pi(st), _p(rbd))
        return out, st, rbd


def hoqp(tasks):
    """the product's general HoQp kernel (csrc/kernels/k_hoqp.h, behind qmhip_hoqp_solve) on the host emulator: one cascade, tasks = [dict(A, b, D, f), ...]"""
    build(); lib = C.CDLL(_LIB)
    n = int(np.asarray(tasks[0]["A"]).shape[1]) if np.asarray(tasks[0]["A"]).size else int(np.asarray(tasks[0]["D"]).shape[1])
    ma = np.array([np.asarray(t["A"]).reshape(-1, n).shape[0] for t in tasks], np.int32); md = np.array([np.asarray(t["D"]).reshape(-1, n).shape[0] for t in tasks], np.int32)
    cat = lambda k: np.ascontiguousarray(np.concatenate([np.asarray(t[k], float).ravel() for t in tasks] + [np.zeros(1)]))
    A, b, D, f = cat("A"), cat("b"), cat("D"), cat("f")
    x = np.zeros(n); st = np.zeros(len(tasks), np.int32)
    rc = lib.emu_hoqp(C.c_int(1), C.c_int(len(tasks)), C.c_int(n), _pi(ma), _pi(md), _p(A), _p(b), _p(D), _p(f), _p(x), _pi(st))
    assert rc == 0
    return x, st
