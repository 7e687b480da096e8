"""Synthetic workloads C1–C5 of BASELINE.md §3 / SURVEY.md §8(d), as plain numpy arrays.

Everything here is input construction for the hot path: initial states, 2-knot target trajectories in
the layout produced by the reference's target publisher
(qm_controllers/src/QmTargetTrajectoriesPublisher_node.cpp:44-68) and contact-mode schedules built by
tiling the gait templates of qm_controllers/config/gait.info the way the reference's gait schedule
does it (event times accumulated by repeated f64 addition).
"""
import json
import os
import numpy as np

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")

STANCE, LF_RH, RF_LH = 15, 9, 6
# layout offsets (include/qmhip_layout.h)
MB_QLO, MB_QHI, MB_QNOM = 288, 306, 667
MB_ROBOTMASS, MB_INOM, MB_RNOM = 654, 655, 664
ST_XINIT, ST_SQP_DT = 930, 991
EE_NOMINAL_POS = np.array([0.52, 0.09, 0.38 + 0.4])          # QMController.cpp:107 (+ base height)
EE_NOMINAL_QUAT = np.array([0.5, -0.5, 0.5, -0.5])           # Quaternion(w=-.5,.5,-.5,.5).coeffs() -> xyzw
COM_HEIGHT = 0.4                                             # reference.info: comHeight


def load_blobs():
    mb = np.load(os.path.join(_DATA, "model_blob.npy"))
    st = np.load(os.path.join(_DATA, "settings_blob.npy"))
    return mb, st


def load_gaits():
    with open(os.path.join(_DATA, "gaits.json")) as fh:
        return json.load(fh)["gaits"]


def tile_gait(template_times, template_modes, t_start, t_until, first_mode=STANCE, last_mode=STANCE):
    """[first_mode | template tiled from t_start | last_mode]; event times by repeated addition.

    Mirrors GaitSchedule::tileModeSequenceTemplate [upstream]: eventTimes.push_back(eventTimes.back() +
    (T[i+1]-T[i])) until the last event >= t_until.
    """
    ev = [float(t_start)]
    modes = [first_mode]
    n = len(template_modes)
    while ev[-1] < t_until:
        for i in range(n):
            modes.append(int(template_modes[i]))
            ev.append(ev[-1] + (template_times[i + 1] - template_times[i]))
    modes.append(last_mode)
    # layout: modes[0] before ev[0]; modes[k] on (ev[k-1], ev[k]]; modes[-1] after the last event
    assert len(modes) == len(ev) + 1
    return np.array(ev), np.array(modes, dtype=np.int32)


def stance_schedule(t0, horizon):
    """mode 15 throughout: initialModeSchedule of reference.info:28-39 (STANCE,STANCE, one event)."""
    return np.array([t0 - 2.0 * horizon - 1.0]), np.array([STANCE, STANCE], dtype=np.int32)


def trot_schedule(t_until):
    g = load_gaits()["trot"]
    return tile_gait(g["switchingTimes"], g["modeSequence"], 0.0, t_until)


def trot_stance_trot_schedule(ts1, ts2, t_until):
    """trot from t=0, stance on [ts1, ts2], trot again from ts2 (C5)."""
    g = load_gaits()["trot"]
    T, Mo = g["switchingTimes"], g["modeSequence"]
    ev = [0.0]
    modes = [STANCE]
    i = 0
    while True:
        nxt = ev[-1] + (T[i % 2 + 1] - T[i % 2])
        modes.append(int(Mo[i % 2]))
        if nxt >= ts1:
            ev.append(float(ts1))
            break
        ev.append(nxt)
        i += 1
    modes.append(STANCE)
    ev.append(float(ts2))
    i = 0
    while ev[-1] < t_until:
        modes.append(int(Mo[i % 2]))
        ev.append(ev[-1] + (T[i % 2 + 1] - T[i % 2]))
        i += 1
    modes.append(STANCE)
    return np.array(ev), np.array(modes, dtype=np.int32)


def gait_schedule(name, t_until):
    """Any template of gait.info tiled from t = 0 until t_until (initial and final STANCE as the reference's schedule has them)."""
    g = load_gaits()[name]
    return tile_gait(g["switchingTimes"], g["modeSequence"], 0.0, t_until)


def gait_config(gait, batch=4, n_intervals=30, seed=7):
    """C3-style random instances walking `gait` (all 12 templates of qm_controllers/config/gait.info:1-255): same dict as make_config."""
    mb, st = load_blobs()
    xbar = st[ST_XINIT:ST_XINIT + 30].copy()
    qnom = mb[MB_QNOM:MB_QNOM + 18].copy()
    horizon = n_intervals * st[ST_SQP_DT]
    g = load_gaits()[gait]
    rng = np.random.default_rng(seed)
    d = np.zeros((batch, 30))
    d[:, 0:6] = rng.uniform(-0.05, 0.05, (batch, 6)); d[:, 6:9] = rng.uniform(-0.02, 0.02, (batch, 3)); d[:, 9:12] = rng.uniform(-0.03, 0.03, (batch, 3))
    d[:, 12:24] = rng.uniform(-0.05, 0.05, (batch, 12)); d[:, 24:30] = rng.uniform(-0.1, 0.1, (batch, 6))
    x0 = np.tile(xbar, (batch, 1)) + d
    t0 = rng.uniform(0.0, g["switchingTimes"][-1], batch)
    ee_nom = np.concatenate([EE_NOMINAL_POS, EE_NOMINAL_QUAT])
    evs, mos = [], []
    ref_t = np.zeros((batch, 2)); ref_x = np.zeros((batch, 2, 37))
    for b in range(batch):
        e, m = gait_schedule(gait, g["switchingTimes"][-1] + 3.0 * horizon)
        goal = x0[b, 6:12].copy(); goal[0] += 0.1; goal[2] = COM_HEIGHT; goal[4] = 0.0; goal[5] = 0.0
        now = x0[b, 6:12].copy(); now[2] = COM_HEIGHT; now[4] = 0.0; now[5] = 0.0
        ref_t[b], ref_x[b] = make_target(t0[b], horizon, now, goal, qnom, ee_nom, ee_nom)
        evs.append(e); mos.append(m)
    ev, modes = _pad_schedules(evs, mos)
    return dict(name="gait:" + gait, B=batch, n_intervals=n_intervals, horizon=horizon, t0=t0, x0=x0, ref_t=ref_t, ref_x=ref_x, ev=ev, modes=modes,
                period=0.002, time=20.0)


def make_target(t0, horizon, base_now, base_goal, q_nom, ee_now, ee_goal):
    """targetPoseToTargetTrajectories: 2 knots of [0_6, base pose(6), defaultJointState(18), EE pose(7)]."""
    ref_t = np.array([t0, t0 + horizon])
    x = np.zeros((2, 37))
    x[0, 6:12] = base_now
    x[1, 6:12] = base_goal
    x[:, 12:30] = q_nom
    x[0, 30:37] = ee_now
    x[1, 30:37] = ee_goal
    return ref_t, x


def _pad_schedules(evs, modes):
    nev = max(len(e) for e in evs)
    B = len(evs)
    ev = np.zeros((B, nev))
    mo = np.full((B, nev + 1), STANCE, dtype=np.int32)
    for b in range(B):
        n = len(evs[b])
        ev[b, :n] = evs[b]
        mo[b, :n + 1] = modes[b]
        # padding: extra events far in the future, stance
        for k in range(n, nev):
            ev[b, k] = evs[b][-1] + 1.0e3 * (k - n + 1)
    return ev, mo


def _rotvec_to_quat(r):
    th = np.linalg.norm(r)
    if th < 1e-12:
        return np.array([0.0, 0.0, 0.0, 1.0])
    a = r / th
    return np.concatenate([a * np.sin(th / 2), [np.cos(th / 2)]])


def _quat_mul(a, b):  # xyzw
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw,
                     aw * bw - ax * bx - ay * by - az * bz])


def make_config(name, batch=None, n_intervals=None, seed=None):
    """Returns dict(t0[B], horizon, x0[B,30], ref_t[B,2], ref_x[B,2,37], ev[B,nev], modes[B,nev+1],
    period, time, n_intervals)."""
    mb, st = load_blobs()
    xbar = st[ST_XINIT:ST_XINIT + 30].copy()
    qnom = mb[MB_QNOM:MB_QNOM + 18].copy()
    dt = st[ST_SQP_DT]
    defaults = dict(C1=(1, 20, None), C2=(1, 100, None), C3=(1024, 100, 1234), C4=(8192, 100, 1235), C5=(4096, 150, 1236))
    B0, N0, s0 = defaults[name]
    B = B0 if batch is None else batch
    N = N0 if n_intervals is None else n_intervals
    seed = s0 if seed is None else seed
    horizon = N * dt
    ee_nom = np.concatenate([EE_NOMINAL_POS, EE_NOMINAL_QUAT])
    # C2: t0 = 0.1 (inside the first trot phase). t0 = 0 would put a gait event exactly at the initial time,
    # for which the reference's primal solution has no input at node 0 (event node) — a degenerate case.
    t0 = np.full(B, 0.1) if name == "C2" else np.zeros(B)
    x0 = np.tile(xbar, (B, 1))
    if name in ("C3", "C4", "C5"):
        rng = np.random.default_rng(seed)
        d = np.zeros((B, 30))
        d[:, 0:6] = rng.uniform(-0.1, 0.1, (B, 6))
        d[:, 6:9] = rng.uniform(-0.02, 0.02, (B, 3))
        d[:, 9:12] = rng.uniform(-0.05, 0.05, (B, 3))
        d[:, 12:24] = rng.uniform(-0.1, 0.1, (B, 12))
        d[:, 24:30] = rng.uniform(-0.2, 0.2, (B, 6))
        x0 = x0 + d
        lo = mb[MB_QLO + 12:MB_QLO + 18] + 0.05
        hi = mb[MB_QHI + 12:MB_QHI + 18] - 0.05
        x0[:, 24:30] = np.clip(x0[:, 24:30], lo, hi)
        t0 = rng.uniform(0.0, 0.7, B)
    evs, mos = [], []
    ref_t = np.zeros((B, 2))
    ref_x = np.zeros((B, 2, 37))
    if name == "C5":
        rng2 = np.random.default_rng(seed + 100)
        near = rng2.uniform(0, 1, B) < 0.25
        lo = mb[MB_QLO + 12:MB_QLO + 18]
        for b in np.nonzero(near)[0]:
            x0[b, 25] = lo[1] + rng2.uniform(0.0, 0.1)
            x0[b, 26] = lo[2] + rng2.uniform(0.0, 0.1)
    for b in range(B):
        if name == "C1":
            e, m = stance_schedule(t0[b], horizon)
            goal = xbar[6:12].copy()
        elif name == "C5":
            e, m = trot_stance_trot_schedule(t0[b] + 0.7, t0[b] + 1.4, t0[b] + 2.0 * horizon + 0.7)
            goal = x0[b, 6:12].copy(); goal[0] += 0.3; goal[2] = COM_HEIGHT; goal[4] = 0.0; goal[5] = 0.0
        else:
            e, m = trot_schedule(0.7 + 3.0 * horizon)
            goal = x0[b, 6:12].copy(); goal[0] += 0.3; goal[2] = COM_HEIGHT; goal[4] = 0.0; goal[5] = 0.0
        now = x0[b, 6:12].copy(); now[2] = COM_HEIGHT; now[4] = 0.0; now[5] = 0.0
        ee_goal = ee_nom.copy()
        if name == "C5":
            ee_goal[:3] += rng2.uniform(-0.15, 0.15, 3)
            ee_goal[3:] = _quat_mul(_rotvec_to_quat(rng2.uniform(-0.3, 0.3, 3)), EE_NOMINAL_QUAT)
        ref_t[b], ref_x[b] = make_target(t0[b], horizon, now, goal, qnom, ee_nom, ee_goal)
        evs.append(e); mos.append(m)
    ev, modes = _pad_schedules(evs, mos)
    return dict(name=name, B=B, n_intervals=N, horizon=horizon, t0=t0, x0=x0, ref_t=ref_t, ref_x=ref_x, ev=ev, modes=modes,
                period=0.002, time=20.0)
