"""Randomised parity of K0 (time discretisation with events, contact-mode lookup): the integer outputs of the hot path must be bit-exact
(BASELINE.json north_star).  Device kernels on the host emulator vs the oracle's restatement of [upstream timeDiscretizationWithEvents],
over schedules with events at / next to t0 and tf, events closer than dt, bursts of events and no events at all."""
import numpy as np
import pytest
import emu_harness

MODES = np.array([15, 9, 6, 15, 15, 9, 6, 15])      # every swing phase enclosed by stance (the swing planner's precondition)


def _random_case(rng, nev):
    t0 = float(rng.uniform(0.0, 3.0)); horizon = float(rng.choice([0.3, 0.45, 1.0, 1.5]))
    kind = rng.integers(0, 5)
    if kind == 0:   ev = np.sort(rng.uniform(t0 - 0.5, t0 + horizon + 0.5, nev))
    elif kind == 1: ev = t0 + np.arange(1, nev + 1) * 0.015 * rng.uniform(0.2, 3.0)                      # multiples / fractions of dt
    elif kind == 2: ev = np.sort(np.concatenate([[t0, t0 + horizon], rng.uniform(t0, t0 + horizon, nev - 2)]))   # exactly at the ends
    elif kind == 3: ev = np.sort(t0 + 0.2 + rng.uniform(0.0, 1e-3, nev))                                 # burst: gaps far below dt
    else:           ev = t0 + horizon + 1.0 + np.arange(nev) * 10.0                                      # nothing inside the window
    ev = np.maximum.accumulate(ev + np.arange(nev) * 1e-9)                                               # strictly increasing
    return t0, horizon, ev


def test_grid_and_modes_bit_exact(blobs, oracle):
    from qm_control_amd import scenarios
    rng = np.random.default_rng(2024)
    B, nev = 16, 7
    base = scenarios.make_config("C3", batch=B, n_intervals=20)
    e = emu_harness.Emu(blobs[0], blobs[1], B, 160, base["ref_t"].shape[1], nev)
    for rep in range(4):
        for horizon in (0.3, 0.45, 1.0, 1.5):
            cases = [_random_case(rng, nev) for _ in range(B)]
            cfg = dict(base); cfg["B"] = B; cfg["horizon"] = horizon
            cfg["t0"] = np.array([c[0] for c in cases]); cfg["ev"] = np.stack([c[2] for c in cases]); cfg["modes"] = np.tile(MODES, (B, 1)).astype(np.int32)
            cfg["ref_t"] = np.stack([[c[0], c[0] + horizon] for c in cases])
            e.grid_only(cfg)
            n = e.buf("n_nodes", (B,), np.int32); t = e.node_arr("node_t", 1); ev = e.node_arr("node_ev", 1, np.int32); md = e.node_arr("node_mode", 1, np.int32)
            st = e.buf("status", (B,), np.int32)
            for b in range(B):
                rt, rev = oracle.time_grid(cfg["t0"][b], cfg["t0"][b] + horizon, 0.015, cfg["ev"][b])
                assert st[b] in (0, -2)                      # -2: a swing phase cut by the window edge (reported, grid still valid)
                assert n[b] == len(rt), (rep, b)
                assert np.array_equal(t[:n[b], b], rt) and np.array_equal(ev[:n[b], b], rev), (rep, b)
                oracle.set_schedule(cfg["ev"][b], cfg["modes"][b])
                for i in range(n[b]):
                    ts = rt[i] + (1e-6 if rev[i] == 2 else 0.0)
                    assert md[i, b] == oracle.mode_at(ts), (rep, b, i)


# ---- a shooting node inside (event − weakEpsilon, event): the interval in front of the PreEvent node has a NEGATIVE adapted duration (the reference's
#      mpcThread_ runs on continuous ROS time, QMController.cpp:315-330, so this happens about once per 3700 solves at 100 Hz).  The solve must SURVIVE it.
DEGENERATE_OFFSETS = (-9e-7, -5e-7, -1e-7, -1e-9, -1e-12)


def degenerate_cases(cfg1, n_grid=5, full=True):
    """instances of C2 (one per event inside the horizon and per offset): t0 chosen so that grid node `n_grid` lands `off` seconds before that event.
    Returns a batch config (same schedule, references shifted with t0) and the list of (event index, offset).  full=False: every offset at the first event and
    one offset at every other event (the host-emulated run; the -m gpu twin runs the full matrix)."""
    ev = cfg1["ev"][0]; t00 = float(cfg1["t0"][0]); horizon = float(cfg1["horizon"])
    evs = [k for k in range(len(ev)) if t00 + 0.2 < ev[k] < t00 + horizon - 0.05]
    cases = [(k, off) for k in evs for off in DEGENERATE_OFFSETS if full or k == evs[0] or off == DEGENERATE_OFFSETS[1]]
    B = len(cases)
    cfg = {k: (np.repeat(v, B, axis=0) if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] == 1 else v) for k, v in cfg1.items()}
    cfg["B"] = B
    t0 = np.array([ev[k] + off - n_grid * 0.015 for k, off in cases])
    # the node really is inside the window in floating point (t0 + n dt accumulates rounding: walk the additions the grid makes)
    for b, (k, off) in enumerate(cases):
        t = t0[b]
        for _ in range(n_grid): t = t + 0.015
        assert ev[k] - 1e-6 < t < ev[k], (k, off, t - ev[k])
    cfg["t0"] = t0; cfg["ref_t"] = cfg1["ref_t"][0][None, :] + (t0 - t00)[:, None]
    return cfg, cases


def check_degenerate_against_robust(res, rob, tol, what):
    """the survived solve against the solve on the ROBUST grid (node merged into the event node): same nodes except the degenerate one, integers identical, x* / u* within
    `tol` per block everywhere except the input of the degenerate interval (and its copy at the PreEvent node) — the one casualty of a stage that lasts −0.5 µs"""
    from conftest import assert_blocks
    n, nr = len(res["t"]), len(rob["t"])
    assert n == nr + 1, (what, n, nr)
    k = next(i for i in range(n - 1) if res["ev"][i + 1] == 1 and 0.0 < res["t"][i + 1] - res["t"][i] < 1e-6)      # the degenerate node
    keep = [i for i in range(n) if i != k]
    assert np.array_equal(res["ev"][keep], rob["ev"]) and np.array_equal(res["mode"][keep], rob["mode"]) and np.array_equal(res["t"][keep], rob["t"]), what
    assert res["mode"][k] == res["mode"][k - 1] and res["ev"][k] == 0, what
    assert_blocks(res["x"][keep], rob["x"], "x", tol, what + " x vs robust grid")
    assert_blocks(res["x"][k], res["x"][k + 1], "x", 1e-4, what + " x across the degenerate interval")      # continuity over <= 1 µs (|xdot| · 1e-6 against the block floors)
    ku = [i for i in keep if i != k + 1]; kr = [j for j in range(nr) if j != k]
    assert_blocks(res["u"][ku], rob["u"][kr], "u", tol, what + " u vs robust grid")
    assert np.isfinite(res["u"][k]).all() and np.abs(res["u"][k]).max() < 1e3, what                  # the casualty stays a sane input (the policy interpolates towards it)
    return k


def test_degenerate_interval_survives(blobs, oblobs):
    """every gait event of C2 (trot, N = 100) x offsets {-9e-7 ... -1e-12}: oracle AND product (host-emulated kernels) finish the solve on [upstream]'s grid with status >= 0
    and the warning bit, agree with each other to 1e-6 per block on the WHOLE trajectories, keep today's integers, and agree with the solve on the robust grid on every
    node but the degenerate interval's input.  Tolerance against the robust grid: 5e-6 — the two grids differ by the 1 µs the neighbouring interval is longer, which moves
    x* by ~2e-6 (an offset of -2e-6, i.e. NO degenerate interval, differs from the robust grid by the same 2e-6)."""
    import pyoracle
    from conftest import assert_blocks
    from qm_control_amd import scenarios, layout as L
    cfg1 = scenarios.make_config("C2", batch=1, n_intervals=100)
    cfg, cases = degenerate_cases(cfg1, full=False)
    B = cfg["B"]; assert B >= 7, B
    st = blobs[1].copy(); assert st[L.ST_GRID_DT_MIN] == L.QM_GRID_DT_MIN_UPSTREAM and st[L.ST_RICCATI_STRICT] == 0.0
    strob = st.copy(); strob[L.ST_GRID_DT_MIN] = L.QM_GRID_DT_MIN_ROBUST
    nmax = 128
    e = emu_harness.Emu(blobs[0], st, B, nmax, cfg["ref_t"].shape[1], cfg["ev"].shape[1])
    er = emu_harness.Emu(blobs[0], strob, B, nmax, cfg["ref_t"].shape[1], cfg["ev"].shape[1])
    outs = []
    for em in (e, er):
        em.mpc_step(cfg)
        n = em.buf("n_nodes", (B,), np.int32); status = em.buf("status", (B,), np.int32); si = em.buf("step_info", (B, 4))
        t = em.node_arr("node_t", 1); evt = em.node_arr("node_ev", 1, np.int32); md = em.node_arr("node_mode", 1, np.int32); xs = em.node_arr("xs", 30); us = em.node_arr("us", 30)
        outs.append([dict(t=t[:n[b], b], ev=evt[:n[b], b], mode=md[:n[b], b], x=xs[:n[b], b], u=us[:n[b], b], status=int(status[b]), pivot=float(si[b, 3])) for b in range(B)])
    dev, devrob = outs
    ost = oblobs[1].copy(); ostrob = ost.copy(); ostrob[L.ST_GRID_DT_MIN] = L.QM_GRID_DT_MIN_ROBUST
    for b, (k_ev, off) in enumerate(cases):
        what = "event %d offset %g" % (k_ev, off)
        ora = {}
        for name, s in (("up", ost), ("rob", ostrob)):
            o = pyoracle.Oracle(oblobs[0], s); o.set_schedule(cfg["ev"][b], cfg["modes"][b]); o.set_target(cfg["ref_t"][b], cfg["ref_x"][b])
            ora[name] = o.mpc_step(cfg["t0"][b], cfg["t0"][b] + cfg["horizon"], cfg["x0"][b])          # raises on a failed solve
        assert ora["up"]["warn"] == L.QM_MPC_WARN_PIVOT and ora["rob"]["warn"] == 0, what
        assert dev[b]["status"] == 0 and dev[b]["pivot"] == 1.0 and devrob[b]["status"] == 0 and devrob[b]["pivot"] == 0.0, what      # (the C ABI turns pivot into status QM_MPC_WARN_PIVOT)
        # product vs oracle: integers bit-exact, whole trajectories (the degenerate interval's input included) to 1e-6 per block
        for r, o_ in ((dev[b], ora["up"]), (devrob[b], ora["rob"])):
            assert np.array_equal(r["t"], o_["t"]) and np.array_equal(r["ev"], o_["ev"]) and np.array_equal(r["mode"], o_["mode"]), what
            assert_blocks(r["x"], o_["x"], "x", 1e-6, what); assert_blocks(r["u"], o_["u"], "u", 1e-6, what)
        check_degenerate_against_robust(ora["up"], ora["rob"], 5e-6, what + " (oracle)")
        check_degenerate_against_robust(dev[b], devrob[b], 5e-6, what + " (product)")


def test_strict_pivot_setting_reports_the_failure(oblobs):
    """ST_RICCATI_STRICT = 1: the rounds 1-3 behaviour — a non-positive pivot is a failed solve"""
    import pyoracle
    from qm_control_amd import scenarios, layout as L
    cfg, cases = degenerate_cases(scenarios.make_config("C2", batch=1, n_intervals=100))
    st = oblobs[1].copy(); st[L.ST_RICCATI_STRICT] = 1.0
    o = pyoracle.Oracle(oblobs[0], st); o.set_schedule(cfg["ev"][1], cfg["modes"][1]); o.set_target(cfg["ref_t"][1], cfg["ref_x"][1])
    with pytest.raises(RuntimeError):
        o.mpc_step(cfg["t0"][1], cfg["t0"][1] + cfg["horizon"], cfg["x0"][1])


def test_nan_observation_and_indefinite_stage_are_failures_not_warnings(blobs, oblobs):
    """Only the BENIGN case is a warning (finite non-positive pivots on a stage of non-positive duration).  A NaN in the observation, or a stage of positive duration whose
    Huu is not positive definite (here: a negated input weight R), must come back as a FAILED solve from the oracle (exception) and from the product (the status mapping of
    qmhip_mpc_download on the emulated pipeline's K0 status + step_info: -4), never as QM_MPC_WARN_PIVOT with a policy made of NaNs ([upstream] SqpSolver
    throws on HPIPM's NaN status)."""
    import ctypes as C
    import pyoracle
    from qm_control_amd import scenarios, layout as L
    B = 2
    cfg = scenarios.make_config("C3", batch=B, n_intervals=20); cfg["B"] = B
    status_of = lambda e, b: int(e.lib.emu_mpc_status(C.c_int(int(e.buf("status", (B,), np.int32)[b])), e.buf("step_info", (B, 4))[b].ctypes.data_as(C.POINTER(C.c_double)), C.c_int(0)))
    # (i) clean solve: status 0
    e = emu_harness.Emu(blobs[0], blobs[1], B, 48, cfg["ref_t"].shape[1], cfg["ev"].shape[1]); e.mpc_step(cfg)
    assert [status_of(e, b) for b in range(B)] == [0, 0]
    # (ii) NaN in instance 0's observation: instance 0 fails, instance 1 is untouched
    bad = dict(cfg); bad["x0"] = cfg["x0"].copy(); bad["x0"][0, 7] = np.nan
    e.mpc_step(bad)
    assert status_of(e, 0) == -4 and status_of(e, 1) == 0, [status_of(e, b) for b in range(B)]
    o = pyoracle.Oracle(*oblobs); o.set_schedule(cfg["ev"][0], cfg["modes"][0]); o.set_target(cfg["ref_t"][0], cfg["ref_x"][0])
    with pytest.raises(RuntimeError): o.mpc_step(cfg["t0"][0], cfg["t0"][0] + cfg["horizon"], bad["x0"][0])
    # (iii) Huu indefinite on stages of POSITIVE duration (negated input weight): hard failure on both sides, although every pivot is finite
    st = blobs[1].copy(); st[L.ST_R:L.ST_R + 900] *= -1.0; ost = oblobs[1].copy(); ost[L.ST_R:L.ST_R + 900] *= -1.0
    e2 = emu_harness.Emu(blobs[0], st, B, 48, cfg["ref_t"].shape[1], cfg["ev"].shape[1]); e2.mpc_step(cfg, max_trials=1)
    assert [status_of(e2, b) for b in range(B)] == [-4, -4] and (e2.buf("step_info", (B, 4))[:, 3].astype(int) & 2).all()
    o2 = pyoracle.Oracle(oblobs[0], ost); o2.set_schedule(cfg["ev"][0], cfg["modes"][0]); o2.set_target(cfg["ref_t"][0], cfg["ref_x"][0])
    with pytest.raises(RuntimeError): o2.mpc_step(cfg["t0"][0], cfg["t0"][0] + cfg["horizon"], cfg["x0"][0])


@pytest.mark.parametrize("robust", [False, True])
def test_grid_minimum_step_setting(blobs, oblobs, robust):
    """ST_GRID_DT_MIN: [upstream]'s dt_min (10 limitEpsilon, the ingestion's default) keeps a node that falls 5e-7 s before a gait event; the opt-in robust
    minimum step (QM_GRID_DT_MIN_ROBUST) merges it into the event node.  Device kernel (emulated) and oracle agree bit for bit under either setting, and the
    whole MPC iteration reports what the resulting grid deserves: with the upstream default the interval in front of the event has a NEGATIVE adapted duration
    (event − weakEpsilon − node) and both sides flag the zeroed pivots of that stage (a warning, the solve completes); with the robust setting neither does."""
    import pyoracle
    from qm_control_amd import scenarios, layout as L
    B, nev = 2, 7
    dt_min = L.QM_GRID_DT_MIN_ROBUST if robust else L.QM_GRID_DT_MIN_UPSTREAM
    assert blobs[1][L.ST_GRID_DT_MIN] == L.QM_GRID_DT_MIN_UPSTREAM and oblobs[1][L.ST_GRID_DT_MIN] == L.QM_GRID_DT_MIN_UPSTREAM      # both ingestions default to upstream
    st = blobs[1].copy(); st[L.ST_GRID_DT_MIN] = dt_min
    ost = oblobs[1].copy(); ost[L.ST_GRID_DT_MIN] = dt_min
    base = scenarios.make_config("C3", batch=B, n_intervals=20)
    cfg = dict(base); cfg["B"] = B; horizon = cfg["horizon"] = 0.3
    t0 = np.array([0.2, 1.0]); cfg["t0"] = t0
    # instance 0: the grid node t0 + 4 dt lands 5e-7 s BEFORE the first event; instance 1: nothing special
    ev = np.stack([np.array([t0[0] + 4 * 0.015 + 5e-7, 0.9, 1.6, 2.3, 3.0, 3.7, 4.4]), np.array([0.5, 1.135, 1.6, 2.3, 3.0, 3.7, 4.4])])
    cfg["ev"] = ev; cfg["modes"] = np.tile(MODES, (B, 1)).astype(np.int32); cfg["ref_t"] = np.stack([[t, t + horizon] for t in t0])
    cfg["x0"] = np.tile(st[L.ST_XINIT:L.ST_XINIT + 30], (B, 1)); cfg["ref_x"] = base["ref_x"][:B]
    e = emu_harness.Emu(blobs[0], st, B, 64, base["ref_t"].shape[1], nev)
    oracle = pyoracle.Oracle(oblobs[0], ost)
    e.grid_only(cfg)
    n = e.buf("n_nodes", (B,), np.int32); t = e.node_arr("node_t", 1); evt = e.node_arr("node_ev", 1, np.int32)
    for b in range(B):
        rt, rev = oracle.time_grid(t0[b], t0[b] + horizon, 0.015, ev[b], dt_min)
        assert n[b] == len(rt) and np.array_equal(t[:n[b], b], rt) and np.array_equal(evt[:n[b], b], rev), b
    rt0, rev0 = oracle.time_grid(t0[0], t0[0] + horizon, 0.015, ev[0], dt_min)
    k = int(np.nonzero(rev0 == 1)[0][0])                     # the PreEvent node
    close = rt0[k] - rt0[k - 1]
    assert (close > 1e-3) if robust else (0.0 < close < 1e-6)      # merged / kept
    # the whole iteration on that grid
    e.mpc_step(cfg); status = e.buf("status", (B,), np.int32).copy(); si = e.buf("step_info", (B, 4))
    assert (status == 0).all()
    dev_clean = [si[b, 3] == 0.0 for b in range(B)]
    ora_clean = []
    for b in range(B):
        oracle.set_schedule(ev[b], cfg["modes"][b]); oracle.set_target(cfg["ref_t"][b], cfg["ref_x"][b])
        ora_clean.append(oracle.mpc_step(t0[b], t0[b] + horizon, cfg["x0"][b])["warn"] == 0)
    assert dev_clean == ora_clean, (dev_clean, ora_clean)
    assert dev_clean[1] and (dev_clean[0] == robust), dev_clean
