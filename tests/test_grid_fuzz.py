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


@pytest.mark.parametrize("robust", [False, True])
def test_grid_minimum_step_setting(blobs, oblobs, robust):
    """ST_GRID_DT_MIN: [upstream]'s dt_min (10 limitEpsilon, the ingestion's default) keeps a node that falls 5e-7 s before a gait event; the opt-in robust
    minimum step (QM_GRID_DT_MIN_ROBUST) merges it into the event node.  Device kernel (emulated) and oracle agree bit for bit under either setting, and the
    whole MPC iteration reports what the resulting grid deserves: with the upstream default the interval in front of the event has a NEGATIVE adapted duration
    (event − weakEpsilon − node) and the solve fails on both sides; with the robust setting it succeeds on both."""
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
    dev_ok = [(status[b] == 0 and si[b, 3] == 0.0) for b in range(B)]
    ora_ok = []
    for b in range(B):
        oracle.set_schedule(ev[b], cfg["modes"][b]); oracle.set_target(cfg["ref_t"][b], cfg["ref_x"][b])
        try:
            oracle.mpc_step(t0[b], t0[b] + horizon, cfg["x0"][b]); ora_ok.append(True)
        except RuntimeError:
            ora_ok.append(False)
    assert dev_ok == ora_ok, (dev_ok, ora_ok)
    assert dev_ok[1] and (dev_ok[0] == robust), dev_ok
