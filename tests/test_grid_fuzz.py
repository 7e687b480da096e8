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
