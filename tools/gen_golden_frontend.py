"""tools/gen_golden_frontend.py — freeze the oracle's reference / gait front-end outputs (oracle/frontend.py) on a deterministic command
stream as tests/golden/frontend_stream.npz.  Self-generated like the other goldens (the reference ships none): it pins the oracle
against regressions and gives the emulator / GPU tests a fixture that does not depend on the oracle's code."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import frontend as fe
from qm_control_amd import scenarios

B, STEPS, HORIZON, SEED, CAP = 16, 24, 1.2, 41, 96


def stream():
    """the deterministic command stream (shared with the tests): per step the observation times and the gait requested per instance (-1: none)"""
    names = list(scenarios.load_gaits().keys())
    rng = np.random.default_rng(SEED)
    t = rng.uniform(0.0, 0.4, B); ts = []; reqs = []
    for _ in range(STEPS):
        ts.append(t.copy()); reqs.append(np.array([rng.integers(0, len(names)) if rng.uniform() < 0.3 else -1 for _ in range(B)], dtype=np.int32))
        t = t + rng.uniform(0.02, 0.45, B)
    return names, np.array(ts), np.array(reqs)


def main():
    gaits = scenarios.load_gaits(); names, ts, reqs = stream()
    g0 = gaits["stance"]
    orc = [fe.GaitSchedule([0.5], [15, 15], g0["switchingTimes"], g0["modeSequence"], 0.1) for _ in range(B)]
    n = np.zeros((STEPS, B), np.int32); ev = np.zeros((STEPS, B, CAP)); mo = np.zeros((STEPS, B, CAP + 1), np.int32)
    for s in range(STEPS):
        for b in range(B):
            if reqs[s, b] >= 0:
                g = gaits[names[reqs[s, b]]]; orc[b].pre_solver_run_insert(g["switchingTimes"], g["modeSequence"], ts[s, b], ts[s, b] + HORIZON)
            e, m = orc[b].modify_references(ts[s, b], HORIZON)
            assert len(e) <= CAP
            n[s, b] = len(e); ev[s, b, :len(e)] = e; mo[s, b, :len(m)] = m
    # targets: one command of each kind on a fixed observation
    mb, st = scenarios.load_blobs(); rng = np.random.default_rng(SEED + 1)
    x0 = np.tile(st[scenarios.ST_XINIT:scenarios.ST_XINIT + 30], (6, 1)) + rng.uniform(-0.05, 0.05, (6, 30)); t0 = rng.uniform(0.5, 3.0, 6)
    ee = np.zeros((6, 7)); ee[:, :3] = np.array([0.52, 0.09, 0.78]) + rng.uniform(-0.2, 0.2, (6, 3)); q = rng.normal(size=(6, 4)); ee[:, 3:] = q / np.linalg.norm(q, axis=1, keepdims=True)
    kind = np.array([1, 1, 2, 2, 3, 3], np.int32); cmd = np.zeros((6, 7))
    cmd[:2, :4] = rng.uniform(-0.5, 0.5, (2, 4)); cmd[2:4, :3] = rng.uniform(-0.2, 0.2, (2, 3)); cmd[4:, :3] = ee[4:, :3] + rng.uniform(-0.3, 0.3, (2, 3)); g = rng.normal(size=(2, 4)); cmd[4:, 3:] = g / np.linalg.norm(g, axis=1, keepdims=True)
    rt = np.zeros((6, 2)); rx = np.zeros((6, 2, 37)); last = np.zeros((6, 7))
    for b in range(6):
        pub = fe.TargetPublisher(mb[scenarios.MB_QNOM:scenarios.MB_QNOM + 18], 0.4, 0.3, 0.1, 1.0)
        rt[b], rx[b] = {1: pub.cmd_vel, 2: pub.ee_cmd_vel, 3: pub.ee_goal}[int(kind[b])](cmd[b], t0[b], x0[b], ee[b]); last[b] = pub.last_ee
    out = os.path.join(ROOT, "tests", "golden", "frontend_stream.npz")
    np.savez_compressed(out, n=n, ev=ev, mo=mo, tgt_x0=x0, tgt_t0=t0, tgt_ee=ee, tgt_kind=kind, tgt_cmd=cmd, tgt_rt=rt, tgt_rx=rx, tgt_last=last)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
