"""tests/wbc_cases.py — random WBC inputs shared by the CPU and GPU parity tests."""
import numpy as np
from qm_control_amd import layout as L

MODES = [15, 9, 6, 7, 15, 9, 6, 14]        # stance, the two trot phases, three-leg support (LF resp. RH in the air)


def random_wbc_inputs(oracle, blobs, n, seed, vel_scale, modes=(15, 9, 6, 15, 9, 6)):
    mb, st = blobs
    rng = np.random.default_rng(seed)
    xbar = st[L.ST_XINIT:L.ST_XINIT + 30]
    cases = []
    for k in range(n):
        mode = modes[k % len(modes)]
        q = xbar[6:30] + 0.1 * rng.normal(size=24); q[18:] = xbar[24:] + 0.05 * rng.normal(size=6)
        v = vel_scale * rng.normal(size=24)
        rbd = oracle.rbd_from_q(q, v)
        xd = xbar + 0.05 * rng.normal(size=30); xd[24:] = xbar[24:] + 0.02 * rng.normal(size=6)
        ud = np.zeros(30); fl = [(mode >> 3) & 1, (mode >> 2) & 1, (mode >> 1) & 1, mode & 1]
        for c in range(4):
            if fl[c]:
                ud[3 * c:3 * c + 3] = [5 * rng.normal(), 5 * rng.normal(), mb[L.MB_ROBOTMASS] * 9.81 / sum(fl) + 10 * rng.normal()]
        ud[12:] = vel_scale * rng.normal(size=18)
        il = vel_scale * rng.normal(size=30)
        cases.append(dict(mode=mode, rbd=rbd, xd=xd, ud=ud, il=il, time=20.0 if k % 4 != 3 else 5.0))
    return cases


def hard_wbc_inputs(oracle, blobs, n, seed, modes=MODES):
    """cases that drive the level-0 soft rows (torque limits, friction pyramids) into their slacks: large tangential force
    requests, large planned joint-velocity jumps (= large desired accelerations through (u − inputLast_) / period), fast motion"""
    cases = random_wbc_inputs(oracle, blobs, n, seed, 0.3, modes)
    rng = np.random.default_rng(seed + 1000)
    for k, c in enumerate(cases):
        fl = [(c["mode"] >> 3) & 1, (c["mode"] >> 2) & 1, (c["mode"] >> 1) & 1, c["mode"] & 1]
        for i in range(4):
            if fl[i]:
                c["ud"][3 * i:3 * i + 2] += rng.uniform(-1.0, 1.0, 2) * 0.6 * c["ud"][3 * i + 2]      # beyond the 0.3 pyramid of task.info:346-349
        if k % 2:
            c["ud"][12:] += rng.normal(size=18) * 0.4                                                # 200 rad/s² requested at period 0.002
    return cases
