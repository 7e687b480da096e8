"""Batched rigid-body plant on the MI355X through the C ABI (qmhip_sim_*) against the CPU oracle's restatement (oracle/src/sim.h)."""
import numpy as np
import pytest
from conftest import rel_err
from test_sim import random_cases, nominal_q, stand_height

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nsub", [1, 2])
def test_plant_vs_oracle(blobs, oracle, nsub):
    from qm_control_amd import api
    mb, st = blobs
    cases = random_cases(oracle, st, 12, 7)
    B = len(cases); arr = lambda k: np.array([c[k] for c in cases])
    itf = api.QMInterface(blobs=blobs, max_batch=B, max_nodes=8, max_ref_knots=2, max_events=2)
    sim = api.QMHWSim(itf)
    sim.reset(arr("q"), arr("v"), 1.0); sim.setCommand(arr("pos"), arr("vel"), arr("kp"), arr("kd"), arr("ff"))
    steps = 12; out = []
    for _ in range(steps):
        rbd, contact = sim.step(0.001, nsub); s = sim.state(); s["rbd"] = rbd; s["contact"] = contact; out.append(s)
    for b, c in enumerate(cases):
        oracle.sim_params(); oracle.sim_reset(c["q"], c["v"], 1.0); oracle.sim_command(c["pos"], c["vel"], c["kp"], c["kd"], c["ff"])
        for k in range(steps):
            r = oracle.sim_step(0.001, nsub)
            assert r["status"] == 0 and out[k]["status"][b] == 0
            assert rel_err(out[k]["q"][b], r["q"]) < 1e-9 and rel_err(out[k]["v"][b], r["v"]) < 1e-7, (b, k)
            assert rel_err(out[k]["force"][b], r["force"]) < 1e-6 and list(out[k]["contact"][b]) == list(r["contact"]), (b, k)
            assert rel_err(out[k]["rbd"][b], r["rbd"]) < 1e-7 and abs(out[k]["time"][b] - r["time"]) < 1e-12, (b, k)
    itf.close()


def test_command_delay_and_full_batch(blobs, oracle):
    """a feed-forward torque step reaches the joints `delay` after it was commanded (QMHWSim.cpp:100-113); 1024 instances stay finite while standing"""
    from qm_control_amd import api
    mb, st = blobs
    B = 1024; itf = api.QMInterface(blobs=blobs, max_batch=B, max_nodes=8, max_ref_knots=2, max_events=2)
    sim = api.QMHWSim(itf, delay=0.009, saturate_effort=0.0)
    q = np.tile(nominal_q(st, 2.0), (B, 1)); sim.reset(q, np.zeros((B, 24)), 0.0); sim.setCommand(0, 0, 0, 0, 0)
    for _ in range(3):
        sim.step(0.001, 1, download=False)
    ff = np.zeros(18); ff[17] = 1.0; sim.setCommand(0, 0, 0, 0, ff)
    vs = [0.0]
    for _ in range(14):
        sim.step(0.001, 1, download=False); vs.append(sim.state()["v"][5, 23])
    dv = np.diff(np.array(vs)); first = int(np.argmax(np.abs(dv) > 1e-3 * np.abs(dv).max()))
    assert first in (8, 9), (first, dv)
    # standing batch with the joint gains the reference commands (legs kd 3, arm kd 0.5) plus a position hold
    rng = np.random.default_rng(3); z0 = stand_height(oracle, st)
    q = np.tile(nominal_q(st, z0 - 0.002), (B, 1)); q[:, 6:] += 0.02 * rng.normal(size=(B, 18))
    sim.set_params(delay=0.009, saturate_effort=1.0); sim.reset(q, np.zeros((B, 24)), 0.0)
    kp = np.concatenate([np.full(12, 300.0), np.full(6, 20.0)]); kd = np.concatenate([np.full(12, 3.0), np.full(6, 0.5)])
    sim.setCommand(q[:, 6:], 0.0, kp, kd, 0.0)
    for _ in range(200):
        rbd, contact = sim.step(0.001, 2)
    s = sim.state()
    assert (s["status"] == 0).all() and np.isfinite(s["q"]).all() and contact.all()
    assert np.abs(s["q"][:, 2] - z0).max() < 0.02 and np.abs(s["q"][:, 3:6]).max() < 0.1
    itf.close()
