"""GPU tests of the file-based construction path: qmhip_create(urdf, task.info, reference.info) — the product's own C++ ingestion —
executed on hardware from the shipped input files (tests/data), and a plain-C client of the C ABI (tests/c_abi_smoke.c) that goes
create-from-files -> qmhip_mpc_step -> qmhip_policy_eval -> qmhip_wbc_step without Python in the loop."""
import os
import subprocess
import numpy as np
import pytest
from conftest import ROOT, assert_blocks

pytestmark = pytest.mark.gpu
DATA = os.path.join(ROOT, "tests", "data")
URDF, TASK, REFI = (os.path.join(DATA, f) for f in ("robot.urdf", "task.info", "reference.info"))


def test_create_from_files_round_trips_the_committed_blobs(blobs, oracle):
    from qm_control_amd import api, scenarios
    cfg = scenarios.make_config("C3", batch=4, n_intervals=20)
    kw = dict(max_batch=4, max_nodes=48, max_ref_knots=2, max_events=cfg["ev"].shape[1])
    itf_f = api.QMInterface(TASK, URDF, REFI, **kw)                       # qm::QMInterface(taskFile, urdfFile, referenceFile)
    assert np.array_equal(itf_f.model_blob, blobs[0]) and np.array_equal(itf_f.settings_blob, blobs[1])   # qmhip_export_blobs round trip, bit-equal
    itf_b = api.QMInterface(blobs=blobs, **kw)
    outs = []
    for itf in (itf_f, itf_b):
        mpc = api.SqpMpc(itf); wbc = api.HierarchicalWbc(itf)
        mpc.set_problem(cfg["t0"], cfg["x0"], cfg["ref_t"], cfg["ref_x"], cfg["ev"], cfg["modes"]); wbc.reset()
        mpc.control_step_resident(cfg["horizon"], cfg["period"], cfg["time"])
        res = mpc.download(); out, st = wbc.download(4)
        assert (res["status"] == 0).all() and (st == 0).all()
        outs.append((res["x"].copy(), res["u"].copy(), out.copy()))
        itf.close()
    for a, b in zip(*outs):
        assert np.array_equal(a, b)                                       # same blobs, same kernels: bit-equal
    oracle.set_schedule(cfg["ev"][0], cfg["modes"][0]); oracle.set_target(cfg["ref_t"][0], cfg["ref_x"][0])
    r = oracle.mpc_step(cfg["t0"][0], cfg["t0"][0] + cfg["horizon"], cfg["x0"][0]); n = len(r["t"])
    assert_blocks(outs[0][0][0, :n], r["x"], "x", 1e-6); assert_blocks(outs[0][1][0, :n], r["u"], "u", 1e-6)
    with pytest.raises(ValueError, match="Task file not found"):
        api.QMInterface("/nonexistent/task.info", URDF, REFI, **kw)


def test_plain_c_client_of_the_c_abi(oracle):
    """gcc-compiled C program linked against libqmhip.so: create from files, one MPC step, policy evaluation, one WBC step; its printed
    torques are checked against the oracle here"""
    from qm_control_amd import scenarios
    exe = os.path.join(ROOT, "tests", "_build", "c_abi_smoke")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    libdir = os.path.join(ROOT, "qm_control_amd")
    subprocess.check_call(["gcc", "-O1", "-std=c99", "-Wall", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c_abi_smoke.c"),
                           "-L" + libdir, "-lqmhip", "-Wl,-rpath," + libdir, "-lm", "-o", exe])
    p = subprocess.run([exe, URDF, TASK, REFI], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    vals = {}
    for line in p.stdout.splitlines():
        k, _, v = line.partition(":")
        vals[k.strip()] = np.array([float(t) for t in v.split()]) if v.strip() else None
    assert int(vals["status"][0]) == 0 and (vals["qp_status"] == 0).all()
    # the same problem on the oracle: C1 (stance, N = 20, nominal state), policy at t0, WBC on the measured state built from x0
    cfg = scenarios.make_config("C1")
    oracle.set_schedule(cfg["ev"][0], cfg["modes"][0]); oracle.set_target(cfg["ref_t"][0], cfg["ref_x"][0])
    r = oracle.mpc_step(0.0, cfg["horizon"], cfg["x0"][0])
    assert int(vals["num_nodes"][0]) == len(r["t"])
    xd, ud, mode = oracle.eval_policy(0.0)
    assert_blocks(vals["x_des"], xd, "x", 1e-6); assert_blocks(vals["u_des"], ud, "u", 1e-6); assert int(vals["mode"][0]) == mode
    oracle.wbc_reset()
    ref, st = oracle.wbc(xd, ud, oracle.rbd_from_q(cfg["x0"][0][6:30]), mode, 0.002, 20.0)
    assert_blocks(vals["wbc_out"], ref, "wbc", 1e-6)
