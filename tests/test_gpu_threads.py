"""GPU test of the boundary under the reference's thread layout: mpcThread_ (QMController.cpp:315-333) beside the ros_control thread's
WbcBase::update (QMController.cpp:128-147), driven from plain C + pthreads (tests/c_abi_threads.c) — no Python in the timed loops."""
import os
import subprocess
import numpy as np
import pytest
from conftest import ROOT, assert_blocks

pytestmark = pytest.mark.gpu
DATA = os.path.join(ROOT, "tests", "data")
URDF, TASK, REFI = (os.path.join(DATA, f) for f in ("robot.urdf", "task.info", "reference.info"))


def _build():
    exe = os.path.join(ROOT, "tests", "_build", "c_abi_threads")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    libdir = os.path.join(ROOT, "qm_control_amd")
    subprocess.check_call(["gcc", "-O1", "-std=c99", "-Wall", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c_abi_threads.c"),
                           "-L" + libdir, "-lqmhip", "-Wl,-rpath," + libdir, "-lm", "-lpthread", "-o", exe])
    return exe


def _parse(stdout):
    phases, misc = {}, {}
    for line in stdout.splitlines():
        k, _, v = line.partition(":")
        toks = v.split()
        if k.startswith(("alone_", "threads_")):
            phases[k] = {toks[i]: float(toks[i + 1]) for i in range(0, len(toks), 2)}
        else:
            misc[k.strip()] = toks
    return phases, misc


def test_mpc_thread_beside_control_ticks_bit_exact_and_fast():
    """thread A: warm MPC solves at 100 Hz; thread B: qmhip_wbc_step at 500 Hz for 2.5 s, on its own WBC context and on the shared one.  Every WBC output
    equals the single-threaded run bit for bit; on its own context a tick stays below 1 ms while MPC solves are in flight."""
    exe = _build()
    p = subprocess.run([exe, URDF, TASK, REFI, "2.5"], capture_output=True, text=True, timeout=900)
    print(p.stdout)
    assert p.returncode == 0, p.stdout + p.stderr
    ph, misc = _parse(p.stdout)
    for name in ("threads_two_contexts", "threads_one_context", "alone_shared_context"):
        assert ph[name]["mismatches"] == 0 and ph[name]["tick_errors"] == 0 and ph[name]["bad_qp"] == 0, (name, ph[name])
    two = ph["threads_two_contexts"]
    assert two["ticks"] >= 1000 and two["mpc_solves"] >= 200 and two["mpc_bad_status"] == 0 and two["mpc_errors"] == 0, two     # >= 2 s of ticks beside ~100 Hz solves
    # a control tick never queues behind the MPC solve in flight: the undisturbed tick takes ~0.3 ms (0.42 ms at most), and beside 100 Hz solves it stays there —
    # 99.9 % of the ticks below 1 ms; a lone host-side outlier (thread wake-up / runtime lock, seen once in 1250 ticks at 1.3 ms) must stay inside the 2 ms period
    assert two["wbc_ms_p999"] < 1.0 and two["ticks_over_1ms"] <= 2 and two["wbc_ms_max"] < 2.0, two
    assert two["wbc_ms_mean"] < 1.25 * ph["alone_wbc_context"]["wbc_ms_mean"] + 0.02, (two, ph["alone_wbc_context"])
    assert two["late_ticks"] <= 0.02 * two["ticks"], two      # host timer jitter only (a tick that started more than one period late)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "threads_report.txt"), "w") as fh:
        fh.write(p.stdout)


def test_wbc_context_matches_oracle_and_refuses_mpc_calls(blobs, oracle):
    from qm_control_amd import api, scenarios
    cfg = scenarios.make_config("C3", batch=4, n_intervals=20)
    itf = api.QMInterface(blobs=blobs, max_batch=4, max_nodes=48, max_ref_knots=2, max_events=cfg["ev"].shape[1])
    witf = itf.wbc_context()
    mpc = api.SqpMpc(itf)
    mpc.set_problem(cfg["t0"], cfg["x0"], cfg["ref_t"], cfg["ref_x"], cfg["ev"], cfg["modes"]); mpc.solve_resident(cfg["horizon"])
    xd, ud, mode = mpc.evaluatePolicy(cfg["t0"])
    rbd = np.stack([oracle.rbd_from_q(cfg["x0"][b][6:30]) for b in range(4)])
    wbc = api.HierarchicalWbc(witf); wbc.reset()
    out, st = wbc.update(xd, ud, rbd, mode, 0.002, np.full(4, 20.0))
    wbc0 = api.HierarchicalWbc(itf); wbc0.reset()
    out0, st0 = wbc0.update(xd, ud, rbd, mode, 0.002, np.full(4, 20.0))
    assert np.array_equal(out, out0) and np.array_equal(st, st0)                    # same kernel, same model values: bit-equal across contexts
    for b in range(4):
        oracle.wbc_reset()
        ref, _ = oracle.wbc(xd[b], ud[b], rbd[b], int(mode[b]), 0.002, 20.0)
        assert_blocks(out[b], ref, "wbc", 1e-6)
    # partial batches take the five-copy path
    wbc.reset(); out2, _ = wbc.update(xd[:2], ud[:2], rbd[:2], mode[:2], 0.002, np.full(2, 20.0))
    assert np.array_equal(out2, out[:2])
    with pytest.raises(api.QmhipError, match="WBC-only"):
        api.SqpMpc(witf).set_problem(cfg["t0"], cfg["x0"], cfg["ref_t"], cfg["ref_x"], cfg["ev"], cfg["modes"])
    witf.close(); itf.close()


def test_fixed_rate_mpc_loop_never_fails():
    """60 s of controller time per raster (6000 warm MPC solves each, back to back) through the C ABI from plain C (tests/c_abi_fixed_rate.c): the review's 10 ms raster
    offset by 1.3 ms, a raster shared with the gait events, and a raster that puts a shooting node 5e-7 s in front of EVERY gait event.  No solve may fail (call error or
    negative status) — the reference's mpcThread_ answers a failed MPC_BASE::run by stopping the controller (QMController.cpp:327-330); the degenerate stages surface as
    warnings (status QM_MPC_WARN_PIVOT), one per gait event on the third raster."""
    exe = os.path.join(ROOT, "tests", "_build", "c_abi_fixed_rate"); os.makedirs(os.path.dirname(exe), exist_ok=True); libdir = os.path.join(ROOT, "qm_control_amd")
    subprocess.check_call(["gcc", "-O1", "-std=c99", "-Wall", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c_abi_fixed_rate.c"),
                           "-L" + libdir, "-lqmhip", "-Wl,-rpath," + libdir, "-lm", "-o", exe])
    p = subprocess.run([exe, URDF, TASK, REFI, "60"], capture_output=True, text=True, timeout=900)
    print(p.stdout)
    assert p.returncode == 0, p.stdout + p.stderr
    rows = {}
    for line in p.stdout.splitlines():
        k, _, v = line.partition(":"); toks = v.split()
        if k.startswith("raster_"):
            rows[k] = {toks[i]: float(toks[i + 1]) for i in range(0, len(toks), 2)}
    assert len(rows) == 3
    for k, r in rows.items():
        assert r["solves"] == 6000 and r["failed"] == 0 and r["call_errors"] == 0, (k, r)
    assert rows["raster_0"]["warnings"] == 0 and rows["raster_2"]["warnings"] >= 168, rows          # 60 s / 0.35 s = 171 gait events
    assert rows["raster_2"]["smallest_gap_before_an_event"] < 1e-6
    with open(os.path.join(ROOT, "gpurun_out", "fixed_rate_report.txt"), "w") as fh:
        fh.write(p.stdout)
