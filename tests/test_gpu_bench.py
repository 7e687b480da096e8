"""GPU test of bench.py's own launcher and distributed branch on the one GPU of the test box: `--gpus N` self-launch is exercised as far as one device
allows — a world-size-1 RCCL process group under torch.distributed.run, the path N ranks take — and the line is checked against the contract."""
import json
import os
import socket
import subprocess
import sys

import pytest
from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _check_line(line, world, B, K, W, scaling="weak"):
    assert line["metric"].startswith("MPC+WBC control steps/sec") and line["unit"] == "steps/s" and line["dtype"] == "f64" and line["data"] == "synthetic"
    assert line["n_gpus"] == world and line["steps"] == K and line["warmup"] == W and line["scaling"] == scaling and line["higher_is_better"] is True and line["vs_baseline"] is None
    assert line["config"]["global_batch"] == world * B and line["roofline"]["counters"]["flops_pmc"] in ("current", "stale", None)
    assert line["config"]["all_status_ok"] and line["config"]["instances_per_gpu"] == B and line["config"]["engine"] == "hip"
    assert abs(line["value"] - world * B * K / (line["ms_per_step"] * K / 1e3)) <= 1e-6 * line["value"]
    r = line["roofline"]
    assert r["bound"] in ("hbm", "mfma") and 0.0 < r["frac"] < 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["avg_launch_ms"] > 0
    assert len(line["per_rank"]["seconds"]) == world


@pytest.mark.parametrize("mode", ["weak", "strong"])
def test_bench_rccl_branch_world_1(mode):
    B, K, W = 64, 2, 1
    env = dict(os.environ, QM_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", str(K), "--warmup", str(W)] + (["--batch", str(B)] if mode == "weak" else ["--global-batch", str(B)]) + ["--no-cpu-baseline", "--no-secondary"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    _check_line(json.loads(lines[0]), 1, B, K, W, mode)


def test_bench_default_line_with_secondaries_and_cpu_baseline():
    """plain `python bench.py` (small batch): roofline, cpu_baseline incl. the single-instance 1 / 3 thread figures, latency, C5 line"""
    B, K, W = 64, 3, 1
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(K), "--warmup", str(W), "--batch", str(B)], capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    _check_line(line, 1, B, K, W)
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1
    si = cb["single_instance_ms"]
    assert si["mpc_ms_1thread"] > 0 and si["mpc_ms_3threads"] > 0 and si["wbc_ms_1thread"] > 0
    assert line["latency_ms"]["B1_C2"] > 0 and line["latency_ms"]["C2_status_ok"]
    c5 = line["config_C5"]; assert c5["all_status_ok"] and c5["value"] > 0
    assert line["closed_loop_warm_start"]["all_status_ok"] and line["closed_loop_plant"]["all_status_ok"]
    # round 3: per-block parity figure, GPU counterparts of the reference's two timers, the strong-scaling point of SURVEY.md §8(d) C4
    assert set(cb["parity_on_sample"]) == {"vdot", "contact_forces", "torques"} and max(cb["parity_on_sample"].values()) < 1e-6
    sg = line["single_instance_ms_gpu"]; assert 0 < sg["wbc_ms_B1"] < sg["mpc_ms_B1"] < 50
    pi = line["pcie_inclusive"]; assert 0 < pi["value"] < line["value"] * 1.2 and pi["ms_per_step"] > pi["ms_step_only_unpipelined"] > 0      # the hand-over costs time on top of the step
    ss = line["strong_scaling_C4"]; assert ss["scaling"] == "strong" and ss["global_batch"] == 8192 and ss["instances_per_gpu"] == 8192 and ss["value"] > 0
    # round 4: the whole BASELINE config-4 batch solves — no failed instance; the one whose shooting node falls within 1e-6 s in front of a gait event carries the warning
    assert ss["all_status_ok"] and ss["instances_with_failed_mpc_status"] == 0 and ss["instances_with_nonzero_wbc_status"] == 0 and 1 <= ss["instances_with_mpc_warning"] <= 4
