"""N>1 path on CPU (world size 2, gloo): bench.py's OWN distributed code path — process-group init from the torch.distributed.run environment,
contiguous shards, barrier + max-over-ranks around the timed region, gathered per-rank vector, n_gpus / value arithmetic — run on the product's
kernels compiled for the host (tests/emu), and checked against the oracle on the unsharded batch."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
from conftest import ROOT, assert_blocks


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def test_shard_bounds():
    from qm_control_amd import sharding
    assert [sharding.shard_bounds(r, 2, 5) for r in range(2)] == [(0, 3), (3, 5)]
    assert [sharding.shard_bounds(r, 8, 8192) for r in (0, 7)] == [(0, 1024), (7168, 8192)]


@pytest.mark.parametrize("mode", ["weak", "strong"])
def test_bench_code_path_two_ranks_over_gloo(oracle, mode):
    """weak: --batch B per rank (the default headline); strong: --global-batch 2 B cut into two contiguous shards — the same instances either way"""
    import emu_harness
    emu_harness.build()                                                    # once, before two ranks race for the build
    B, N, K, W = 2, 8, 2, 1
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "dist_bench_driver.py"), "--gpus", "2", "--steps", str(K), "--warmup", str(W)] + (["--batch", str(B)] if mode == "weak" else ["--global-batch", str(2 * B)]) + ["--n-intervals", str(N), "--no-cpu-baseline", "--no-secondary"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    payload = [l for l in p.stdout.splitlines() if l.startswith("BENCH_LINE ")]
    assert len(payload) == 1, p.stdout[-2000:]                             # ONE line, from rank 0
    d = json.loads(payload[0][len("BENCH_LINE "):]); line = d["line"]; allw = np.array(d["all_out"])
    assert line["n_gpus"] == 2 and line["steps"] == K and line["warmup"] == W and line["scaling"] == mode and line["config"]["engine"] == "emu" and line["config"]["global_batch"] == 2 * B
    assert line["config"]["all_status_ok"] and line["config"]["instances_per_gpu"] == B and line["config"]["parallelism"] == "shard2"
    secs = line["per_rank"]["seconds"]
    assert len(secs) == 2 and len(line["per_rank"]["intervals_per_launch"]) == 2
    # whole-job aggregate over both ranks, priced on the SLOWEST rank's clock
    assert abs(line["ms_per_step"] * K / 1e3 - max(secs)) <= 0.25 * max(secs) + 0.05
    assert abs(line["value"] - 2 * B * K / (line["ms_per_step"] * K / 1e3)) <= 1e-6 * line["value"]
    # the shards are the contiguous halves of the C4 batch of 2 B instances: every rank's torques equal the oracle's on the unsharded batch
    from qm_control_amd import scenarios
    import pyoracle
    cfg = scenarios.make_config("C4", batch=2 * B, n_intervals=N)
    bad, xf, uf, w = pyoracle.batch_step(*pyoracle.load_blobs(), 2, cfg["t0"], cfg["horizon"], cfg["x0"], cfg["ref_t"], cfg["ref_x"], cfg["ev"], cfg["modes"], cfg["period"], cfg["time"])
    assert bad == 0 and allw.shape == (2 * B, 54)
    for b in range(2 * B):
        assert_blocks(allw[b], w[b], "wbc", 1e-6, b)
