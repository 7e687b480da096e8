"""N>1 path on CPU: 2 processes over gloo exercise the sharding / max-over-ranks / gather helpers bench.py uses,
with the oracle standing in for the per-rank solve (the GPU path is identical per shard)."""
import os
import numpy as np
import torch.multiprocessing as mp


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "oracle"))
    import pyoracle
    from qm_control_amd import scenarios, sharding
    blobs = scenarios.load_blobs()
    cfg_all = scenarios.make_config("C3", batch=5, n_intervals=12)        # 5 instances over 2 ranks: 3 + 2
    cfg = sharding.shard_config(cfg_all, rank, world)
    bad, xf, uf, w = pyoracle.batch_step(*pyoracle.load_blobs(), 1, cfg["t0"], cfg["horizon"], cfg["x0"], cfg["ref_t"], cfg["ref_x"], cfg["ev"], cfg["modes"], cfg["period"], cfg["time"])
    dist.barrier()
    tmax = sharding.max_over_ranks(10.0 + rank, dist)
    pad = np.zeros((3, 54)); pad[:cfg["B"]] = w                            # all_gather needs equal shapes
    allw = sharding.gather_rows(pad, dist)
    if rank == 0:
        q.put((bad, tmax, allw, cfg["B"]))
    else:
        q.put((bad, tmax, None, cfg["B"]))
    dist.destroy_process_group()


def test_two_rank_sharding_over_gloo(blobs):
    from qm_control_amd import scenarios, sharding
    assert [sharding.shard_bounds(r, 2, 5) for r in range(2)] == [(0, 3), (3, 5)]
    assert [sharding.shard_bounds(r, 8, 8192) for r in (0, 7)] == [(0, 1024), (7168, 8192)]
    ctx = mp.get_context("spawn"); q = ctx.Queue(); port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = [q.get(timeout=300) for _ in range(2)]
    for p in procs: p.join(timeout=60)
    assert all(r[0] == 0 for r in res) and all(r[1] == 11.0 for r in res)
    allw = [r[2] for r in res if r[2] is not None][0]
    assert sorted(r[3] for r in res) == [2, 3]
    # unsharded reference
    import pyoracle
    cfg = scenarios.make_config("C3", batch=5, n_intervals=12)
    bad, xf, uf, w = pyoracle.batch_step(*pyoracle.load_blobs(), 2, cfg["t0"], cfg["horizon"], cfg["x0"], cfg["ref_t"], cfg["ref_x"], cfg["ev"], cfg["modes"], cfg["period"], cfg["time"])
    got = np.concatenate([allw[0:3], allw[3:5]])
    assert np.array_equal(got, w)                                            # sharding does not change any result
