"""tests/dist_bench_driver.py — one rank of bench.py's distributed code path on CPU: launched by tests/test_dist_gloo.py as
`python -m torch.distributed.run --nproc-per-node 2 ... tests/dist_bench_driver.py`.  bench.run() is the very function `python bench.py` executes;
only the engine is swapped: the product's kernels and launch sequence compiled for the host (tests/emu) instead of libqmhip.so on a GPU,
and gloo instead of RCCL.  Rank 0 prints the bench line plus the gathered torques of every rank as one JSON object."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
import bench  # noqa: E402


class EmuEngine:
    name = "emu"

    def __init__(self, cfg, local_rank):
        import emu_harness
        from qm_control_amd import scenarios
        mb, st = scenarios.load_blobs()
        self.cfg = cfg; self.B = cfg["B"]
        self.e = emu_harness.Emu(mb, st, self.B, cfg["n_intervals"] + 12, cfg["ref_t"].shape[1], cfg["ev"].shape[1])
        self.out = None; self.st = None

    def step(self):
        self.e.wbc_reset()                                   # like HipEngine.step: every step is the same cold problem (inputLast_ = 0)
        self.out, self.st, _ = self.e.control_step(self.cfg)

    def sync(self):
        pass

    def results(self):
        n = self.e.buf("n_nodes", (self.B,), np.int32); ev = self.e.node_arr("node_ev", 1, np.int32)
        n_intervals = int(sum(int(n[b]) - 1 - int((ev[:n[b], b] == 1).sum()) for b in range(self.B)))
        status = self.e.buf("status", (self.B,), np.int32)
        return dict(ok=bool((status == 0).all() and (self.st == 0).all()), out=self.out, n_intervals=n_intervals, ls_trials=0)

    def close(self):
        pass


def main():
    from qm_control_amd import sharding
    args = bench.parse_args(sys.argv[1:])
    holder = {}

    def make(cfg, local):
        holder["eng"] = EmuEngine(cfg, local); return holder["eng"]

    # bench.run destroys the process group at its end: gather the torques through a hook on close()
    import torch.distributed as dist
    orig_close = EmuEngine.close

    def close_and_gather(self):
        pad = np.zeros((self.B, 54)); pad[:self.B] = self.out      # equal shards in both modes (--batch per rank / --global-batch over the ranks)
        holder["all_out"] = sharding.gather_rows(pad, dist if dist.is_initialized() else None, "cpu")
        orig_close(self)
    EmuEngine.close = close_and_gather
    line = bench.run(args, make_engine=make, backend="gloo", device="cpu")
    if line is not None:
        print("BENCH_LINE " + json.dumps({"line": line, "all_out": holder["all_out"].tolist()}))


if __name__ == "__main__":
    main()
