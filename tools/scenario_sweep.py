import sys; sys.path.insert(0, "/root/repo")
import numpy as np, time
from qm_control_amd import api, scenarios
for name, B, N in (("C5", 512, 150), ("C3", 2048, 40), ("C2", 64, 100), ("C1", 256, 20)):
    cfg = scenarios.make_config(name, batch=B, n_intervals=N)
    itf = api.QMInterface(blobs=scenarios.load_blobs(), max_batch=B, max_nodes=N + 60, max_ref_knots=cfg["ref_t"].shape[1], max_events=cfg["ev"].shape[1])
    mpc = api.SqpMpc(itf); wbc = api.HierarchicalWbc(itf)
    mpc.set_problem(cfg["t0"], cfg["x0"], cfg["ref_t"], cfg["ref_x"], cfg["ev"], cfg["modes"])
    wbc.reset(); mpc.control_step_resident(cfg["horizon"], cfg["period"], cfg["time"]); itf.synchronize()
    t = time.perf_counter()
    for _ in range(5): wbc.reset(); mpc.control_step_resident(cfg["horizon"], cfg["period"], cfg["time"])
    itf.synchronize(); t = (time.perf_counter() - t) / 5
    res = mpc.download(); out, qps = wbc.download(B)
    print(name, B, N, "status ok", bool((res["status"] == 0).all()), "qp status counts", np.bincount(qps.ravel(), minlength=3).tolist(), "ls_trials", res["ls_trials"], "ms/step %.3f" % (t * 1e3), "steps/s %.0f" % (B / t), "finite", bool(np.isfinite(out).all()))
    itf.close()
