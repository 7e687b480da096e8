"""tools/plant_multi_ctx.py [contexts] [batch] [ticks] — the whole controller around the plant (1 ms ticks, MPC every 10th) with the batch split over several contexts, one host
thread and stream set each: a context's tick chain [plant -> estimate -> policy -> WBC -> command] advances as soon as ITS slowest WBC instance is done instead of the batch's,
so the sub-batches fill each other's tails.  Prints instance-ticks per second for 1 context and for the split."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
from qm_control_amd import api
from sim_closed_loop_demo import setup

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4; B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024; ticks = int(sys.argv[3]) if len(sys.argv) > 3 else 200
horizon = 1.0


def make(Bk, seed):
    rng = np.random.default_rng(seed)
    c = setup("trot", Bk, horizon)
    q = np.tile(c["xbar"][6:30], (Bk, 1)); q[:, 2] = 0.385; q[:, 6:18] += 0.03 * rng.normal(size=(Bk, 12)); q[:, 18:] += 0.1 * rng.normal(size=(Bk, 6)); q[:, 5] += 0.1 * rng.normal(size=Bk)
    itf = api.QMInterface(blobs=(c["mb"], c["st"]), max_batch=Bk, max_nodes=128, max_ref_knots=2, max_events=c["ev"].shape[1])
    mpc = api.SqpMpc(itf); wbc = api.HierarchicalWbc(itf); sim = api.QMHWSim(itf, robust_grid=True)
    sim.reset(q, np.zeros((Bk, 24)), 20.0); rbd0, _ = sim.step(1e-9, 1)
    for b in range(Bk):
        c["ref_x"][b, :, 30:37] = rbd0[b, 48:55]; c["ref_x"][b, :, 11] = q[b, 5]; c["ref_x"][b, :, 9] = 0.0
    mpc.set_problem(c["t0"], c["x0"], c["ref_t"], c["ref_x"], c["ev"], c["modes"]); wbc.reset(); sim.reset(q, np.zeros((Bk, 24)), 20.0)
    sim.closed_loop(600, 0.001, horizon, n_substeps=2, mpc_every=10); itf.synchronize()      # into the trot
    return dict(itf=itf, mpc=mpc, wbc=wbc, sim=sim, B=Bk)


def run(ctxs, n):
    def work(c):
        c["sim"].closed_loop(n, 0.001, horizon, n_substeps=2, mpc_every=10); c["itf"].synchronize()
    th = [threading.Thread(target=work, args=(c,)) for c in ctxs]
    t = time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    return time.perf_counter() - t


one = [make(B, 7)]
t1 = run(one, ticks); print("1 context  x %4d instances: %.0f instance-ticks/s (%.3f ms per tick)" % (B, B * ticks / t1, t1 / ticks * 1e3))
ok1 = bool((one[0]["mpc"].download()["status"] == 0).all()); one[0]["itf"].close()
many = [make(B // K, 7 + k) for k in range(K)]
tk = run(many, ticks); print("%d contexts x %4d instances: %.0f instance-ticks/s (%.3f ms per tick of the whole batch)  ratio %.2f" % (K, B // K, B * ticks / tk, tk / ticks * 1e3, t1 / tk))
print("status ok:", ok1, all(bool((c["mpc"].download()["status"] == 0).all()) for c in many))
