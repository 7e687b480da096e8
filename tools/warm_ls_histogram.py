"""tools/warm_ls_histogram.py [n_instances] [n_steps] — why does the warm-started solve need more than one line-search trial?  (round-5 review, What's weak #3)

Runs the bench's `closed_loop_warm_start` leg on the CPU ORACLE (test infrastructure; the device loop equals it to 1e-12 per step, tests/test_gpu_mpc.py): BASELINE config 3
instances, one cold solve, then warm-started solves with the observation advanced along the policy by mpc_dt = 0.01 s (qm_advance_kernel's perfect-tracking plant).  For every
solve it records the filter line search's trials — step length, merit, constraint violation theta, the branch of the filter that decided, accepted or not — and prints the
histogram of accepted step lengths and of the REJECTING branch.  Writes profiles/r06_warm_ls_histogram.json."""
import json, os, sys, collections
import numpy as np
from concurrent.futures import ProcessPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))


def run_instance(args):
    b, n_steps = args
    import pyoracle
    from qm_control_amd import scenarios
    cfg = scenarios.make_config("C3", batch=max(b + 1, 64))
    mb, st = pyoracle.load_blobs(); o = pyoracle.Oracle(mb, st); o.set_threads(1)
    o.set_schedule(cfg["ev"][b], cfg["modes"][b]); o.set_target(cfg["ref_t"][b], cfg["ref_x"][b])
    t0 = float(cfg["t0"][b]); x0 = cfg["x0"][b].copy(); rows = []
    for k in range(n_steps):
        if k > 0:
            t0 += 0.01; x0, _, _ = o.eval_policy(t0)
        r = o.mpc_step(t0, t0 + cfg["horizon"], x0, warm=(k > 0)); tr = o.ls_trace()
        rows.append(dict(step=k, alpha=float(r["alpha"]), trials=int(r["ls_trials"]), trace=tr.tolist()))
    return b, rows


def main():
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 128; ns = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    with ProcessPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        res = list(ex.map(run_instance, [(b, ns) for b in range(nb)]))
    br_names = {0: "theta > g_max: needs theta < (1 - 1e-6) theta0", 1: "theta, theta0 < g_min and descent: Armijo on the cost", 2: "otherwise: cost < cost0 - 1e-6 theta0 OR theta < (1 - 1e-6) theta0"}
    out = dict(instances=nb, steps=ns, per_step=[])
    for k in range(ns):
        acc = collections.Counter(); rej = collections.Counter(); trials = collections.Counter(); th0 = []; worst = None
        for b, rows in res:
            r = rows[k]; acc["%g" % r["alpha"]] += 1; trials[r["trials"]] += 1; th0.append(r["trace"][0][2])
            for t in r["trace"][1:]:
                if t[4] == 0.0: rej[br_names[int(t[3])]] += 1
            if r["trials"] > 1 and (worst is None or r["trials"] > worst["trials"]): worst = dict(instance=b, trials=r["trials"], trace=r["trace"])
        out["per_step"].append(dict(step=k, warm=k > 0, accepted_alpha=dict(acc), trials=dict(trials), rejections_by_branch=dict(rej), theta0_median=float(np.median(th0)), theta0_max=float(np.max(th0)), worst=worst))
        print("step %2d %s  alpha %s  trials %s  rejected-by %s  theta0 med %.2e max %.2e" % (k, "warm" if k else "cold", dict(acc), dict(trials), dict(rej), np.median(th0), np.max(th0)))
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "profiles", "r06_warm_ls_histogram.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
