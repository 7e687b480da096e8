"""tools/import_reference_goldens.py — output of adaptors/tools/dump_reference_goldens.cpp (run by a maintainer inside the reference's catkin workspace) ->
tests/golden_ref/*.npz in the layout of tests/golden (keys t_b, ev_b, mode_b, x_b, u_b, perf_b, policy_*_b, wbc_b) + golden_ref/wbc_cases.npz.

  python tools/import_reference_goldens.py reference_goldens.txt [--cases adaptors/tools/reference_cases.txt] [--out tests/golden_ref]

The dump carries what ocs2::PrimalSolution holds: interpolation times (PreEvent / PostEvent nodes nudged by ∓ limitEpsilon, [upstream] getInterpolationTime) and
postEventIndices_.  The importer restores the solver's node view: event tags from the post-event indices (node i PostEvent => node i − 1 PreEvent), node times of event
nodes snapped to the schedule's event time, node modes = modeAtTime(interval start) on the case's schedule."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tokens(path):
    with open(path) as fh:
        for line in fh:
            yield line.split()


def read_cases(path):
    mpc, wbc = {}, {}; cur = None
    rows = [r for r in _tokens(path) if r]
    assert rows[0][0] == "QM_REFERENCE_CASES"
    i = 1
    while i < len(rows):
        r = rows[i]
        if r[0] == "END": break
        if r[0] == "MPC":
            c = dict(name=r[1], instance=int(r[2]), intervals=int(r[4]), horizon=float(r[6]), t0=float(r[8]), period=float(r[10]), time=float(r[12]))
            c["x0"] = np.array(rows[i + 1][1:], float); n = int(rows[i + 2][1])
            ref = np.array([rows[i + 3 + k] for k in range(n)], float); c["ref_t"] = ref[:, 0]; c["ref_x"] = ref[:, 1:]
            m = int(rows[i + 3 + n][1]); c["ev"] = np.array(rows[i + 4 + n], float); c["modes"] = np.array(rows[i + 5 + n], int); assert len(c["ev"]) == m and len(c["modes"]) == m + 1
            mpc[(c["name"], c["instance"])] = c; i += 6 + n
        elif r[0] == "WBC":
            c = dict(name=r[1], variant=int(r[3]), mode=int(r[5]), period=float(r[7]), time=float(r[9]))
            for k, key in enumerate(("xd", "ud", "rbd", "il")): c[key] = np.array(rows[i + 1 + k][1:], float)
            wbc[c["name"]] = c; i += 5
        else:
            raise ValueError("case file: unknown record %s" % r[0])
    return mpc, wbc


def read_dump(path):
    rows = [r for r in _tokens(path) if r]
    assert rows[0][0] == "QM_REFERENCE_GOLDENS" and rows[0][1] == "1", rows[0]
    source = rows[0][3] if len(rows[0]) > 3 else "?"
    mpc, wbc = {}, {}; i = 1
    while i < len(rows):
        r = rows[i]
        if r[0] == "END": break
        if r[0] == "MPC":
            name, inst, n = r[1], int(r[2]), int(r[4]); post = [int(v) for v in rows[i + 1][2:]]; assert int(rows[i + 1][1]) == len(post)
            body = np.array(rows[i + 2:i + 2 + n], float); assert body.shape == (n, 61), body.shape
            perf = np.array(rows[i + 2 + n][1:], float); pol = rows[i + 3 + n]; step = np.array(rows[i + 4 + n][1:], float)
            assert rows[i + 2 + n][0] == "PERF" and pol[0] == "POLICY" and rows[i + 4 + n][0] == "STEPWBC"
            mpc[(name, inst)] = dict(tt=body[:, 0], x=body[:, 1:31], u=body[:, 31:61], post=post, perf=perf, policy_mode=int(pol[1]), policy_x=np.array(pol[2:32], float), policy_u=np.array(pol[32:62], float), stepwbc=step)
            i += 5 + n
        elif r[0] == "WBC":
            assert rows[i + 1][0] == "OUT"; wbc[r[1]] = dict(variant=int(r[3]), out=np.array(rows[i + 1][1:], float)); i += 2
        else:
            raise ValueError("dump: unknown record %s" % r[0])
    return source, mpc, wbc


def node_view(case, rec):
    """solver-node view of a dumped primal solution: node times, event tags (0 none, 1 PreEvent, 2 PostEvent), node modes"""
    n = len(rec["tt"]); ev = np.zeros(n, np.int32)
    for i in rec["post"]:
        ev[i] = 2; ev[i - 1] = 1
    t = rec["tt"].copy(); sched = case["ev"]
    for i in np.nonzero(ev)[0]:
        k = int(np.argmin(np.abs(sched - t[i]))); assert abs(sched[k] - t[i]) < 1e-9, (t[i], sched[k]); t[i] = sched[k]
    ts = t + np.where(ev == 2, 1e-6, 0.0)                                     # interval start of a PostEvent node: + weakEpsilon
    mode = case["modes"][np.searchsorted(sched, ts, side="left")].astype(np.int32)      # modeAtTime: an event AT t still belongs to the earlier phase
    return t, ev, mode


def main(argv):
    if not argv: print(__doc__); return 2
    dump = argv[0]
    cases = argv[argv.index("--cases") + 1] if "--cases" in argv else os.path.join(ROOT, "adaptors", "tools", "reference_cases.txt")
    out = argv[argv.index("--out") + 1] if "--out" in argv else os.path.join(ROOT, "tests", "golden_ref")
    cm, cw = read_cases(cases); source, dm, dw = read_dump(dump)
    os.makedirs(out, exist_ok=True)
    groups = {}
    for (name, inst), rec in dm.items():
        groups.setdefault(name, {})[inst] = rec
    for name, insts in groups.items():
        B = len(insts); assert sorted(insts) == list(range(B)); N = cm[(name, 0)]["intervals"]; arrays = {"source": np.array(source)}
        for b in range(B):
            rec = insts[b]; t, ev, mode = node_view(cm[(name, b)], rec)
            arrays.update({"t_%d" % b: t, "ev_%d" % b: ev, "mode_%d" % b: mode, "x_%d" % b: rec["x"], "u_%d" % b: rec["u"], "perf_after_%d" % b: rec["perf"],
                           "policy_x_%d" % b: rec["policy_x"], "policy_u_%d" % b: rec["policy_u"], "policy_mode_%d" % b: np.array(rec["policy_mode"]), "wbc_%d" % b: rec["stepwbc"]})
        np.savez_compressed(os.path.join(out, "%s_B%d_N%d.npz" % (name, B, N)), **arrays); print("wrote %s_B%d_N%d.npz (source %s)" % (name, B, N, source))
    if dw:
        arrays = {"source": np.array(source), "names": np.array(sorted(dw))}
        for nm in sorted(dw):
            c = cw[nm]; assert c["variant"] == dw[nm]["variant"]
            arrays.update({nm + "_out": dw[nm]["out"], nm + "_xd": c["xd"], nm + "_ud": c["ud"], nm + "_rbd": c["rbd"], nm + "_il": c["il"],
                           nm + "_meta": np.array([c["variant"], c["mode"], c["period"], c["time"]], float)})
        np.savez_compressed(os.path.join(out, "wbc_cases.npz"), **arrays); print("wrote wbc_cases.npz (%d cases)" % len(dw))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
