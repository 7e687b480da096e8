"""tools/export_reference_cases.py — the inputs of the reference-side golden generator (adaptors/tools/dump_reference_goldens.cpp).

Writes adaptors/tools/reference_cases.txt: the MPC problems C1 / C2 / C5 of BASELINE.md (exactly the arrays tools/gen_golden.py feeds the oracle: observation, 2-knot
target trajectories, mode schedule, horizon) and WBC update inputs (both hierarchies, the time < 10 branch included) in a line-oriented text format a C++ program reads
without a JSON / numpy dependency.  Numbers are printed with 17 significant digits (exact f64 round trip).

  python tools/export_reference_cases.py                       # (re)write adaptors/tools/reference_cases.txt
  python tools/export_reference_cases.py --oracle-dump OUT     # ALSO run the CPU oracle on the cases and write OUT in the generator's OUTPUT format — a stand-in dump
                                                               # that exercises tools/import_reference_goldens.py + tests/test_reference_goldens.py end to end
                                                               # (tests/test_reference_goldens.py::test_pipeline_round_trip); it is NOT reference data
"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from qm_control_amd import scenarios

MPC_CASES = [("C1", 1, 20), ("C2", 1, 100), ("C5", 3, 150)]       # name, instances, N — the fixtures of tests/golden without the C3 batch
F = lambda a: " ".join("%.17g" % v for v in np.ravel(a))


def mpc_cases():
    for name, B, N in MPC_CASES:
        cfg = scenarios.make_config(name, batch=B, n_intervals=N)
        for b in range(B):
            yield name, b, N, cfg


def wbc_cases(oracle, blobs):
    """three measured / desired state pairs per hierarchy: stance, the two trot phases; one of them at time 5 (the arm-joint branch of HierarchicalWbc.cpp:24-30)"""
    from wbc_cases import random_wbc_inputs
    out = []
    for variant in (0, 1):
        for k, c in enumerate(random_wbc_inputs(oracle, blobs, 3, 71 + variant, 0.05, modes=(15, 9, 6))):
            c = dict(c); c["time"] = 5.0 if (variant == 0 and k == 2) else 20.0; c["variant"] = variant; c["name"] = "W%d_%d" % (variant, k); out.append(c)
    return out


def write_cases(path, wcases):
    with open(path, "w") as fh:
        fh.write("QM_REFERENCE_CASES 1\n")
        for name, b, N, cfg in mpc_cases():
            nref = cfg["ref_t"].shape[1]; nev = cfg["ev"].shape[1]
            fh.write("MPC %s %d intervals %d horizon %.17g t0 %.17g period %.17g time %.17g\n" % (name, b, N, cfg["horizon"], cfg["t0"][b], cfg["period"], cfg["time"]))
            fh.write("X0 %s\n" % F(cfg["x0"][b]))
            fh.write("TARGET %d\n" % nref)
            for k in range(nref):
                fh.write("%.17g %s\n" % (cfg["ref_t"][b, k], F(cfg["ref_x"][b, k])))
            fh.write("SCHEDULE %d\n%s\n%s\n" % (nev, F(cfg["ev"][b]), " ".join(str(int(m)) for m in cfg["modes"][b])))
        for c in wcases:
            fh.write("WBC %s variant %d mode %d period %.17g time %.17g\n" % (c["name"], c["variant"], c["mode"], 0.002, c["time"]))
            fh.write("XDES %s\nUDES %s\nRBD %s\nINPUTLAST %s\n" % (F(c["xd"]), F(c["ud"]), F(c["rbd"]), F(c["il"])))
        fh.write("END\n")


def oracle_dump(path, wcases):
    """the generator's output format, filled by the ORACLE (pipeline self-test only)"""
    import pyoracle
    o = pyoracle.Oracle(*pyoracle.load_blobs())
    with open(path, "w") as fh:
        fh.write("QM_REFERENCE_GOLDENS 1 source oracle-stand-in\n")
        for name, b, N, cfg in mpc_cases():
            o.set_schedule(cfg["ev"][b], cfg["modes"][b]); o.set_target(cfg["ref_t"][b], cfg["ref_x"][b])
            r = o.mpc_step(cfg["t0"][b], cfg["t0"][b] + cfg["horizon"], cfg["x0"][b])
            n = len(r["t"]); post = [i for i in range(n) if r["ev"][i] == 2]
            # PrimalSolution::timeTrajectory_ carries the INTERPOLATION times ([upstream] getInterpolationTime: PreEvent − limitEpsilon, PostEvent + limitEpsilon)
            tt = r["t"] + np.where(r["ev"] == 2, 2.220446049250313e-16, np.where(r["ev"] == 1, -2.220446049250313e-16, 0.0))
            fh.write("MPC %s %d nodes %d\n" % (name, b, n))
            fh.write("POSTEVENT %d %s\n" % (len(post), " ".join(map(str, post))))
            for i in range(n):
                fh.write("%.17g %s %s\n" % (tt[i], F(r["x"][i]), F(r["u"][i])))
            fh.write("PERF %s\n" % F(r["perf"][4:8]))
            xd, ud, mode = o.eval_policy(cfg["t0"][b])
            fh.write("POLICY %d %s %s\n" % (mode, F(xd), F(ud)))
            rbd = o.rbd_from_q(cfg["x0"][b][6:30]); o.wbc_reset(); w, st = o.wbc(xd, ud, rbd, mode, cfg["period"], cfg["time"])
            fh.write("STEPWBC %s\n" % F(w))
        for c in wcases:
            o.wbc_reset(); o.wbc(c["xd"], c["il"], c["rbd"], c["mode"], 0.002, c["time"], mpc_variant=bool(c["variant"]))
            w, st = o.wbc(c["xd"], c["ud"], c["rbd"], c["mode"], 0.002, c["time"], mpc_variant=bool(c["variant"]))
            fh.write("WBC %s variant %d\nOUT %s\n" % (c["name"], c["variant"], F(w)))
        fh.write("END\n")


def main(argv):
    import pyoracle
    blobs = scenarios.load_blobs(); o = pyoracle.Oracle(*pyoracle.load_blobs())
    wc = wbc_cases(o, blobs)
    out = os.path.join(ROOT, "adaptors", "tools", "reference_cases.txt")
    write_cases(out, wc); print("wrote", out)
    if "--oracle-dump" in argv:
        p = argv[argv.index("--oracle-dump") + 1]; oracle_dump(p, wc); print("wrote", p, "(oracle stand-in, NOT reference data)")


if __name__ == "__main__":
    main(sys.argv[1:])
