"""bench.py — MPC+WBC control steps/sec of the 24-DoF quadruped-manipulator at horizon N=100 (BASELINE.json metric).

One "step" = one multiple-shooting SQP iteration over the horizon + policy evaluation at t0 + one 3-level
hierarchical WBC solve, for one instance.  Workload per GPU: configuration C3/C4 of
BASELINE.md — trot gait, N=100, 1024 independent instances with random initial states (seeded), inputs resident
in HBM before the timed region.  Instances are independent, so N GPUs run N contiguous shards with NO data-path
collective (weak scaling); torch.distributed (RCCL) carries the barrier, the max-over-ranks time and a gathered
per-rank timing vector.

Usage:  python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--no-cpu-baseline] [--no-secondary]
        --gpus N > 1 without WORLD_SIZE in the environment: bench.py re-launches itself as N ranks under
        `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` (what the driver does itself).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Per-unit work model of the three big kernels: unit = one non-event shooting interval (lq, riccati) or one instance (wbc).
#   flop : the survey's dense FP64 ESTIMATE of SURVEY.md §8(d) (LQ 190 k + projection 410 k; Riccati 250 k; WBC 2.0 M) — only the fall-back: the flops the kernels
#          actually ISSUE are measured with SQ PMC counters (tools/gpu_round_profile.sh, tools/flops_pmc_digest.py -> profiles/flops_pmc.json) and replace the
#          estimate whenever that file is present; the estimate stays in the line as `survey_dense_*`
#   bytes: HBM bytes the design moves per unit (DESIGN.md §4 derives every figure; `doubles` below are f64 counts)

def _kernel_model():
    # K1b writes one compact stage record per interval and reads the kin record + node inputs
    from qm_control_amd import record_model as rm
    return {
        "lq": {"flop": 190e3 + 410e3, "bytes": (rm.LQ_WRITE_DOUBLES + rm.LQ_READ_DOUBLES) * 8.0, "unit": "interval"},
        "riccati": {"flop": 250e3, "bytes": (rm.RICCATI_BWD_READ_DOUBLES + rm.RICCATI_BWD_WRITE_DOUBLES + rm.RICCATI_FWD_READ_DOUBLES + rm.RICCATI_FWD_IO_DOUBLES) * 8.0, "unit": "interval"},
        "wbc": {"flop": 2.0e6, "bytes": rm.WBC_IO_DOUBLES * 8.0, "unit": "instance"},
    }


FP64_PEAK_TFLOPS = 78.6                 # MI355X dense FP64 matrix peak = FP64 vector peak (AMD public figure; measured on the device by this bench: fp64_peak_measured)
HBM_PEAK_TBS = 8.0                      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (≈6.3 TB/s achievable)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=1024, help="instances per GPU (weak scaling: the global batch grows with --gpus)")
    ap.add_argument("--global-batch", type=int, default=0, help="STRONG scaling (SURVEY.md §8(d) C4 / BASELINE.json config 4): a fixed global batch, e.g. 8192, cut into "
                    "contiguous shards of global/G instances for G GPUs; overrides --batch")
    ap.add_argument("--n-intervals", type=int, default=100, help="horizon N (the metric is quoted at 100; other values are for tests)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary figures (closed loops, latency, C5)")
    ap.add_argument("--no-plant-loop", action="store_true", help="skip only the closed-loop-around-the-plant figure")
    return ap.parse_args(argv)


def launch_ranks(n, argv):
    """--gpus N given to a plain `python bench.py`: become N ranks, one per GPU, under torch.distributed.run (RCCL over xGMI)"""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


class HipEngine:
    """the product: libqmhip.so through the C ABI (qm_control_amd/api.py); inputs resident in HBM after construction"""
    name = "hip"

    def __init__(self, cfg, local_rank, max_nodes=None):
        from qm_control_amd import api, scenarios
        self.api = api; self.cfg = cfg; self.B = cfg["B"]
        nm = max_nodes or (cfg["n_intervals"] + 28)
        import torch
        torch.cuda.synchronize(local_rank); free0 = torch.cuda.mem_get_info(local_rank)[0]       # device-wide free bytes: sees the library's own hipMalloc calls
        self.max_nodes = nm
        self.itf = api.QMInterface(blobs=scenarios.load_blobs(), device=local_rank, max_batch=self.B, max_nodes=nm, max_ref_knots=2, max_events=cfg["ev"].shape[1])
        self.mpc = api.SqpMpc(self.itf); self.wbc = api.HierarchicalWbc(self.itf)
        self.upload(cfg)
        self.itf.synchronize(); self.device_bytes = int(free0 - torch.cuda.mem_get_info(local_rank)[0])    # MEASURED footprint of the contexts + uploaded inputs
        for key in ("riccati_skip", "wbc_stop", "lq_prof", "lq_debug", "lds_pad"):      # profiling-only switches make results meaningless: they must all be off
            assert self.itf.debug_get(key) == 0, key
        for key, want in (("ls_device_tail", 1), ("fused_policy", 1), ("wbc_defer", 0), ("filler_at_lq", 0), ("r_dense", 0)):      # the product's launch order, not one of the A/B orders
            assert self.itf.debug_get(key) == want, key

    def upload(self, cfg):
        self.cfg = cfg
        self.mpc.set_problem(cfg["t0"], cfg["x0"], cfg["ref_t"], cfg["ref_x"], cfg["ev"], cfg["modes"])

    def step(self):
        self.wbc.reset()
        self.mpc.control_step_resident(self.cfg["horizon"], self.cfg["period"], self.cfg["time"])

    def sync(self):
        self.itf.synchronize()

    def results(self):
        res = self.mpc.download(); out, qps = self.wbc.download(self.cfg["B"])
        n_intervals = int(sum(int(res["num_nodes"][b]) - 1 - int((res["event"][b, :res["num_nodes"][b]] == 1).sum()) for b in range(self.cfg["B"])))
        # status >= 0 is success (include/qmhip.h): > 0 are warning bits on a valid solution (QM_MPC_WARN_PIVOT: zeroed Riccati pivots on the stage in front of a gait event)
        return dict(ok=bool((res["status"] >= 0).all() and (qps == 0).all()), out=out, n_intervals=n_intervals, ls_trials=int(res["ls_trials"]),
                    n_bad_mpc=int((res["status"] < 0).sum()), n_warn_mpc=int((res["status"] > 0).sum()), n_bad_wbc=int((qps != 0).any(axis=1).sum()))

    def close(self):
        self.itf.close()


def timed_region(engine, steps, dist, device):
    """EXACTLY `steps` steps between barrier + device synchronisation on both sides; returns (max over ranks of the elapsed seconds, this rank's seconds)"""
    from qm_control_amd import sharding
    sharding.barrier(dist, device); engine.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        engine.step()
    engine.sync(); sharding.barrier(dist, device)
    mine = time.perf_counter() - t0
    return sharding.max_over_ranks(mine, dist, device), mine


def run(args, make_engine=HipEngine, backend="nccl", device="cuda"):
    """the whole benchmark on this rank; returns the JSON line (dict) on rank 0, None elsewhere.  `make_engine`, `backend`, `device` exist so that the
    world-size-2 gloo test (tests/test_dist_gloo.py) drives THIS code path with the host-emulated kernels."""
    import numpy as np
    from qm_control_amd import scenarios, sharding

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = sharding.init_distributed(backend, local) if (world > 1 or os.environ.get("QM_BENCH_FORCE_DIST")) else None     # (the env switch exercises the RCCL path on one GPU)
    strong = getattr(args, "global_batch", 0) > 0
    if strong and args.global_batch % world:
        raise SystemExit("--global-batch %d is not divisible by %d ranks" % (args.global_batch, world))
    B = args.global_batch // world if strong else args.batch
    # C4: seed 1235, contiguous shard of the global batch for this rank
    cfg = sharding.shard_config(scenarios.make_config("C4", batch=B * world, n_intervals=args.n_intervals), rank, world)
    eng = make_engine(cfg, local)
    hip = getattr(eng, "name", "") == "hip"
    itf = eng.itf if hip else None

    peak_mfma = peak_fma = None
    if hip:
        # roofline denominators measured on this very device before anything is timed ("microbenchmark it first"): FP64 matrix and
        # vector FMA throughput, ≈ 0.6 s of sustained FP64 work (which also brings a freshly booted device to its sustained clocks)
        peak_mfma = max(itf.microbench_fp64(True) for _ in range(12)); peak_fma = max(itf.microbench_fp64(False) for _ in range(6))
    for _ in range(args.warmup):
        eng.step()
    eng.sync()
    kms_all = {}
    if hip:
        # per-kernel times of every kernel from a short untimed pass; inside the timed region only the three modelled kernels (lq, riccati, wbc — the roofline's
        # avg_launch_ms) carry HIP-event spans: two event records cost about one launch, 22 of them per step would cost 2 % of the headline
        itf.set_profiling(True); itf.reset_kernel_ms()
        for _ in range(5):
            eng.step()
        eng.sync(); itf.set_profiling(False)
        # "lq" = both product instances of the LQ kernel (qm_lq_kernel: nodes with m <= 16; qm_lq_m18_kernel: stance nodes, launched only when the grid has a phase with
        # three or four stance feet — never on the headline trot workload): the roofline credits flops and bytes for ALL intervals, so the time must cover both launches
        lq_both = lambda: (itf.kernel_ms("lq")[0] + itf.kernel_ms("lq_m18")[0], itf.kernel_ms("lq")[1])
        kms_all = {k: (lq_both() if k == "lq" else itf.kernel_ms(k)) for k in ("grid", "lq_kin", "lq", "riccati", "ls_eval", "ls_misc", "policy", "wbc")}
        # Inside the timed region only the DOMINANT kernel (the LQ kernel, whose avg_launch_ms the roofline is priced on) carries HIP-event spans — round 6; rounds 1-5 also
        # spanned riccati and wbc there: four more event records per step on two streams (~ 0.5 % of the headline).  Their averages come from the five untimed steps above
        # (same process, same buffers, seconds earlier); QM_BENCH_SPANS=2 restores the three-kernel spans for an A/B.
        span_level = int(os.environ.get("QM_BENCH_SPANS", "3"))
        itf.set_profiling(span_level); itf.reset_kernel_ms()
    elapsed, mine = timed_region(eng, args.steps, dist, device)
    kms = {}
    if hip:
        itf.set_profiling(False)      # per-kernel HIP-event times over the timed region (events recorded on the stream each kernel runs on)
        kms = {k: (lq_both() if k == "lq" else (itf.kernel_ms(k) if span_level == 2 else kms_all[k])) for k in ("lq", "riccati", "wbc")}
    res = eng.results()
    avg = lambda k: (kms[k][0] / max(1, kms[k][1])) if k in kms else 0.0
    # every rank's {seconds of the timed region, avg launch ms of the modelled kernels, all statuses ok, intervals per launch}: what RCCL is used for here
    per_rank = sharding.gather_rows(np.array([[mine, avg("lq"), avg("riccati"), avg("wbc"), float(res["ok"]), float(res["n_intervals"])]]), dist, device)

    # ---- secondary figures (NOT the headline value) ----
    sec = {}
    if hip and not args.no_secondary:
        sec = secondary_figures(args, eng, cfg, dist, device, world, rank)

    line = None
    if rank == 0:
        KM = _kernel_model()
        n_intervals = res["n_intervals"]

        def roof(name):
            ms = avg(name); mdl = KM[name]; units = n_intervals if mdl["unit"] == "interval" else B
            tf = mdl["flop"] * units / (ms * 1e-3) / 1e12 if ms > 0 else 0.0; tb = mdl["bytes"] * units / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            return {"kernel": "qm_%s_kernel" % name, "avg_launch_ms": ms, "units_per_launch": units, "unit": mdl["unit"], "flop_per_launch": mdl["flop"] * units, "bytes_per_launch": mdl["bytes"] * units,
                    "tflops": tf, "frac_fp64": tf / FP64_PEAK_TFLOPS, "tbs": tb, "frac_hbm": tb / HBM_PEAK_TBS}
        roofline = None; roofs = {}; flops_stale = traffic_stale = None
        from qm_control_amd import record_model as _rm
        src_hash = _rm.kernel_source_hash()
        if hip:
            roofs = {k: roof(k) for k in KM}
            # flops the kernels ISSUE, from the SQ instruction counters of this same command (tools/gpu_round_profile.sh -> profiles/flops_pmc.json: a STATIC file of the
            # named profile round, PMC collection cannot run inside the timed bench).  When present it REPLACES the survey's dense estimate as the kernel's flop count:
            # frac_fp64 then is an achieved-flops figure (tile padding and masked MFMA slots included, i.e. still an upper bound of the useful flops)
            flop_src = "SURVEY.md §8(d) dense estimate (profiles/flops_pmc.json not found)"
            try:
                with open(os.path.join(ROOT, "profiles", "flops_pmc.json")) as fh:
                    fp = json.load(fh)
                flops_stale = fp.get("kernel_source_hash") != src_hash
                for k, v in roofs.items():
                    f = fp["kernels"].get(v["kernel"], {}).get("flops_per_launch")
                    if f and v["avg_launch_ms"] > 0 and B == 1024 and args.n_intervals == 100:      # the counters were collected on the default workload
                        v["survey_dense_flop_per_launch"] = v["flop_per_launch"]; v["survey_dense_frac_fp64"] = v["frac_fp64"]
                        v["flop_per_launch"] = f; v["tflops"] = f / (v["avg_launch_ms"] * 1e-3) / 1e12; v["frac_fp64"] = v["tflops"] / FP64_PEAK_TFLOPS
                    share = fp["kernels"].get(v["kernel"], {}).get("valu_non_f64_share")
                    if share is not None: v["valu_non_f64_share"] = share      # share of the kernel's vector instructions that are not FP64 arithmetic (addressing, masks, moves): same counter passes
                if B == 1024 and args.n_intervals == 100: flop_src = "instrumented: SQ_INSTS_VALU_{FMA,MUL,ADD,TRANS}_F64 + SQ_INSTS_VALU_MFMA_MOPS_F64 per launch, profiles/flops_pmc.json round %s (same command, B=1024; not measured in this run)%s" % (
                    fp.get("round", "?"), " — STALE: collected on kernel sources %s, this tree is %s" % (fp.get("kernel_source_hash"), src_hash) if flops_stale else "")
            except (OSError, KeyError, ValueError):
                pass
            dom = max(roofs, key=lambda k: roofs[k]["avg_launch_ms"]); rd = roofs[dom]      # the dominant kernel = the longest average launch among the modelled ones
            if rd["frac_hbm"] >= rd["frac_fp64"]:
                roofline = {"bound": "hbm", "kernel": rd["kernel"], "achieved": rd["tbs"] * 1e3, "peak": HBM_PEAK_TBS * 1e3, "unit": "GB/s", "frac": rd["frac_hbm"], "traffic": None}
            else:
                roofline = {"bound": "mfma", "kernel": rd["kernel"], "achieved": rd["tflops"], "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": rd["frac_fp64"], "traffic": None}
            roofline.update({"avg_launch_ms": rd["avg_launch_ms"], "flop_per_launch": rd["flop_per_launch"], "bytes_per_launch": rd["bytes_per_launch"], "flop_model": flop_src,
                             "frac_hbm": rd["frac_hbm"], "frac_fp64": rd["frac_fp64"],
                             "avg_launch_ms_source": ("HIP events around every launch of this kernel over the timed region, on its stream" if (span_level == 2 or dom == "lq") else
                                                      "HIP events over five untimed steps right before the timed region (the timed region spans the LQ kernel only; QM_BENCH_SPANS=2 spans all three)"),
                             "note": "bound = the nearer of the two roofs for this kernel; neither is close: the kernel is limited by instruction issue (DESIGN.md §4)",
                             # K1b and K3 take the same time to within the run-to-run spread: which of them is `the longest` flips from run to run, so both are named here
                             "co_dominant": {v["kernel"]: {"avg_launch_ms": v["avg_launch_ms"], "frac_hbm": v["frac_hbm"], "frac_fp64": v["frac_fp64"]}
                                             for v in roofs.values() if v["avg_launch_ms"] >= 0.97 * rd["avg_launch_ms"]}})
            # measured HBM bytes per launch of that kernel: the separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command, digested into
            # profiles/hbm_traffic.json by tools/digest_round_profile.sh — a STATIC file of the named profile round
            try:
                with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as fh:
                    ht = json.load(fh)
                roofline["traffic"] = ht["kernels"][rd["kernel"]]["traffic_bytes"]
                traffic_stale = ht.get("kernel_source_hash") != src_hash
                roofline["traffic_source"] = "profiles/hbm_traffic.json, round %s (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, B=1024; not measured in this run)" % ht.get("round", "r01 v18")
            except (OSError, KeyError, ValueError):
                pass
            # the counter files are static: say so when they were collected on other kernel sources than the ones that just ran
            roofline["kernel_source_hash"] = src_hash
            roofline["counters"] = {"flops_pmc": None if flops_stale is None else ("stale" if flops_stale else "current"),
                                    "hbm_traffic": None if traffic_stale is None else ("stale" if traffic_stale else "current")}
            if flops_stale or traffic_stale:
                roofline["stale"] = True
        total_steps = B * world * args.steps
        line = {
            "metric": "MPC+WBC control steps/sec (24-DoF quadruped-manipulator, SQP horizon N=100)", "value": total_steps / elapsed, "unit": "steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "C3/C4: trot gait, horizon N=%d (dt 0.015), %d random initial states per GPU (seed 1235), cold start, 1 SQP iteration + policy eval + 3-level WBC; "
                                   "back-to-back steps, the WBC of a step on its own stream beside the next step's MPC kernels (pipelined THROUGHPUT: the per-kernel times add up to more "
                                   "than ms_per_step; the unpipelined step latency is latency_ms)" % (args.n_intervals, B),
                       "instances_per_gpu": B, "global_batch": B * world, "parallelism": "shard%d" % world, "all_status_ok": bool(per_rank[:, 4].all()), "ls_trials": res["ls_trials"], "engine": getattr(eng, "name", "?"),
                       "max_nodes": getattr(eng, "max_nodes", None), "device_gb_measured": round(getattr(eng, "device_bytes", 0) / 1e9, 3)},
            "per_rank": {"seconds": [float(v) for v in per_rank[:, 0]], "lq_ms": [float(v) for v in per_rank[:, 1]], "riccati_ms": [float(v) for v in per_rank[:, 2]],
                         "wbc_ms": [float(v) for v in per_rank[:, 3]], "intervals_per_launch": [int(v) for v in per_rank[:, 5]]},
        }
        if hip:
            line["roofline"] = roofline
            line["fp64_peak_measured"] = {"mfma_f64_16x16x4": peak_mfma, "vector_fma": peak_fma, "unit": "TFLOP/s", "note": "roofline.peak stays the 78.6 TFLOP/s data-sheet figure"}
            keys = ("avg_launch_ms", "flop_per_launch", "tflops", "frac_fp64", "survey_dense_flop_per_launch", "survey_dense_frac_fp64", "bytes_per_launch", "tbs", "frac_hbm", "valu_non_f64_share")
            line["roofline_all"] = {k: {kk: v[kk] for kk in keys if kk in v} for k, v in roofs.items()}
            line["kernel_ms_per_step"] = {k: v[0] / 5 for k, v in kms_all.items()}
        line.update(sec)
        if hip and not args.no_cpu_baseline and world == 1:      # the CPU baseline is a property of the box: reported on the single-GPU line only
            line["cpu_baseline"] = cpu_baseline(cfg, res["out"], B)
    eng.close()
    if dist is not None:
        dist.destroy_process_group()
    return line


def secondary_figures(args, eng, cfg, dist, device, world, rank):
    import numpy as np
    from qm_control_amd import api, scenarios, sharding
    itf, mpc, wbc = eng.itf, eng.mpc, eng.wbc; B = cfg["B"]; out = {}
    # (1) unpipelined step LATENCY: every step followed by a device synchronisation (no WBC(k) / MPC(k+1) overlap); B = 1024 here, B = 1 below
    eng.sync(); t = time.perf_counter()
    for _ in range(10):
        eng.step(); eng.sync()
    lat_b = (time.perf_counter() - t) / 10 * 1e3
    # (2) SURVEY.md §8(f) rank 1: the same step run as a receding-horizon closed loop on the device — every MPC call warm-started from the previous
    #     primal solution, the observation advanced along the policy, no host data movement
    cl_steps, cl_dt = 10, 0.01
    wbc.reset(); mpc.closed_loop_resident(2, cl_dt, cfg["horizon"], cfg["period"], cfg["time"]); itf.synchronize()
    tcl = time.perf_counter(); mpc.closed_loop_resident(cl_steps, cl_dt, cfg["horizon"], cfg["period"], cfg["time"]); itf.synchronize(); tcl = time.perf_counter() - tcl
    res_cl = mpc.download(); _, qps_cl = wbc.download(B)
    out["closed_loop_warm_start"] = {"value": B * cl_steps / tcl, "unit": "steps/s per GPU", "steps": cl_steps, "mpc_dt": cl_dt, "ms_per_step": tcl / cl_steps * 1e3,
                                     "all_status_ok": bool((res_cl["status"] >= 0).all() and (qps_cl == 0).all()), "ls_trials_last": int(res_cl["ls_trials"])}
    # (3) SURVEY.md §8(f) rank 3: the whole controller around the batched rigid-body plant, device resident — per 1 ms tick
    #     [state estimate -> MPC every 10 ticks (warm) -> policy -> WBC -> updateControlLaw -> plant step with the 9 ms command delay]
    if not args.no_plant_loop and world == 1:
        sim = api.QMHWSim(itf, robust_grid=True); t_shift = 20.0                                  # the reference switches the legs on at time > 10 (QMController.cpp:179)
        mpc.set_problem(cfg["t0"] + t_shift, cfg["x0"], cfg["ref_t"] + t_shift, cfg["ref_x"], cfg["ev"] + t_shift, cfg["modes"]); wbc.reset()
        q0 = np.array(cfg["x0"][:, 6:30]); q0[:, 0:2] = 0.0; q0[:, 2] = 0.385; q0[:, 3:6] = 0.0
        sim.reset(q0, np.zeros((B, 24)), cfg["t0"] + t_shift)
        sim.closed_loop(10, 0.001, cfg["horizon"], n_substeps=2, mpc_every=10); itf.synchronize()
        n_ticks = 30; tp = time.perf_counter(); sim.closed_loop(n_ticks, 0.001, cfg["horizon"], n_substeps=2, mpc_every=10); itf.synchronize(); tp = time.perf_counter() - tp
        sp = sim.state(); res_p = mpc.download(); _, qps_p = wbc.download(B)
        out["closed_loop_plant"] = {"value": B * n_ticks / tp, "unit": "plant + controller ticks/s per GPU (1 ms ticks, MPC every 10th)", "ticks": n_ticks, "ms_per_tick": tp / n_ticks * 1e3,
                                    "realtime_factor_per_instance": n_ticks * 0.001 / tp, "all_status_ok": bool((res_p["status"] >= 0).all() and (qps_p == 0).all() and (sp["status"] == 0).all()),
                                    "all_finite": bool(np.isfinite(sp["q"]).all()), "base_height_range": [float(sp["q"][:, 2].min()), float(sp["q"][:, 2].max())]}
    # (4) BASELINE.json config 5: EE-tracking task, trot -> stance -> trot, N = 150, arm near its joint limits, 512 instances per GPU (4096 over 8), seed 1236
    B5 = min(512, B); steps5 = max(2, args.steps // 5)
    cfg5 = sharding.shard_config(scenarios.make_config("C5", batch=B5 * world), rank, world)
    e5 = HipEngine(cfg5, int(os.environ.get("LOCAL_RANK", "0")), max_nodes=192)
    for _ in range(2):
        e5.step()
    el5, _ = timed_region(e5, steps5, dist, device); r5 = e5.results(); e5.close()
    out["config_C5"] = {"workload": "C5: EE-tracking target, trot -> stance -> trot, N = 150, arm near joint limits, %d instances per GPU (seed 1236)" % B5, "value": B5 * world * steps5 / el5,
                        "unit": "steps/s", "n_gpus": world, "steps": steps5, "ms_per_step": el5 / steps5 * 1e3, "all_status_ok": r5["ok"], "ls_trials": r5["ls_trials"]}
    # (4b) BASELINE.json config 4 as SURVEY.md §8(d) defines it — STRONG scaling: global batch 8192 cut into 8192 / G contiguous shards for G GPUs (the headline `value`
    #      keeps 1024 instances per GPU, i.e. weak scaling; the two coincide at G = 8).  `python bench.py --global-batch 8192` makes this the headline instead.
    if not getattr(args, "global_batch", 0) and 8192 % world == 0:
        Bs = 8192 // world; steps_s = max(2, args.steps // 10)
        cfgs = sharding.shard_config(scenarios.make_config("C4", batch=8192, n_intervals=args.n_intervals), rank, world)
        es = HipEngine(cfgs, int(os.environ.get("LOCAL_RANK", "0")))
        es.step(); es.sync()
        els, _ = timed_region(es, steps_s, dist, device); rs = es.results(); es_nodes = es.max_nodes; es_gb = round(es.device_bytes / 1e9, 3); es.close()
        out["strong_scaling_C4"] = {"workload": "C4: trot, N = %d, global batch 8192 (seed 1235) in contiguous shards of %d instances per GPU" % (args.n_intervals, Bs), "value": 8192 * steps_s / els,
                                    "unit": "steps/s", "n_gpus": world, "global_batch": 8192, "instances_per_gpu": Bs, "max_nodes": es_nodes, "device_gb_measured": es_gb, "steps": steps_s, "ms_per_step": els / steps_s * 1e3, "scaling": "strong",
                                    "all_status_ok": rs["ok"], "instances_with_failed_mpc_status": rs["n_bad_mpc"], "instances_with_mpc_warning": rs["n_warn_mpc"], "instances_with_nonzero_wbc_status": rs["n_bad_wbc"],
                                    "note": "a warning (status 1 = QM_MPC_WARN_PIVOT) marks an instance whose observation time puts a shooting node within 1e-6 s in front of a gait event "
                                            "(instance 2453 of this batch): the negative-duration stage there is solved with zeroed pivots in product and oracle alike "
                                            "(tests/test_grid_fuzz.py, tests/test_gpu_fullsize.py); the solution is valid"}
    # (5) BASELINE.json config 2: a single instance (B = 1): the dependency-chain latency of one control step
    if rank == 0:
        cfg2 = scenarios.make_config("C2"); e2 = HipEngine(cfg2, int(os.environ.get("LOCAL_RANK", "0")), max_nodes=128)
        for _ in range(3):
            e2.step()
        e2.sync(); t = time.perf_counter()
        for _ in range(20):
            e2.step(); e2.sync()
        lat1 = (time.perf_counter() - t) / 20 * 1e3; r2 = e2.results()
        # the two timers the reference prints (ocs2 benchmark::RepeatedTimer: mpcTimer_ around MPC_BASE::run, QMController.cpp:321-323; wbcTimer_ around WbcBase::update,
        # QMController.cpp:145-147), on the GPU for ONE robot: a cold MPC iteration alone (solve + device synchronisation), and one WBC update through the control-tick
        # entry point qmhip_wbc_step on a WBC-only context (host inputs in, torques out: what the ros_control thread sees)
        e2.sync(); t = time.perf_counter()
        for _ in range(20):
            e2.mpc.solve_resident(cfg2["horizon"]); e2.sync()
        mpc1 = (time.perf_counter() - t) / 20 * 1e3
        e2.mpc.solve_resident(cfg2["horizon"], warm=True); e2.sync(); t = time.perf_counter()
        for _ in range(20):
            e2.mpc.solve_resident(cfg2["horizon"], warm=True); e2.sync()
        mpc1w = (time.perf_counter() - t) / 20 * 1e3
        xd2, ud2, md2 = e2.mpc.evaluatePolicy(cfg2["t0"])
        rbd2 = np.zeros((1, 55)); rbd2[0, 0:3] = cfg2["x0"][0, 9:12]; rbd2[0, 3:6] = cfg2["x0"][0, 6:9]; rbd2[0, 6:24] = cfg2["x0"][0, 12:30]
        witf = e2.itf.wbc_context(); w2 = api.HierarchicalWbc(witf); w2.reset()
        for _ in range(3):
            w2.update(xd2, ud2, rbd2, md2, cfg2["period"], np.full(1, cfg2["time"]))
        t = time.perf_counter()
        for _ in range(50):
            w2.update(xd2, ud2, rbd2, md2, cfg2["period"], np.full(1, cfg2["time"]))
        wbc1 = (time.perf_counter() - t) / 50 * 1e3
        witf.close(); e2.close()
        out["single_instance_ms_gpu"] = {"mpc_ms_B1": mpc1, "mpc_ms_B1_warm": mpc1w, "wbc_ms_B1": wbc1, "workload": "BASELINE.json config 2 (trot, N = 100, one robot)",
                                         "note": "mpc: one SQP iteration + device synchronisation (cold / warm-started), no host copies; wbc: qmhip_wbc_step on its own context incl. the host "
                                                 "copies of its 116 input and 54 output doubles and the ctypes call (the C client of tests/c_abi_threads.c sees ~0.3 ms); counterparts of "
                                                 "cpu_baseline.single_instance_ms"}
        out["latency_ms"] = {"B1_C2": lat1, "B%d_unpipelined" % B: lat_b, "note": "one control step with a device synchronisation after every step (no stream overlap between steps); "
                             "B1_C2 = BASELINE.json config 2 (single instance, trot, N = 100): %.0f Hz" % (1e3 / lat1), "C2_status_ok": r2["ok"]}
    # (6) PCIe-INCLUSIVE rate: the boundary hands over host buffers (qmhip_mpc_upload / qmhip_mpc_download / qmhip_wbc_download), the headline keeps its inputs resident.
    #     One whole hand-over per step: upload of (t0, x0, targets, schedule), the step, download of the node arrays + primal solution + WBC output — never `value`
    if rank == 0:
        eng.sync(); reps = 3; t = time.perf_counter()
        for _ in range(reps):
            mpc.set_problem(cfg["t0"], cfg["x0"], cfg["ref_t"], cfg["ref_x"], cfg["ev"], cfg["modes"]); wbc.reset()
            mpc.control_step_resident(cfg["horizon"], cfg["period"], cfg["time"]); r_ = mpc.download(); o_, q_ = wbc.download(B)
        dt_all = (time.perf_counter() - t) / reps
        t = time.perf_counter()
        for _ in range(reps):
            wbc.reset(); mpc.control_step_resident(cfg["horizon"], cfg["period"], cfg["time"]); eng.sync()
        dt_step = (time.perf_counter() - t) / reps
        out["pcie_inclusive"] = {"value": B / dt_all, "unit": "steps/s", "ms_per_step": dt_all * 1e3, "ms_step_only_unpipelined": dt_step * 1e3,
                                 "bytes_in_per_instance": int(8 * (1 + 30 + cfg["ref_t"].shape[1] * 38 + cfg["ev"].shape[1]) + 4 * (cfg["ev"].shape[1] + 1)),
                                 "bytes_out_per_instance": int(eng.itf.max_nodes * (8 + 4 + 4 + 8 * 60) + 8 * 10 + 4 + 8 * 54 + 12),
                                 "note": "host -> device inputs, one unpipelined step, device -> host of every output array (node times / tags / modes, x*, u* on max_nodes nodes, perf, WBC output) "
                                         "through the C ABI's synchronous copies and the host-side node-major -> instance-major gather of qmhip_mpc_download; NOT the headline value"}

    return out


def cpu_baseline(cfg, gpu_out, B):
    """the oracle (C++ restatement of the reference's algorithm, kind "port") timed on this box's host cores, on a bounded sample of the same workload:
    (i) batch mode: threads over instances; (ii) ONE instance the way the reference runs it: 1 thread and sqp.nThreads = 3 worker threads over the
    shooting nodes (task.info:77), MPC iteration and WBC update timed separately (the two timers QMController prints, QMController.cpp:145-147, 321-323)"""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle
    from qm_control_amd import scenarios
    ob = pyoracle.load_blobs()
    cores = min(os.cpu_count() or 1, 64); S = min(B, 512)      # ≈ 15 s of CPU work (≈ 28 ms per instance with the seeded forward mode)
    tb = time.perf_counter()
    bad, _, _, w = pyoracle.batch_step(*ob, cores, cfg["t0"][:S], cfg["horizon"], cfg["x0"][:S], cfg["ref_t"][:S], cfg["ref_x"][:S], cfg["ev"][:S], cfg["modes"][:S], cfg["period"], cfg["time"])
    tcpu = time.perf_counter() - tb
    # per-block relative errors (accelerations / contact forces / torques each on its own scale: one figure over the 54-vector would let 134 N of contact
    # force hide a joint acceleration)
    blocks = {"vdot": (slice(0, 24), 1e-2), "contact_forces": (slice(24, 36), 1.0), "torques": (slice(36, 54), 1e-1)}
    errs = {k: float(np.abs(gpu_out[:S][:, sl] - w[:, sl]).max() / max(float(np.abs(w[:, sl]).max()), fl)) for k, (sl, fl) in blocks.items()}
    # single instance = BASELINE.json config 2 (trot, N = 100)
    o = pyoracle.Oracle(*ob); c2 = scenarios.make_config("C2")
    o.set_schedule(c2["ev"][0], c2["modes"][0]); o.set_target(c2["ref_t"][0], c2["ref_x"][0])
    single = {}
    for nt in (1, 3):
        o.set_threads(nt); best = 1e9; ph = None
        for _ in range(3):
            t = time.perf_counter(); o.mpc_step(c2["t0"][0], c2["t0"][0] + c2["horizon"], c2["x0"][0]); dt_ = (time.perf_counter() - t) * 1e3
            if dt_ < best:
                best = dt_; ph = o.phase_ms()
        single["mpc_ms_%dthread%s" % (nt, "" if nt == 1 else "s")] = best
        single["mpc_phases_ms_%dthread%s" % (nt, "" if nt == 1 else "s")] = {"lq_approximation": float(ph[0]), "riccati": float(ph[1]), "line_search": float(ph[2])}
    o.set_threads(1)
    xd, ud, mode = o.eval_policy(c2["t0"][0]); rbd = o.rbd_from_q(c2["x0"][0][6:30]); o.wbc_reset(); best = 1e9
    for _ in range(5):
        t = time.perf_counter(); o.wbc(xd, ud, rbd, mode, 0.002, 20.0); best = min(best, (time.perf_counter() - t) * 1e3)
    single["wbc_ms_1thread"] = best
    return {"value": S / tcpu, "unit": "steps/s", "cores": cores, "kind": "port",
            "sample": "first %d instances of the same batch, %d threads over instances (thread count = min(os.cpu_count(), 64)); max rel diff GPU vs oracle on the sample per block: %s" % (S, cores, ", ".join("%s %.1e" % kv for kv in errs.items())),
            "parity_on_sample": errs,
            "single_instance_ms": single,
            "ad": "seeded forward mode (flow map 33 slots, feet 36, end-effector error 12; bit-identical to the 60-slot evaluation of rounds 1-4, tests/test_oracle.py)",
            "note": "the oracle is a RESTATEMENT (forward-mode AD Jacobians, an O(n^4) Lagrangian mass matrix in the WBC); since round 5 its single-instance MPC iteration on 3 threads is "
                    "in the range SURVEY.md a11 expects of the OCS2 / Pinocchio / HPIPM binary the reference runs (~5-10 ms on 3 cores; rounds 1-4: 32 ms).  A reported baseline, never a speed-up claim"}


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    line = run(args)
    if line is not None:
        print(json.dumps(line))


if __name__ == "__main__":
    main()
