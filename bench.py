"""bench.py — MPC+WBC control steps/sec of the 24-DoF quadruped-manipulator at horizon N=100 (BASELINE.json metric).

One "step" = one multiple-shooting SQP iteration over the horizon + policy evaluation at t0 + one 3-level
hierarchical WBC solve, for one instance.  Workload per GPU: configuration C3/C4 of
BASELINE.md — trot gait, N=100, 1024 independent instances with random initial states (seeded), inputs resident
in HBM before the timed region.  Instances are independent, so N GPUs run N shards with no data-path
collective (weak scaling); torch.distributed (RCCL) only carries the barrier and the max-over-ranks time.

Usage:  python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--no-cpu-baseline]
        N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic FP64 work and HBM bytes per unit of each kernel (DESIGN.md §5 derives every figure)
#   unit = one non-event shooting interval (lq, riccati) or one instance (wbc)
KERNEL_MODEL = {
    # K1b: LQ approximation + projection. bytes: stage record written (Ap Bp, the upper tiles of Qp, Pp Rp, the 12 non-zero rows of Px, vectors, swing blocks = 3533 doubles; Pu is not stored) + kin record / inputs read (~620)
    "lq": {"flop": 190e3 + 410e3, "bytes": (3533 + 620) * 8.0, "unit": "interval"},
    # K3: backward sweep reads [Ap|bp] Bp [Qp|qp] [Pp|rp] Rp (3282) and writes L W y (882); forward rollout reads the 12 momentum / base-pose rows of Ap Bp, W L,
    #     the 12 non-zero rows of Px, vectors and the swing blocks (1952; joint rows are x_j + dt u_j, Pu is rebuilt from the contact mode), x/dx/du (120)
    "riccati": {"flop": 250e3, "bytes": (3282 + 882 + 1952 + 120) * 8.0, "unit": "interval"},
    # K5-K7: rigid-body pass + 3-level cascade; bytes: inputs/outputs + tip/Jacobian scratch (~0.9k doubles)
    "wbc": {"flop": 2.0e6, "bytes": 900 * 8.0, "unit": "instance"},
}
FP64_PEAK_TFLOPS = 78.6                 # MI355X dense FP64 matrix peak = FP64 vector peak (AMD public figure; v_mfma_f64_16x16x4 micro-benchmark: 77.7, profiles/)
HBM_PEAK_TBS = 8.0                      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (≈6.3 TB/s achievable)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=1024, help="instances per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-plant-loop", action="store_true", help="skip the secondary closed-loop-around-the-plant figure")
    args = ap.parse_args()

    import numpy as np
    from qm_control_amd import api, scenarios, sharding

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1 or os.environ.get("QM_BENCH_FORCE_DIST"):      # (the env switch exercises the RCCL path on a single GPU)
        import torch
        import torch.distributed as dist_mod
        torch.cuda.set_device(local)
        dist_mod.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        dist = dist_mod

    def barrier():
        if dist is not None:
            import torch
            dist.barrier(); torch.cuda.synchronize()

    B = args.batch
    blobs = scenarios.load_blobs()
    # C4: seed 1235, contiguous shard of the global batch for this rank
    cfg = sharding.shard_config(scenarios.make_config("C4", batch=B * world), rank, world)
    itf = api.QMInterface(blobs=blobs, device=local, max_batch=B, max_nodes=128, max_ref_knots=2, max_events=cfg["ev"].shape[1])
    mpc = api.SqpMpc(itf); wbc = api.HierarchicalWbc(itf)
    mpc.set_problem(cfg["t0"], cfg["x0"], cfg["ref_t"], cfg["ref_x"], cfg["ev"], cfg["modes"])     # inputs resident in HBM from here on

    def step():
        wbc.reset()
        mpc.control_step_resident(cfg["horizon"], cfg["period"], cfg["time"])

    # roofline denominators measured on this very device before anything is timed ("microbenchmark it first"): FP64 matrix and
    # vector FMA throughput, ≈ 0.6 s of sustained FP64 work (which also brings a freshly booted device to its sustained clocks)
    peak_mfma = max(itf.microbench_fp64(True) for _ in range(12)); peak_fma = max(itf.microbench_fp64(False) for _ in range(6))
    for _ in range(args.warmup):
        step()
    itf.synchronize()
    # per-kernel times of every kernel from a short untimed pass; inside the timed region only the three modelled kernels (lq, riccati, wbc — the roofline's
    # avg_launch_ms) carry HIP-event spans: two event records cost about one launch, 22 of them per step would cost 2 % of the headline
    itf.set_profiling(True); itf.reset_kernel_ms()
    for _ in range(5):
        step()
    itf.synchronize(); itf.set_profiling(False)
    kms_all = {k: itf.kernel_ms(k) for k in ("grid", "lq_kin", "lq", "riccati", "ls_eval", "ls_misc", "policy", "wbc")}
    itf.set_profiling(2); itf.reset_kernel_ms()
    barrier(); itf.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    itf.synchronize(); barrier()
    elapsed = time.perf_counter() - t0
    elapsed = sharding.max_over_ranks(elapsed, dist, "cuda")

    # per-kernel HIP-event times over the timed region (events recorded on the stream the kernels run on)
    itf.set_profiling(False)
    kms = {k: itf.kernel_ms(k) for k in ("lq", "riccati", "wbc")}
    res = mpc.download(); out, qps = wbc.download(B)
    # secondary figure (NOT the headline value): the same step run as a receding-horizon closed loop on the device —
    # every MPC call warm-started from the previous primal solution, the observation advanced along the policy, no host data movement
    cl_steps, cl_dt = 10, 0.01
    wbc.reset(); mpc.closed_loop_resident(2, cl_dt, cfg["horizon"], cfg["period"], cfg["time"]); itf.synchronize()
    tcl = time.perf_counter(); mpc.closed_loop_resident(cl_steps, cl_dt, cfg["horizon"], cfg["period"], cfg["time"]); itf.synchronize(); tcl = time.perf_counter() - tcl
    res_cl = mpc.download(); _, qps_cl = wbc.download(B)
    closed_loop = {"value": B * cl_steps / tcl, "unit": "steps/s per GPU", "steps": cl_steps, "mpc_dt": cl_dt, "ms_per_step": tcl / cl_steps * 1e3,
                   "all_status_ok": bool((res_cl["status"] == 0).all() and (qps_cl == 0).all()), "ls_trials_last": int(res_cl["ls_trials"])}
    # secondary figure (NOT the headline value): the whole controller around the batched rigid-body plant, device resident —
    # per 1 ms tick [state estimate -> MPC every 10 ticks (warm) -> policy -> WBC -> updateControlLaw -> plant step with the 9 ms command delay]
    plant = None
    if not args.no_plant_loop:
        sim = api.QMHWSim(itf); t_shift = 20.0                                  # the reference switches the legs on at time > 10 (QMController.cpp:179)
        mpc.set_problem(cfg["t0"] + t_shift, cfg["x0"], cfg["ref_t"] + t_shift, cfg["ref_x"], cfg["ev"] + t_shift, cfg["modes"]); wbc.reset()
        q0 = np.array(cfg["x0"][:, 6:30]); q0[:, 0:2] = 0.0; q0[:, 2] = 0.385; q0[:, 3:6] = 0.0
        sim.reset(q0, np.zeros((B, 24)), cfg["t0"] + t_shift)
        sim.closed_loop(10, 0.001, cfg["horizon"], n_substeps=2, mpc_every=10); itf.synchronize()
        n_ticks = 30; tp = time.perf_counter(); sim.closed_loop(n_ticks, 0.001, cfg["horizon"], n_substeps=2, mpc_every=10); itf.synchronize(); tp = time.perf_counter() - tp
        sp = sim.state(); res_p = mpc.download(); _, qps_p = wbc.download(B)
        plant = {"value": B * n_ticks / tp, "unit": "plant + controller ticks/s per GPU (1 ms ticks, MPC every 10th)", "ticks": n_ticks, "ms_per_tick": tp / n_ticks * 1e3,
                 "realtime_factor_per_instance": n_ticks * 0.001 / tp, "all_status_ok": bool((res_p["status"] == 0).all() and (qps_p == 0).all() and (sp["status"] == 0).all()),
                 "all_finite": bool(np.isfinite(sp["q"]).all()), "base_height_range": [float(sp["q"][:, 2].min()), float(sp["q"][:, 2].max())]}
    ok = bool((res["status"] == 0).all() and (qps == 0).all())
    n_intervals = int(sum(int(res["num_nodes"][b]) - 1 - int((res["event"][b, :res["num_nodes"][b]] == 1).sum()) for b in range(B)))
    # roofline of the dominant kernel (largest average launch duration among the modelled kernels), both ceilings priced
    def roof(name):
        ms = kms[name][0] / max(1, kms[name][1]); mdl = KERNEL_MODEL[name]; units = n_intervals if mdl["unit"] == "interval" else B
        tf = mdl["flop"] * units / (ms * 1e-3) / 1e12 if ms > 0 else 0.0; tb = mdl["bytes"] * units / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        return {"kernel": "qm_%s_kernel" % name, "avg_launch_ms": ms, "units_per_launch": units, "unit": mdl["unit"], "flop_per_launch": mdl["flop"] * units, "bytes_per_launch": mdl["bytes"] * units,
                "tflops": tf, "frac_fp64": tf / FP64_PEAK_TFLOPS, "tbs": tb, "frac_hbm": tb / HBM_PEAK_TBS}
    roofs = {k: roof(k) for k in KERNEL_MODEL}
    dom = max(roofs, key=lambda k: roofs[k]["avg_launch_ms"]); rd = roofs[dom]
    if rd["frac_hbm"] >= rd["frac_fp64"]:
        roofline = {"bound": "hbm", "kernel": rd["kernel"], "achieved": rd["tbs"] * 1e3, "peak": HBM_PEAK_TBS * 1e3, "unit": "GB/s", "frac": rd["frac_hbm"], "traffic": None}
    else:
        roofline = {"bound": "mfma", "kernel": rd["kernel"], "achieved": rd["tflops"], "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": rd["frac_fp64"], "traffic": None}
    roofline.update({"avg_launch_ms": rd["avg_launch_ms"], "flop_per_launch": rd["flop_per_launch"], "bytes_per_launch": rd["bytes_per_launch"]})
    # measured HBM bytes per launch of that kernel: the separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command,
    # summarised in profiles/hbm_traffic.json (tools/gpu_round_profile.sh; PMC collection cannot run inside the timed bench itself)
    try:
        with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as fh:
            roofline["traffic"] = json.load(fh)["kernels"][rd["kernel"]]["traffic_bytes"]
            roofline["traffic_source"] = "profiles/hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, B=1024)"
    except (OSError, KeyError, ValueError):
        pass

    if rank == 0:
        total_steps = B * world * args.steps
        line = {
            "metric": "MPC+WBC control steps/sec (24-DoF quadruped-manipulator, SQP horizon N=100)", "value": total_steps / elapsed, "unit": "steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "C3/C4: trot gait, horizon N=100 (dt 0.015), %d random initial states per GPU (seed 1235), cold start, 1 SQP iteration + policy eval + 3-level WBC; back-to-back steps, the WBC of a step on its own stream beside the next step's MPC kernels" % B,
                       "instances_per_gpu": B, "parallelism": "shard%d" % world, "all_status_ok": ok, "ls_trials": int(res["ls_trials"])},
            "roofline": roofline,
            "fp64_peak_measured": {"mfma_f64_16x16x4": peak_mfma, "vector_fma": peak_fma, "unit": "TFLOP/s", "note": "roofline.peak stays the 78.6 TFLOP/s data-sheet figure"},
            "roofline_all": {k: {kk: v[kk] for kk in ("avg_launch_ms", "tflops", "frac_fp64", "tbs", "frac_hbm")} for k, v in roofs.items()},
            "kernel_ms_per_step": {k: v[0] / 5 for k, v in kms_all.items()},
            "closed_loop_warm_start": closed_loop,
            "closed_loop_plant": plant,
        }
        if not args.no_cpu_baseline and world == 1:      # the CPU baseline is a property of the box: reported on the single-GPU line only
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import pyoracle
            cores = min(os.cpu_count() or 1, 64); S = min(B, 128)
            tb = time.perf_counter()
            bad, _, _, w = pyoracle.batch_step(*pyoracle.load_blobs(), cores, cfg["t0"][:S], cfg["horizon"], cfg["x0"][:S], cfg["ref_t"][:S], cfg["ref_x"][:S], cfg["ev"][:S], cfg["modes"][:S], cfg["period"], cfg["time"])
            tcpu = time.perf_counter() - tb
            err = float(np.abs(out[:S] - w).max() / np.abs(w).max())
            line["cpu_baseline"] = {"value": S / tcpu, "unit": "steps/s", "cores": cores, "kind": "port",
                                    "sample": "first %d instances of the same batch, CPU oracle (C++ restatement, AD Jacobians), %d threads over instances; max rel diff of GPU torques on the sample %.1e" % (S, cores, err)}
        print(json.dumps(line))
    itf.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
