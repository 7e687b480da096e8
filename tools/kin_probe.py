import sys; sys.path.insert(0, "/root/repo")
from qm_control_amd import api, scenarios
B = 1024
cfg = scenarios.make_config("C4", batch=B)
itf = api.QMInterface(blobs=scenarios.load_blobs(), max_batch=B, max_nodes=128, max_ref_knots=2, max_events=cfg["ev"].shape[1])
mpc = api.SqpMpc(itf); mpc.set_problem(cfg["t0"], cfg["x0"], cfg["ref_t"], cfg["ref_x"], cfg["ev"], cfg["modes"])
mpc.solve_resident(cfg["horizon"]); itf.synchronize()
for prof in (0, 2):
    itf.debug_set("lq_prof", prof)
    itf.set_profiling(True); itf.reset_kernel_ms()
    for _ in range(5): mpc.solve_resident(cfg["horizon"])
    print(prof, {k: round(itf.kernel_ms(k)[0] / max(1, itf.kernel_ms(k)[1]), 3) for k in ("lq_kin", "lq", "riccati", "ls_eval")})
