"""tools/readme_experiment.py — ablation of the README's end-effector stability experiment (/root/reference/README.md:109-116, docs/position_err.png: the base backs away
0.31 m in the −x direction within 10 s under the gamepad's cmd_vel while the end-effector is commanded to hold its pose; EE deviation <= 3.5 mm / 2.6 deg) on the CPU oracle's
closed loop (tools/readme_experiment_cpu.py; the device loop reproduces it to the digit: profiles/r04_sim_closed_loop_demo.txt = cell r04_demo/default).  One process per cell.
usage: python tools/readme_experiment.py [out.json]"""
import json, os, subprocess, sys, concurrent.futures as cf
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CELLS = [("r04_demo", "pose target +0.30 m in 3 s (the round-4 demo)", dict(ticks=6000)),
         ("r04_demo", "same, perfect-tracking plant", dict(ticks=6000, plant="perfect"))]
for vx in (-0.0345, -0.1, -0.2, -0.3):
    for kd in (0.5, 0.05, 0.0):
        CELLS.append(("readme_drive", "cmd_vel %.4f m/s in -x for 10 s, kd_arm_wbc %.2f" % (vx, kd), dict(ticks=11000, drive="cmdvel", vx=vx, walk_s=10.0, arm_kd=kd)))
BASE = dict(ticks=11000, drive="cmdvel", vx=-0.1, walk_s=10.0)
for name, kw in (("contact stiffness x 0.25", dict(stiffness=1.0e4)), ("contact stiffness x 4", dict(stiffness=1.6e5, nsub=4)), ("contact damping x 4", dict(damping=800.0)), ("8 plant sub-steps", dict(nsub=8)),
                 ("no command delay", dict(delay=0.0)), ("MPC every 20 ticks", dict(mpc_every=20)), ("MPC pipelined (one period of latency)", dict(pipelined=1)), ("horizon 1.5 s", dict(horizon=1.5)),
                 ("arm kp 20 (kd 0.5)", dict(arm_kp=20.0)), ("perfect-tracking plant", dict(plant="perfect"))):
    CELLS.append(("ablation_at_-0.1", name, dict(BASE, **kw)))


def run_cell(cell):
    group, name, kw = cell
    cmd = [sys.executable, os.path.join(ROOT, "tools", "readme_experiment_cpu.py"), "threads=1", "log_every=100000"] + ["%s=%s" % kv for kv in kw.items()]
    out = subprocess.run(cmd, capture_output=True, text=True).stdout
    line = [l for l in out.split("\n") if l.startswith('{"config')][0]; d = json.loads(line)
    return dict(group=group, cell=name, config=d["config"], base_travel_m=round(d["base_travel_m"], 4), ee_dev_max_mm=round(d["ee_dev_max_mm"], 2), ee_dev_max_deg=round(d["ee_dev_max_deg"], 2),
                planned_ee_dev_max_mm=round(d["plan_ee_dev_max_mm"], 2), ee_vs_plan_max_mm=round(d["ee_vs_plan_max_mm"], 2))


if __name__ == "__main__":
    with cf.ThreadPoolExecutor(max_workers=max(1, (os.cpu_count() or 2) - 1)) as ex:
        rows = list(ex.map(run_cell, CELLS))
    res = dict(readme=dict(base_travel_m=-0.31, seconds=10.0, ee_dev_max_mm=3.5, ee_dev_max_deg=2.6, source="/root/reference/README.md:116, docs/position_err.png (right axis: distance the base moves in the -x direction)"),
               loop="CPU oracle: QMController::update around oracle/src/sim.h (tests/test_sim.py), 1 kHz ticks, MPC every 10 ticks synchronous unless stated, trot after 0.5 s of stance", cells=rows)
    for r in rows: print("%-18s %-58s travel %+.3f m  EE %5.1f mm %5.2f deg  planned %5.1f mm  EE-vs-plan %.2f mm" % (r["group"], r["cell"], r["base_travel_m"], r["ee_dev_max_mm"], r["ee_dev_max_deg"], r["planned_ee_dev_max_mm"], r["ee_vs_plan_max_mm"]))
    if len(sys.argv) > 1: json.dump(res, open(sys.argv[1], "w"), indent=1)
