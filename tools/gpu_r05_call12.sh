#!/bin/bash
# round 5: robustness evidence on the final kernels (status sweep over all scenario families / gaits, 1024 perturbed robots on the plant, occupancy / batch sweep)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python tools/status_sweep.py > gpurun_out/r05_status_sweep.txt 2>&1; tail -6 gpurun_out/r05_status_sweep.txt
timeout 300 python tools/sim_robustness.py 3000 1024 > gpurun_out/r05_sim_robustness.txt 2>&1; tail -4 gpurun_out/r05_sim_robustness.txt
timeout 600 python tools/occupancy_sweep.py > gpurun_out/r05_occupancy_sweep.json 2> gpurun_out/r05_occupancy_sweep.err; tail -c 1500 gpurun_out/r05_occupancy_sweep.json; tail -3 gpurun_out/r05_occupancy_sweep.err
cp gpurun_out/threads_report.txt gpurun_out/r05_threads_report.txt 2>/dev/null; cp gpurun_out/fixed_rate_report.txt gpurun_out/r05_fixed_rate_report.txt 2>/dev/null
