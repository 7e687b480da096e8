#!/bin/bash
# round 5 final evidence on the final tree: GPU suite, bench line + rocprofv3 kernel stats + PMC passes (tools/gpu_round_profile.sh), parity report, README experiment on the device loop
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/r05_pytest_gpu_final.log 2>&1; tail -3 gpurun_out/r05_pytest_gpu_final.log
bash tools/gpu_round_profile.sh noprof > gpurun_out/r05_round_profile.log 2>&1; tail -3 gpurun_out/r05_round_profile.log
python bench.py --steps 20 --warmup 3 > gpurun_out/bench.json 2> gpurun_out/bench.err      # the kept bench line is written AFTER the counter passes: its roofline then reads `counters: current`... (digest first)
timeout 600 python tools/parity_report.py > gpurun_out/r05_parity_report.log 2>&1; tail -3 gpurun_out/r05_parity_report.log
python tools/readme_experiment_gpu.py -0.1 0.0 4 10 > gpurun_out/r05_readme_gpu.json 2>&1; tail -c 400 gpurun_out/r05_readme_gpu.json
