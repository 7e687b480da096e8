#!/bin/bash
# round 6 evidence on the final tree: GPU suite, bench line + rocprofv3 kernel stats + PMC passes (tools/gpu_round_profile.sh), same-box A/B of the line search's two drivers
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/r06_pytest_gpu_final.log 2>&1; tail -3 gpurun_out/r06_pytest_gpu_final.log
bash tools/gpu_round_profile.sh noprof > gpurun_out/r06_round_profile.log 2>&1; tail -3 gpurun_out/r06_round_profile.log
python tools/warm_step_ab.py 2 > gpurun_out/r06_ab_ls_device_tail_final.log 2>&1; tail -4 gpurun_out/r06_ab_ls_device_tail_final.log
