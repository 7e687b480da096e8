#!/bin/bash
# round 5: compact stage record (i.so) against the previous tree (prev.so): per-kernel times, full step; GPU suite on the new tree
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
AB_ROUNDS=3 bash tools/ab_env.sh "tools/_build/prev.so" "tools/_build/i.so" > gpurun_out/r05_ab_record_compact.log 2>&1; cat gpurun_out/r05_ab_record_compact.log
AB_ROUNDS=2 QM_MPC_ONLY= bash tools/ab_env.sh "tools/_build/prev.so" "tools/_build/i.so" >> gpurun_out/r05_ab_record_compact.log 2>&1; tail -4 gpurun_out/r05_ab_record_compact.log
python -m pytest tests -m gpu -q > gpurun_out/r05_pytest_gpu_13.log 2>&1; tail -4 gpurun_out/r05_pytest_gpu_13.log
