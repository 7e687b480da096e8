#!/bin/bash
# round 5, GPU call 2: K1a with cooperative (coalesced) loads / record stores (d.so) against the tree before it (c.so)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -k "mpc or lq_records or fullsize or golden" > gpurun_out/r05_pytest_gpu_2.log 2>&1; tail -3 gpurun_out/r05_pytest_gpu_2.log
AB_ROUNDS=2 bash tools/ab_env.sh "tools/_build/c.so" "tools/_build/d.so" > gpurun_out/r05_ab_kin_coalesced_mpc.log 2>&1; cat gpurun_out/r05_ab_kin_coalesced_mpc.log
AB_ROUNDS=2 QM_MPC_ONLY= bash tools/ab_env.sh "tools/_build/c.so" "tools/_build/d.so" > gpurun_out/r05_ab_kin_coalesced_full.log 2>&1; cat gpurun_out/r05_ab_kin_coalesced_full.log
