#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
AB_ROUNDS=2 bash tools/ab_env.sh "tools/_build/g.so" "tools/_build/h.so" "tools/_build/h2.so" > gpurun_out/r05_ab_lb1.log 2>&1; cat gpurun_out/r05_ab_lb1.log
AB_ROUNDS=1 QM_MPC_ONLY= bash tools/ab_env.sh "tools/_build/g.so" "tools/_build/h.so" "tools/_build/h2.so" > gpurun_out/r05_ab_lb1_full.log 2>&1; cat gpurun_out/r05_ab_lb1_full.log
for V in g h; do rm -rf gpurun_out/prof_q_$V; QM_AB_LIB=tools/_build/$V.so QM_MPC_ONLY=1 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_q_$V -- python tools/quick_kernel_ms.py > gpurun_out/prof_q_$V.log 2>&1; python tools/rocpd_kernel_stats.py "$(ls -t gpurun_out/prof_q_$V/*/*_results.db | head -1)" gpurun_out/r05_kernel_stats_quick_$V.csv | grep -i "kin\|ls_eval\|lq_kernel\|riccati" | cut -c1-120; tail -1 gpurun_out/prof_q_$V.log | cut -c1-300; done
