#!/bin/bash
# round 5, GPU call 1: GPU tests on the tree without the inlined sin/cos library fallback, same-box A/B (a.so = round-4 kernels, c.so = this tree), instruction-cache counters of both
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r05_pytest_gpu_1.log 2>&1; tail -4 gpurun_out/r05_pytest_gpu_1.log
AB_ROUNDS=2 bash tools/ab_env.sh "tools/_build/a.so" "tools/_build/c.so" > gpurun_out/r05_ab_sincos_mpc.log 2>&1; cat gpurun_out/r05_ab_sincos_mpc.log
AB_ROUNDS=2 QM_MPC_ONLY= bash tools/ab_env.sh "tools/_build/a.so" "tools/_build/c.so" > gpurun_out/r05_ab_sincos_full.log 2>&1; cat gpurun_out/r05_ab_sincos_full.log
for V in a c; do
  rm -rf gpurun_out/pmc_ic_$V
  QM_AB_LIB=tools/_build/$V.so timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-trace -d gpurun_out/pmc_ic_$V -- python tools/quick_kernel_ms.py > gpurun_out/pmc_ic_$V.log 2>&1
  tail -2 gpurun_out/pmc_ic_$V.log
  python tools/rocpd_pmc_summary.py "$(ls -t gpurun_out/pmc_ic_$V/*/*_results.db | head -1)" gpurun_out/r05_pmc_icache_$V.csv > /dev/null 2>&1
  grep -i "lq_kin\|ls_eval\|wbc_kernel\|riccati_kernel\|qm_lq_kernel" gpurun_out/r05_pmc_icache_$V.csv | sort | cut -c1-160
done
