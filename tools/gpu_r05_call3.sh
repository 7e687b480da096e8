#!/bin/bash
# round 5, GPU call 3: K4 with cooperative loads + structured R (e.so) against d.so (K1a coalesced only) and c.so; SQ / TA counters of the thread-per-node kernels on e.so
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -k "mpc or lq_records or fullsize or golden or ilqr" > gpurun_out/r05_pytest_gpu_3.log 2>&1; tail -3 gpurun_out/r05_pytest_gpu_3.log
AB_ROUNDS=2 bash tools/ab_env.sh "tools/_build/c.so" "tools/_build/d.so" "tools/_build/e.so" > gpurun_out/r05_ab_ls_coalesced_mpc.log 2>&1; cat gpurun_out/r05_ab_ls_coalesced_mpc.log
AB_ROUNDS=2 QM_MPC_ONLY= bash tools/ab_env.sh "tools/_build/c.so" "tools/_build/e.so" > gpurun_out/r05_ab_ls_coalesced_full.log 2>&1; cat gpurun_out/r05_ab_ls_coalesced_full.log
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_WAIT_ANY"
P2="SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU"
P3="TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_WAVEFRONTS_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
n=1
for P in "$P1" "$P2" "$P3"; do
  rm -rf gpurun_out/pmc_tpn_$n
  QM_AB_LIB=tools/_build/e.so QM_MPC_ONLY=1 timeout 300 rocprofv3 --pmc $P --kernel-trace -d gpurun_out/pmc_tpn_$n -- python tools/quick_kernel_ms.py > gpurun_out/pmc_tpn_$n.log 2>&1
  python tools/rocpd_pmc_summary.py "$(ls -t gpurun_out/pmc_tpn_$n/*/*_results.db | head -1)" gpurun_out/r05_pmc_tpn_$n.csv > /dev/null 2>&1
  grep -i "lq_kin\|ls_eval" gpurun_out/r05_pmc_tpn_$n.csv | sort | cut -c1-150
  n=$((n+1))
done
