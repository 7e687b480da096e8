#!/bin/bash
# tools/digest_round_profile.sh <round> <tag> — after `gpurun -- bash tools/gpu_round_profile.sh`: turn the NEWEST rocprofv3 databases under gpurun_out/ into the
# summaries kept in profiles/ (kernel stats, FETCH_SIZE / WRITE_SIZE per kernel, HBM traffic digest, issued-flops digest, SQ activity, bench line).
# Usage: bash tools/digest_round_profile.sh r02 v19
set -e
cd "$(dirname "$0")/.."
rnd=${1:?round like r02}; tag=${2:?tag like v19}
newest() { ls -t "$1"/runc/*_results.db | head -1; }
python tools/rocpd_kernel_stats.py "$(newest gpurun_out/prof_stats)" profiles/${rnd}_rocprofv3_kernel_stats_${tag}.csv | head -12
python tools/rocpd_pmc_summary.py "$(newest gpurun_out/pmc_fetch)" profiles/${rnd}_pmc_fetch_size_${tag}.csv > /dev/null
python tools/rocpd_pmc_summary.py "$(newest gpurun_out/pmc_write)" profiles/${rnd}_pmc_write_size_${tag}.csv > /dev/null
python tools/hbm_traffic_digest.py profiles/${rnd}_pmc_fetch_size_${tag}.csv profiles/${rnd}_pmc_write_size_${tag}.csv "${rnd} ${tag}" profiles/hbm_traffic.json | head -6
if [ -d gpurun_out/pmc_flops_a ]; then
  python tools/rocpd_pmc_summary.py "$(newest gpurun_out/pmc_flops_a)" profiles/${rnd}_pmc_flops_a_${tag}.csv > /dev/null
  python tools/rocpd_pmc_summary.py "$(newest gpurun_out/pmc_flops_b)" profiles/${rnd}_pmc_flops_b_${tag}.csv > /dev/null
  python tools/flops_pmc_digest.py profiles/${rnd}_pmc_flops_a_${tag}.csv profiles/${rnd}_pmc_flops_b_${tag}.csv "${rnd} ${tag}" 32 profiles/flops_pmc.json
fi
if [ -d gpurun_out/pmc_sq ]; then python tools/rocpd_pmc_summary.py "$(newest gpurun_out/pmc_sq)" profiles/${rnd}_pmc_sq_activity_${tag}.csv > /dev/null; fi
if [ -d gpurun_out/pmc_lds ]; then python tools/rocpd_pmc_summary.py "$(newest gpurun_out/pmc_lds)" profiles/${rnd}_pmc_lds_mix_${tag}.csv > /dev/null; fi
cp gpurun_out/bench.json profiles/${rnd}_bench_${tag}.json
