#!/bin/bash
# tools/digest_round_profile.sh <tag> — after `gpurun -- bash tools/gpu_round_profile.sh`: turn the NEWEST rocprofv3 databases under gpurun_out/ into the
# summaries kept in profiles/ (kernel stats, FETCH_SIZE / WRITE_SIZE per kernel, HBM traffic digest, bench line).  Usage: bash tools/digest_round_profile.sh v14
set -e
cd "$(dirname "$0")/.."
tag=${1:?tag like v14}
newest() { ls -t "$1"/runc/*_results.db | head -1; }
python tools/rocpd_kernel_stats.py "$(newest gpurun_out/prof_stats)" profiles/r01_rocprofv3_kernel_stats_${tag}.csv | head -12
python tools/rocpd_pmc_summary.py "$(newest gpurun_out/pmc_fetch)" profiles/r01_pmc_fetch_size_${tag}.csv > /dev/null
python tools/rocpd_pmc_summary.py "$(newest gpurun_out/pmc_write)" profiles/r01_pmc_write_size_${tag}.csv > /dev/null
python tools/hbm_traffic_digest.py profiles/r01_pmc_fetch_size_${tag}.csv profiles/r01_pmc_write_size_${tag}.csv profiles/hbm_traffic.json | head -6
cp gpurun_out/bench.json profiles/r01_bench_${tag}.json
