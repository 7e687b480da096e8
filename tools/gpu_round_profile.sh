#!/bin/bash
# tools/gpu_round_profile.sh — run on the GPU box through gpurun: tests, bench line, rocprofv3 kernel stats and the two HBM PMC passes.
# Everything lands under gpurun_out/; the summaries worth keeping are copied into profiles/ afterwards (tools/rocpd_kernel_stats.py).
set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; tail -2 gpurun_out/pytest_gpu.log
python bench.py --steps 20 --warmup 3 > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -c 600 gpurun_out/bench.json
rm -rf gpurun_out/prof_stats gpurun_out/pmc_fetch gpurun_out/pmc_write
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_fetch -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_write -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_write.log 2>&1
ls -R gpurun_out | head -40
