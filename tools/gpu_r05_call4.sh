#!/bin/bash
# round 5, GPU call 4: full GPU suite on the tree with the interior-point solver and the README experiment test; bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r05_pytest_gpu_4.log 2>&1; tail -6 gpurun_out/r05_pytest_gpu_4.log
python bench.py --steps 20 --warmup 3 > gpurun_out/r05_bench_4.json 2> gpurun_out/r05_bench_4.err; tail -c 2500 gpurun_out/r05_bench_4.json; tail -3 gpurun_out/r05_bench_4.err
python tools/readme_experiment_gpu.py -0.1 0.0 4 10 > gpurun_out/r05_readme_gpu.json 2>&1; tail -c 600 gpurun_out/r05_readme_gpu.json
