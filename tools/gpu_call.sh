#!/bin/bash
# tools/gpu_call.sh <tag> <command ...> — one gpurun recipe instead of a script per call: runs the command from the repo root on the GPU box with TMPDIR set,
# stdout + stderr into gpurun_out/<tag>.log (its tail is what gpurun prints back).  Round 6 replaces the near-identical tools/gpu_r05_call*.sh with this.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=$1; shift
bash -c "$*" > gpurun_out/$TAG.log 2>&1
echo "exit $?" >> gpurun_out/$TAG.log
tail -n ${GPU_CALL_TAIL:-25} gpurun_out/$TAG.log
