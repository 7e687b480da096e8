#!/bin/bash
# tools/ab_kernel_ms.sh libA.so libB.so [rounds] — per-kernel times of two builds of libqmhip.so, alternating on ONE GPU box (boxes differ by a few per cent, which is
# more than most kernel changes are worth: never compare numbers of two gpurun calls).  Paths relative to the repo root, e.g. tools/_build/a.so.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
A=$1; Bb=$2; N=${3:-3}
for i in $(seq 1 $N); do
  for L in $A $Bb; do echo -n "$L  "; QM_AB_LIB=$L QM_MPC_ONLY=1 python tools/quick_kernel_ms.py 2>&1 | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernel_ms'], d['ok'])"; done
done
