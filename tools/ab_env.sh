#!/bin/bash
# tools/ab_env.sh "<lib> [ENV=val ...]" ...  — per-kernel times of several (build, environment) variants alternating on ONE GPU box (tools/quick_kernel_ms.py; MPC-only steps)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
N=${AB_ROUNDS:-2}
for i in $(seq 1 $N); do
  for V in "$@"; do set -- $V; L=$1; shift; echo -n "$L $*  "; env "$@" QM_AB_LIB=$L QM_MPC_ONLY=${QM_MPC_ONLY-1} python tools/quick_kernel_ms.py 2>&1 | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernel_ms'], d['ok'], d['tau_checksum'])"; set -- "$@"; done
done
