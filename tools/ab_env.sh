#!/bin/bash
# tools/ab_env.sh "<lib> [ENV=val ...]" ...  — per-kernel times of several (build, environment) variants alternating on ONE GPU box (tools/quick_kernel_ms.py; MPC-only steps
# unless QM_MPC_ONLY is set empty).  AB_ROUNDS rounds over all variants.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
N=${AB_ROUNDS:-2}
for i in $(seq 1 $N); do
  for V in "$@"; do
    read -r -a W <<< "$V"; L=${W[0]}; E=("${W[@]:1}")
    echo -n "$V  "; env "${E[@]}" QM_AB_LIB=$L QM_MPC_ONLY=${QM_MPC_ONLY-1} python tools/quick_kernel_ms.py 2>&1 | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernel_ms'], d['ok'], d['tau_checksum'])"
  done
done
