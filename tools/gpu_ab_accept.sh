#!/bin/bash
# tools/gpu_ab_accept.sh <base.so> <new.so> — acceptance of a kernel change on ONE box: per-kernel times of the two builds alternating (MPC-only and full step), then every output
# array of a cold step AND of warm-started receding-horizon steps compared entry by entry (a 16-digit checksum of a cold step does not see rounding-level changes, profiles/r05_build_bisect.txt)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export QM_DUMP_DIR=/tmp; A=$1; B=$2
echo "== MPC-only steps"; AB_ROUNDS=3 bash tools/ab_env.sh "$A" "$B"
echo "== full control steps"; QM_MPC_ONLY= AB_ROUNDS=3 bash tools/ab_env.sh "$A" "$B"
for n in 1 3; do
  QM_AB_LIB=$A python tools/warm_step_dump.py a_$n $n; QM_AB_LIB=$B python tools/warm_step_dump.py b_$n $n
  echo "== $n receding-horizon step(s), $A vs $B"; python tools/cold_step_compare.py /tmp a_$n b_$n
done
