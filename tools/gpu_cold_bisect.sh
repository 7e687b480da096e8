#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export QM_DUMP_DIR=/tmp
for n in 1 2 3 10; do
  python tools/warm_step_dump.py head_$n $n; python tools/warm_step_dump.py headb_$n $n
  QM_AB_LIB=tools/_build/libqmhip_3b1b2fa.so python tools/warm_step_dump.py old_$n $n
  echo "== $n receding-horizon step(s): head vs head (rerun) vs 3b1b2fa"; python tools/cold_step_compare.py /tmp head_$n headb_$n old_$n
done
