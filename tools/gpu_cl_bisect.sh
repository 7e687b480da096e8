#!/bin/bash
# same box: the device loop of the README experiment on three builds, the current one twice (run-to-run determinism)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python tools/closed_loop_trace.py head_a

QM_AB_LIB=tools/_build/libqmhip_3124eb4.so python tools/closed_loop_trace.py c3124eb4
QM_AB_LIB=tools/_build/libqmhip_d57968a.so python tools/closed_loop_trace.py cd57968a
