#!/bin/bash
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
OUT=gpurun_out/r6_final4
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_topology.py tests/test_cabi.py -m gpu -x -q -k "event or topology or csr or rollout or segment or speculative or cabi" > $OUT/pytest2.log 2>&1 || { tail -40 $OUT/pytest2.log; exit 1; }
tail -2 $OUT/pytest2.log
GGNN_EVENT_GRAPHS=0 timeout -k 10 500 python tests/fuzz_events.py 48 > $OUT/fuzz_off_48.log 2>&1; echo rc $?; tail -1 $OUT/fuzz_off_48.log
