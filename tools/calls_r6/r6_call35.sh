#!/bin/bash
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
OUT=gpurun_out/r6F
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "segment_graphs or step_events or speculative or event_rollout" > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
