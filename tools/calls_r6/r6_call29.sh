#!/bin/bash
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
OUT=gpurun_out/r6C
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "step_events or event or parameter_updates" > $OUT/pytest.log 2>&1 || { tail -40 $OUT/pytest.log; exit 1; }
tail -2 $OUT/pytest.log
timeout -k 10 400 python tests/bench_event_step.py > $OUT/event.txt 2>&1 || { tail -20 $OUT/event.txt; exit 1; }
tail -1 $OUT/event.txt
