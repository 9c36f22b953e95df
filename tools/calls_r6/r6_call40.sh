#!/bin/bash
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
OUT=gpurun_out/r6H
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_topology.py -m gpu -x -q -k "segment or speculative_blocks" > $OUT/pytest.log 2>&1 || { tail -60 $OUT/pytest.log; exit 1; }
tail -2 $OUT/pytest.log
timeout -k 10 400 python tests/bench_event_step.py > $OUT/event.json 2> $OUT/event.err || { tail -30 $OUT/event.err; exit 1; }
tail -1 $OUT/event.json | python -c "
import sys, json
d=json.loads(sys.stdin.read())
print(d['eventful_step_ms'], d['of_which'], d['quiet_over_static'])
for s in d['steady_eventful']: print('  ', s['workload'], s['ms_per_step_median'], s['ms_per_step_min_max'], s['steps_per_s'], s['quiet_step_events_ms_per_step_afterwards'])"
