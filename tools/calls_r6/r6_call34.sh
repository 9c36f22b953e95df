#!/bin/bash
# event mode with a topology that changes in place (graphs survive events): event tests, fuzz, timing
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
OUT=gpurun_out/r6E
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_topology.py -m gpu -x -q -k "event or topology or step_events or rollout or cfg4 or refresh" > $OUT/pytest.log 2>&1 || { tail -60 $OUT/pytest.log; exit 1; }
tail -2 $OUT/pytest.log
timeout -k 10 500 python tests/fuzz_events.py > $OUT/fuzz_events.log 2>&1 || { tail -30 $OUT/fuzz_events.log; exit 1; }
tail -2 $OUT/fuzz_events.log
timeout -k 10 400 python tests/bench_event_step.py > $OUT/event.json 2> $OUT/event.err || { tail -30 $OUT/event.err; exit 1; }
tail -1 $OUT/event.json | cut -c1-700
GGNN_EVENT_GRAPHS=0 timeout -k 10 400 python tests/bench_event_step.py > $OUT/event_off.json 2> $OUT/event_off.err || { tail -30 $OUT/event_off.err; exit 1; }
tail -1 $OUT/event_off.json | cut -c1-700
