#!/bin/bash
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
OUT=gpurun_out/r6_final3
mkdir -p $OUT
timeout -k 10 1100 python tests/fuzz_events.py 500 > $OUT/fuzz_events_500.log 2>&1; echo rc $?
grep -n "case 63\|identical\|Error\|apart" $OUT/fuzz_events_500.log | cut -c1-400 | tail -6
