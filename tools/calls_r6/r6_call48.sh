#!/bin/bash
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
OUT=gpurun_out/r6_final3
mkdir -p $OUT
timeout -k 10 900 python tests/fuzz_events.py 160 > $OUT/fuzz_events_160.log 2>&1; echo rc $?
grep -n "case 63\|identical\|Error\|apart" $OUT/fuzz_events_160.log | cut -c1-400 | tail -6
timeout -k 10 300 python tools/probes/fuzz63.py 2>&1 | grep -v amdgpu.ids | grep "^step 4\|^chunks (5" | cut -c1-300
