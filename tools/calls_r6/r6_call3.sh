#!/bin/bash
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
OUT=gpurun_out/r6c
mkdir -p $OUT
for m in static events; do
  timeout -k 10 200 python3 tools/probes/evquiet.py $m 400 > $OUT/plain_$m.log 2>&1 || { tail -5 $OUT/plain_$m.log; exit 1; }
  tail -1 $OUT/plain_$m.log
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_$m -- python3 tools/probes/evquiet.py $m 120 > $OUT/trace_$m.log 2>&1 || { tail -5 $OUT/trace_$m.log; exit 1; }
  f=$(find $OUT/trace_$m -name "*kernel_trace.csv" | head -1)
  python3 tools/timeline.py $f > $OUT/timeline_$m.txt 2>&1
  rm -rf $OUT/trace_$m
  cat $OUT/timeline_$m.txt
done
