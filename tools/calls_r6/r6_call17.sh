#!/bin/bash
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
bash tools/profile_round.sh r6_v3 stats bench
OUT=gpurun_out/r6q
mkdir -p $OUT
for m in static events; do
  timeout -k 10 200 python3 tools/probes/evquiet.py $m 400 >> $OUT/plain_$m.log 2>&1
  timeout -k 10 200 python3 tools/probes/evquiet.py $m 400 >> $OUT/plain_$m.log 2>&1
  grep "per step" $OUT/plain_$m.log
done
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_events -- python3 tools/probes/evquiet.py events 120 > $OUT/trace_events.log 2>&1
f=$(find $OUT/trace_events -name "*kernel_trace.csv" | head -1)
python3 tools/timeline.py $f > $OUT/timeline_events.txt 2>&1
rm -rf $OUT/trace_events
cat $OUT/timeline_events.txt
