#!/bin/bash
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
OUT=gpurun_out/r6r
mkdir -p $OUT
for rep in 1 2 3; do
  for st in none enc dec; do
    GGNN_C_AFTER=$st timeout -k 10 300 python bench.py --steps 500 --warmup 20 --no-cpu-baseline 2>> $OUT/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C_AFTER=$st', d['value'], d['value_median_of_repeats'])" | tee -a $OUT/ab.txt
  done
done
GGNN_C_AFTER=enc timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 tools/probes/evquiet.py static 120 > $OUT/trace.log 2>&1
f=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python3 tools/timeline.py $f > $OUT/timeline_enc.txt 2>&1
rm -rf $OUT/trace
cat $OUT/timeline_enc.txt
