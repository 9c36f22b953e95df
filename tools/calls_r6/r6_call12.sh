#!/bin/bash
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
OUT=gpurun_out/r6l
mkdir -p $OUT
python tools/probes/dccheck.py | tee -a $OUT/check.txt
echo "expected         d5004fae9e2c4239ae44dea050573d0f"
for rep in 1 2 3; do timeout -k 10 200 python tools/dcbench.py >> $OUT/dcbench.txt 2>> $OUT/dcbench.err || exit 1; done
cat $OUT/dcbench.txt
for rep in 1 2 3; do
  timeout -k 10 300 python bench.py --steps 500 --warmup 20 --no-cpu-baseline 2>> $OUT/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['value_median_of_repeats'], d['roofline']['avg_launch_us'])" | tee -a $OUT/ab.txt
done
