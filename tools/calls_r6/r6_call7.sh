#!/bin/bash
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
OUT=gpurun_out/r6g
mkdir -p $OUT
for rep in 1 2 3; do
  for lay in rows block; do
    GGNN_VLAYOUT=$lay timeout -k 10 300 python bench.py --steps 500 --warmup 20 --no-cpu-baseline 2> $OUT/bench_${lay}_$rep.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lay', $rep, d['value'], d['value_median_of_repeats'], d['roofline']['avg_launch_us'], [g['avg_launch_us'] for g in d['roofline_gemm'][:1]])" | tee -a $OUT/ab.txt
  done
done
for lay in rows block; do
  GGNN_VLAYOUT=$lay timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>> $OUT/bench_drv.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver command $lay', d['value'], d['value_median_of_repeats'])" | tee -a $OUT/ab.txt
  GGNN_VLAYOUT=$lay timeout -k 10 300 python bench.py --workload gen368 --steps 100 --warmup 10 --no-cpu-baseline 2>> $OUT/bench_gen.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('gen368 $lay', d['value'], d['value_median_of_repeats'])" | tee -a $OUT/ab.txt
done
