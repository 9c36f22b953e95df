#!/bin/bash
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
OUT=gpurun_out/r6o
mkdir -p $OUT
for rep in 1 2 3; do
  for v in pxold ""; do
    if [ -z "$v" ]; then unset GGNN_LIB_PATH; else export GGNN_LIB_PATH=$PWD/graingraphnn_amd/libggnn_$v.so; fi
    python tools/probes/pxcheck.py | tee -a $OUT/px.txt
  done
done
for rep in 1 2 3; do
  for v in pxold ""; do
    if [ -z "$v" ]; then unset GGNN_LIB_PATH; else export GGNN_LIB_PATH=$PWD/graingraphnn_amd/libggnn_$v.so; fi
    timeout -k 10 300 python bench.py --steps 500 --warmup 20 --no-cpu-baseline 2>> $OUT/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=$v', d['value'], d['value_median_of_repeats'], [g['avg_launch_us'] for g in d['roofline_gemm'][:1]])" | tee -a $OUT/ab.txt
  done
done
