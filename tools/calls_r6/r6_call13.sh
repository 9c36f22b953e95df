#!/bin/bash
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
OUT=gpurun_out/r6m
mkdir -p $OUT
for v in "" ecdma2 ecdma4; do
  if [ -z "$v" ]; then unset GGNN_LIB_PATH; else export GGNN_LIB_PATH=$PWD/graingraphnn_amd/libggnn_$v.so; fi
  python tools/probes/eccheck.py | tee -a $OUT/check.txt
done
for rep in 1 2; do
  for v in "" ecdma2 ecdma4; do
    if [ -z "$v" ]; then unset GGNN_LIB_PATH; else export GGNN_LIB_PATH=$PWD/graingraphnn_amd/libggnn_$v.so; fi
    timeout -k 10 200 python tools/ecbench.py >> $OUT/ecbench.txt 2>> $OUT/ecbench.err || exit 1
  done
done
cat $OUT/ecbench.txt
for rep in 1 2; do
  for v in "" ecdma2 ecdma4; do
    if [ -z "$v" ]; then unset GGNN_LIB_PATH; else export GGNN_LIB_PATH=$PWD/graingraphnn_amd/libggnn_$v.so; fi
    timeout -k 10 300 python bench.py --steps 500 --warmup 20 --no-cpu-baseline 2>> $OUT/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=$v', d['value'], d['value_median_of_repeats'], d['roofline_encoder_cell']['avg_launch_us'])" | tee -a $OUT/ab.txt
  done
done
