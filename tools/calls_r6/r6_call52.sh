#!/bin/bash
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
OUT=gpurun_out/r6_final4
mkdir -p $OUT
GGNN_BENCH_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 2 --steps 20 --warmup 5 > $OUT/bench_gloo2.json 2> $OUT/bench_gloo2.err; echo rc $?
tail -1 $OUT/bench_gloo2.json | cut -c1-400
GGNN_BENCH_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 4 --workload cfg4 --steps 20 --warmup 5 > $OUT/bench_gloo4_cfg4.json 2> $OUT/bench_gloo4_cfg4.err; echo rc $?
tail -1 $OUT/bench_gloo4_cfg4.json | cut -c1-300
