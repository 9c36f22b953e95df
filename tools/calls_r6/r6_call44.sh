#!/bin/bash
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
OUT=gpurun_out/r6_v5
mkdir -p $OUT
timeout 700 python3 bench.py > $OUT/bench_default_b.json 2> $OUT/bench_default_b.err
tail -1 $OUT/bench_default_b.json | cut -c1-200
