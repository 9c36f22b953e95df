#!/bin/bash
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
OUT=gpurun_out/r6_final3
mkdir -p $OUT
timeout -k 10 900 python tests/fuzz_training.py --n 60 > $OUT/fuzz_training_60.log 2>&1; echo rc $?; tail -2 $OUT/fuzz_training_60.log | cut -c1-900
