#!/bin/bash
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
OUT=gpurun_out/r6t
mkdir -p $OUT
timeout -k 10 300 python tools/probes/train_ops.py > $OUT/train_ops.txt 2>&1
tail -5 $OUT/train_ops.txt
