#!/bin/bash
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
OUT=gpurun_out/r6I
mkdir -p $OUT
for period in 4 16 64; do
  for mode in 1 0; do
    GGNN_EVENT_GRAPHS=$mode timeout -k 10 300 python tools/probes/evsparse.py $period $((256 / period + 4)) 2>&1 | grep "in place" | tee -a $OUT/sparse.txt
  done
done
