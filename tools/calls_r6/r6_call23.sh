#!/bin/bash
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
OUT=gpurun_out/r6w
mkdir -p $OUT
for mode in none h; do
  for rep in 1 2 3 4 5 6; do
    GGNN_TRAIN_STREAMS=$mode timeout -k 10 300 python -m pytest tests/test_training.py -m gpu -x -q -k "two_ranks_on_one_gpu" > $OUT/ddp_${mode}_$rep.txt 2>&1
    echo "mode $mode rep $rep rc $?"
    grep -o "AssertionError: ([0-9]*, \[.*" $OUT/ddp_${mode}_$rep.txt | cut -c1-700 | head -2
  done
done
