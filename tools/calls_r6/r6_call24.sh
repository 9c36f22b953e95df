#!/bin/bash
# the 2-rank DDP test repeated on the PREVIOUS commit (worktree _old): is its rare failure older than ABI 25?
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run through gpurun}/_old" || exit 1
OUT=../gpurun_out/r6x
mkdir -p $OUT
for rep in 1 2 3 4 5 6 7 8 9 10 11 12 13 14; do
  timeout -k 10 300 python -m pytest tests/test_training.py -m gpu -x -q -k "two_ranks_on_one_gpu" > $OUT/old_$rep.txt 2>&1
  echo "old rep $rep rc $?"
done
