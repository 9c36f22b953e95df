#!/bin/bash
# the cell backward's weight gradients on a second stream: tests, A/B of the replayed step
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
OUT=gpurun_out/r6v
mkdir -p $OUT
GGNN_TRAIN_STREAMS=h timeout -k 10 900 python -m pytest tests/test_training.py -m gpu -x -q > $OUT/pytest_training.log 2>&1 || { tail -40 $OUT/pytest_training.log; exit 1; }
tail -2 $OUT/pytest_training.log
for rep in 1 2; do
  for st in none h; do
    GGNN_TRAIN_STREAMS=$st timeout -k 10 300 python tests/bench_train_step.py --cfg3 --steps 100 --no-cpu --graph --ggnn-adam 2>&1 | tail -1 | sed "s/^/streams=$st /" | tee -a $OUT/ab.txt
  done
done
GGNN_TRAIN_STREAMS=h timeout -k 10 300 python tests/bench_train_step.py --cfg3 --steps 100 --no-cpu --graph --ggnn-adam --bf16 2>&1 | tail -1 | tee -a $OUT/ab.txt
GGNN_TRAIN_STREAMS=h timeout -k 10 300 python tests/bench_train_step.py --steps 100 --no-cpu --graph --ggnn-adam 2>&1 | tail -1 | tee -a $OUT/ab.txt
GGNN_TRAIN_STREAMS=none timeout -k 10 300 python tests/bench_train_step.py --steps 100 --no-cpu --graph --ggnn-adam 2>&1 | tail -1 | tee -a $OUT/ab.txt
GGNN_TRAIN_STREAMS=h bash tools/profile_train.sh r6v/prof --graph --ggnn-adam > $OUT/profile_train.log 2>&1
tail -3 $OUT/prof/train_timeline.txt
