#!/bin/bash
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
OUT=gpurun_out/r6D
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_training.py -m gpu -x -q > $OUT/pytest.log 2>&1 || { tail -40 $OUT/pytest.log; exit 1; }
tail -2 $OUT/pytest.log
bash tools/profile_train.sh r6D/train --graph --ggnn-adam > $OUT/profile_train.log 2>&1
grep "aggregate_bwd\|kernels per step" $OUT/train/train_kernel_table.txt; tail -1 $OUT/train/train_timeline.txt
for rep in 1 2; do timeout -k 10 300 python tests/bench_train_step.py --cfg3 --steps 100 --no-cpu --graph --ggnn-adam 2>&1 | tail -1; done
