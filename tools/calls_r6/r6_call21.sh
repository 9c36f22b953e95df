#!/bin/bash
# ABI 25 (training step with fewer launches): the new unit tests, the training tests, the step's wall time and kernel table.
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
OUT=gpurun_out/r6u
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_training.py -m gpu -x -q > $OUT/pytest_training.log 2>&1 || { tail -40 $OUT/pytest_training.log; exit 1; }
tail -3 $OUT/pytest_training.log
timeout -k 10 300 python tests/bench_train_step.py --cfg3 --steps 50 --no-cpu --graph --ggnn-adam > $OUT/wall_graph.txt 2>&1 || { tail -20 $OUT/wall_graph.txt; exit 1; }
tail -3 $OUT/wall_graph.txt
timeout -k 10 300 python tests/bench_train_step.py --cfg3 --steps 50 --no-cpu --graph --ggnn-adam --bf16 > $OUT/wall_graph_bf16.txt 2>&1
tail -2 $OUT/wall_graph_bf16.txt
timeout -k 10 300 python tools/probes/train_ops.py > $OUT/train_ops.txt 2>&1
grep -c "aten\." $OUT/train_ops.txt
bash tools/profile_train.sh r6u/prof --graph --ggnn-adam > $OUT/profile_train.log 2>&1
tail -30 $OUT/profile_train.log
