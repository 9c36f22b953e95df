#!/bin/bash
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
OUT=gpurun_out/r6_final4
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_training.py -m gpu -x -q > $OUT/pytest.log 2>&1 || { tail -40 $OUT/pytest.log; exit 1; }
tail -2 $OUT/pytest.log
for rep in 1 2; do timeout -k 10 300 python tests/bench_train_step.py --cfg3 --steps 50 --no-cpu --graph --ggnn-adam --classifier 2>&1 | tail -1; done
timeout -k 10 300 python tests/bench_train_step.py --steps 50 --no-cpu --graph --ggnn-adam --classifier 2>&1 | tail -1
timeout -k 10 600 python tests/fuzz_training.py --n 12 > $OUT/fuzz_training_12.log 2>&1; echo rc $?; tail -1 $OUT/fuzz_training_12.log | cut -c1-300
