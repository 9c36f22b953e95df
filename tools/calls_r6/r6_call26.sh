#!/bin/bash
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
OUT=gpurun_out/r6z
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_training.py tests/test_hip_parity.py -m gpu -x -q -k "training or rowgemm or wgrad or lstm or sum or pack or train" > $OUT/pytest.log 2>&1 || { tail -40 $OUT/pytest.log; exit 1; }
tail -2 $OUT/pytest.log
for rep in 1 2; do
  timeout -k 10 300 python tests/bench_train_step.py --cfg3 --steps 100 --no-cpu --graph --ggnn-adam 2>&1 | tail -1 | tee -a $OUT/ab.txt
done
timeout -k 10 300 python tests/bench_train_step.py --cfg3 --steps 100 --no-cpu --graph --ggnn-adam --bf16 2>&1 | tail -1 | tee -a $OUT/ab.txt
timeout -k 10 300 python tests/bench_train_step.py --steps 100 --no-cpu --graph --ggnn-adam 2>&1 | tail -1 | tee -a $OUT/ab.txt
timeout -k 10 300 python tests/bench_train_step.py --cfg3 --steps 50 --no-cpu --graph --ggnn-adam --classifier 2>&1 | tail -1 | tee -a $OUT/ab.txt
