#!/bin/bash
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
OUT=gpurun_out/r6j
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1
rc=$?
echo "pytest rc $rc"; tail -4 $OUT/pytest.log
if [ $rc -ne 0 ]; then grep -n "Error\|FAILED" $OUT/pytest.log | head; exit $rc; fi
for extra in "--graph --ggnn-adam --cfg3" "--graph --ggnn-adam --cfg3 --bf16" "--graph --ggnn-adam"; do
  timeout -k 10 300 python tests/bench_train_step.py --no-cpu --json --steps 20 $extra 2>> $OUT/train.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$extra', d.get('ms_per_step'))" | tee -a $OUT/train.txt
done
bash tools/profile_train.sh r6j_train --graph --ggnn-adam > $OUT/profile_train.log 2>&1
head -50 gpurun_out/r6j_train/train_kernel_table.txt
