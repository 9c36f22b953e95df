#!/bin/bash
# Training-step kernel table (gpurun -- bash tools/profile_train.sh TAG [bench_train_step flags, default: eager foreach]):
# rocprofv3 kernel trace (rocpd .db) of `tests/bench_train_step.py --cfg3 --steps 20 --no-cpu FLAGS`
# -> gpurun_out/TAG/train_kernel_table.txt + train_sequence.txt + the wall figures without the profiler.
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the repository copy on the GPU box)}" || exit 1
OUT=gpurun_out/${1:-train}
shift || true
mkdir -p $OUT
timeout 400 rocprofv3 --kernel-trace --output-format rocpd -d $OUT/tr -- python3 tests/bench_train_step.py --cfg3 --steps 20 --no-cpu "$@" > $OUT/train.log 2>&1
DB=$(find $OUT/tr -name "*.db" | head -1)
python3 tools/train_kernels.py $DB 20 3 > $OUT/train_kernel_table.txt 2>&1
python3 tools/train_sequence.py $DB 20 3 > $OUT/train_sequence.txt 2>&1
python3 tools/train_timeline.py $DB > $OUT/train_timeline.txt 2>&1
rm -rf $OUT/tr
timeout 300 python3 tests/bench_train_step.py --cfg3 --steps 20 --no-cpu > $OUT/train_wall.txt 2>&1
timeout 300 python3 tests/bench_train_step.py --cfg3 --steps 20 --no-cpu --graph --fused > $OUT/train_wall_graph.txt 2>&1
head -60 $OUT/train_kernel_table.txt; tail -3 $OUT/train_wall.txt; tail -3 $OUT/train_wall_graph.txt
