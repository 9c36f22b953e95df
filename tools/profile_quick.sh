#!/bin/bash
# Quick per-kernel table of the step (gpurun -- bash tools/profile_quick.sh TAG [extra bench flags]):
# rocprofv3 --kernel-trace --stats of `bench.py --profile --no-graph --serial` -> gpurun_out/TAG_kernel_table.txt
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the repository copy on the GPU box)}" || exit 1
TAG=${1:-quick}
shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/serial -- python3 bench.py --steps 48 --warmup 4 --profile --no-graph --serial "$@" > $OUT/serial.log 2>&1
find $OUT/serial -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/serial_kernel_stats.csv
rm -rf $OUT/serial
python3 tools/kernel_table.py $OUT/serial_kernel_stats.csv > $OUT/kernel_table.txt
cat $OUT/kernel_table.txt
