#!/bin/bash
# GPU box: per-kernel durations of the rollout step in its launch modes (rocprofv3 kernel trace;
# bench.py --profile traces nothing but rollout steps).  Usage: bash tools/profile_step.sh TAG
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-ps}
mkdir -p $OUT
for mode in joint serial two-streams; do
  flag=""; [ $mode != joint ] && flag="--$mode"; [ $mode == joint ] && flag="--joint"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$mode -- python3 bench.py --steps 48 --warmup 4 --profile --no-graph $flag > $OUT/$mode.log 2>&1
  find $OUT/$mode -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${mode}_kernel_stats.csv
  tail -1 $OUT/$mode.log | cut -c1-160
done
python3 tools/kernel_table.py $OUT/joint_kernel_stats.csv $OUT/serial_kernel_stats.csv $OUT/two-streams_kernel_stats.csv
