#!/bin/bash
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
OUT=gpurun_out/r6_final2
mkdir -p $OUT
timeout -k 10 300 python bench.py --events --steps 100 --warmup 10 --no-cpu-baseline > $OUT/bench_events.json 2> $OUT/bench_events.err; echo rc $?
tail -1 $OUT/bench_events.json | cut -c1-600
timeout -k 10 300 python bench.py --events-quiet --steps 200 --warmup 10 --no-cpu-baseline > $OUT/bench_events_quiet.json 2> $OUT/bench_events_quiet.err; echo rc $?
tail -1 $OUT/bench_events_quiet.json | cut -c1-300
timeout -k 10 300 python bench.py --workload cfg2 --events --steps 60 --warmup 5 --no-cpu-baseline > $OUT/bench_events_cfg2.json 2> $OUT/bench_events_cfg2.err; echo rc $?
tail -1 $OUT/bench_events_cfg2.json | cut -c1-300
