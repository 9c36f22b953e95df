#!/bin/bash
# final library of round 6 (ABI 25): fuzz logs, training step profile, event-step record
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
OUT=gpurun_out/r6_final
mkdir -p $OUT
timeout -k 10 500 python tests/fuzz_training.py > $OUT/fuzz_training.log 2>&1 || { tail -20 $OUT/fuzz_training.log; exit 1; }
tail -2 $OUT/fuzz_training.log
timeout -k 10 500 python tests/fuzz_events.py > $OUT/fuzz_events.log 2>&1 || { tail -20 $OUT/fuzz_events.log; exit 1; }
tail -2 $OUT/fuzz_events.log
timeout -k 10 500 python tests/fuzz_forward.py > $OUT/fuzz_forward.log 2>&1 || { tail -20 $OUT/fuzz_forward.log; exit 1; }
tail -2 $OUT/fuzz_forward.log
bash tools/profile_train.sh r6_final/train --graph --ggnn-adam > $OUT/profile_train.log 2>&1
head -8 $OUT/train/train_kernel_table.txt; tail -1 $OUT/train/train_timeline.txt
timeout -k 10 300 python tests/bench_train_step.py --cfg3 --steps 100 --no-cpu --graph --ggnn-adam --bf16 2>&1 | tail -1 | tee $OUT/train_wall_bf16.txt
timeout -k 10 300 python tests/bench_train_step.py --cfg3 --steps 50 --no-cpu --graph --ggnn-adam --classifier 2>&1 | tail -1 | tee $OUT/train_wall_classifier.txt
timeout -k 10 300 python tests/bench_train_step.py --steps 100 --no-cpu --graph --ggnn-adam 2>&1 | tail -1 | tee $OUT/train_wall_batch4.txt
timeout -k 10 400 python tests/bench_event_step.py > $OUT/event_step.json 2> $OUT/event_step.err || { tail -20 $OUT/event_step.err; exit 1; }
tail -1 $OUT/event_step.json | cut -c1-400
