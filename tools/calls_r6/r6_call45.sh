#!/bin/bash
# end-of-round stability pass: a longer event fuzz on the in-place topology, and the GPU suite once more on another box
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
OUT=gpurun_out/r6_final2
mkdir -p $OUT
timeout -k 10 900 python tests/fuzz_events.py 120 > $OUT/fuzz_events_120.log 2>&1 || { tail -30 $OUT/fuzz_events_120.log; exit 1; }
tail -1 $OUT/fuzz_events_120.log
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1 || { tail -40 $OUT/pytest_gpu.log; exit 1; }
tail -2 $OUT/pytest_gpu.log
