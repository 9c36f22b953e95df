#!/bin/bash
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
OUT=gpurun_out/r6y
mkdir -p $OUT
REPRO_PASSES=400 timeout -k 10 500 python tools/probes/repro2.py 2 > $OUT/repro2_400.txt 2>&1
echo "two processes x 400 plain passes: $(grep -c ' pass ' $OUT/repro2_400.txt) deviation lines"; grep " pass " $OUT/repro2_400.txt | cut -c1-250 | head -6
REPRO_PASSES=800 timeout -k 10 500 python tools/probes/repro2.py 1 > $OUT/repro1_800.txt 2>&1
echo "one process x 800 plain passes: $(grep -c ' pass ' $OUT/repro1_800.txt) deviation lines"; grep " pass " $OUT/repro1_800.txt | cut -c1-250 | head -6
