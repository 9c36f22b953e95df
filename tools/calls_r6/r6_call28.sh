#!/bin/bash
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
OUT=gpurun_out/r6B
mkdir -p $OUT
timeout -k 10 400 python tools/probes/evprofile.py > $OUT/evprofile.txt 2>&1
grep -c . $OUT/evprofile.txt
