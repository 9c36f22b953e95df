#!/bin/bash
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
OUT=gpurun_out/r6n
mkdir -p $OUT
for rep in 1 2; do
  for v in "" prio dmamid0 ahead2; do
    if [ -z "$v" ]; then unset GGNN_LIB_PATH; else export GGNN_LIB_PATH=$PWD/graingraphnn_amd/libggnn_$v.so; fi
    timeout -k 10 200 python tools/dcbench.py >> $OUT/dcbench.txt 2>> $OUT/dcbench.err || exit 1
  done
done
unset GGNN_LIB_PATH
cat $OUT/dcbench.txt
bash tools/profile_round.sh r6_v3 pmc
