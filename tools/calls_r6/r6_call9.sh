#!/bin/bash
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
OUT=gpurun_out/r6i
mkdir -p $OUT
python tools/probes/dccheck.py | tee -a $OUT/check.txt
GGNN_LIB_PATH=$PWD/graingraphnn_amd/libggnn_bf.so python tools/probes/dccheck.py | tee -a $OUT/check.txt
GGNN_LIB_PATH=$PWD/graingraphnn_amd/libggnn_bf.so timeout -k 10 400 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "decoder_cell or forward_golden or cfg3_ten" > $OUT/pytest_bf.log 2>&1
echo "pytest (bf) rc $?"; tail -2 $OUT/pytest_bf.log
for rep in 1 2 3; do
  for v in "" bf; do
    if [ -z "$v" ]; then unset GGNN_LIB_PATH; else export GGNN_LIB_PATH=$PWD/graingraphnn_amd/libggnn_$v.so; fi
    timeout -k 10 200 python tools/dcbench.py >> $OUT/dcbench.txt 2>> $OUT/dcbench.err || { echo "dcbench $v failed"; tail -5 $OUT/dcbench.err; exit 1; }
  done
done
unset GGNN_LIB_PATH
cat $OUT/dcbench.txt
for rep in 1 2; do
  for v in "" bf; do
    if [ -z "$v" ]; then unset GGNN_LIB_PATH; else export GGNN_LIB_PATH=$PWD/graingraphnn_amd/libggnn_$v.so; fi
    timeout -k 10 300 python bench.py --steps 500 --warmup 20 --no-cpu-baseline 2>> $OUT/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=$v', d['value'], d['value_median_of_repeats'], d['roofline']['avg_launch_us'])" | tee -a $OUT/ab.txt
  done
done
