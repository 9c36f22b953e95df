#!/bin/bash
# Round 6: value rows block-major (GGNN_VLAYOUT=block) against row-major, same box -> gpurun_out/r6f/
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
OUT=gpurun_out/r6f
mkdir -p $OUT
GGNN_VLAYOUT=block timeout -k 10 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "golden or cfg3_ten or cfg3_full or cfg4 or generated or pipelined or fused_decoder or decoder_cell" > $OUT/pytest_block.log 2>&1
rc=$?
echo "pytest (block) rc $rc"; tail -4 $OUT/pytest_block.log
if [ $rc -ne 0 ]; then grep -n "Error\|FAILED\|assert" $OUT/pytest_block.log | head -20; exit $rc; fi
for rep in 1 2 3; do
  for lay in rows block; do
    GGNN_VLAYOUT=$lay timeout -k 10 300 python bench.py --steps 500 --warmup 20 --no-cpu-baseline 2> $OUT/bench_${lay}_$rep.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lay', $rep, d['value'], d['value_median_of_repeats'], d['roofline']['avg_launch_us'], [g['avg_launch_us'] for g in d['roofline_gemm'][:1]])" | tee -a $OUT/ab.txt
  done
done
for lay in rows block; do
  GGNN_VLAYOUT=$lay timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/serial_$lay -- python3 bench.py --steps 48 --warmup 4 --profile --no-graph --serial > $OUT/serial_$lay.log 2>&1
  find $OUT/serial_$lay -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/serial_${lay}_kernel_stats.csv
  rm -rf $OUT/serial_$lay
  echo "== $lay"; grep -E "dec_cell|project_x6|enc_cell" $OUT/serial_${lay}_kernel_stats.csv | cut -d, -f1-5
done
