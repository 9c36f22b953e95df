#!/bin/bash
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
OUT=gpurun_out/r6k
mkdir -p $OUT
python tools/probes/dccheck.py | tee -a $OUT/check.txt
echo "expected         d5004fae9e2c4239ae44dea050573d0f (round-5 fold, branch-free fold)"
timeout -k 10 500 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "decoder_cell or golden or cfg3_ten or block_major or cfg4 or generated" > $OUT/pytest.log 2>&1
echo "pytest rc $?"; tail -2 $OUT/pytest.log
for rep in 1 2 3; do timeout -k 10 200 python tools/dcbench.py >> $OUT/dcbench.txt 2>> $OUT/dcbench.err || exit 1; done
cat $OUT/dcbench.txt
for rep in 1 2 3; do
  timeout -k 10 300 python bench.py --steps 500 --warmup 20 --no-cpu-baseline 2>> $OUT/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['value_median_of_repeats'], d['roofline']['avg_launch_us'])" | tee -a $OUT/ab.txt
done
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>> $OUT/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver', d['value'], d['value_median_of_repeats'])" | tee -a $OUT/ab.txt
