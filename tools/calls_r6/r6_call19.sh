#!/bin/bash
# The classifier-behind-the-encoder launch order as the default: GPU suite, A/B of the driver's command and of the event
# loop against GGNN_C_AFTER=none, the step timeline.
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
OUT=gpurun_out/r6s
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1 || { tail -20 $OUT/pytest.log; exit 1; }
tail -2 $OUT/pytest.log
for rep in 1 2 3; do
  for st in none enc; do
    GGNN_C_AFTER=$st timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>> $OUT/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver C_AFTER=$st', d['value'], d['value_median_of_repeats'])" | tee -a $OUT/ab.txt
    GGNN_C_AFTER=$st timeout -k 10 300 python bench.py --no-cpu-baseline 2>> $OUT/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default C_AFTER=$st', d['value'], d['value_median_of_repeats'])" | tee -a $OUT/ab.txt
  done
done
for st in none enc; do
  GGNN_C_AFTER=$st timeout -k 10 400 python tests/bench_event_step.py > $OUT/event_$st.txt 2>&1 || exit 1
  echo "== events C_AFTER=$st"; tail -25 $OUT/event_$st.txt
done
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 tools/probes/evquiet.py static 120 > $OUT/trace.log 2>&1
f=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python3 tools/timeline.py $f > $OUT/timeline_enc.txt 2>&1
rm -rf $OUT/trace
