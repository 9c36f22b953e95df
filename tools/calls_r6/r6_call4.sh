#!/bin/bash
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
OUT=gpurun_out/r6d
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "rollout or event or speculative or cfg3 or cfg4 or pipelined or glue or generated" > $OUT/pytest.log 2>&1
rc=$?
echo "pytest rc $rc"; tail -5 $OUT/pytest.log
if [ $rc -ne 0 ]; then exit $rc; fi
for m in static events; do
  for rep in 1 2; do
  timeout -k 10 200 python3 tools/probes/evquiet.py $m 400 >> $OUT/plain_$m.log 2>&1 || { tail -5 $OUT/plain_$m.log; exit 1; }
  done
  grep "per step" $OUT/plain_$m.log
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_$m -- python3 tools/probes/evquiet.py $m 120 > $OUT/trace_$m.log 2>&1 || { tail -5 $OUT/trace_$m.log; exit 1; }
  f=$(find $OUT/trace_$m -name "*kernel_trace.csv" | head -1)
  python3 tools/timeline.py $f > $OUT/timeline_$m.txt 2>&1
  rm -rf $OUT/trace_$m
  cat $OUT/timeline_$m.txt
done
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_command.json 2> $OUT/bench_driver_command.err || { tail -5 $OUT/bench_driver_command.err; exit 1; }
timeout -k 10 300 python bench.py --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default.err || { tail -5 $OUT/bench_default.err; exit 1; }
python - <<'PY'
import json
for n in ("bench_driver_command", "bench_default"):
    d = json.loads(open(f"gpurun_out/r6d/{n}.json").read().strip().splitlines()[-1])
    print(n, d["value"], d.get("value_median_of_repeats"), d["ms_per_step"], (d.get("roofline") or {}).get("frac"))
PY
