#!/bin/bash
# Round 6, second GPU call: GPU tests on the current tree, the event-loop records, the encoder-cell variants -> gpurun_out/r6b/
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
OUT=gpurun_out/r6b
mkdir -p $OUT
timeout -k 10 800 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1
rc=$?
echo "pytest rc $rc"; tail -8 $OUT/pytest.log
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 500 python tests/bench_event_step.py > $OUT/event_step.json 2> $OUT/event_step.err || { tail -20 $OUT/event_step.err; exit 1; }
cat $OUT/event_step.json
for rep in 1 2; do
  for v in "" ecw5 ecw6 ecw7 ecprio; do
    if [ -z "$v" ]; then unset GGNN_LIB_PATH; else export GGNN_LIB_PATH=$PWD/graingraphnn_amd/libggnn_$v.so; fi
    timeout -k 10 200 python tools/ecbench.py >> $OUT/ecbench.txt 2>> $OUT/ecbench.err || { echo "ecbench $v failed"; tail -5 $OUT/ecbench.err; exit 1; }
  done
done
unset GGNN_LIB_PATH
cat $OUT/ecbench.txt
timeout -k 10 400 python tests/fuzz_events.py 24 > $OUT/fuzz_events.log 2>&1 || { tail -20 $OUT/fuzz_events.log; exit 1; }
tail -2 $OUT/fuzz_events.log
timeout -k 10 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --events-quiet > $OUT/bench_events_quiet.json 2> $OUT/bench_events_quiet.err || { tail -5 $OUT/bench_events_quiet.err; exit 1; }
timeout -k 10 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline > $OUT/bench_static200.json 2> $OUT/bench_static200.err || exit 1
python - <<'PY'
import json
for n in ("bench_events_quiet", "bench_static200"):
    d = json.loads(open(f"gpurun_out/r6b/{n}.json").read().strip().splitlines()[-1])
    print(n, d["value"], d["ms_per_step"])
PY
