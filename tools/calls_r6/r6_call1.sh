#!/bin/bash
# Round 6, first GPU call (gpurun -- bash tools/r6_call1.sh): the GPU tests, the decoder-cell variants of
# profiles/r6_dec_cell_experiments.txt, the bench lines of the new workloads and the event-loop records -> gpurun_out/r6a/
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
OUT=gpurun_out/r6a
mkdir -p $OUT
timeout -k 10 800 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1
rc=$?
echo "pytest rc $rc"; tail -15 $OUT/pytest.log
if [ $rc -ge 124 ]; then exit $rc; fi
for rep in 1 2; do
  for v in "" prio dmamid0 dmamid2 priomid; do
    if [ -z "$v" ]; then unset GGNN_LIB_PATH; else export GGNN_LIB_PATH=$PWD/graingraphnn_amd/libggnn_$v.so; fi
    timeout -k 10 200 python tools/dcbench.py >> $OUT/dcbench.txt 2>> $OUT/dcbench.err || { echo "dcbench $v failed"; tail -5 $OUT/dcbench.err; exit 1; }
  done
done
unset GGNN_LIB_PATH
cat $OUT/dcbench.txt
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_command.json 2> $OUT/bench_driver_command.err || { tail -5 $OUT/bench_driver_command.err; exit 1; }
timeout -k 10 300 python bench.py --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default.err || { tail -5 $OUT/bench_default.err; exit 1; }
timeout -k 10 400 python bench.py --workload gen368 --steps 100 --warmup 10 --no-cpu-baseline > $OUT/bench_gen368.json 2> $OUT/bench_gen368.err || { tail -5 $OUT/bench_gen368.err; exit 1; }
python - <<'PY'
import json
for n in ("bench_driver_command", "bench_default", "bench_gen368"):
    d = json.loads(open(f"gpurun_out/r6a/{n}.json").read().strip().splitlines()[-1])
    print(n, d["value"], d.get("value_median_of_repeats"), d["ms_per_step"], (d.get("roofline") or {}).get("frac"))
PY
timeout -k 10 500 python tests/bench_event_step.py > $OUT/event_step.json 2> $OUT/event_step.err || { tail -20 $OUT/event_step.err; exit 1; }
cat $OUT/event_step.json
timeout -k 10 400 python tests/fuzz_events.py 24 > $OUT/fuzz_events.log 2>&1 || { tail -20 $OUT/fuzz_events.log; exit 1; }
tail -3 $OUT/fuzz_events.log
