#!/bin/bash
# Round 6: full GPU tests on the current tree, event-loop records, event fuzz, the N > 1 rehearsal through gloo on one GPU
# (six rank processes: the box allows at most six processes on its card) -> gpurun_out/r6e/
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
OUT=gpurun_out/r6e
mkdir -p $OUT
timeout -k 10 800 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1
rc=$?
echo "pytest rc $rc"; tail -4 $OUT/pytest.log
if [ $rc -ne 0 ]; then grep -n "Error\|FAILED" $OUT/pytest.log | head; exit $rc; fi
timeout -k 10 500 python tests/bench_event_step.py > $OUT/event_step.json 2> $OUT/event_step.err || { tail -20 $OUT/event_step.err; exit 1; }
cat $OUT/event_step.json
timeout -k 10 400 python tests/fuzz_events.py 36 > $OUT/fuzz_events.log 2>&1 || { tail -20 $OUT/fuzz_events.log; exit 1; }
tail -1 $OUT/fuzz_events.log
export GGNN_BENCH_BACKEND=gloo
timeout -k 10 500 python bench.py --gpus 6 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_gloo6_cfg3.json 2> $OUT/bench_gloo6_cfg3.err || { tail -20 $OUT/bench_gloo6_cfg3.err; exit 1; }
timeout -k 10 500 python bench.py --gpus 4 --steps 20 --warmup 5 --no-cpu-baseline --workload cfg4 > $OUT/bench_gloo4_cfg4.json 2> $OUT/bench_gloo4_cfg4.err || { tail -20 $OUT/bench_gloo4_cfg4.err; exit 1; }
unset GGNN_BENCH_BACKEND
python - <<'PY'
import json
for n in ("bench_gloo6_cfg3", "bench_gloo4_cfg4"):
    d = json.loads(open(f"gpurun_out/r6e/{n}.json").read().strip().splitlines()[-1])
    print(n, d["value"], d["n_gpus"], d["rccl_ranks"], d["collective_backend"], d["config"]["results_finite"])
PY
